#!/usr/bin/env python3
"""Run the UNMODIFIED reference training driver (main_bayesian.py of kumar-shridhar/PyTorch-BayesianCNN) on the
MI355X-native layers.

    python run_reference.py --reference /path/to/PyTorch-BayesianCNN --net_type alexnet --dataset CIFAR10 \
        [--epochs 2] [--synthetic 2048] [--set layer_type=bbb --set batch_size=128] [--literal]

--literal executes main_bayesian.py ITSELF as `__main__` (runpy: its own argparse block, main_bayesian.py:137-142, parses
--net_type / --dataset); without it the module is imported and `run(dataset, net_type)` called -- the same code either way.
--set NAME=VALUE assigns attributes of the upstream `config_bayesian` module (the user's configuration file) before the run.

Nothing upstream is edited.  This launcher only prepares the interpreter before `main_bayesian` is imported:
  * sys.path: this package first, so `from layers import ...` in models/BayesianModels/*.py resolves to ours;
  * shims for today's library versions (the reference pins nothing and was written for torch 1.3 / numpy 1.x):
    `np.Inf`, and a `ReduceLROnPlateau` that accepts the removed `verbose=` keyword (main_bayesian.py:118-119);
  * when torchvision is absent (or --synthetic N is given) a stand-in `data` module with the reference's
    `getDataset` / `getDataloader` interface (data/data.py:36,169) serving random tensors of the dataset's shape.
Checkpoints are written by the reference's own `torch.save(net.state_dict(), ...)` with the reference's key names.
"""
import argparse
import importlib
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))

_SHAPES = {"MNIST": (1, 10), "CIFAR10": (3, 10), "CIFAR100": (3, 100)}


def install_version_shims():
    import numpy as np
    import torch
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    sched = torch.optim.lr_scheduler
    base = sched.ReduceLROnPlateau
    if not getattr(base, "_bbb_accepts_verbose", False):
        class ReduceLROnPlateau(base):
            _bbb_accepts_verbose = True

            def __init__(self, optimizer, *args, verbose=False, **kwargs):
                super().__init__(optimizer, *args, **kwargs)

        sched.ReduceLROnPlateau = ReduceLROnPlateau


def install_torchvision_stub():
    try:
        import torchvision  # noqa: F401
        return False
    except ImportError:
        for name in ("torchvision", "torchvision.transforms", "torchvision.datasets"):
            sys.modules.setdefault(name, types.ModuleType(name))
        sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
        sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
        return True


def make_synthetic_data_module(n_train):
    """A module with data.getDataset / data.getDataloader of the reference (data/data.py:36-186): 32x32 images."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    mod = types.ModuleType("data")

    def getDataset(dataset):
        if dataset not in _SHAPES:
            raise ValueError("synthetic data supports MNIST / CIFAR10 / CIFAR100")
        cin, ncls = _SHAPES[dataset]
        g = torch.Generator().manual_seed(0)
        mk = lambda n: TensorDataset(torch.rand(n, cin, 32, 32, generator=g), torch.randint(0, ncls, (n,), generator=g))
        return mk(n_train), mk(max(n_train // 5, 1)), cin, ncls

    def getDataloader(trainset, testset, valid_size, batch_size, num_workers):
        n = len(trainset)
        split = int(valid_size * n)
        idx = torch.randperm(n)
        tr = torch.utils.data.Subset(trainset, idx[split:].tolist())
        va = torch.utils.data.Subset(trainset, idx[:split].tolist())
        mk = lambda d: DataLoader(d, batch_size=batch_size, shuffle=False, num_workers=0, drop_last=False)
        return mk(tr), mk(va), mk(testset)

    mod.getDataset, mod.getDataloader = getDataset, getDataloader
    return mod


def prepare(reference, synthetic=0, import_driver=True):
    """Everything that has to happen before `import main_bayesian`.  Returns the imported module (None with
    import_driver=False: the caller executes the file itself)."""
    reference = os.path.abspath(reference)
    if not os.path.isfile(os.path.join(reference, "main_bayesian.py")):
        raise SystemExit(f"{reference} does not look like a PyTorch-BayesianCNN checkout")
    sys.dont_write_bytecode = True
    for p in (reference, HERE):                      # HERE ends up first: our `layers` shadows the reference's
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    for name in [m for m in sys.modules if m == "layers" or m.startswith("layers.")]:
        if not getattr(sys.modules[name], "__file__", "").startswith(HERE):
            del sys.modules[name]
    install_version_shims()
    stubbed = install_torchvision_stub()
    if synthetic or stubbed:
        sys.modules["data"] = make_synthetic_data_module(synthetic or 2048)
    return importlib.import_module("main_bayesian") if import_driver else None


def _parse_value(text):
    import ast
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def run_literal(reference, net_type, dataset, synthetic=0, overrides=None):
    """main_bayesian.py executed as a script, unmodified, in this interpreter: `python main_bayesian.py --net_type N --dataset D`."""
    import runpy
    prepare(reference, synthetic, import_driver=False)
    cfg = importlib.import_module("config_bayesian")
    for k, v in (overrides or {}).items():
        if not hasattr(cfg, k):
            raise SystemExit(f"config_bayesian has no attribute {k!r}")
        setattr(cfg, k, v)
    saved = sys.argv
    sys.argv = ["main_bayesian.py", "--net_type", net_type, "--dataset", dataset]
    try:
        return runpy.run_path(os.path.join(os.path.abspath(reference), "main_bayesian.py"), run_name="__main__")
    finally:
        sys.argv = saved


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="path to the upstream checkout")
    ap.add_argument("--net_type", default="lenet")
    ap.add_argument("--dataset", default="MNIST")
    ap.add_argument("--epochs", type=int, default=None, help="override config_bayesian.n_epochs")
    ap.add_argument("--synthetic", type=int, default=0, help="serve N random training images instead of torchvision data")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE", help="assign config_bayesian.NAME (repeatable)")
    ap.add_argument("--literal", action="store_true", help="execute main_bayesian.py itself as __main__ instead of importing it")
    args = ap.parse_args()
    overrides = {}
    for item in args.set:
        k, sep, v = item.partition("=")
        if not sep:
            raise SystemExit("--set expects NAME=VALUE")
        overrides[k] = _parse_value(v)
    if args.epochs is not None:
        overrides["n_epochs"] = args.epochs
    if args.literal:
        run_literal(args.reference, args.net_type, args.dataset, args.synthetic, overrides)
        return
    mb = prepare(args.reference, args.synthetic)
    for k, v in overrides.items():
        if not hasattr(mb.cfg, k):
            raise SystemExit(f"config_bayesian has no attribute {k!r}")
        setattr(mb.cfg, k, v)
    mb.run(args.dataset, args.net_type)


if __name__ == "__main__":
    main()
