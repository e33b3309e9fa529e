"""TEST INFRASTRUCTURE (oracle side) -- never imported by the product (`pytorch-bayesiancnn_amd/`); only tests/, bench.py's
cpu_baseline leg and __graft_entry__ may use it.

The reference (kumar-shridhar/PyTorch-BayesianCNN) is pure Python, so "building the reference for the oracle" means carrying
its UNMODIFIED hot-path files to wherever the checks run.  `/root/reference` exists in the build container only; the GPU box
gets the build products under `oracle/_ref/` (git-ignored, not gpurun-ignored -- exactly like an `oracle/_ref/*.so` of a
compiled reference).  The recipe:

    build_snapshot()   (called by __graft_entry__.build() when /root/reference is present)
        packs the files of the path (layers/, models/BayesianModels/, metrics.py, utils.py,
        config_bayesian.py, main_bayesian.py + data/ and uncertainty_estimation.py, which main_bayesian / section 8f import)
        byte for byte into ONE archive, oracle/_ref/upstream_snapshot.zip, with a manifest of their sha256.  No reference source
        is written into the repository as a source file, and nothing of it is committed.

    checkout()
        a directory holding the unmodified files: /root/reference itself when it exists, else the archive unpacked into a
        fresh temporary directory (verified against the manifest), else None.

Users: bench.py `cpu_baseline` (kind "reference": the upstream nn.Modules timed on the host cores of the GPU box),
tests/test_gpu_driver.py (the literal main_bayesian.py executed on the MI355X over the drop-in layers), tests marked
`reference`.
"""
import hashlib
import json
import os
import shutil
import tempfile
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
UPSTREAM = "/root/reference"
ARCHIVE = os.path.join(HERE, "_ref", "upstream_snapshot.zip")
MANIFEST = "MANIFEST.json"

# what travels: the files on / next to the hot path, nothing else of the upstream tree
_FILES = ("main_bayesian.py", "config_bayesian.py", "metrics.py", "utils.py", "uncertainty_estimation.py", "__init__.py")
_DIRS = ("layers", os.path.join("models", "BayesianModels"), "data")

_unpacked = None


def _members(root):
    out = [f for f in _FILES if os.path.isfile(os.path.join(root, f))]
    for d in _DIRS:
        for dp, dn, fn in os.walk(os.path.join(root, d)):
            dn[:] = sorted(x for x in dn if x != "__pycache__")
            for f in sorted(fn):
                if f.endswith(".py"):
                    out.append(os.path.relpath(os.path.join(dp, f), root))
    return sorted(out)


def _sha(b):
    return hashlib.sha256(b).hexdigest()


def build_snapshot(upstream=UPSTREAM, archive=ARCHIVE):
    """Pack the unmodified upstream files into `archive`; returns the manifest (None when there is no upstream checkout)."""
    if not os.path.isfile(os.path.join(upstream, "main_bayesian.py")):
        return None
    names = _members(upstream)
    blobs = {}
    for n in names:
        with open(os.path.join(upstream, n), "rb") as f:
            blobs[n] = f.read()
    manifest = {"origin": "kumar-shridhar/PyTorch-BayesianCNN (read-only checkout, packed unmodified)",
                "files": {n: {"sha256": _sha(b), "bytes": len(b)} for n, b in blobs.items()}}
    manifest["tree_sha256"] = _sha(json.dumps(manifest["files"], sort_keys=True).encode())
    if os.path.isfile(archive):                      # unchanged upstream -> leave the file (and its mtime) alone
        try:
            with zipfile.ZipFile(archive) as z:
                if json.loads(z.read(MANIFEST))["tree_sha256"] == manifest["tree_sha256"]:
                    return manifest
        except Exception:
            pass
    os.makedirs(os.path.dirname(archive), exist_ok=True)
    tmp = archive + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for n in names:
            zi = zipfile.ZipInfo(n, date_time=(2020, 1, 1, 0, 0, 0))     # reproducible archive
            zi.compress_type = zipfile.ZIP_DEFLATED
            z.writestr(zi, blobs[n])
        zi = zipfile.ZipInfo(MANIFEST, date_time=(2020, 1, 1, 0, 0, 0))
        z.writestr(zi, json.dumps(manifest, indent=1, sort_keys=True))
    os.replace(tmp, archive)
    return manifest


def manifest(archive=ARCHIVE):
    if not os.path.isfile(archive):
        return None
    with zipfile.ZipFile(archive) as z:
        return json.loads(z.read(MANIFEST))


def checkout(prefer_upstream=True):
    """Directory with the unmodified upstream files, or None.  (kind, path): kind is "checkout" or "snapshot"."""
    global _unpacked
    if prefer_upstream and os.path.isfile(os.path.join(UPSTREAM, "main_bayesian.py")):
        return "checkout", UPSTREAM
    if _unpacked and os.path.isfile(os.path.join(_unpacked, "main_bayesian.py")):
        return "snapshot", _unpacked
    m = manifest()
    if m is None:
        return None, None
    d = tempfile.mkdtemp(prefix="bbb_upstream_")
    with zipfile.ZipFile(ARCHIVE) as z:
        for n, meta in m["files"].items():
            b = z.read(n)
            if _sha(b) != meta["sha256"]:
                shutil.rmtree(d, ignore_errors=True)
                raise RuntimeError(f"oracle/_ref/upstream_snapshot.zip: {n} does not match its manifest")
            p = os.path.join(d, n)
            os.makedirs(os.path.dirname(p), exist_ok=True)
            with open(p, "wb") as f:
                f.write(b)
    _unpacked = d
    import atexit
    atexit.register(shutil.rmtree, d, True)
    return "snapshot", d


if __name__ == "__main__":
    m = build_snapshot()
    print("no upstream checkout" if m is None else f"{ARCHIVE}: {len(m['files'])} files, tree {m['tree_sha256'][:16]}")
