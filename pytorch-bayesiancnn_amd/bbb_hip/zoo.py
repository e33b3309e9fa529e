"""BBBLeNet / BBBAlexNet / BBB3Conv3FC for hosts without the reference checkout (e.g. the GPU box).

Same constructor surface and attribute / state_dict names as models/BayesianModels/{BayesianLeNet,
BayesianAlexNet,Bayesian3Conv3FC}.py upstream -- `Model(outputs, inputs, priors, layer_type='lrt',
activation_type='softplus')`, `.num_classes` -- but generated from a topology table instead of
hand-written per model.  Like upstream the classes define no forward: ModuleWrapper.forward walks the
children in registration order, so the table order IS the graph.
"""
import torch.nn as nn

from layers import BBB_Linear, BBB_Conv2d, BBB_LRT_Linear, BBB_LRT_Conv2d, FlattenLayer, ModuleWrapper

# (attribute, kind, args)   kinds: conv(cout, k, stride, pad) | act | pool(k, s) | flatten(n) | fc(out or None=classes)
_TOPOLOGY = {
    "lenet": [
        ("conv1", "conv", (6, 5, 1, 0)), ("act1", "act", ()), ("pool1", "pool", (2, 2)),
        ("conv2", "conv", (16, 5, 1, 0)), ("act2", "act", ()), ("pool2", "pool", (2, 2)),
        ("flatten", "flatten", (5 * 5 * 16,)),
        ("fc1", "fc", (120,)), ("act3", "act", ()), ("fc2", "fc", (84,)), ("act4", "act", ()), ("fc3", "fc", (None,)),
    ],
    "alexnet": [
        ("conv1", "conv", (64, 11, 4, 5)), ("act1", "act", ()), ("pool1", "pool", (2, 2)),
        ("conv2", "conv", (192, 5, 1, 2)), ("act2", "act", ()), ("pool2", "pool", (2, 2)),
        ("conv3", "conv", (384, 3, 1, 1)), ("act3", "act", ()),
        ("conv4", "conv", (256, 3, 1, 1)), ("act4", "act", ()),
        ("conv5", "conv", (128, 3, 1, 1)), ("act5", "act", ()), ("pool3", "pool", (2, 2)),
        ("flatten", "flatten", (1 * 1 * 128,)), ("classifier", "fc", (None,)),
    ],
    "3conv3fc": [
        ("conv1", "conv", (32, 5, 1, 2)), ("act1", "act", ()), ("pool1", "pool", (3, 2)),
        ("conv2", "conv", (64, 5, 1, 2)), ("act2", "act", ()), ("pool2", "pool", (3, 2)),
        ("conv3", "conv", (128, 5, 1, 1)), ("act3", "act", ()), ("pool3", "pool", (3, 2)),
        ("flatten", "flatten", (2 * 2 * 128,)),
        ("fc1", "fc", (1000,)), ("act4", "act", ()), ("fc2", "fc", (1000,)), ("act5", "act", ()), ("fc3", "fc", (None,)),
    ],
}


class _TableNet(ModuleWrapper):
    _net_type = None

    def __init__(self, outputs, inputs, priors, layer_type="lrt", activation_type="softplus"):
        super().__init__()
        self.num_classes = outputs
        self.layer_type = layer_type
        self.priors = priors
        if layer_type == "lrt":
            Linear, Conv = BBB_LRT_Linear, BBB_LRT_Conv2d
        elif layer_type == "bbb":
            Linear, Conv = BBB_Linear, BBB_Conv2d
        else:
            raise ValueError("Undefined layer_type")
        if activation_type == "softplus":
            self.act = nn.Softplus
        elif activation_type == "relu":
            self.act = nn.ReLU
        else:
            raise ValueError("Only softplus or relu supported")
        cin, feat = inputs, None
        for name, kind, args in _TOPOLOGY[self._net_type]:
            if kind == "conv":
                cout, k, stride, pad = args
                mod = Conv(cin, cout, k, stride=stride, padding=pad, bias=True, priors=self.priors)
                cin = cout
            elif kind == "act":
                mod = self.act()
            elif kind == "pool":
                mod = nn.MaxPool2d(kernel_size=args[0], stride=args[1])
            elif kind == "flatten":
                feat = args[0]
                mod = FlattenLayer(feat)
            else:
                out = outputs if args[0] is None else args[0]
                mod = Linear(feat, out, bias=True, priors=self.priors)
                feat = out
            setattr(self, name, mod)


class BBBLeNet(_TableNet):
    _net_type = "lenet"


class BBBAlexNet(_TableNet):
    _net_type = "alexnet"


class BBB3Conv3FC(_TableNet):
    _net_type = "3conv3fc"


def getModel(net_type, inputs, outputs, priors, layer_type, activation_type):
    """Same selector as main_bayesian.getModel (main_bayesian.py:22-30)."""
    table = {"lenet": BBBLeNet, "alexnet": BBBAlexNet, "3conv3fc": BBB3Conv3FC}
    if net_type not in table:
        raise ValueError("Network should be either [LeNet / AlexNet / 3Conv3FC")
    return table[net_type](outputs, inputs, priors, layer_type, activation_type)
