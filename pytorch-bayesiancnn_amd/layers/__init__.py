"""Drop-in replacement for the reference's ``layers`` package (layers/__init__.py:1-7 upstream).

Put this directory's parent ahead of the reference checkout on sys.path and
models/BayesianModels/*.py, main_bayesian.getModel / train_model / validate_model run unchanged on the
MI355X-native kernels.
"""
from .bbb import BBBLinear as BBB_Linear
from .bbb import BBBConv2d as BBB_Conv2d
from .lrt import BBBLinear as BBB_LRT_Linear
from .lrt import BBBConv2d as BBB_LRT_Conv2d
from .misc import FlattenLayer, ModuleWrapper

__all__ = ["BBB_Linear", "BBB_Conv2d", "BBB_LRT_Linear", "BBB_LRT_Conv2d", "FlattenLayer", "ModuleWrapper"]
