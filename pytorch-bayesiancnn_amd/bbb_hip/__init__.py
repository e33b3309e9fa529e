"""MI355X-native Bayes-by-Backprop hot path: host side (Python) over libbbb_hip.so.

  _lib      ctypes binding of the C ABI (include/bbb_hip.h); no fallback if the .so is missing
  rng       seed / call-counter contract of the on-chip Philox noise
  ops       tensor-level entry points + autograd wiring
  ensemble  the num_ens Monte-Carlo loop of main_bayesian.py:43-53 / 73-80, batched over draws and
            sharded over GPUs
  fast_train  one autograd node for the batched forward on the inference kernels (training extension)
  train     train_step / FusedAdam / GraphedTrainStep / data-parallel gradient averaging
  metrics   ELBO, get_beta, device-side accuracy (metrics.py upstream, without the per-step host sync)
  zoo       BBBLeNet / BBBAlexNet / BBB3Conv3FC built from a topology table (same constructor surface as
            the reference's models/BayesianModels/*.py, for hosts without the reference checkout)
The drop-in boundary itself is the sibling package ``layers``.
"""
import os as _os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin, and hipGraph branches take slots too: with 4,
# the lanes of a GraphedPipeline end up sharing a queue depending on how many streams the process created before (measured: the
# same 4-lane pipeline 0.087 or 0.115 ms per step).  8 queues keep every lane on its own.  Read by the HIP runtime when it
# initialises (the first device call), so this must run before that; a value the user exported wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# A sharded step records its one all_gather into the step's hipGraph (ensemble.collective_capture_ok).  torch's NCCL event cache
# makes that unsafe: the watchdog thread may query an event it still holds for a finished eager collective after the cache gave
# the same event to a collective under capture -> hipErrorCapturedEvent -> terminate().  Read when a process group is created, so
# it must be set before that; if it was not "0" then, ensemble keeps collectives outside its graphs.  A value the user exported wins.
# What was true BEFORE this import decides (ensemble.collective_capture_ok): torch reads the variable when a ProcessGroupNCCL is
# constructed, so a group that already existed while the variable was not "0" still runs with the cache on, whatever we set now.
_event_cache_off_before_import = _os.environ.get("TORCH_NCCL_CUDA_EVENT_CACHE") == "0"
_os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")


def _dist_initialized_now():
    try:
        import torch.distributed as _dist
        return bool(_dist.is_available() and _dist.is_initialized())
    except Exception:
        return False


_dist_initialized_before_import = _dist_initialized_now()


def nccl_event_cache_known_off():
    """True when every NCCL process group of this process was (or will be) constructed with TORCH_NCCL_CUDA_EVENT_CACHE=0: the
    variable was already "0" when this package was imported, or no process group existed yet at that point (the import set it) and
    nobody changed it since.  False = unknown -> collectives stay outside hipGraphs (the eager protocol)."""
    if _os.environ.get("TORCH_NCCL_CUDA_EVENT_CACHE") != "0":
        return False
    return _event_cache_off_before_import or not _dist_initialized_before_import

from . import _lib, rng, ops  # noqa: E402,F401
from ._lib import BBBHipError, LIB_PATH  # noqa: E402,F401
