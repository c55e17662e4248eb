"""mpg_amd - MI355X-native (gfx950) hot path of Mixed Policy Gradient (idthanm/mpg).

Python host side that keeps the reference's duck-typed class/method names (train_script.py:39-51) on top of
the C ABI in include/mpg_hip.h (libmpg_hip.so, hand-written HIP).  PyTorch is used for device memory,
streams and torch.distributed (RCCL) only."""
from ._lib import MpgError, lib  # noqa: F401

__all__ = ['MpgError', 'lib']
