"""Host side of the MI355X-native attention-MIL + patch-graph GNN path.

``lib``   -- ctypes binding of libisic_hip.so (C ABI: include/isic_hip.h)
``ops``   -- torch.autograd Functions over the C ABI (the custom ops)
There is no CPU / eager-PyTorch fallback anywhere in this package.
"""
from . import lib  # noqa: F401
