"""ctypes binding of libgmhip.so (the C-ABI declared in include/gmhip.h)."""
from . import runtime  # noqa: F401
from .array import DeviceArray  # noqa: F401
