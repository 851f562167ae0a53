"""pic1dp_amd -- MI355X-native time-step engine for 1-D electrostatic delta-f PIC
with the behaviour of the hot path of wenjundeng/pic1dp (PIC1D-PETSc).

The product is libpic1dp_hip.so (hand-written gfx950 HIP kernels behind the C
ABI of include/pic1dp_hip.h).  This package is the thin Python host mirroring the
reference's procedure names.  Importing it loads the library; if the library is
missing the import fails -- there is no CPU fallback.
"""
from . import _lib

_lib.load()  # fail loudly, now, if the HIP library is absent

from ._lib import Input, Layout, Pic1dpError  # noqa: E402,F401
from .engine import Pic1dp, device_count, make_input, tuning_build  # noqa: E402,F401
from . import parallel  # noqa: E402,F401

__all__ = ["Pic1dp", "make_input", "device_count", "tuning_build", "Input", "Layout", "Pic1dpError", "parallel"]
