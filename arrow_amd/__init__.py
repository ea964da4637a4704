"""arrow_amd — MI355X (gfx950) execution of the arrow::compute vectorized-kernel hot path.

`arrow_amd.compute` mirrors the reference's FunctionRegistry / call_function / options API over
device-resident arrays (`arrow_amd.array.Array`); all compute happens in libarrow_amd.so
(hand-written HIP kernels behind the C ABI of include/arrow_amd.h).  No CPU fallback exists.
"""
from . import array, compute, ipc, parquet  # noqa: F401
from ._lib import (ArrowAmdError, ArrowDeviceError, ArrowIndexError, ArrowInvalid,  # noqa: F401
                   ArrowNotImplementedError)
from .array import Array, Scalar  # noqa: F401

__version__ = "0.1.0"
