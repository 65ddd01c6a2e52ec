"""fargocpt_amd -- MI355X-native FargoCPT gas update (hot path only).

The product is the HIP library `libfargocpt_hip.so` behind the C ABI of
include/fargocpt_hip.h; this package is the ctypes plumbing around it.  There
is no CPU fallback: importing works anywhere (host-side helpers such as the
grid construction and initial conditions are plain C++), but creating a
context without a HIP device raises.
"""
from __future__ import annotations

import ctypes
import os

from . import binding
from .binding import (Clock, Context, Desc, FcptError, Library, Split)  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FCPT_LIB_PATH") or os.path.join(_HERE, "libfargocpt_hip.so")  # FCPT_LIB_PATH: tuning builds

_lib = None


def load() -> Library:
    """Load the in-tree HIP library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FcptError(
                f"{LIB_PATH} is missing: build it with `make -C fargocpt_amd/csrc` "
                "(or __graft_entry__.build()); there is no fallback path")
        _lib = Library(ctypes.CDLL(LIB_PATH), "fcpt_")
    return _lib
