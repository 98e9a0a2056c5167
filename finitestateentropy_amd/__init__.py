"""finitestateentropy_amd -- MI355X-native block entropy coding (FSE / tANS and Huff0).

The product is the C-ABI shared library ``csrc/libfsehip.so`` (hand-written HIP kernels for gfx950,
declared in ``include/fsehip.h``).  This package is only the thin host-side binding used by the tests and
by ``bench.py``: ctypes calls on raw device pointers, with PyTorch providing device memory, streams and
``torch.distributed``.  There is no CPU fallback: importing :mod:`finitestateentropy_amd.api` raises if the
library has not been built (``python -c 'import __graft_entry__ as g; g.build()'``).
"""
from ._lib import build_library, library_path  # noqa: F401

__all__ = ["build_library", "library_path"]
