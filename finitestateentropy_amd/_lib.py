"""Locate / build / load libfsehip.so.  Fails loudly: no library -> OSError, never a fallback."""
import ctypes
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def library_path():
    if os.environ.get("FSEHIP_LIB"):        # development aid: benchmark a differently configured build of the same library
        return os.environ["FSEHIP_LIB"]
    return os.path.join(_CSRC, "libfsehip.so")


def build_library(force=False):
    """Compile every HIP source for gfx950 into csrc/libfsehip.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _CSRC, "-j8", "all"], stdout=subprocess.DEVNULL)
    return library_path()


_LIB = None


def load():
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise OSError("libfsehip.so is missing (%s): build it with finitestateentropy_amd.build_library(); "
                          "there is no CPU fallback" % path)
        _LIB = ctypes.CDLL(path)
    return _LIB
