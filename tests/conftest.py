import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


# The modules that PIN the restatement (oracle/fse_oracle.c against the compiled reference, the golden vectors and the reference tool): they need
# no GPU and run in the CPU suite; on a box with a GPU they additionally carry the `gpu` marker, so that the driver's `pytest -m gpu` run holds
# the whole parity chain (restatement -> compiled reference -> golden vectors, kernels -> compiled reference) and deselects none of it.
PIN_MODULES = ("test_oracle_vs_ref", "test_oracle_golden", "test_u16_golden", "test_frame_oracle")


def _gpu_box():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    if not _gpu_box():
        return
    for item in items:
        if item.module.__name__.split(".")[-1] in PIN_MODULES and item.get_closest_marker("gpu") is None:
            item.add_marker(pytest.mark.gpu)


@pytest.fixture(scope="session")
def restatement():
    """oracle/fse_oracle.c alone (the CPU restatement): what the pin modules put against the compiled reference and the golden vectors"""
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def oracle(request):
    """What the kernels are compared with.  One hop wherever possible: the COMPILED REFERENCE (oracle/_ref/libfse_ref.so, which travels to the
    GPU box) behind the restatement's interface -- `Checker` routes every codec entry point to it and keeps the restatement only for what the
    reference library does not have (the probagen generator, the .fse frame, the checksums).  Without oracle/_ref it is the restatement.
    (The pin modules override this fixture with `restatement`: there the restatement is the thing under test.)"""
    from oracle.oracle import Checker
    return Checker()


@pytest.fixture(scope="session")
def ref():
    from oracle.oracle import Ref
    if not Ref.available() and not os.path.isdir("/root/reference/lib"):
        pytest.skip("compiled reference (oracle/_ref) not present on this box")
    return Ref()


@pytest.fixture(scope="session")
def checker():
    """the GPU tests' checker: the compiled reference when oracle/_ref is present, the restatement otherwise"""
    from oracle.oracle import Checker
    return Checker()


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"))


@pytest.fixture(scope="session")
def hip():
    """The product: libfsehip.so through its C ABI.  GPU tests fail (not skip) if the library is missing."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from finitestateentropy_amd.api import FseHip
    # Guard mode for the whole GPU suite (SURVEY 8(b) "never write beyond dst + dstCapacity"; the reference's fuzzers check a guard
    # byte, programs/fuzzer.c:217-230, fuzzerHuff0.c:198-212): every destination the binding allocates gets a row stride of
    # capacity + guard with 0xA5 behind each block's capacity, asserted untouched after every batched call.  FSEHIP_TEST_GUARD=0
    # switches it off, an odd value additionally misaligns every slot.
    FseHip.guard = int(os.environ.get("FSEHIP_TEST_GUARD", "64"))
    return FseHip()
