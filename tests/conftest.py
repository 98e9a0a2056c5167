import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


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
