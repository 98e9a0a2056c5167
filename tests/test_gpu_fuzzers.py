"""The reference's OWN fuzzers, compiled unmodified from /root/reference/programs with their hot-path calls re-targeted at
libfsehip.so (oracle/Makefile target `fuzzers`, shim oracle/fse_on_mi355x.h), run on the device:
programs/fuzzer.c:142-273 (FSE_compress / FSE_decompress round trips on five distributions, too-small destinations, bogus
headers and bogus compressed data) and :282-464 (unit tests incl. raw tables through *_usingCTable / *_usingDTable, :420-444);
programs/fuzzerHuff0.c:137-261 (the same for HUF_compress / HUF_decompress).  The binaries exit non-zero on the first failed
check.  They are built in the container that has the reference tree and travel to the GPU box with the snapshot."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def _run(name, *args):
    exe = os.path.join(REFDIR, name)
    if not os.path.exists(exe):
        pytest.skip("%s not built (needs the reference tree: make -C oracle fuzzers)" % name)
    p = subprocess.run([exe, *args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0, out[-2000:]
    return out


def test_reference_fse_fuzzer_on_device(hip):
    out = _run("fuzzer-mi355x", "-s1", "-i2000")
    assert "Error" not in out, out[-2000:]


def test_reference_huff0_fuzzer_on_device(hip):
    out = _run("fuzzerHuff0-mi355x", "-s1", "-i2000")
    assert "Error" not in out, out[-2000:]
