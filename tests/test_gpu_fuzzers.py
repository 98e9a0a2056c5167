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


@pytest.mark.parametrize("name", ["fuzzer-linked", "fuzzerHuff0-linked"])
def test_reference_fuzzers_relinked_against_the_dropin_library(hip, name):
    """LINK-level drop-in: the same fuzzers compiled against the reference's headers alone -- no force-include, no rename macro -- and linked
    with libfse_dropin.so (csrc/dropin_alias.c: the reference's own symbol names forwarding to libfsehip.so) in front of the reference's
    library: FSE_compress / FSE_decompress / HUF_* / HIST_count resolve to the device at link time; the binary must actually depend on it"""
    exe = os.path.join(REFDIR, name)
    if not os.path.exists(exe):
        pytest.skip("%s not built (make -C oracle fuzzers, with libfse_dropin.so)" % name)
    nm = subprocess.run(["nm", "-D", "--undefined-only", exe], stdout=subprocess.PIPE).stdout.decode()
    assert ("FSE_compress" in nm or "HUF_compress" in nm) and "FSEHIP_" not in nm, nm          # bound by the reference's names only
    env = dict(os.environ, LD_DEBUG="bindings")
    p = subprocess.run([exe, "-s1", "-i300"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    out, dbg = p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    assert p.returncode == 0 and "Error" not in out, out[-2000:]
    sym = "FSE_decompress" if name == "fuzzer-linked" else "HUF_decompress"
    bound = [l for l in dbg.splitlines() if "symbol `%s'" % sym in l and "binding file" in l and name in l.split("to")[0]]
    assert bound and all("libfse_dropin" in l for l in bound), bound[:3]                          # ... and resolved to the drop-in at load time


@pytest.mark.parametrize("case", [1, 2, 3, 4, 5, 6, 7, 8, 9, 13, 14, 20, 21, 22, 23, 30, 33, 40, 41, 42, 45, 46, 80, 81])
def test_reference_fullbench_on_device(hip, case):
    """programs/fullbench.c (the reference's per-function speed analyzer; `make test` runs `fullbench -i1`, programs/Makefile:135-139) bound
    to the device like the fuzzers: one timed round of the cases whose function is a hot-path call of libfsehip.so -- 1 / 2 HIST_count with
    limits 255 / 254, 3 HIST_countFast(254), 7 FSE_compress_usingCTable, 8 the same into FSE_BLOCKBOUND - 1 bytes (the careful path,
    programs/fullbench.c:606-610,816-826), 9 FSE_compress, 13 FSE_decompress_usingDTable, 14 FSE_decompress, 20 HUF_compress, 23
    HUF_compress4x_usingCTable, 30 HUF_decompress, 33 HUF_decompress4X_usingDTable on the reference's double-symbol table (:954-965), 42
    HUF_decompress4X1_usingDTable, 46 HUF_decompress1X1_usingDTable (:1027-1043) (programs/fullbench.c:758-771,805-826,851-862,897-905,987-998) -- and,
    since the table glue binds too (oracle/fse_on_mi355x.h: FSEHIP_DROPIN_GLUE_NAMES), 4 FSE_normalizeCount, 5 FSE_writeNCount, 6 FSE_buildCTable,
    80 / 81 FSE_buildDTable at table logs 10 / 9, 21 HUF_buildCTable, 22 HUF_writeCTable, 40 HUF_decompress4X1, 41 HUF_readDTableX1 and 45
    HUF_decompress1X1 on a block whose table, header and stream the set-up produced with the device's HUF_buildCTable, HUF_writeCTable and
    HUF_compress1X_usingCTable (:773-803,878-895,967-985,1013-1025,1165-1186).  The program prints MB/s and the function's return value; a call
    that failed shows as an error code there, and the set-up of the decode cases (compressor -> device decoder) only works if the formats agree."""
    import re
    out = _run("fullbench-mi355x", "-i1", "-b%d" % case)
    m = re.findall(r"([0-9.]+) MB/s\s+\(\s*(\d+)\)", out)
    assert m, out[-1500:]
    speed, code = float(m[-1][0]), int(m[-1][1])
    assert speed > 0
    # the return value the program shows: sizes (HIST_count: the largest count; compressors: compressed size; decompressors: 32768)
    assert (0 if case in (6, 80, 81) else 1) <= code < (1 << 31) - 16, out[-600:]      # (FSE_buildCTable / FSE_buildDTable return 0)
    if case in (13, 14, 30, 33, 40, 42, 45, 46):
        assert code == 32768, out[-600:]
    if case in (6, 80, 81):
        assert code == 0, out[-600:]
    if case == 4:
        assert 5 <= code <= 12, out[-600:]                                                # FSE_normalizeCount returns the table log


def test_host_threads_driver_against_reference(hip):
    """tests/host/san_driver.c on the product library: four host threads through the calls on host pointers (per-thread scratch arenas,
    FSEHIP_releaseScratch), the _wksp names, then frames over the library's own thread pool -- every result byte for byte against the
    reference linked beside it.  scripts/sanitize.sh runs the same program on the -fsanitize=address,undefined build of the library."""
    out = _run("san_driver", "4", "6")
    assert "san_driver OK" in out, out[-2000:]
