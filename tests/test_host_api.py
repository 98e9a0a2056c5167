"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/fsehip.h declares, and the Python
binding refuses arguments the raw-pointer ABI cannot take (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib_path():
    return os.path.join(ROOT, "finitestateentropy_amd", "csrc", "libfsehip.so")


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib_path()):
        import finitestateentropy_amd
        finitestateentropy_amd.build_library()
    lib = ctypes.CDLL(_lib_path())
    header = open(os.path.join(ROOT, "include", "fsehip.h")).read()
    names = sorted(set(re.findall(r"FSEHIP_API[^;]*?\b(FSEHIP_\w+)\s*\(", header)))
    assert len(names) >= 30
    for name in names:
        getattr(lib, name)                      # AttributeError = a declared entry point is missing
    lib.FSEHIP_versionString.restype = ctypes.c_char_p
    assert b"gfx950" in lib.FSEHIP_versionString()
    lib.FSEHIP_getErrorName.restype = ctypes.c_char_p
    lib.FSEHIP_getErrorName.argtypes = [ctypes.c_size_t]
    assert lib.FSEHIP_getErrorName((1 << 64) - 4) == b"Corrupted block detected"     # lib/error_private.h:88-104
    assert lib.FSEHIP_isError(ctypes.c_size_t((1 << 64) - 8)) == 1 and lib.FSEHIP_isError(ctypes.c_size_t(12345)) == 0


def test_dropin_names_cover_the_reference_prototypes():
    """every reference entry point of the hot path has a drop-in macro (lib/hist.h:30, lib/fse.h:76,90,104,174,247, lib/huf.h:66,82,95,...)"""
    header = open(os.path.join(ROOT, "include", "fsehip.h")).read()
    for name in ("HIST_count", "FSE_compress", "FSE_compress2", "FSE_decompress", "FSE_compress_usingCTable", "FSE_decompress_usingDTable",
                 "HUF_compress", "HUF_compress2", "HUF_decompress", "HUF_compress1X_usingCTable", "HUF_compress4X_usingCTable",
                 "HUF_decompress4X_usingDTable", "HUF_decompress4X1_usingDTable",
                 # the _wksp forms the reference's own callers go through (lib/fse.h:315,335, lib/huf.h:95,164,289, lib/hist.h:46,54)
                 "HIST_count_wksp", "HIST_countFast", "HIST_countFast_wksp", "HIST_count_simple", "FSE_compress_wksp", "FSE_decompress_wksp", "HUF_compress4X_wksp", "HUF_compress1X_wksp",
                 "HUF_decompress4X1_DCtx_wksp"):
        assert re.search(r"#define %s FSEHIP_%s\b" % (name, name), header), name


def test_link_level_dropin_library_exports_the_reference_names():
    """libfse_dropin.so (csrc/dropin_alias.c): every hot-path name of lib/fse.h, lib/huf.h, lib/hist.h as a real exported symbol, and a
    dependency on libfsehip.so -- an object compiled against the reference's headers links against it unchanged"""
    import subprocess
    path = os.path.join(ROOT, "finitestateentropy_amd", "csrc", "libfse_dropin.so")
    if not os.path.exists(path):
        import finitestateentropy_amd
        finitestateentropy_amd.build_library()
    out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, check=True).stdout.decode()
    have = set(l.split()[-1] for l in out.splitlines() if " T " in l)
    header = open(os.path.join(ROOT, "include", "fsehip.h")).read()
    names = set(re.findall(r"#define (\w+) FSEHIP_\1\b", header))
    assert len(names) >= 24 and names <= have, sorted(names - have)
    assert {"FSE_isError", "HUF_isError", "FSE_getErrorName", "FSE_compressBound", "HUF_compressBound"} <= have
    assert not any(n.startswith("FSEHIP_") for n in have)
    need = subprocess.run(["readelf", "-d", path], stdout=subprocess.PIPE, check=True).stdout.decode()
    assert "libfsehip.so" in need


def test_binding_rejects_what_the_raw_pointer_abi_cannot_take():
    from finitestateentropy_amd import api
    cpu = torch.zeros((4, 64), dtype=torch.uint8)
    with pytest.raises(TypeError):
        api._blocks(cpu, "src")                                   # host tensor: the kernels would dereference a host address
    with pytest.raises(TypeError):
        api._sizes_arg(torch.zeros(4, dtype=torch.int64))         # host sizes
    p, uni, keep = api._sizes_arg(np.int64(77))                   # numpy integers are uniform sizes
    assert uni.value == 77 and keep is None
    p, uni, keep = api._sizes_arg(None)
    assert uni.value == 0
    assert api.fse_compress_bound(32768) == 33548 and api.huf_compress_bound(32768) == 33033      # lib/fse.h:290-292, lib/huf.h:131-133


def test_mixed_corpus_layout():
    """config-5 corpus rule (block g: P[g mod 3], seed g+1): the strided generator calls cover every row exactly once"""
    k = 3
    for first in (0, 1, 2, 125000, 250001):
        for n in (0, 1, 2, 3, 10):
            seen = {}
            for j in range(k):
                r0 = (j - first) % k
                rows = list(range(r0, n, k))
                for i, row in enumerate(rows):
                    seed = first + r0 + 1 + i * k
                    assert row not in seen
                    seen[row] = (j, seed)
            assert sorted(seen) == list(range(n))
            for row, (j, seed) in seen.items():
                g = first + row
                assert j == g % k and seed == g + 1


def test_shard_range_c_abi_matches_the_python_one():
    """FSEHIP_shardRange (what a C host shards with) = finitestateentropy_amd.shard.shard_range"""
    from finitestateentropy_amd.shard import shard_range
    lib = ctypes.CDLL(_lib_path())
    lib.FSEHIP_shardRange.restype = None
    for n in (0, 1, 7, 8, 100000, 1000000, 1000003):
        for w in (1, 2, 3, 4, 8):
            for r in range(w):
                first, count = ctypes.c_size_t(), ctypes.c_size_t()
                lib.FSEHIP_shardRange(ctypes.c_size_t(n), ctypes.c_int(r), ctypes.c_int(w), ctypes.byref(first), ctypes.byref(count))
                lo, hi = shard_range(n, r, w)
                assert (first.value, count.value) == (lo, hi - lo), (n, w, r)


def test_c_host_example_with_rccl_compiles():
    """examples/shard_rccl.c -- the scatter / code / gather of config 5 written against the C ABI and RCCL -- must compile against
    include/fsehip.h and the ROCm headers without a warning (tests/test_gpu_rccl.py RUNS it on the GPU box: one rank exercises every RCCL call)"""
    import shutil, subprocess, tempfile
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h") or shutil.which("gcc") is None:
        pytest.skip("no RCCL headers / gcc here")
    with tempfile.TemporaryDirectory() as d:
        p = subprocess.run(["gcc", "-D__HIP_PLATFORM_AMD__", "-Wall", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"), "-c",
                            os.path.join(ROOT, "examples", "shard_rccl.c"), "-o", os.path.join(d, "x.o")], capture_output=True, text=True)
        assert p.returncode == 0 and "warning" not in p.stderr, p.stderr[-2000:]
