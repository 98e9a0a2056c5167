"""ctypes bindings for the CPU checkers -- TEST INFRASTRUCTURE ONLY.

``Oracle``  -> oracle/liboracle.so      (our restatement, oracle/fse_oracle.c)
``Ref``     -> oracle/_ref/libfse_ref.so (the unmodified reference compiled by oracle/Makefile; may be
                                          absent on a box that never had /root/reference)

Both expose the same method names so a test can be parametrised over the two.  Nothing under
finitestateentropy_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ERR_NAMES = {1: "GENERIC", 2: "dstSize_tooSmall", 3: "srcSize_wrong", 4: "corruption_detected",
             5: "tableLog_tooLarge", 6: "maxSymbolValue_tooLarge", 7: "maxSymbolValue_tooSmall",
             8: "workSpace_tooSmall"}
SIZE_MAX = (1 << 64) - 1


def is_error(code):
    """lib/error_private.h:79"""
    return int(code) > (1 << 64) - 9


def err_code(code):
    return (1 << 64) - int(code) if is_error(code) else 0


def build(force=False):
    """(Re)build liboracle.so and, when /root/reference exists, _ref/libfse_ref.so."""
    lib = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "fse_oracle.c")
    need = force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src)
    if need:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(_HERE, "_ref", "libfse_ref.so")
    if os.path.isdir("/root/reference/lib") and (force or not os.path.exists(ref)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def fse_compress_bound(n):      # lib/fse.h:290-292
    return 512 + n + (n >> 7) + 4 + 8


def fse_block_bound(n):         # lib/fse.h:291
    return n + (n >> 7) + 4 + 8


def huf_compress_bound(n):      # lib/huf.h:131-133
    return 129 + n + (n >> 8) + 8


def fse_ctable_u32(table_log, max_sv):   # lib/fse.h:295
    return 1 + (1 << (table_log - 1) if table_log else 1) + (max_sv + 1) * 2


def fse_dtable_u32(table_log):           # lib/fse.h:296
    return 1 + (1 << table_log)


class _Base:
    """Common numpy-level API. Subclasses provide self._f(name) -> ctypes function."""

    sz = C.c_size_t
    vp = C.c_void_p

    def _call(self, name, restype, *args):
        fn = self._f(name)
        fn.restype = restype
        return fn(*args)

    # ---- a1 -------------------------------------------------------------------------------
    def hist_count(self, src, max_sv=255):
        src, ps = _u8(src)
        count = np.zeros(256, dtype=np.uint32)
        msv = C.c_uint(max_sv)
        r = self._call(self.N["hist_count"], self.sz, count.ctypes.data_as(self.vp), C.byref(msv), ps, self.sz(src.size))
        return int(r), int(msv.value), count

    # ---- g1..g3 ---------------------------------------------------------------------------
    def fse_optimal_tablelog(self, max_tl, src_size, max_sv, minus=2):
        return int(self._call(self.N["optimal_tablelog"], C.c_uint, C.c_uint(max_tl), self.sz(src_size), C.c_uint(max_sv), C.c_uint(minus)))

    def fse_normalize_count(self, table_log, count, total, max_sv):
        count = np.ascontiguousarray(count, dtype=np.uint32)
        norm = np.zeros(256, dtype=np.int16)
        r = self._call(self.N["normalize"], self.sz, norm.ctypes.data_as(self.vp), C.c_uint(table_log),
                       count.ctypes.data_as(self.vp), self.sz(total), C.c_uint(max_sv))
        return int(r), norm

    def fse_write_ncount(self, cap, norm, max_sv, table_log):
        norm = np.ascontiguousarray(norm, dtype=np.int16)
        out = np.zeros(max(cap, 1) + 8, dtype=np.uint8)
        r = self._call(self.N["write_ncount"], self.sz, out.ctypes.data_as(self.vp), self.sz(cap),
                       norm.ctypes.data_as(self.vp), C.c_uint(max_sv), C.c_uint(table_log))
        return int(r), out

    def fse_read_ncount(self, src, max_sv=255):
        src, ps = _u8(src)
        norm = np.zeros(max(256, max_sv + 1), dtype=np.int16)
        msv = C.c_uint(max_sv)
        tl = C.c_uint(0)
        r = self._call(self.N["read_ncount"], self.sz, norm.ctypes.data_as(self.vp), C.byref(msv), C.byref(tl), ps, self.sz(src.size))
        return int(r), int(msv.value), int(tl.value), norm

    def fse_build_ctable(self, norm, max_sv, table_log):
        norm = np.ascontiguousarray(norm, dtype=np.int16)
        ct = np.zeros(fse_ctable_u32(table_log, max_sv), dtype=np.uint32)
        r = self._call(self.N["build_ctable"], self.sz, ct.ctypes.data_as(self.vp), norm.ctypes.data_as(self.vp), C.c_uint(max_sv), C.c_uint(table_log))
        return int(r), ct

    def fse_build_ctable_raw(self, nb_bits):
        ct = np.zeros(fse_ctable_u32(nb_bits, (1 << nb_bits) - 1), dtype=np.uint32)
        r = self._call(self.N["build_ctable_raw"], self.sz, ct.ctypes.data_as(self.vp), C.c_uint(nb_bits))
        return int(r), ct

    def fse_build_dtable(self, norm, max_sv, table_log):
        norm = np.ascontiguousarray(norm, dtype=np.int16)
        dt = np.zeros(fse_dtable_u32(table_log), dtype=np.uint32)
        r = self._call(self.N["build_dtable"], self.sz, dt.ctypes.data_as(self.vp), norm.ctypes.data_as(self.vp), C.c_uint(max_sv), C.c_uint(table_log))
        return int(r), dt

    def fse_build_dtable_raw(self, nb_bits):
        dt = np.zeros(fse_dtable_u32(nb_bits), dtype=np.uint32)
        r = self._call(self.N["build_dtable_raw"], self.sz, dt.ctypes.data_as(self.vp), C.c_uint(nb_bits))
        return int(r), dt

    # ---- a2 / a3 --------------------------------------------------------------------------
    def _codec(self, key, cap, src, table):
        src, ps = _u8(src)
        table = np.ascontiguousarray(table, dtype=np.uint32)
        out = np.zeros(max(cap, 1) + 16, dtype=np.uint8)
        out[cap:] = 0xA5   # guard
        r = self._call(self.N[key], self.sz, out.ctypes.data_as(self.vp), self.sz(cap), ps, self.sz(src.size), table.ctypes.data_as(self.vp))
        assert (out[cap:] == 0xA5).all(), "%s wrote past dstCapacity" % key
        return int(r), out[:cap]

    def fse_compress_using_ctable(self, src, ct, cap=None):
        return self._codec("compress_using_ctable", fse_compress_bound(len(src)) if cap is None else cap, src, ct)

    def fse_decompress_using_dtable(self, csrc, dt, cap):
        return self._codec("decompress_using_dtable", cap, csrc, dt)

    # ---- g4 -------------------------------------------------------------------------------
    def fse_compress2(self, src, max_sv=255, table_log=11, cap=None):
        src, ps = _u8(src)
        cap = fse_compress_bound(src.size) if cap is None else cap
        out = np.zeros(max(cap, 1) + 16, dtype=np.uint8)
        out[cap:] = 0xA5
        r = self._call(self.N["fse_compress2"], self.sz, out.ctypes.data_as(self.vp), self.sz(cap), ps, self.sz(src.size), C.c_uint(max_sv), C.c_uint(table_log))
        assert (out[cap:] == 0xA5).all()
        return int(r), out[:cap]

    def fse_decompress(self, csrc, cap):
        csrc, ps = _u8(csrc)
        out = np.zeros(max(cap, 1) + 16, dtype=np.uint8)
        out[cap:] = 0xA5
        r = self._call(self.N["fse_decompress"], self.sz, out.ctypes.data_as(self.vp), self.sz(cap), ps, self.sz(csrc.size))
        assert (out[cap:] == 0xA5).all()
        return int(r), out[:cap]

    # ---- Huff0 ----------------------------------------------------------------------------
    def huf_build_ctable(self, count, max_sv, max_nb_bits):
        count = np.ascontiguousarray(count, dtype=np.uint32)
        celt = np.zeros(256, dtype=np.uint32)
        r = self._call(self.N["huf_build_ctable"], self.sz, celt.ctypes.data_as(self.vp), count.ctypes.data_as(self.vp), C.c_uint(max_sv), C.c_uint(max_nb_bits))
        return int(r), celt

    def huf_write_ctable(self, cap, celt, max_sv, huff_log):
        celt = np.ascontiguousarray(celt, dtype=np.uint32)
        out = np.zeros(max(cap, 1) + 8, dtype=np.uint8)
        r = self._call(self.N["huf_write_ctable"], self.sz, out.ctypes.data_as(self.vp), self.sz(cap), celt.ctypes.data_as(self.vp), C.c_uint(max_sv), C.c_uint(huff_log))
        return int(r), out

    def huf_read_dtable_x1(self, src, max_table_log=11):
        src, ps = _u8(src)
        dt = np.zeros(1 + (1 << 12), dtype=np.uint32)
        dt[0] = max_table_log * 0x01000001
        r = self._call(self.N["huf_read_dtable_x1"], self.sz, dt.ctypes.data_as(self.vp), ps, self.sz(src.size))
        return int(r), dt

    def huf_compress1x_using_ctable(self, src, celt, cap=None):
        return self._codec("huf_compress1x", huf_compress_bound(len(src)) if cap is None else cap, src, celt)

    def huf_compress4x_using_ctable(self, src, celt, cap=None):
        return self._codec("huf_compress4x", huf_compress_bound(len(src)) if cap is None else cap, src, celt)

    def huf_decompress4x1_using_dtable(self, csrc, dt, dst_size):
        return self._codec("huf_decompress4x1", dst_size, csrc, dt)

    def huf_decompress1x1_using_dtable(self, csrc, dt, dst_size):
        return self._codec("huf_decompress1x1", dst_size, csrc, dt)

    def huf_compress2(self, src, max_sv=255, huff_log=11, cap=None):
        src, ps = _u8(src)
        cap = huf_compress_bound(src.size) if cap is None else cap
        out = np.zeros(max(cap, 1) + 16, dtype=np.uint8)
        out[cap:] = 0xA5
        r = self._call(self.N["huf_compress2"], self.sz, out.ctypes.data_as(self.vp), self.sz(cap), ps, self.sz(src.size), C.c_uint(max_sv), C.c_uint(huff_log))
        assert (out[cap:] == 0xA5).all()
        return int(r), out[:cap]

    def huf_decompress(self, csrc, dst_size, x1_only=False):
        csrc, ps = _u8(csrc)
        out = np.zeros(max(dst_size, 1) + 16, dtype=np.uint8)
        out[dst_size:] = 0xA5
        key = "huf_decompress4x1_oneshot" if (x1_only and "huf_decompress4x1_oneshot" in self.N) else "huf_decompress"
        r = self._call(self.N[key], self.sz, out.ctypes.data_as(self.vp), self.sz(dst_size), ps, self.sz(csrc.size))
        assert (out[dst_size:] == 0xA5).all()
        return int(r), out[:dst_size]

    # ---- batches (OpenMP) -----------------------------------------------------------------
    def bench_roundtrip(self, codec, src2d, table_log=11, max_sv=255, nthreads=0, dynamic=False, min_seconds=1.0, reps=3):
        """multi-core baseline loop (oracle/cpu_bench.h): returns dict(enc_s, dec_s per pass of the sample, threads, ...)"""
        src2d = np.ascontiguousarray(src2d, dtype=np.uint8)
        n, bs = src2d.shape
        cap = fse_compress_bound(bs) if codec == 0 else huf_compress_bound(bs)
        out = (C.c_double * 8)()
        rc = self._call(self.N["bench_roundtrip"], C.c_int, C.c_int(codec), src2d.ctypes.data_as(self.vp), self.sz(n), self.sz(bs), self.sz(cap),
                        C.c_uint(max_sv), C.c_uint(table_log), C.c_int(nthreads), C.c_int(1 if dynamic else 0), C.c_double(min_seconds), C.c_int(reps), out)
        if rc != 0:
            raise RuntimeError("bench_roundtrip failed: %d" % rc)
        return {"enc_s": out[0], "dec_s": out[1], "threads": int(out[2]), "passes_enc": int(out[3]), "passes_dec": int(out[4]), "mean_csize": out[5]}

    def stream_bandwidth(self, nbytes=1 << 30, nthreads=0, passes=5):
        return float(self._call(self.N["stream_bandwidth"], C.c_double, self.sz(nbytes), C.c_int(nthreads), C.c_int(passes)))

    def compress_batch(self, codec, src2d, table_log=11, max_sv=255, nthreads=0, cap=None):
        """src2d: (nBlocks, blockSize) uint8.  Returns (seconds, results u64[n], dst (n, cap) u8)."""
        src2d = np.ascontiguousarray(src2d, dtype=np.uint8)
        n, bs = src2d.shape
        cap = (fse_compress_bound(bs) if codec == 0 else huf_compress_bound(bs)) if cap is None else cap
        dst = np.zeros((n, cap), dtype=np.uint8)
        res = np.zeros(n, dtype=np.uint64)
        t = self._call(self.N["compress_batch"], C.c_double, C.c_int(codec), src2d.ctypes.data_as(self.vp), self.sz(bs), self.sz(bs),
                       dst.ctypes.data_as(self.vp), self.sz(cap), self.sz(cap), res.ctypes.data_as(self.vp), self.sz(n),
                       C.c_uint(max_sv), C.c_uint(table_log), C.c_int(nthreads))
        return float(t), res, dst

    def decompress_batch(self, codec, csrc2d, csizes, dst_size, nthreads=0):
        csrc2d = np.ascontiguousarray(csrc2d, dtype=np.uint8)
        csizes = np.ascontiguousarray(csizes, dtype=np.uint64)
        n, cs = csrc2d.shape
        dst = np.zeros((n, dst_size), dtype=np.uint8)
        res = np.zeros(n, dtype=np.uint64)
        t = self._call(self.N["decompress_batch"], C.c_double, C.c_int(codec), csrc2d.ctypes.data_as(self.vp), self.sz(cs),
                       csizes.ctypes.data_as(self.vp), dst.ctypes.data_as(self.vp), self.sz(dst_size), self.sz(dst_size),
                       res.ctypes.data_as(self.vp), self.sz(n), C.c_int(nthreads))
        return float(t), res, dst


class Oracle(_Base):
    N = {"hist_count": "orc_hist_count", "optimal_tablelog": "orc_fse_optimal_tablelog", "normalize": "orc_fse_normalize_count",
         "write_ncount": "orc_fse_write_ncount", "read_ncount": "orc_fse_read_ncount", "build_ctable": "orc_fse_build_ctable",
         "build_ctable_raw": "orc_fse_build_ctable_raw", "build_dtable": "orc_fse_build_dtable", "build_dtable_raw": "orc_fse_build_dtable_raw",
         "compress_using_ctable": "orc_fse_compress_using_ctable", "decompress_using_dtable": "orc_fse_decompress_using_dtable",
         "fse_compress2": "orc_fse_compress2", "fse_decompress": "orc_fse_decompress",
         "huf_build_ctable": "orc_huf_build_ctable", "huf_write_ctable": "orc_huf_write_ctable", "huf_read_dtable_x1": "orc_huf_read_dtable_x1",
         "huf_compress1x": "orc_huf_compress1x_using_ctable", "huf_compress4x": "orc_huf_compress4x_using_ctable",
         "huf_decompress4x1": "orc_huf_decompress4x1_using_dtable", "huf_decompress1x1": "orc_huf_decompress1x1_using_dtable",
         "huf_compress2": "orc_huf_compress2", "huf_decompress": "orc_huf_decompress",
         "compress_batch": "orc_compress_batch", "decompress_batch": "orc_decompress_batch",
         "bench_roundtrip": "orc_bench_roundtrip", "stream_bandwidth": "orc_stream_bandwidth"}
    kind = "port"

    def __init__(self):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "liboracle.so"))

    def _f(self, name):
        return getattr(self.lib, name)

    # workload + checksum live only in the restatement
    def probagen_table(self, p):
        t = np.zeros(4096, dtype=np.uint8)
        self.lib.orc_probagen_table.restype = None
        self.lib.orc_probagen_table(t.ctypes.data_as(C.c_void_p), C.c_double(p))
        return t

    def probagen_batch(self, p_percent, n_blocks, block_size=32768, first_seed=1):
        """block b = generate(block_size, p, seed = first_seed + b)  """
        t = self.probagen_table(p_percent / 100.0)
        out = np.zeros((n_blocks, block_size), dtype=np.uint8)
        self.lib.orc_probagen_batch.restype = None
        self.lib.orc_probagen_batch(out.ctypes.data_as(C.c_void_p), C.c_size_t(block_size), C.c_size_t(block_size), C.c_size_t(n_blocks),
                                    t.ctypes.data_as(C.c_void_p), C.c_uint32(first_seed))
        return out

    # ---- .fse frames (only the Oracle class restates them; the Ref class has the CLI binary, see ref_cli_*) -------
    def frame_compress(self, src, block_size_id=5, codec=0):
        src, ps = _u8(src)
        cap = int(self._call("orc_frame_compress_bound", self.sz, self.sz(src.size), C.c_uint(block_size_id)))
        out = np.zeros(cap + 8, dtype=np.uint8)
        r = int(self._call("orc_frame_compress", self.sz, out.ctypes.data_as(self.vp), self.sz(cap), ps, self.sz(src.size),
                           C.c_uint(block_size_id), C.c_int(codec)))
        return r, out

    def frame_decompress(self, frame, cap):
        frame, pf = _u8(frame)
        out = np.zeros(max(cap, 1) + 8, dtype=np.uint8)
        r = int(self._call("orc_frame_decompress", self.sz, out.ctypes.data_as(self.vp), self.sz(cap), pf, self.sz(frame.size)))
        return r, out

    def xxh32(self, data, seed=0):
        data, p = _u8(data)
        self.lib.orc_xxh32.restype = C.c_uint32
        return int(self.lib.orc_xxh32(p, C.c_size_t(data.size), C.c_uint32(seed)))

    def xxh64(self, data, seed=0):
        data, p = _u8(data)
        self.lib.orc_xxh64.restype = C.c_uint64
        return int(self.lib.orc_xxh64(p, C.c_size_t(data.size), C.c_uint64(seed)))


class Ref(_Base):
    N = {"hist_count": "HIST_count", "optimal_tablelog": "FSE_optimalTableLog_internal", "normalize": "FSE_normalizeCount",
         "write_ncount": "FSE_writeNCount", "read_ncount": "FSE_readNCount", "build_ctable": "FSE_buildCTable",
         "build_ctable_raw": "FSE_buildCTable_raw", "build_dtable": "FSE_buildDTable", "build_dtable_raw": "FSE_buildDTable_raw",
         "compress_using_ctable": "FSE_compress_usingCTable", "decompress_using_dtable": "FSE_decompress_usingDTable",
         "fse_compress2": "FSE_compress2", "fse_decompress": "FSE_decompress",
         "huf_build_ctable": "HUF_buildCTable", "huf_write_ctable": "HUF_writeCTable", "huf_read_dtable_x1": "HUF_readDTableX1",
         "huf_compress1x": "HUF_compress1X_usingCTable", "huf_compress4x": "HUF_compress4X_usingCTable",
         "huf_decompress4x1": "HUF_decompress4X1_usingDTable", "huf_decompress1x1": "HUF_decompress1X1_usingDTable",
         "huf_compress2": "HUF_compress2", "huf_decompress": "HUF_decompress", "huf_decompress4x1_oneshot": "HUF_decompress4X1",
         "compress_batch": "ref_compress_batch", "decompress_batch": "ref_decompress_batch",
         "bench_roundtrip": "ref_bench_roundtrip", "stream_bandwidth": "ref_stream_bandwidth"}
    kind = "reference"

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libfse_ref.so"))

    def __init__(self):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", "libfse_ref.so"))

    def _f(self, name):
        return getattr(self.lib, name)

    # double-symbol (X2) decoding tables exist only in the reference (our restatement covers the X1 decoder)
    def huf_read_dtable_x2(self, src, max_table_log=12):
        src, ps = _u8(src)
        dt = np.zeros(1 + (1 << 12), dtype=np.uint32)
        dt[0] = max_table_log * 0x01000001
        r = self._call("HUF_readDTableX2", self.sz, dt.ctypes.data_as(self.vp), ps, self.sz(src.size))
        return int(r), dt

    def huf_decompress4x_using_dtable(self, csrc, dt, dst_size):
        """HUF_decompress4X_usingDTable: dispatches on the table type (lib/huf_decompress.c:980-997)"""
        csrc, ps = _u8(csrc)
        dt = np.ascontiguousarray(dt, dtype=np.uint32)
        out = np.zeros(max(dst_size, 1) + 16, dtype=np.uint8)
        out[dst_size:] = 0xA5
        r = self._call("HUF_decompress4X_usingDTable", self.sz, out.ctypes.data_as(self.vp), self.sz(dst_size), ps, self.sz(csrc.size), dt.ctypes.data_as(self.vp))
        assert (out[dst_size:] == 0xA5).all()
        return int(r), out[:dst_size]

    def huf_decompress1x_using_dtable(self, csrc, dt, dst_size):
        """HUF_decompress1X_usingDTable: one stream, dispatches on the table type (lib/huf_decompress.c:961-975)"""
        csrc, ps = _u8(csrc)
        dt = np.ascontiguousarray(dt, dtype=np.uint32)
        out = np.zeros(max(dst_size, 1) + 16, dtype=np.uint8)
        out[dst_size:] = 0xA5
        r = self._call("HUF_decompress1X_usingDTable", self.sz, out.ctypes.data_as(self.vp), self.sz(dst_size), ps, self.sz(csrc.size), dt.ctypes.data_as(self.vp))
        assert (out[dst_size:] == 0xA5).all()
        return int(r), out[:dst_size]

    # FSE for 16-bit symbols (lib/fseU16.c): only the compiled reference has it (a side path: SURVEY 8(f) rank 4)
    def fse_count_u16(self, src, max_sv=286):
        src = np.ascontiguousarray(src, dtype=np.uint16)
        count = np.zeros(max(max_sv, 286) + 1, dtype=np.uint32)
        msv = C.c_uint(max_sv)
        r = self._call("FSE_countU16", self.sz, count.ctypes.data_as(self.vp), C.byref(msv), src.ctypes.data_as(self.vp), self.sz(src.size))
        return int(r), count, int(msv.value)

    def fse_compress_u16(self, src, max_sv=0, table_log=0, cap=None):
        src = np.ascontiguousarray(src, dtype=np.uint16)
        cap = fse_compress_bound(2 * src.size) if cap is None else cap
        out = np.zeros(max(cap, 1) + 16, dtype=np.uint8)
        r = self._call("FSE_compressU16", self.sz, out.ctypes.data_as(self.vp), self.sz(cap), src.ctypes.data_as(self.vp), self.sz(src.size),
                       C.c_uint(max_sv), C.c_uint(table_log))
        return int(r), out[:cap]

    def fse_build_ctable_u16(self, norm, max_sv, table_log):
        """FSE_buildCTableU16 (lib/fseU16.c:103, fse_compress.c:66-169 compiled for 16-bit symbols): table logs up to 13"""
        norm = np.ascontiguousarray(norm, dtype=np.int16)
        ct = np.zeros(1 + (1 << (max(table_log, 1) - 1)) + (max_sv + 1) * 2 + 8, dtype=np.uint32)
        r = self._call("FSE_buildCTableU16", self.sz, ct.ctypes.data_as(self.vp), norm.ctypes.data_as(self.vp), C.c_uint(max_sv), C.c_uint(table_log))
        return int(r), ct

    def fse_compress_u16_using_ctable(self, src, ct, cap):
        """FSE_compressU16_usingCTable (lib/fseU16.c:150-200)"""
        src = np.ascontiguousarray(src, dtype=np.uint16)
        out = np.zeros(max(cap, 1) + 16, dtype=np.uint8)
        r = self._call("FSE_compressU16_usingCTable", self.sz, out.ctypes.data_as(self.vp), self.sz(cap), src.ctypes.data_as(self.vp), self.sz(src.size),
                       ct.ctypes.data_as(self.vp))
        return int(r), out[:cap]

    def fse_decompress_u16(self, csrc, cap):
        csrc, ps = _u8(csrc)
        out = np.zeros(max(cap, 1) + 8, dtype=np.uint16)
        out[cap:] = 0xA5A5
        r = self._call("FSE_decompressU16", self.sz, out.ctypes.data_as(self.vp), self.sz(cap), ps, self.sz(csrc.size))
        assert (out[cap:] == 0xA5A5).all()
        return int(r), out[:cap]

    def max_threads(self):
        self.lib.ref_max_threads.restype = C.c_int
        return int(self.lib.ref_max_threads())


class Checker(Oracle):
    """What the `-m gpu` parity tests and bench.py compare the kernels with: the COMPILED REFERENCE itself whenever
    oracle/_ref/libfse_ref.so is present (it travels to the GPU box with the snapshot), so kernel -> reference is one hop;
    our restatement otherwise.  The workload generator, the .fse frame and the checksums exist only in the restatement
    (the reference library has no such entry points; the frame restatement is pinned against the reference tool on the CPU)."""

    def __init__(self):
        super().__init__()
        self.ref = Ref() if Ref.available() else None
        if self.ref is not None:
            self.N = Ref.N
            self.kind = "reference"

    def _f(self, name):
        if self.ref is not None and not name.startswith("orc_"):
            return getattr(self.ref.lib, name)
        return getattr(self.lib, name)
