"""Host-side mirror of the reference block API (lib/fse.h, lib/huf.h, lib/hist.h) over libfsehip.so.

Batched functions take / return CUDA (HIP) ``torch.uint8`` tensors and call the C ABI with raw device
pointers on torch's current stream.  ``results`` tensors are ``torch.int64`` views of the reference's
``size_t`` return values: a negative value ``-c`` is the error code ``c`` of lib/error_public.h:45-56.
Single-block functions take numpy arrays (host pointers) and have the reference's exact signatures.
"""
import ctypes as C
import numbers

import numpy as np
import torch

from . import _lib

SZ = C.c_size_t
VP = C.c_void_p

ERROR_NAMES = {1: "GENERIC", 2: "dstSize_tooSmall", 3: "srcSize_wrong", 4: "corruption_detected", 5: "tableLog_tooLarge",
               6: "maxSymbolValue_tooLarge", 7: "maxSymbolValue_tooSmall", 8: "workSpace_tooSmall"}


def fse_compress_bound(n):      # lib/fse.h:290-292
    return 512 + n + (n >> 7) + 4 + 8


def huf_compress_bound(n):      # lib/huf.h:131-133
    return 129 + n + (n >> 8) + 8


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (what, rc))


def _stream():
    return VP(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return VP(t.data_ptr()) if t is not None else VP(0)


def _sizes_arg(sizes, like=None):
    """sizes: None | integer (python or numpy) | integer cuda tensor -> (device pointer or NULL, uniform, keepalive)"""
    if sizes is None or isinstance(sizes, numbers.Integral):
        return VP(0), SZ(int(sizes or 0)), None
    if not isinstance(sizes, torch.Tensor) or not sizes.is_cuda:
        raise TypeError("per-block sizes must be an integer or a CUDA integer tensor (the C ABI takes a device pointer)")
    if like is not None and sizes.device != like.device:
        raise ValueError("sizes live on %s, the blocks on %s" % (sizes.device, like.device))
    t = sizes.to(torch.int64).contiguous()
    return VP(t.data_ptr()), SZ(0), t


def _blocks(t, what):
    """a batch of byte blocks handed to the C ABI as (base pointer, row stride): uint8, on the GPU, rows contiguous"""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.uint8 or t.dim() != 2:
        raise TypeError("%s must be a 2-D CUDA uint8 tensor" % what)
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError("%s: bytes of a block must be contiguous (stride(1) == 1); call .contiguous()" % what)
    return t


class _Guarded:
    """Destination of a batched call in guard mode (FseHip.guard > 0): every block's slot is followed by `guard` bytes of 0xA5 -- the
    row stride is capacity + guard, the capacity handed to the library stays `cap` -- and `check()` asserts after the call that no
    kernel touched them: the device-side form of the reference fuzzers' guard byte (programs/fuzzer.c:217-230,
    programs/fuzzerHuff0.c:198-212; SURVEY 8(b): nothing is written beyond dst + dstCapacity)."""
    FILL = 0xA5

    def __init__(self, n, cap, guard, device, dtype=torch.uint8, zero=False):
        self.cap = cap
        self.full = torch.empty((n, max(cap, 1) + guard), dtype=dtype, device=device)
        self.fill = self.FILL if dtype == torch.uint8 else 0xA5A5 - 0x10000
        self.full.fill_(self.fill)
        self.view = self.full[:, :max(cap, 1)]
        if zero and cap > 0:
            self.view.zero_()

    def check(self, what, sizes=None):
        """sizes: per-block capacities (tensor) when they differ from block to block (Huff0 decoders: the exact regenerated size)"""
        if sizes is None or isinstance(sizes, numbers.Integral):
            cap = self.cap if sizes is None else int(sizes)
            touched = self.full[:, cap:] != self.fill
            caps = None
        else:
            caps = sizes.to(torch.int64).to(self.full.device).clamp(max=self.full.shape[1])
            cols = torch.arange(self.full.shape[1], device=self.full.device)
            touched = (self.full != self.fill) & (cols[None, :] >= caps[:, None])
            cap = 0
        bad = touched.any(dim=1)
        if bool(bad.any().item()):
            rows = torch.nonzero(bad).flatten()[:8].tolist()
            b = rows[0]
            where = (torch.nonzero(touched[b]).flatten()[:8] + cap).tolist()
            raise AssertionError("%s wrote past its destination capacity (%s): blocks %s (block %d at byte offsets %s)"
                                 % (what, self.cap if caps is None else int(caps[b]), rows, b, where))


class FseHip:
    # > 0: the batch helpers below that allocate their own destination put `guard` sentinel bytes (symbols for the 16-bit coder)
    # behind every block's capacity and assert after the call that they are untouched (synchronises; tests only)
    guard = 0

    def _dst(self, n, cap, device, zero=False, dtype=torch.uint8):
        """(destination tensor handed to the library, guard object or None)"""
        if self.guard > 0:
            g = _Guarded(n, cap, self.guard, device, dtype, zero)
            return g.view, g
        alloc = torch.zeros if zero else torch.empty
        return alloc((n, max(cap, 1)), dtype=dtype, device=device), None

    def __init__(self):
        self.lib = _lib.load()
        L = self.lib
        for name in ("FSEHIP_HIST_count", "FSEHIP_FSE_compress_usingCTable", "FSEHIP_FSE_decompress_usingDTable",
                     "FSEHIP_FSE_compress", "FSEHIP_FSE_compress2", "FSEHIP_FSE_decompress",
                     "FSEHIP_HUF_compress1X_usingCTable", "FSEHIP_HUF_compress4X_usingCTable",
                     "FSEHIP_HUF_decompress4X_usingDTable", "FSEHIP_HUF_decompress4X1_usingDTable",
                     "FSEHIP_HUF_decompress1X_usingDTable", "FSEHIP_HUF_decompress1X1_usingDTable",
                     "FSEHIP_HUF_compress", "FSEHIP_HUF_compress2", "FSEHIP_HUF_decompress",
                     "FSEHIP_FSE_compress_batch_workspaceSize", "FSEHIP_FSE_decompress_batch_workspaceSize",
                     "FSEHIP_HUF_compress_batch_workspaceSize", "FSEHIP_HUF_decompress_batch_workspaceSize",
                     "FSEHIP_frame_compressBound", "FSEHIP_frame_compress", "FSEHIP_frame_decompress", "FSEHIP_frame_compress_batch", "FSEHIP_frame_decompress_batch",
                     "FSEHIP_FSE_countU16", "FSEHIP_FSE_compressU16", "FSEHIP_FSE_decompressU16",
                     "FSEHIP_FSE_compressU16_batch_workspaceSize", "FSEHIP_FSE_decompressU16_batch_workspaceSize",
                     "FSEHIP_FSE_buildCTable_batch_workspaceSize", "FSEHIP_FSE_buildDTable_batch_workspaceSize",
                     "FSEHIP_HUF_buildCTable_batch_workspaceSize", "FSEHIP_HUF_readDTableX1_batch_workspaceSize",
                     "FSEHIP_compact_batch_workspaceSize", "FSEHIP_compact_batch_bound", "FSEHIP_HUF_compress1X",
                     "FSEHIP_HUF_decompress4X1", "FSEHIP_HUF_decompress1X1"):
            if hasattr(L, name):
                getattr(L, name).restype = SZ
        L.FSEHIP_getErrorName.restype = C.c_char_p
        L.FSEHIP_versionString.restype = C.c_char_p

    # ------------------------------------------------------------------ info
    def device_info(self):
        class Info(C.Structure):
            _fields_ = [("deviceOrdinal", C.c_int), ("computeUnits", C.c_int), ("ldsBytesPerCU", C.c_int),
                        ("wavefrontSize", C.c_int), ("archName", C.c_char * 64)]
        info = Info()
        _check(self.lib.FSEHIP_deviceInfo(C.byref(info)), "deviceInfo")
        return {"device": info.deviceOrdinal, "cus": info.computeUnits, "lds_per_cu": info.ldsBytesPerCU,
                "wave": info.wavefrontSize, "arch": info.archName.decode()}

    # ------------------------------------------------------------------ workload
    def probagen_table(self, p):
        t = np.zeros(4096, dtype=np.uint8)
        self.lib.FSEHIP_probagen_table(t.ctypes.data_as(VP), C.c_double(p))
        return t

    def probagen_batch(self, p_percent, n_blocks, block_size=32768, first_seed=1, out=None, device="cuda", seed_step=1):
        """block b = probagen(block_size, p, seed=first_seed + b*seed_step)  (programs/probaGenerator.c)"""
        table = self.probagen_table(p_percent / 100.0)
        if out is None:
            out = torch.empty((n_blocks, block_size), dtype=torch.uint8, device=device)
        _check(self.lib.FSEHIP_probagen_batch_ex(_ptr(out), SZ(out.stride(0)), SZ(block_size), SZ(n_blocks),
                                                 table.ctypes.data_as(VP), C.c_uint32(first_seed & 0xFFFFFFFF), C.c_uint32(seed_step), _stream()), "probagen_batch")
        return out

    def probagen_mixed(self, probas, n_blocks, block_size=32768, first_block=0, device="cuda", out=None):
        """BASELINE config 5 corpus: global block g (= first_block + row) is drawn from distribution probas[g mod len(probas)]
        with seed g + 1 -- one strided generator call per distribution (into `out` if given)."""
        if out is None:
            out = torch.empty((n_blocks, block_size), dtype=torch.uint8, device=device)
        k = len(probas)
        for j in range(k):
            r0 = (j - first_block) % k                       # first row whose global index is congruent to j
            rows = out[r0::k]
            if rows.shape[0]:
                self.probagen_batch(probas[j], rows.shape[0], block_size, first_seed=first_block + r0 + 1, out=rows, seed_step=k)
        return out

    # ------------------------------------------------------------------ a1
    def hist_count_batch(self, src, sizes=None, max_symbol_values=None):
        n = _blocks(src, "src").shape[0]
        counts = torch.zeros((n, 256), dtype=torch.int32, device=src.device)
        msv = (torch.full((n,), 255, dtype=torch.int32, device=src.device) if max_symbol_values is None
               else max_symbol_values.to(torch.int32).contiguous().clone())
        res = torch.zeros(n, dtype=torch.int64, device=src.device)
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        _check(self.lib.FSEHIP_HIST_count_batch(_ptr(counts), _ptr(msv), _ptr(res), _ptr(src), SZ(src.stride(0)), ps, uni,
                                                SZ(n), _stream()), "HIST_count_batch")
        return counts, msv, res

    # ------------------------------------------------------------------ a2 / a3
    def fse_compress_using_ctable_batch(self, src, ctables, max_table_log=12, sizes=None, dst_capacity=None, shared_table=False, dst=None, results=None):
        n = _blocks(src, "src").shape[0]
        cap = fse_compress_bound(src.shape[1]) if dst_capacity is None else dst_capacity
        g = None
        if dst is None:
            dst, g = self._dst(n, cap, src.device, zero=True)
        res = torch.zeros(n, dtype=torch.int64, device=src.device) if results is None else results
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        stride = 0 if shared_table else ctables.stride(0)
        _check(self.lib.FSEHIP_FSE_compress_usingCTable_batch(_ptr(dst), SZ(dst.stride(0)), SZ(cap), _ptr(res), _ptr(src), SZ(src.stride(0)),
                                                              ps, uni, _ptr(ctables), SZ(stride), C.c_uint(max_table_log), SZ(n), _stream()),
               "FSE_compress_usingCTable_batch")
        if g:
            g.check("FSE_compress_usingCTable_batch")
        return dst, res

    def fse_decompress_using_dtable_batch(self, csrc, csizes, dtables, dst_capacity, max_table_log=12, shared_table=False, dst=None, results=None):
        n = _blocks(csrc, "csrc").shape[0]
        g = None
        if dst is None:
            dst, g = self._dst(n, dst_capacity, csrc.device, zero=True)
        res = torch.zeros(n, dtype=torch.int64, device=csrc.device) if results is None else results
        ps, uni, keep = _sizes_arg(csizes, csrc)
        stride = 0 if shared_table else dtables.stride(0)
        _check(self.lib.FSEHIP_FSE_decompress_usingDTable_batch(_ptr(dst), SZ(dst.stride(0)), SZ(dst_capacity), _ptr(res), _ptr(csrc),
                                                                SZ(csrc.stride(0)), ps, uni, _ptr(dtables), SZ(stride),
                                                                C.c_uint(max_table_log), SZ(n), _stream()),
               "FSE_decompress_usingDTable_batch")
        if g:
            g.check("FSE_decompress_usingDTable_batch")
        return dst, res

    # ------------------------------------------------------------------ one-shot FSE over a batch
    def fse_workspace(self, n_blocks, table_log=11, decompress=False, device="cuda"):
        fn = self.lib.FSEHIP_FSE_decompress_batch_workspaceSize if decompress else self.lib.FSEHIP_FSE_compress_batch_workspaceSize
        nbytes = int(fn(SZ(n_blocks), C.c_uint(table_log)))
        return torch.empty(nbytes, dtype=torch.uint8, device=device)

    def fse_compress_batch(self, src, table_log=11, max_symbol_value=255, sizes=None, dst=None, dst_capacity=None, results=None, workspace=None):
        n = _blocks(src, "src").shape[0]
        cap = (fse_compress_bound(src.shape[1]) if dst_capacity is None else dst_capacity)
        g = None
        if dst is None:
            dst, g = self._dst(n, cap, src.device)
        if results is None:
            results = torch.empty(n, dtype=torch.int64, device=src.device)
        if workspace is None:
            workspace = self.fse_workspace(n, table_log, False, src.device)
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        _check(self.lib.FSEHIP_FSE_compress_batch(_ptr(dst), SZ(dst.stride(0)), SZ(cap), _ptr(results), _ptr(src), SZ(src.stride(0)), ps, uni,
                                                  C.c_uint(max_symbol_value), C.c_uint(table_log), SZ(n), _ptr(workspace),
                                                  SZ(workspace.numel()), _stream()), "FSE_compress_batch")
        if g:
            g.check("FSE_compress_batch")
        return dst, results

    def fse_decompress_batch(self, csrc, csizes, dst_capacity, max_log=12, dst=None, results=None, workspace=None):
        n = _blocks(csrc, "csrc").shape[0]
        g = None
        if dst is None:
            dst, g = self._dst(n, dst_capacity, csrc.device)
        if results is None:
            results = torch.empty(n, dtype=torch.int64, device=csrc.device)
        if workspace is None:
            workspace = self.fse_workspace(n, max_log, True, csrc.device)
        ps, uni, keep = _sizes_arg(csizes, csrc)
        _check(self.lib.FSEHIP_FSE_decompress_batch(_ptr(dst), SZ(dst.stride(0)), SZ(dst_capacity), _ptr(results), _ptr(csrc), SZ(csrc.stride(0)),
                                                    ps, uni, C.c_uint(max_log), SZ(n), _ptr(workspace), SZ(workspace.numel()), _stream()),
               "FSE_decompress_batch")
        if g:
            g.check("FSE_decompress_batch")
        return dst, results

    # ------------------------------------------------------------------ tables built on the device (g1-g3)
    def fse_build_ctable_batch(self, src, table_log=11, max_symbol_value=255, sizes=None, header_capacity=512, ctables=None):
        """FSE_buildCTable_batch: (ctables (n, FSE_CTABLE_SIZE_U32(max(table_log, 9), 255)) int32, headers (n, header_capacity) uint8, results)"""
        n = _blocks(src, "src").shape[0]
        tl = min(max(table_log or 11, 9), 12)
        ctw = 1 + (1 << (tl - 1)) + 512
        ct = torch.zeros((n, ctw), dtype=torch.int32, device=src.device) if ctables is None else ctables
        hdr, g = self._dst(n, header_capacity, src.device, zero=True)
        res = torch.zeros(n, dtype=torch.int64, device=src.device)
        ws = torch.empty(int(self.lib.FSEHIP_FSE_buildCTable_batch_workspaceSize(SZ(n))), dtype=torch.uint8, device=src.device)
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        _check(self.lib.FSEHIP_FSE_buildCTable_batch(_ptr(ct), SZ(ct.stride(0)), _ptr(hdr), SZ(hdr.stride(0)), SZ(header_capacity), _ptr(res), _ptr(src),
                                                     SZ(src.stride(0)), ps, uni, C.c_uint(max_symbol_value), C.c_uint(table_log), SZ(n), _ptr(ws),
                                                     SZ(ws.numel()), _stream()), "FSE_buildCTable_batch")
        if g:
            g.check("FSE_buildCTable_batch")
        return ct, hdr, res

    def fse_build_dtable_batch(self, headers, header_sizes, max_log=12):
        """FSE_buildDTable_batch: (dtables (n, FSE_DTABLE_SIZE_U32(max_log)) int32 in the reference layout, results = header bytes or error)"""
        n = _blocks(headers, "headers").shape[0]
        dt = torch.zeros((n, 1 + (1 << max_log)), dtype=torch.int32, device=headers.device)
        res = torch.zeros(n, dtype=torch.int64, device=headers.device)
        ws = torch.empty(int(self.lib.FSEHIP_FSE_buildDTable_batch_workspaceSize(SZ(n), C.c_uint(max_log))), dtype=torch.uint8, device=headers.device)
        ps, uni, keep = _sizes_arg(header_sizes, headers)
        _check(self.lib.FSEHIP_FSE_buildDTable_batch(_ptr(dt), SZ(dt.stride(0)), _ptr(res), _ptr(headers), SZ(headers.stride(0)), ps, uni,
                                                     C.c_uint(max_log), SZ(n), _ptr(ws), SZ(ws.numel()), _stream()), "FSE_buildDTable_batch")
        return dt, res

    # ------------------------------------------------------------------ table glue, step by step (lib/fse.h:137-156, 222-229)
    def fse_normalize_count_batch(self, counts, totals, max_symbol_values, table_log):
        """FSE_normalizeCount per row of `counts` (n, 256) int32: (norms (n, 256) int16, results = tableLog or error)"""
        n = counts.shape[0]
        norms = torch.zeros((n, 256), dtype=torch.int16, device=counts.device)
        res = torch.zeros(n, dtype=torch.int64, device=counts.device)
        _check(self.lib.FSEHIP_FSE_normalizeCount_batch(_ptr(norms), SZ(256), C.c_uint(table_log), _ptr(counts), SZ(counts.stride(0)), _ptr(totals),
                                                        _ptr(max_symbol_values), SZ(n), _ptr(res), _stream()), "FSE_normalizeCount_batch")
        return norms, res

    def fse_write_ncount_batch(self, norms, max_symbol_values, table_log, capacity=512, stride=None):
        """FSE_writeNCount per row of `norms` (n, 256) int16: (headers (n, stride) uint8 pre-filled with 0xA5, results = header bytes or error)"""
        n = norms.shape[0]
        hdr = torch.full((n, stride or max(capacity, 1)), 0xA5, dtype=torch.uint8, device=norms.device)
        res = torch.zeros(n, dtype=torch.int64, device=norms.device)
        _check(self.lib.FSEHIP_FSE_writeNCount_batch(_ptr(hdr), SZ(hdr.stride(0)), SZ(capacity), _ptr(norms), SZ(norms.stride(0)), _ptr(max_symbol_values),
                                                     C.c_uint(table_log), SZ(n), _ptr(res), _stream()), "FSE_writeNCount_batch")
        return hdr, res

    def fse_read_ncount_batch(self, headers, header_sizes, max_symbol_values):
        """FSE_readNCount per row: (norms (n, 256) int16, maxSymbolValues out, tableLogs, results = bytes read or error)"""
        n = _blocks(headers, "headers").shape[0]
        norms = torch.zeros((n, 256), dtype=torch.int16, device=headers.device)
        msv = max_symbol_values.clone()
        tls = torch.zeros(n, dtype=torch.int32, device=headers.device)
        res = torch.zeros(n, dtype=torch.int64, device=headers.device)
        ps, uni, keep = _sizes_arg(header_sizes, headers)
        _check(self.lib.FSEHIP_FSE_readNCount_batch(_ptr(norms), SZ(256), _ptr(msv), _ptr(tls), _ptr(headers), SZ(headers.stride(0)), ps, uni, SZ(n), _ptr(res),
                                                    _stream()), "FSE_readNCount_batch")
        return norms, msv, tls, res

    def fse_build_ctable_from_norm_batch(self, norms, max_symbol_values, table_log):
        """FSE_buildCTable per row of `norms` (n, 256) int16: (ctables (n, FSE_CTABLE_SIZE_U32(table_log, 255)) int32, results = 0 or error)"""
        n = norms.shape[0]
        ct = torch.zeros((n, 1 + (1 << max(min(table_log, 12) - 1, 0)) + 512), dtype=torch.int32, device=norms.device)
        res = torch.zeros(n, dtype=torch.int64, device=norms.device)
        _check(self.lib.FSEHIP_FSE_buildCTable_fromNorm_batch(_ptr(ct), SZ(ct.stride(0)), _ptr(norms), SZ(norms.stride(0)), _ptr(max_symbol_values),
                                                              C.c_uint(table_log), SZ(n), _ptr(res), _stream()), "FSE_buildCTable_fromNorm_batch")
        return ct, res

    def fse_build_dtable_from_norm_batch(self, norms, max_symbol_values, table_log):
        """FSE_buildDTable per row of `norms` (n, 256) int16: (dtables (n, FSE_DTABLE_SIZE_U32(table_log)) int32, results = 0 or error)"""
        n = norms.shape[0]
        dt = torch.zeros((n, 1 + (1 << max(min(table_log, 12), 1))), dtype=torch.int32, device=norms.device)
        res = torch.zeros(n, dtype=torch.int64, device=norms.device)
        self.lib.FSEHIP_FSE_buildDTable_fromNorm_batch_workspaceSize.restype = SZ
        ws = torch.empty(int(self.lib.FSEHIP_FSE_buildDTable_fromNorm_batch_workspaceSize(SZ(n), C.c_uint(table_log))), dtype=torch.uint8, device=norms.device)
        _check(self.lib.FSEHIP_FSE_buildDTable_fromNorm_batch(_ptr(dt), SZ(dt.stride(0)), _ptr(norms), SZ(norms.stride(0)), _ptr(max_symbol_values),
                                                              C.c_uint(table_log), SZ(n), _ptr(res), _ptr(ws), SZ(ws.numel()), _stream()), "FSE_buildDTable_fromNorm_batch")
        return dt, res

    # ------------------------------------------------------------------ packed (variable-length) batches
    def compact_batch(self, slots, results, src, sizes=None, packed=None, offsets=None):
        """FSEHIP_compact_batch: (packed uint8 (capacity,), offsets int64 (n + 1,)); offsets[n] = the packed size"""
        n = _blocks(slots, "slots").shape[0]
        _blocks(src, "src")
        if packed is None:
            packed = torch.empty(max(n * src.shape[1], 1), dtype=torch.uint8, device=src.device)
        if offsets is None:
            offsets = torch.empty(n + 1, dtype=torch.int64, device=src.device)
        self.lib.FSEHIP_compact_batch_workspaceSize.restype = SZ
        ws = torch.empty(int(self.lib.FSEHIP_compact_batch_workspaceSize(SZ(n))), dtype=torch.uint8, device=src.device)
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        _check(self.lib.FSEHIP_compact_batch(_ptr(packed), SZ(packed.numel()), _ptr(offsets), _ptr(slots), SZ(slots.stride(0)), _ptr(results),
                                             _ptr(src), SZ(src.stride(0)), ps, uni, SZ(n), _ptr(ws), SZ(ws.numel()), _stream()), "compact_batch")
        return packed, offsets

    def fse_decompress_packed_batch(self, packed, offsets, orig_sizes, dst_capacity, max_log=12, dst=None, results=None, workspace=None):
        n = offsets.numel() - 1
        g = None
        if dst is None:
            dst, g = self._dst(n, dst_capacity, packed.device)
        if results is None:
            results = torch.empty(n, dtype=torch.int64, device=packed.device)
        if workspace is None:
            workspace = self.fse_workspace(n, max_log, True, packed.device)
        po, uo, keep = _sizes_arg(orig_sizes, packed)
        _check(self.lib.FSEHIP_FSE_decompress_packed_batch(_ptr(dst), SZ(dst.stride(0)), SZ(dst_capacity), _ptr(results), _ptr(packed), _ptr(offsets), po, uo,
                                                           C.c_uint(max_log), SZ(n), _ptr(workspace), SZ(workspace.numel()), _stream()), "FSE_decompress_packed_batch")
        if g:
            g.check("FSE_decompress_packed_batch")
        return dst, results

    # ------------------------------------------------------------------ layer 1 (host pointers, reference signatures)
    def _single(self, fname, cap, src, *extra):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        out = np.zeros(max(cap, 1) + 16, dtype=np.uint8)
        out[cap:] = 0xA5
        r = int(getattr(self.lib, fname)(out.ctypes.data_as(VP), SZ(cap), src.ctypes.data_as(VP), SZ(src.size), *extra))
        assert (out[cap:] == 0xA5).all(), "%s wrote past dstCapacity" % fname
        return r, out[:cap]

    def hist_count(self, src, max_sv=255):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        count = np.zeros(256, dtype=np.uint32)
        msv = C.c_uint(max_sv)
        r = int(self.lib.FSEHIP_HIST_count(count.ctypes.data_as(VP), C.byref(msv), src.ctypes.data_as(VP), SZ(src.size)))
        return r, int(msv.value), count

    def fse_compress_using_ctable(self, src, ct, cap=None):
        ct = np.ascontiguousarray(ct, dtype=np.uint32)
        return self._single("FSEHIP_FSE_compress_usingCTable", fse_compress_bound(len(src)) if cap is None else cap, src, ct.ctypes.data_as(VP))

    def fse_decompress_using_dtable(self, csrc, dt, cap):
        dt = np.ascontiguousarray(dt, dtype=np.uint32)
        return self._single("FSEHIP_FSE_decompress_usingDTable", cap, csrc, dt.ctypes.data_as(VP))

    # the table glue on host pointers (lib/fse.h:119-163, :222-241): same argument order as the reference, numpy in / out
    def fse_optimal_tablelog(self, max_tl, src_size, max_sv):
        self.lib.FSEHIP_FSE_optimalTableLog.restype = C.c_uint
        return int(self.lib.FSEHIP_FSE_optimalTableLog(C.c_uint(max_tl), SZ(src_size), C.c_uint(max_sv)))

    def fse_ncount_write_bound(self, max_sv, table_log):
        self.lib.FSEHIP_FSE_NCountWriteBound.restype = SZ
        return int(self.lib.FSEHIP_FSE_NCountWriteBound(C.c_uint(max_sv), C.c_uint(table_log)))

    def fse_normalize_count(self, table_log, count, total, max_sv):
        count = np.ascontiguousarray(count, dtype=np.uint32)
        norm = np.full(max(256, max_sv + 1) + 4, 0x5A5A, dtype=np.int16)
        self.lib.FSEHIP_FSE_normalizeCount.restype = SZ
        r = int(self.lib.FSEHIP_FSE_normalizeCount(norm.ctypes.data_as(VP), C.c_uint(table_log), count.ctypes.data_as(VP), SZ(total), C.c_uint(max_sv)))
        assert (norm[max(max_sv, 0) + 1:] == 0x5A5A).all() or max_sv > 255, "FSE_normalizeCount wrote past normalizedCounter[maxSymbolValue]"
        return r, norm[:256]

    def fse_write_ncount(self, cap, norm, max_sv, table_log):
        norm = np.ascontiguousarray(norm, dtype=np.int16)
        out = np.full(max(cap, 1) + 8, 0xA5, dtype=np.uint8)
        self.lib.FSEHIP_FSE_writeNCount.restype = SZ
        r = int(self.lib.FSEHIP_FSE_writeNCount(out.ctypes.data_as(VP), SZ(cap), norm.ctypes.data_as(VP), C.c_uint(max_sv), C.c_uint(table_log)))
        assert (out[cap:] == 0xA5).all(), "FSE_writeNCount wrote past bufferSize"
        return r, out[:cap]

    def fse_read_ncount(self, src, max_sv=255):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        norm = np.zeros(max(256, max_sv + 1), dtype=np.int16)
        msv, tl = C.c_uint(max_sv), C.c_uint(0)
        self.lib.FSEHIP_FSE_readNCount.restype = SZ
        r = int(self.lib.FSEHIP_FSE_readNCount(norm.ctypes.data_as(VP), C.byref(msv), C.byref(tl), src.ctypes.data_as(VP), SZ(src.size)))
        return r, int(msv.value), int(tl.value), norm

    def fse_build_ctable(self, norm, max_sv, table_log, wksp_bytes=None):
        norm = np.ascontiguousarray(norm, dtype=np.int16)
        words = 1 + (1 << max(min(table_log, 12) - 1, 0)) + 2 * (min(max_sv, 255) + 1)
        ct = np.zeros(words + 4, dtype=np.uint32)
        ct[words:] = 0xA5A5A5A5
        if wksp_bytes is None:
            self.lib.FSEHIP_FSE_buildCTable.restype = SZ
            r = int(self.lib.FSEHIP_FSE_buildCTable(ct.ctypes.data_as(VP), norm.ctypes.data_as(VP), C.c_uint(max_sv), C.c_uint(table_log)))
        else:
            ws = np.zeros(max(wksp_bytes, 1), dtype=np.uint8)
            self.lib.FSEHIP_FSE_buildCTable_wksp.restype = SZ
            r = int(self.lib.FSEHIP_FSE_buildCTable_wksp(ct.ctypes.data_as(VP), norm.ctypes.data_as(VP), C.c_uint(max_sv), C.c_uint(table_log), ws.ctypes.data_as(VP), SZ(wksp_bytes)))
        assert (ct[words:] == 0xA5A5A5A5).all(), "FSE_buildCTable wrote past FSE_CTABLE_SIZE_U32(tableLog, maxSymbolValue)"
        return r, ct[:words]

    def fse_build_dtable(self, norm, max_sv, table_log):
        norm = np.ascontiguousarray(norm, dtype=np.int16)
        words = 1 + (1 << max(min(table_log, 12), 0))
        dt = np.zeros(words + 4, dtype=np.uint32)
        dt[words:] = 0xA5A5A5A5
        self.lib.FSEHIP_FSE_buildDTable.restype = SZ
        r = int(self.lib.FSEHIP_FSE_buildDTable(dt.ctypes.data_as(VP), norm.ctypes.data_as(VP), C.c_uint(max_sv), C.c_uint(table_log)))
        assert (dt[words:] == 0xA5A5A5A5).all(), "FSE_buildDTable wrote past FSE_DTABLE_SIZE_U32(tableLog)"
        return r, dt[:words]

    def fse_compress2(self, src, max_sv=255, table_log=11, cap=None):
        return self._single("FSEHIP_FSE_compress2", fse_compress_bound(len(src)) if cap is None else cap, src, C.c_uint(max_sv), C.c_uint(table_log))

    def fse_decompress(self, csrc, cap):
        return self._single("FSEHIP_FSE_decompress", cap, csrc)


# ---------------------------------------------------------------------------------------------------------
#  Huff0 (lib/huf.h)
# ---------------------------------------------------------------------------------------------------------
def _huf_methods():
    def huf_workspace(self, n_blocks, decompress=False, device="cuda"):
        fn = self.lib.FSEHIP_HUF_decompress_batch_workspaceSize if decompress else self.lib.FSEHIP_HUF_compress_batch_workspaceSize
        return torch.empty(int(fn(SZ(n_blocks))), dtype=torch.uint8, device=device)

    def huf_compress_batch(self, src, table_log=11, max_symbol_value=255, sizes=None, dst=None, dst_capacity=None, results=None, workspace=None):
        n = _blocks(src, "src").shape[0]
        cap = huf_compress_bound(src.shape[1]) if dst_capacity is None else dst_capacity
        g = None
        if dst is None:
            dst, g = self._dst(n, cap, src.device)
        if results is None:
            results = torch.empty(n, dtype=torch.int64, device=src.device)
        if workspace is None:
            workspace = self.huf_workspace(n, False, src.device)
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        _check(self.lib.FSEHIP_HUF_compress_batch(_ptr(dst), SZ(dst.stride(0)), SZ(cap), _ptr(results), _ptr(src), SZ(src.stride(0)), ps, uni,
                                                  C.c_uint(max_symbol_value), C.c_uint(table_log), SZ(n), _ptr(workspace),
                                                  SZ(workspace.numel()), _stream()), "HUF_compress_batch")
        if g:
            g.check("HUF_compress_batch")
        return dst, results

    def huf_decompress_batch(self, csrc, csizes, dst_sizes, dst=None, results=None, workspace=None):
        n = _blocks(csrc, "csrc").shape[0]
        width = int(dst_sizes) if isinstance(dst_sizes, numbers.Integral) else int(dst_sizes.max().item())
        g = None
        if dst is None:
            dst, g = self._dst(n, width, csrc.device)
        if results is None:
            results = torch.empty(n, dtype=torch.int64, device=csrc.device)
        if workspace is None:
            workspace = self.huf_workspace(n, True, csrc.device)
        pc, unic, keepc = _sizes_arg(csizes, csrc)
        pd, unid, keepd = _sizes_arg(dst_sizes, csrc)
        _check(self.lib.FSEHIP_HUF_decompress_batch(_ptr(dst), SZ(dst.stride(0)), pd, unid, _ptr(results), _ptr(csrc), SZ(csrc.stride(0)), pc, unic,
                                                    SZ(n), _ptr(workspace), SZ(workspace.numel()), _stream()), "HUF_decompress_batch")
        if g:
            g.check("HUF_decompress_batch", dst_sizes)
        return dst, results

    def huf_compress4x_using_ctable_batch(self, src, ctables, sizes=None, dst_capacity=None, shared_table=False, dst=None, results=None):
        """ctables: (n, 256) int32/uint32 HUF_CElt entries (val | nbBits << 16)"""
        n = _blocks(src, "src").shape[0]
        cap = huf_compress_bound(src.shape[1]) if dst_capacity is None else dst_capacity
        g = None
        if dst is None:
            dst, g = self._dst(n, cap, src.device, zero=True)
        res = torch.zeros(n, dtype=torch.int64, device=src.device) if results is None else results
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        stride = 0 if shared_table else ctables.stride(0)
        _check(self.lib.FSEHIP_HUF_compress4X_usingCTable_batch(_ptr(dst), SZ(dst.stride(0)), SZ(cap), _ptr(res), _ptr(src), SZ(src.stride(0)),
                                                                ps, uni, _ptr(ctables), SZ(stride), SZ(n), _stream()),
               "HUF_compress4X_usingCTable_batch")
        if g:
            g.check("HUF_compress4X_usingCTable_batch")
        return dst, res

    def huf_compress1x_using_ctable_batch(self, src, ctables, sizes=None, dst_capacity=None, shared_table=False, dst=None, results=None):
        """HUF_compress1X_usingCTable over a batch (lib/huf.h:290): one stream per block"""
        n = _blocks(src, "src").shape[0]
        cap = huf_compress_bound(src.shape[1]) if dst_capacity is None else dst_capacity
        g = None
        if dst is None:
            dst, g = self._dst(n, cap, src.device, zero=True)
        res = torch.zeros(n, dtype=torch.int64, device=src.device) if results is None else results
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        stride = 0 if shared_table else ctables.stride(0)
        _check(self.lib.FSEHIP_HUF_compress1X_usingCTable_batch(_ptr(dst), SZ(dst.stride(0)), SZ(cap), _ptr(res), _ptr(src), SZ(src.stride(0)),
                                                                ps, uni, _ptr(ctables), SZ(stride), SZ(n), _stream()),
               "HUF_compress1X_usingCTable_batch")
        if g:
            g.check("HUF_compress1X_usingCTable_batch")
        return dst, res

    def huf_decompress4x_using_dtable_batch(self, csrc, csizes, dtables, dst_sizes, max_table_log=12, shared_table=False, dst=None, results=None):
        """HUF_decompress4X_usingDTable over a batch: X1 (tableType 0) and X2 (tableType 1) tables, chosen per block"""
        return self.huf_decompress4x1_using_dtable_batch(csrc, csizes, dtables, dst_sizes, max_table_log, shared_table, dst=dst, results=results,
                                                         _fn="FSEHIP_HUF_decompress4X_usingDTable_batch")

    def huf_decompress4x1_using_dtable_batch(self, csrc, csizes, dtables, dst_sizes, max_table_log=12, shared_table=False, dst=None, results=None,
                                             _fn="FSEHIP_HUF_decompress4X1_usingDTable_batch"):
        n = _blocks(csrc, "csrc").shape[0]
        g = None
        if dst is None:
            width = int(dst_sizes) if isinstance(dst_sizes, numbers.Integral) else int(dst_sizes.max().item())
            dst, g = self._dst(n, width, csrc.device, zero=not self.guard)
        res = torch.zeros(n, dtype=torch.int64, device=csrc.device) if results is None else results
        pc, unic, keepc = _sizes_arg(csizes, csrc)
        pd, unid, keepd = _sizes_arg(dst_sizes, csrc)
        stride = 0 if shared_table else dtables.stride(0)
        _check(getattr(self.lib, _fn)(_ptr(dst), SZ(dst.stride(0)), pd, unid, _ptr(res), _ptr(csrc), SZ(csrc.stride(0)),
                                      pc, unic, _ptr(dtables), SZ(stride), C.c_uint(max_table_log), SZ(n), _stream()), _fn)
        if g:
            g.check(_fn, dst_sizes)
        return dst, res

    def huf_build_ctable_batch(self, src, table_log=11, max_symbol_value=255, sizes=None, header_capacity=256):
        """HUF_buildCTable_batch: (ctables (n, 256) int32 HUF_CElt, headers (n, header_capacity) uint8, results)"""
        n = _blocks(src, "src").shape[0]
        ct = torch.zeros((n, 256), dtype=torch.int32, device=src.device)
        hdr, g = self._dst(n, header_capacity, src.device, zero=True)
        res = torch.zeros(n, dtype=torch.int64, device=src.device)
        ws = torch.empty(int(self.lib.FSEHIP_HUF_buildCTable_batch_workspaceSize(SZ(n))), dtype=torch.uint8, device=src.device)
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        _check(self.lib.FSEHIP_HUF_buildCTable_batch(_ptr(ct), SZ(ct.stride(0)), _ptr(hdr), SZ(hdr.stride(0)), SZ(header_capacity), _ptr(res), _ptr(src),
                                                     SZ(src.stride(0)), ps, uni, C.c_uint(max_symbol_value), C.c_uint(table_log), SZ(n), _ptr(ws),
                                                     SZ(ws.numel()), _stream()), "HUF_buildCTable_batch")
        if g:
            g.check("HUF_buildCTable_batch")
        return ct, hdr, res

    def huf_read_dtable_x1_batch(self, csrc, csizes, max_table_log=11):
        """HUF_readDTableX1_batch: (dtables (n, 1 + (1 << max_table_log)) int32, results = header bytes or error)"""
        n = _blocks(csrc, "csrc").shape[0]
        dt = torch.zeros((n, 1 + (1 << max_table_log)), dtype=torch.int32, device=csrc.device)
        res = torch.zeros(n, dtype=torch.int64, device=csrc.device)
        ws = torch.empty(int(self.lib.FSEHIP_HUF_readDTableX1_batch_workspaceSize(SZ(n))), dtype=torch.uint8, device=csrc.device)
        ps, uni, keep = _sizes_arg(csizes, csrc)
        _check(self.lib.FSEHIP_HUF_readDTableX1_batch(_ptr(dt), SZ(dt.stride(0)), C.c_uint(max_table_log), _ptr(res), _ptr(csrc), SZ(csrc.stride(0)), ps, uni,
                                                      SZ(n), _ptr(ws), SZ(ws.numel()), _stream()), "HUF_readDTableX1_batch")
        return dt, res

    def huf_decompress_packed_batch(self, packed, offsets, dst_sizes, dst=None, results=None, workspace=None):
        n = offsets.numel() - 1
        width = int(dst_sizes) if isinstance(dst_sizes, numbers.Integral) else int(dst_sizes.max().item())
        g = None
        if dst is None:
            dst, g = self._dst(n, width, packed.device)
        if results is None:
            results = torch.empty(n, dtype=torch.int64, device=packed.device)
        if workspace is None:
            workspace = self.huf_workspace(n, True, packed.device)
        pd, unid, keepd = _sizes_arg(dst_sizes, packed)
        _check(self.lib.FSEHIP_HUF_decompress_packed_batch(_ptr(dst), SZ(dst.stride(0)), pd, unid, _ptr(results), _ptr(packed), _ptr(offsets), SZ(n),
                                                           _ptr(workspace), SZ(workspace.numel()), _stream()), "HUF_decompress_packed_batch")
        if g:
            g.check("HUF_decompress_packed_batch", dst_sizes)
        return dst, results

    def huf_decompress1x1_using_dtable_batch(self, csrc, csizes, dtables, dst_sizes, max_table_log=12, shared_table=False, dst=None, results=None):
        """HUF_decompress1X1_usingDTable over a batch: one stream per block"""
        return self.huf_decompress4x1_using_dtable_batch(csrc, csizes, dtables, dst_sizes, max_table_log, shared_table, dst=dst, results=results,
                                                         _fn="FSEHIP_HUF_decompress1X1_usingDTable_batch")

    def huf_decompress1x_using_dtable_batch(self, csrc, csizes, dtables, dst_sizes, max_table_log=12, shared_table=False, dst=None, results=None):
        return self.huf_decompress4x1_using_dtable_batch(csrc, csizes, dtables, dst_sizes, max_table_log, shared_table, dst=dst, results=results,
                                                         _fn="FSEHIP_HUF_decompress1X_usingDTable_batch")

    def huf_decompress1x1_using_dtable(self, csrc, dt, dst_size):
        dt = np.ascontiguousarray(dt, dtype=np.uint32)
        return self._single("FSEHIP_HUF_decompress1X1_usingDTable", dst_size, csrc, dt.ctypes.data_as(VP))

    def huf_decompress1x_using_dtable(self, csrc, dt, dst_size):
        dt = np.ascontiguousarray(dt, dtype=np.uint32)
        return self._single("FSEHIP_HUF_decompress1X_usingDTable", dst_size, csrc, dt.ctypes.data_as(VP))

    # the Huff0 advanced flow on host pointers (lib/huf.h:141-167,204-218,288,299-304): same argument order as the reference, numpy in / out
    def huf_build_ctable(self, count, max_sv, max_nb_bits=0):
        """HUF_buildCTable: (table log or error, HUF_CElt[256] as uint32 -- entries beyond max_sv stay zero)"""
        count = np.ascontiguousarray(count, dtype=np.uint32)
        celt = np.zeros(256, dtype=np.uint32)
        self.lib.FSEHIP_HUF_buildCTable.restype = SZ
        return int(self.lib.FSEHIP_HUF_buildCTable(celt.ctypes.data_as(VP), count.ctypes.data_as(VP), C.c_uint(max_sv), C.c_uint(max_nb_bits))), celt

    def huf_write_ctable(self, celt, max_sv, huff_log, cap=256):
        celt = np.ascontiguousarray(celt, dtype=np.uint32)
        out = np.full(max(cap, 1) + 8, 0xA5, dtype=np.uint8)
        self.lib.FSEHIP_HUF_writeCTable.restype = SZ
        r = int(self.lib.FSEHIP_HUF_writeCTable(out.ctypes.data_as(VP), SZ(cap), celt.ctypes.data_as(VP), C.c_uint(max_sv), C.c_uint(huff_log)))
        assert (out[cap:] == 0xA5).all(), "HUF_writeCTable wrote past maxDstSize"
        return r, out[:cap]

    def huf_read_dtable_x1(self, src, max_table_log=12):
        """HUF_readDTableX1 into a DTable made by HUF_CREATE_STATIC_DTABLEX1(DTable, max_table_log): (header size or error, DTable)"""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        dt = np.zeros(1 + (1 << 11), dtype=np.uint32)
        dt[0] = (max_table_log - 1) * 0x01000001
        self.lib.FSEHIP_HUF_readDTableX1.restype = SZ
        return int(self.lib.FSEHIP_HUF_readDTableX1(dt.ctypes.data_as(VP), src.ctypes.data_as(VP), SZ(src.size))), dt

    def huf_compress1x(self, src, max_sv=255, huff_log=11, cap=None):
        return self._single("FSEHIP_HUF_compress1X", huf_compress_bound(len(src)) if cap is None else cap, src, C.c_uint(max_sv), C.c_uint(huff_log))

    def huf_decompress4x1(self, csrc, dst_size):
        self.lib.FSEHIP_HUF_decompress4X1.restype = SZ
        return self._single("FSEHIP_HUF_decompress4X1", dst_size, csrc)

    def huf_decompress1x1(self, csrc, dst_size):
        self.lib.FSEHIP_HUF_decompress1X1.restype = SZ
        return self._single("FSEHIP_HUF_decompress1X1", dst_size, csrc)

    # layer 1
    def huf_compress2(self, src, max_sv=255, huff_log=11, cap=None):
        return self._single("FSEHIP_HUF_compress2", huf_compress_bound(len(src)) if cap is None else cap, src, C.c_uint(max_sv), C.c_uint(huff_log))

    def huf_decompress(self, csrc, dst_size):
        return self._single("FSEHIP_HUF_decompress", dst_size, csrc)

    def huf_compress1x_using_ctable(self, src, celt, cap=None):
        celt = np.ascontiguousarray(celt, dtype=np.uint32)
        return self._single("FSEHIP_HUF_compress1X_usingCTable", huf_compress_bound(len(src)) if cap is None else cap, src, celt.ctypes.data_as(VP))

    def huf_compress4x_using_ctable(self, src, celt, cap=None):
        celt = np.ascontiguousarray(celt, dtype=np.uint32)
        return self._single("FSEHIP_HUF_compress4X_usingCTable", huf_compress_bound(len(src)) if cap is None else cap, src, celt.ctypes.data_as(VP))

    def huf_decompress4x1_using_dtable(self, csrc, dt, dst_size):
        dt = np.ascontiguousarray(dt, dtype=np.uint32)
        return self._single("FSEHIP_HUF_decompress4X1_usingDTable", dst_size, csrc, dt.ctypes.data_as(VP))

    def huf_decompress4x_using_dtable(self, csrc, dt, dst_size):
        dt = np.ascontiguousarray(dt, dtype=np.uint32)
        return self._single("FSEHIP_HUF_decompress4X_usingDTable", dst_size, csrc, dt.ctypes.data_as(VP))

    for f in (huf_decompress1x1_using_dtable_batch, huf_decompress1x_using_dtable_batch, huf_decompress1x1_using_dtable, huf_decompress1x_using_dtable,
              huf_decompress_packed_batch, huf_build_ctable_batch, huf_read_dtable_x1_batch, huf_workspace, huf_compress_batch, huf_decompress_batch, huf_compress4x_using_ctable_batch, huf_compress1x_using_ctable_batch,
              huf_decompress4x1_using_dtable_batch, huf_decompress4x_using_dtable_batch, huf_compress2, huf_decompress, huf_compress1x_using_ctable,
              huf_compress4x_using_ctable, huf_decompress4x1_using_dtable, huf_decompress4x_using_dtable,
              huf_build_ctable, huf_write_ctable, huf_read_dtable_x1, huf_compress1x, huf_decompress4x1, huf_decompress1x1):
        setattr(FseHip, f.__name__, f)


_huf_methods()


def _frame_methods():
    # .fse frames on host buffers (programs/fileio.c): FSEHIP_frame_compress / FSEHIP_frame_decompress
    def frame_compress(self, src, block_size_id=5, codec=0, cap=None):
        src = np.ascontiguousarray(src, dtype=np.uint8)
        self.lib.FSEHIP_frame_compressBound.restype = C.c_size_t
        bound = int(self.lib.FSEHIP_frame_compressBound(SZ(src.size), C.c_uint(block_size_id)))
        if cap is None:
            cap = bound if bound < (1 << 62) else 16
        return self._single("FSEHIP_frame_compress", cap, src, C.c_uint(block_size_id), C.c_int(codec))

    def frame_decompress(self, frame, cap):
        return self._single("FSEHIP_frame_decompress", cap, frame)

    def _frames(self, fname, srcs, caps, n_threads, *extra):
        # many frames per call: arrays of host pointers / sizes; every destination carries a 0xA5 tail like _single's
        srcs = [np.ascontiguousarray(x, dtype=np.uint8) for x in srcs]
        n = len(srcs)
        outs = [np.zeros(max(c, 1) + 16, dtype=np.uint8) for c in caps]
        for o, c in zip(outs, caps):
            o[c:] = 0xA5
        PA, SA = C.c_void_p * max(n, 1), C.c_size_t * max(n, 1)
        dsts = PA(*[o.ctypes.data for o in outs]); dcap = SA(*caps)
        sp = PA(*[x.ctypes.data for x in srcs]); ssz = SA(*[x.size for x in srcs])
        res = SA()
        fn = getattr(self.lib, fname)
        fn.restype = C.c_size_t
        r = int(fn(dsts, dcap, sp, ssz, res, SZ(n), *extra, C.c_uint(n_threads)))
        assert r == 0, "%s: %#x" % (fname, r)
        for o, c in zip(outs, caps):
            assert (o[c:] == 0xA5).all(), "%s wrote past dstCapacity" % fname
        return [(int(res[i]), outs[i][:caps[i]]) for i in range(n)]

    def frame_compress_batch(self, srcs, block_size_id=5, codec=0, caps=None, n_threads=0):
        self.lib.FSEHIP_frame_compressBound.restype = C.c_size_t
        if caps is None:
            caps = [int(self.lib.FSEHIP_frame_compressBound(SZ(np.asarray(x).size), C.c_uint(block_size_id))) for x in srcs]
            caps = [c if c < (1 << 62) else 16 for c in caps]
        return self._frames("FSEHIP_frame_compress_batch", srcs, caps, n_threads, C.c_uint(block_size_id), C.c_int(codec))

    def frame_decompress_batch(self, frames, caps, n_threads=0):
        return self._frames("FSEHIP_frame_decompress_batch", frames, caps, n_threads)

    for f in (_frames, frame_compress_batch, frame_decompress_batch, frame_compress, frame_decompress):
        setattr(FseHip, f.__name__, f)


_frame_methods()


# ---------------------------------------------------------------------------------------------------------
#  FSE for 16-bit symbols (lib/fseU16.h): sizes of the uncompressed side are in symbols
# ---------------------------------------------------------------------------------------------------------
FSEU16_MAX_SYMBOL_VALUE = 286


def fse_u16_compress_bound(n_symbols):      # FSE_compressBound over the bytes of the symbols (what programs/bench.c:221 hands over)
    return fse_compress_bound(2 * n_symbols)


def _u16_methods():
    def _blocks16(t, what):
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.int16 or t.dim() != 2:
            raise TypeError("%s must be a 2-D CUDA int16 tensor (16-bit symbols; torch has no uint16 arithmetic, the bits are what counts)" % what)
        if t.shape[1] > 1 and t.stride(1) != 1:
            raise ValueError("%s: symbols of a block must be contiguous" % what)
        return t

    def fse_count_u16_batch(self, src, sizes=None, max_symbol_value=FSEU16_MAX_SYMBOL_VALUE):
        n = _blocks16(src, "src").shape[0]
        counts = torch.empty((n, FSEU16_MAX_SYMBOL_VALUE + 1), dtype=torch.int32, device=src.device)
        maxsv = torch.empty(n, dtype=torch.int32, device=src.device)
        results = torch.empty(n, dtype=torch.int64, device=src.device)
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        _check(self.lib.FSEHIP_FSE_countU16_batch(_ptr(counts), _ptr(maxsv), _ptr(results), _ptr(src), SZ(2 * src.stride(0)), ps, uni,
                                                  C.c_uint(max_symbol_value), SZ(n), _stream()), "FSE_countU16_batch")
        return counts, maxsv, results

    def fse_compress_u16_batch(self, src, table_log=0, max_symbol_value=0, sizes=None, dst=None, dst_capacity=None, results=None, workspace=None):
        n = _blocks16(src, "src").shape[0]
        cap = fse_u16_compress_bound(src.shape[1]) if dst_capacity is None else dst_capacity
        g = None
        if dst is None:
            dst, g = self._dst(n, cap, src.device)
        if results is None:
            results = torch.empty(n, dtype=torch.int64, device=src.device)
        if workspace is None:
            workspace = torch.empty(int(self.lib.FSEHIP_FSE_compressU16_batch_workspaceSize(SZ(n))), dtype=torch.uint8, device=src.device)
        ps, uni, keep = _sizes_arg(src.shape[1] if sizes is None else sizes, src)
        _check(self.lib.FSEHIP_FSE_compressU16_batch(_ptr(dst), SZ(dst.stride(0)), SZ(cap), _ptr(results), _ptr(src), SZ(2 * src.stride(0)), ps, uni,
                                                     C.c_uint(max_symbol_value), C.c_uint(table_log), SZ(n), _ptr(workspace), SZ(workspace.numel()), _stream()),
               "FSE_compressU16_batch")
        if g:
            g.check("FSE_compressU16_batch")
        return dst, results

    def fse_decompress_u16_batch(self, csrc, csizes, dst_capacity, dst=None, results=None, workspace=None):
        n = _blocks(csrc, "csrc").shape[0]
        g = None
        if dst is None:
            dst, g = self._dst(n, dst_capacity, csrc.device, dtype=torch.int16)
        if results is None:
            results = torch.empty(n, dtype=torch.int64, device=csrc.device)
        if workspace is None:
            workspace = torch.empty(int(self.lib.FSEHIP_FSE_decompressU16_batch_workspaceSize(SZ(n))), dtype=torch.uint8, device=csrc.device)
        ps, uni, keep = _sizes_arg(csizes, csrc)
        _check(self.lib.FSEHIP_FSE_decompressU16_batch(_ptr(dst), SZ(2 * dst.stride(0)), SZ(dst_capacity), _ptr(results), _ptr(csrc), SZ(csrc.stride(0)), ps, uni,
                                                       SZ(n), _ptr(workspace), SZ(workspace.numel()), _stream()), "FSE_decompressU16_batch")
        if g:
            g.check("FSE_decompressU16_batch")
        return dst, results

    # host pointers, reference signatures
    def fse_count_u16(self, src, max_sv=FSEU16_MAX_SYMBOL_VALUE):
        src = np.ascontiguousarray(src, dtype=np.uint16)
        count = np.zeros(max(max_sv, FSEU16_MAX_SYMBOL_VALUE) + 1, dtype=np.uint32)
        msv = C.c_uint(max_sv)
        r = int(self.lib.FSEHIP_FSE_countU16(count.ctypes.data_as(VP), C.byref(msv), src.ctypes.data_as(VP), SZ(src.size)))
        return r, count, int(msv.value)

    def fse_compress_u16(self, src, max_sv=0, table_log=0, cap=None):
        src = np.ascontiguousarray(src, dtype=np.uint16)
        cap = fse_u16_compress_bound(src.size) if cap is None else cap
        out = np.zeros(max(cap, 1) + 16, dtype=np.uint8)
        out[cap:] = 0xA5
        r = int(self.lib.FSEHIP_FSE_compressU16(out.ctypes.data_as(VP), SZ(cap), src.ctypes.data_as(VP), SZ(src.size), C.c_uint(max_sv), C.c_uint(table_log)))
        assert (out[cap:] == 0xA5).all(), "FSE_compressU16 wrote past dstCapacity"
        return r, out[:cap]

    def fse_decompress_u16(self, csrc, cap):
        csrc = np.ascontiguousarray(csrc, dtype=np.uint8)
        out = np.zeros(max(cap, 1) + 8, dtype=np.uint16)
        out[cap:] = 0xA5A5
        r = int(self.lib.FSEHIP_FSE_decompressU16(out.ctypes.data_as(VP), SZ(cap), csrc.ctypes.data_as(VP), SZ(csrc.size)))
        assert (out[cap:] == 0xA5A5).all(), "FSE_decompressU16 wrote past dstCapacity"
        return r, out[:cap]

    for f in (fse_count_u16_batch, fse_compress_u16_batch, fse_decompress_u16_batch, fse_count_u16, fse_compress_u16, fse_decompress_u16):
        setattr(FseHip, f.__name__, f)


_u16_methods()
