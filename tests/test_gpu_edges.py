"""Over-read checks of the batched calls (SURVEY 8(b) ownership: the library reads src[0, srcSize) / cSrc[0, cSrcSize) and the
tables it is handed, nothing behind them).  The input of the LAST block of a batch is placed flush against unmapped address
space (oracle/vmm_edge.c: a reservation of which only the first half is mapped), so a load behind it is a GPU page fault and
kills the process; every case therefore runs in a child process (`python tests/test_gpu_edges.py <case>`), and the child
also compares results and output bytes with the same call on ordinary memory.  The write side of the contract is covered by
guard mode (tests/conftest.py: FseHip.guard), which the child switches on as well."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CASES = ["hist", "fse_compress", "fse_decompress", "fse_using_ctable", "fse_using_dtable", "huf_compress", "huf_decompress",
         "huf_using_ctable", "huf_using_dtable", "u16_compress", "u16_decompress"]


@pytest.mark.parametrize("case", CASES)
def test_no_read_behind_the_last_block(case):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    p = subprocess.run([sys.executable, os.path.abspath(__file__), case], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "EDGE_OK" in p.stdout, "%s: rc %d\n%s\n%s" % (case, p.returncode, p.stdout[-2000:], p.stderr[-3000:])


# ------------------------------------------------------------------------------------------------ child process
class _Edge(C.Structure):
    _fields_ = [("va", C.c_void_p), ("vaBytes", C.c_size_t), ("mapped", C.c_size_t), ("handle", C.c_void_p), ("user", C.c_void_p)]


class EdgeMem:
    """nbytes of device memory ending exactly where mapped memory ends; .tensor is a flat uint8 torch view of it"""
    _lib = None

    def __init__(self, nbytes):
        import torch
        if EdgeMem._lib is None:
            EdgeMem._lib = C.CDLL(os.path.join(ROOT, "oracle", "libvmm_edge.so"))
        self.rec = _Edge()
        rc = EdgeMem._lib.vmm_edge_alloc(C.c_size_t(nbytes), C.byref(self.rec))
        assert rc == 0, "vmm_edge_alloc failed: hipError %d" % rc
        self.nbytes = nbytes
        self.__cuda_array_interface__ = {"shape": (max(nbytes, 1),), "typestr": "|u1", "data": (int(self.rec.user) - (0 if nbytes else 1), False), "version": 2}
        self.tensor = torch.as_tensor(self, device="cuda")
        assert self.tensor.data_ptr() == self.__cuda_array_interface__["data"][0], "torch copied the edge memory instead of wrapping it"

    def free(self):
        """the mapping is left in place until the child process exits (a few MB per case): unmapping a range and mapping the next one
        at the recycled address showed stale contents through torch views on this stack"""
        self.tensor = None


def at_edge(t, row_bytes=None):
    """copy of the 2-D uint8 / int16 CUDA tensor `t` whose LAST row's first `row_bytes` bytes (default: the whole row) end at the
    edge; rows keep t's row stride.  Returns (tensor view, EdgeMem keepalive)."""
    import torch
    flat = t.contiguous().view(torch.uint8).reshape(t.shape[0], -1)
    n, w = flat.shape
    last = w if row_bytes is None else int(row_bytes)
    total = (n - 1) * w + last
    mem = EdgeMem(total)
    mem.tensor[:total].copy_(flat.reshape(-1)[:total])
    view = torch.as_strided(mem.tensor, (n, max(last, 1)), (w, 1))
    if t.dtype == torch.int16:
        view = torch.as_strided(mem.tensor.view(torch.int16), (n, max(last // 2, 1)), (w // 2, 1))
    return view, mem


def _same(a, b, what):
    import torch
    assert torch.equal(a, b), what


def _run(case):
    import numpy as np
    import torch
    from finitestateentropy_amd.api import FseHip
    from oracle.oracle import Checker, fse_ctable_u32, fse_dtable_u32, is_error
    FseHip.guard = 64
    hip, orc = FseHip(), Checker()
    torch.cuda.set_device(0)
    n = 70

    def blocks(size, seed=1):
        out = torch.empty((n, size), dtype=torch.uint8, device="cuda")
        for j, P in enumerate((14, 80, 2)):
            rows = out[j::3]
            hip.probagen_batch(P, rows.shape[0], size, first_seed=seed + j, out=rows, seed_step=3)
        return out

    sizes_list = (32768, 4097, 1001, 2048, 15)
    if case == "hist":
        for size in sizes_list + (1, 16, 17, 63, 64, 65):
            src = blocks(size)
            e, keep = at_edge(src)
            a = hip.hist_count_batch(src)
            b = hip.hist_count_batch(e)
            for x, y in zip(a, b):
                _same(x, y, ("hist", size))
            keep.free()
    elif case in ("fse_compress", "huf_compress"):
        fn = hip.fse_compress_batch if case == "fse_compress" else hip.huf_compress_batch
        for size in sizes_list:
            for tl in (11, 12, 9):
                src = blocks(size)
                e, keep = at_edge(src)
                d0, r0 = fn(src, table_log=tl)
                d1, r1 = fn(e, table_log=tl)
                _same(r0, r1, (case, size, tl))
                ok = (r0 > 1) & (r0 < (1 << 40))
                cols = torch.arange(d0.shape[1], device="cuda")[None, :] < (r0 * ok)[:, None]
                _same(d0 * cols, d1 * cols, (case, size, tl, "bytes"))
                keep.free()
    elif case in ("fse_decompress", "huf_decompress"):
        codec = 0 if case == "fse_decompress" else 1
        for size in sizes_list:
            for tl in (11, 12):
                src = blocks(size)
                comp, res = (hip.fse_compress_batch if codec == 0 else hip.huf_compress_batch)(src, table_log=tl)
                good = ((res > 1) & (res < (1 << 40))).nonzero().flatten()
                if good.numel() == 0:
                    continue
                comp, res, want = comp[good].contiguous(), res[good].contiguous(), src[good]
                e, keep = at_edge(comp, row_bytes=int(res[-1].item()))
                if codec == 0:
                    o0, q0 = hip.fse_decompress_batch(comp, res, size)
                    o1, q1 = hip.fse_decompress_batch(e, res, size)
                else:
                    o0, q0 = hip.huf_decompress_batch(comp, res, size)
                    o1, q1 = hip.huf_decompress_batch(e, res, size)
                _same(q0, q1, (case, size, tl)); _same(o0, o1, (case, size, tl, "bytes"))
                assert bool((q1 == size).all()) and torch.equal(o1, want), (case, size, tl, "round trip")
                # truncated streams: the last block's stream cut short, still flush with the edge
                for cut in (1, 2, 9):
                    rs = res.clone(); rs[-1] = max(int(res[-1].item()) - cut, 1)
                    e2, keep2 = at_edge(comp, row_bytes=int(rs[-1].item()))
                    if codec == 0:
                        _, qa = hip.fse_decompress_batch(comp, rs, size); _, qb = hip.fse_decompress_batch(e2, rs, size)
                    else:
                        _, qa = hip.huf_decompress_batch(comp, rs, size); _, qb = hip.huf_decompress_batch(e2, rs, size)
                    _same(qa, qb, (case, size, tl, "cut", cut))
                    keep2.free()
                keep.free()
    elif case in ("fse_using_ctable", "fse_using_dtable"):
        for size in sizes_list[:4]:
            src = blocks(size)
            host = src.cpu().numpy()
            for tl in (11, 12):
                cts, dts, rows = [], [], []
                for b in range(n):
                    mx, msv, cnt = orc.hist_count(host[b])
                    t = orc.fse_optimal_tablelog(tl, size, msv, 2)
                    r, norm = orc.fse_normalize_count(t, cnt, size, msv)
                    if is_error(r) or r == 0 or mx == size:
                        continue
                    _, ct = orc.fse_build_ctable(norm, msv, t)
                    _, dt = orc.fse_build_dtable(norm, msv, t)
                    cts.append((ct, msv, t)); dts.append(dt); rows.append(b)
                # every table at its exact size (FSE_CTABLE_SIZE_U32(tableLog, maxSymbolValue), FSE_DTABLE_SIZE_U32(tableLog)) in a
                # slot of the widest; the last block's table ends at the edge
                ctw = max(len(c[0]) for c in cts); dtw = max(len(d) for d in dts)
                ct_np = np.zeros((len(rows), ctw), np.uint32); dt_np = np.zeros((len(rows), dtw), np.uint32)
                for i in range(len(rows)):
                    ct_np[i, :len(cts[i][0])] = cts[i][0]; dt_np[i, :len(dts[i])] = dts[i]
                d_ct = torch.from_numpy(ct_np.view(np.int32)).cuda(); d_dt = torch.from_numpy(dt_np.view(np.int32)).cuda()
                sub = src[torch.tensor(rows, device="cuda")].contiguous()
                c0, r0 = hip.fse_compress_using_ctable_batch(sub, d_ct, max_table_log=12)
                if case == "fse_using_ctable":
                    e, keep = at_edge(sub)
                    exact = fse_ctable_u32(cts[-1][2], cts[-1][1])
                    ect, keep2 = at_edge(d_ct.view(torch.uint8).reshape(len(rows), -1), row_bytes=4 * exact)
                    ect = torch.as_strided(ect.view(torch.int32), (len(rows), exact), (ctw, 1))
                    c1, r1 = hip.fse_compress_using_ctable_batch(e, ect, max_table_log=12)
                    _same(r0, r1, (case, size, tl))
                    cols = torch.arange(c0.shape[1], device="cuda")[None, :] < r0[:, None]
                    _same(c0 * cols, c1 * cols, (case, size, tl, "bytes"))
                    keep.free(); keep2.free()
                else:
                    good = (r0 > 0).nonzero().flatten()
                    comp, res = c0[good].contiguous(), r0[good].contiguous()
                    tabs = d_dt[good].contiguous()
                    o0, q0 = hip.fse_decompress_using_dtable_batch(comp, res, tabs, size, max_table_log=12)
                    e, keep = at_edge(comp, row_bytes=int(res[-1].item()))
                    exact = fse_dtable_u32(cts[int(good[-1].item())][2])
                    edt, keep2 = at_edge(tabs.view(torch.uint8).reshape(tabs.shape[0], -1), row_bytes=4 * exact)
                    edt = torch.as_strided(edt.view(torch.int32), (tabs.shape[0], exact), (dtw, 1))
                    o1, q1 = hip.fse_decompress_using_dtable_batch(e, res, edt, size, max_table_log=12)
                    _same(q0, q1, (case, size, tl)); _same(o0, o1, (case, size, tl, "bytes"))
                    assert bool((q1 == size).all()) and torch.equal(o1, sub[good]), (case, size, tl, "round trip")
                    keep.free(); keep2.free()
    elif case in ("huf_using_ctable", "huf_using_dtable"):
        for size in sizes_list[:4]:
            src = blocks(size)
            host = src.cpu().numpy()
            celts, dts, rows = [], [], []
            for b in range(n):
                mx, msv, cnt = orc.hist_count(host[b])
                if mx == size or msv == 0:
                    continue
                hl = orc.fse_optimal_tablelog(11, size, msv, 1)
                mb, celt = orc.huf_build_ctable(cnt, msv, hl)
                hs, hdr = orc.huf_write_ctable(256, celt, msv, mb)
                if is_error(mb) or is_error(hs):
                    continue
                r, dt = orc.huf_read_dtable_x1(hdr[:hs], 11)
                if is_error(r):
                    continue
                celt = celt.copy(); celt[msv + 1:] = 0
                celts.append((celt, msv)); dts.append((dt, mb)); rows.append(b)
            ce_np = np.stack([c[0] for c in celts]).astype(np.uint32)
            dtw = max(len(d[0]) for d in dts)
            dt_np = np.zeros((len(rows), dtw), np.uint32)
            for i in range(len(rows)):
                dt_np[i, :len(dts[i][0])] = dts[i][0]
            d_ce = torch.from_numpy(ce_np.view(np.int32)).cuda(); d_dt = torch.from_numpy(dt_np.view(np.int32)).cuda()
            sub = src[torch.tensor(rows, device="cuda")].contiguous()
            c0, r0 = hip.huf_compress4x_using_ctable_batch(sub, d_ce)
            if case == "huf_using_ctable":
                e, keep = at_edge(sub)
                exact = 256                # a HUF_CElt table has no header: the batched calls read HUF_CTABLE_SIZE_U32(255) entries (fsehip.h)
                ece, keep2 = at_edge(d_ce.view(torch.uint8).reshape(len(rows), -1), row_bytes=4 * exact)
                ece = torch.as_strided(ece.view(torch.int32), (len(rows), exact), (256, 1))
                c1, r1 = hip.huf_compress4x_using_ctable_batch(e, ece)
                _same(r0, r1, (case, size))
                cols = torch.arange(c0.shape[1], device="cuda")[None, :] < r0[:, None]
                _same(c0 * cols, c1 * cols, (case, size, "bytes"))
                x0, s0 = hip.huf_compress1x_using_ctable_batch(sub, d_ce)
                x1, s1 = hip.huf_compress1x_using_ctable_batch(e, ece)
                _same(s0, s1, (case, size, "1X"))
                cols = torch.arange(x0.shape[1], device="cuda")[None, :] < s0[:, None]
                _same(x0 * cols, x1 * cols, (case, size, "1X bytes"))
                keep.free(); keep2.free()
            else:
                good = (r0 > 0).nonzero().flatten()
                if good.numel() == 0:
                    continue
                comp, res, tabs = c0[good].contiguous(), r0[good].contiguous(), d_dt[good].contiguous()
                o0, q0 = hip.huf_decompress4x1_using_dtable_batch(comp, res, tabs, size, max_table_log=12)
                e, keep = at_edge(comp, row_bytes=int(res[-1].item()))
                exact = 1 + (1 << (dts[int(good[-1].item())][1] - 1))    # 4-byte descriptor + 2-byte cells
                edt, keep2 = at_edge(tabs.view(torch.uint8).reshape(tabs.shape[0], -1), row_bytes=4 * exact)
                edt = torch.as_strided(edt.view(torch.int32), (tabs.shape[0], exact), (dtw, 1))
                o1, q1 = hip.huf_decompress4x1_using_dtable_batch(e, res, edt, size, max_table_log=12)
                _same(q0, q1, (case, size)); _same(o0, o1, (case, size, "bytes"))
                assert bool((q1 == size).all()) and torch.equal(o1, sub[good]), (case, size, "round trip")
                keep.free(); keep2.free()
    elif case in ("u16_compress", "u16_decompress"):
        rng = np.random.default_rng(5)
        for nsym in (16384, 4097, 2048, 300):
            table = np.zeros(4096, np.uint16)
            remaining, pos, val = 4096, 0, 240
            while remaining:
                k = int(remaining * 0.08) + 1
                table[pos:pos + k] = val
                pos += k; remaining -= k; val = val + 1 if val + 1 < 286 else 1
            host = table[rng.integers(0, 4096, (n, nsym))]
            src = torch.from_numpy(host.view(np.int16)).cuda()
            c0, r0 = hip.fse_compress_u16_batch(src)
            if case == "u16_compress":
                e, keep = at_edge(src)
                c1, r1 = hip.fse_compress_u16_batch(e)
                _same(r0, r1, (case, nsym))
                cols = torch.arange(c0.shape[1], device="cuda")[None, :] < r0[:, None]
                _same(c0 * cols, c1 * cols, (case, nsym, "bytes"))
                keep.free()
            else:
                good = ((r0 > 1) & (r0 < (1 << 40))).nonzero().flatten()
                comp, res = c0[good].contiguous(), r0[good].contiguous()
                o0, q0 = hip.fse_decompress_u16_batch(comp, res, nsym)
                e, keep = at_edge(comp, row_bytes=int(res[-1].item()))
                o1, q1 = hip.fse_decompress_u16_batch(e, res, nsym)
                _same(q0, q1, (case, nsym)); _same(o0, o1, (case, nsym, "symbols"))
                assert bool((q1 == nsym).all()) and torch.equal(o1, src[good]), (case, nsym, "round trip")
                keep.free()
    else:
        raise SystemExit("unknown case " + case)
    torch.cuda.synchronize()
    print("EDGE_OK", case)


if __name__ == "__main__":
    _run(sys.argv[1])


def test_guard_mode_detects_a_write_past_the_capacity():
    """the guard the whole GPU suite runs under (tests/conftest.py) does fire: one byte behind a block's capacity is reported"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from finitestateentropy_amd.api import FseHip, _Guarded
    g = _Guarded(5, 100, 64, "cuda", zero=True)
    g.check("untouched")
    g.full[3, 100] = 7
    with pytest.raises(AssertionError, match="block 3"):
        g.check("touched")
    g2 = _Guarded(5, 100, 64, "cuda")
    sizes = torch.tensor([100, 90, 100, 7, 0], device="cuda")
    g2.view[1, :90] = 1; g2.view[3, :7] = 1
    g2.check("exact", sizes)
    g2.view[3, 7] = 1
    with pytest.raises(AssertionError, match="block 3"):
        g2.check("one past", sizes)
    old = FseHip.guard
    try:
        FseHip.guard = 64
        hip = FseHip()
        src = hip.probagen_batch(14, 8, 4096)
        dst, res = hip.fse_compress_batch(src)
        assert dst.stride(0) == dst.shape[1] + 64       # the library was handed slots with a gap behind each
    finally:
        FseHip.guard = old
