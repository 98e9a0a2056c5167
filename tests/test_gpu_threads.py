"""Re-entrancy (SURVEY 8(b) "Threading"): the reference's functions are pure; the device library keeps that -- host threads calling
the batched entry points concurrently, each on its own stream with its own buffers, get the results a lone caller gets."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_concurrent_callers(hip):
    n, size, nthreads, rounds = 512, 32768, 4, 6
    srcs = [hip.probagen_batch(p, n, size, first_seed=1 + 1000 * i) for i, p in enumerate((14, 80, 2, 50))]
    expect = []
    for s in srcs:
        fd, fr = hip.fse_compress_batch(s, 11)
        hd, hr = hip.huf_compress_batch(s)
        expect.append((fd.clone(), fr.clone(), hd.clone(), hr.clone()))
    torch.cuda.synchronize()
    errors = []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(rounds):
                    fd, fr = hip.fse_compress_batch(srcs[i], 11)
                    out, dr = hip.fse_decompress_batch(fd, fr, size)
                    hd, hr = hip.huf_compress_batch(srcs[i])
                    hout, hdr = hip.huf_decompress_batch(hd, hr, size)
                    st.synchronize()
                    e = expect[i]
                    assert torch.equal(fr, e[1]) and torch.equal(hr, e[3])
                    assert torch.equal(out, srcs[i]) and torch.equal(hout, srcs[i])
                    k = int(fr.max())
                    assert torch.equal(fd[:, :k] * (torch.arange(k, device="cuda")[None, :] < fr[:, None]), e[0][:, :k] * (torch.arange(k, device="cuda")[None, :] < e[1][:, None]))
        except Exception as ex:  # noqa: BLE001
            errors.append((i, repr(ex)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def test_host_call_scratch_arena_grows_and_is_released(hip, checker):
    """the calls on host pointers carve their device buffers from a per-thread arena (no hipMalloc per call): results must not depend
    on what the arena held before -- a large call, small ones, a release in between, a call larger than the arena's first size"""
    import ctypes as C
    small = checker.probagen_batch(14, 1, 3000, 5)[0]
    big = checker.probagen_batch(14, 1, 1 << 21, 6)[0]                  # 2 MiB: beyond the arena's initial megabyte
    want_s, want_b = checker.fse_compress2(small), checker.fse_compress2(big, 255, 12)
    for rounds in range(2):
        for blk, want, tl in ((small, want_s, 11), (big, want_b, 12), (small, want_s, 11)):
            r, out = hip.fse_compress2(blk, 255, tl)
            assert r == want[0] and (out[:r] == want[1][:r]).all()
            r2, back = hip.fse_decompress(out[:r], blk.size)
            assert r2 == blk.size and (back[:r2] == blk).all()
            hr, ho = hip.huf_compress2(blk[:100000]), checker.huf_compress2(blk[:100000])
            assert hr[0] == ho[0] and (hr[1][:hr[0]] == ho[1][:ho[0]]).all()
        assert hip.lib.FSEHIP_releaseScratch() == 0
