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
