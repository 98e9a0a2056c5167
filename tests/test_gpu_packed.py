"""GPU tests of the packed (variable-length) batches: FSEHIP_compact_batch against the plain concatenation of the reference's blocks
(compressed bytes, raw blocks for result 0, one byte for result 1: programs/bench.c:393-406), and the decoders that read a packed batch
where it lies (FSEHIP_FSE_decompress_packed_batch, FSEHIP_HUF_decompress_packed_batch): every block regenerated, whatever its kind."""
import numpy as np
import pytest
import torch

from oracle.oracle import is_error
from test_gpu_fse import mixed_blocks, s64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle(checker):
    return checker


@pytest.mark.parametrize("size", [1, 2, 7, 100, 1001, 4097, 32768])
@pytest.mark.parametrize("codec", ["fse", "huf"])
def test_compact_and_packed_decode(hip, oracle, codec, size):
    n = 45
    blocks = mixed_blocks(oracle, n, size, seed=size)            # compressible, incompressible (-> 0) and single-byte (-> 1) blocks
    src = torch.from_numpy(blocks).cuda()
    comp = hip.fse_compress_batch if codec == "fse" else hip.huf_compress_batch
    slots, res = comp(src)
    packed, offsets = hip.compact_batch(slots, res, src)
    res_h, off_h, packed_h, slots_h = res.cpu().numpy(), offsets.cpu().numpy(), packed.cpu().numpy(), slots.cpu().numpy()
    _, ores, odst = oracle.compress_batch(0 if codec == "fse" else 1, blocks)
    assert (res_h == ores.astype(np.int64)).all()
    expect = []
    for b in range(n):
        r = int(ores[b])
        if is_error(r):
            rec = np.zeros(0, np.uint8)
        elif r == 0:
            rec = blocks[b]
        elif r == 1:
            rec = blocks[b][:1]
        else:
            rec = odst[b][:r]
        expect.append(rec)
    lens = np.array([len(e) for e in expect])
    assert off_h[0] == 0 and (np.diff(off_h) == lens).all(), (codec, size, np.diff(off_h)[:10], lens[:10])
    cat = np.concatenate(expect) if lens.sum() else np.zeros(0, np.uint8)
    assert (packed_h[:off_h[n]] == cat).all(), (codec, size)
    assert {0, 1} <= set(int(r) for r in res_h) or size < 64      # (the mix does contain declined and single-byte blocks)
    # decode the packed batch where it lies
    if codec == "fse":
        out, dres = hip.fse_decompress_packed_batch(packed, offsets, size, size)
    else:
        out, dres = hip.huf_decompress_packed_batch(packed, offsets, size)
    out_h, dres_h = out.cpu().numpy(), dres.cpu().numpy()
    for b in range(n):
        r = int(ores[b])
        if r in (0, 1) or codec == "huf":
            want, _ = (size, None) if r in (0, 1) else oracle.huf_decompress(expect[b], size)
        else:
            want, _ = oracle.fse_decompress(expect[b], size)
        assert dres_h[b] == s64(want), (codec, size, b, r, dres_h[b], want)
        if not is_error(want):
            assert (out_h[b][:size] == blocks[b]).all(), (codec, size, b, r)


def test_compact_ragged_sizes_errors_and_small_buffers(hip, oracle):
    n, width = 40, 5000
    blocks = mixed_blocks(oracle, n, width, seed=5)
    src = torch.from_numpy(blocks).cuda()
    sizes = torch.tensor([width - 113 * (b % 9) for b in range(n)], dtype=torch.int64, device="cuda")
    sizes[3] = 0; sizes[4] = 1
    hs = sizes.cpu().numpy()
    slots, res = hip.fse_compress_batch(src, sizes=sizes)
    res2 = res.clone(); res2[7] = -3                              # an error result: no record
    packed, offsets = hip.compact_batch(slots, res2, src, sizes=sizes)
    off_h, packed_h, res_h = offsets.cpu().numpy(), packed.cpu().numpy(), res2.cpu().numpy()
    pos = 0
    for b in range(n):
        r = int(res_h[b])
        rec = np.zeros(0, np.uint8) if r < 0 else blocks[b][:hs[b]] if r == 0 else blocks[b][:min(1, hs[b])] if r == 1 else slots[b].cpu().numpy()[:r]
        assert off_h[b] == pos, b
        assert (packed_h[pos:pos + len(rec)] == rec).all(), b
        pos += len(rec)
    assert off_h[n] == pos
    # decode with per-block regenerated sizes (block 7 has no record: it is told apart by its size 0 != orig)
    out, dres = hip.fse_decompress_packed_batch(packed, offsets, sizes, width)
    out_h, dres_h = out.cpu().numpy(), dres.cpu().numpy()
    for b in range(n):
        if b == 7 or hs[b] == 0:
            continue
        assert dres_h[b] == hs[b] and (out_h[b][:hs[b]] == blocks[b][:hs[b]]).all(), b
    # a packed buffer that is too small: the offsets still describe the whole batch, what fits is written
    small = torch.zeros(int(off_h[n]) // 2, dtype=torch.uint8, device="cuda")
    packed2, offsets2 = hip.compact_batch(slots, res2, src, sizes=sizes, packed=small)
    assert torch.equal(offsets2, offsets)
    fit = int((off_h[1:] <= small.numel()).sum())
    assert (packed2.cpu().numpy()[:off_h[fit]] == packed_h[:off_h[fit]]).all()


def test_compact_full_size_batch(hip, oracle):
    """100k-block shape in small: 20,000 mixed P02 / P14 / P80 blocks: packed size = sum of the sizes, round trip through the packed decoders"""
    n = 20000
    src = hip.probagen_mixed((2, 14, 80), n)
    for codec in ("fse", "huf"):
        slots, res = (hip.fse_compress_batch if codec == "fse" else hip.huf_compress_batch)(src)
        packed, offsets = hip.compact_batch(slots, res, src)
        total = int(offsets[n].item())
        assert total == int(res.sum().item()) and bool((torch.diff(offsets) == res).all())
        if codec == "fse":
            out, dres = hip.fse_decompress_packed_batch(packed, offsets, 32768, 32768)
        else:
            out, dres = hip.huf_decompress_packed_batch(packed, offsets, 32768)
        assert bool((dres == 32768).all()) and torch.equal(out, src), codec


def test_compact_ignores_results_that_cannot_belong_to_the_slots(hip, oracle):
    """a results array that does not belong to these slots (a value beyond the slot stride that is no error code) yields no record instead of
    a read behind the slot; the other records are unaffected"""
    blocks = mixed_blocks(oracle, 12, 4097, seed=3)
    src = torch.from_numpy(blocks).cuda()
    slots, res = hip.fse_compress_batch(src)
    good_p, good_o = hip.compact_batch(slots, res, src)
    bad = res.clone()
    bad[5] = slots.stride(0) + 1
    bad[9] = 1 << 40
    p, o = hip.compact_batch(slots, bad, src)
    oh, gh = o.cpu().numpy(), good_o.cpu().numpy()
    lens, glens = np.diff(oh), np.diff(gh)
    assert lens[5] == 0 and lens[9] == 0
    keep = [b for b in range(12) if b not in (5, 9)]
    assert (lens[keep] == glens[keep]).all()
    ph, gph = p.cpu().numpy(), good_p.cpu().numpy()
    for b in keep:
        assert (ph[oh[b]:oh[b + 1]] == gph[gh[b]:gh[b + 1]]).all()


def test_prepare_device_is_idempotent(hip):
    assert hip.lib.FSEHIP_prepareDevice() == 0 and hip.lib.FSEHIP_prepareDevice() == 0
