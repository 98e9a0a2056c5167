"""GPU parity of the .fse frame calls (FSEHIP_frame_compress / FSEHIP_frame_decompress, host buffers) against the CPU
oracle, whose frames are pinned against the reference's command-line tool (tests/test_frame_oracle.py)."""
import numpy as np
import pytest

from oracle.oracle import is_error
from test_frame_oracle import _inputs, oversize_frames
from test_gpu_fse import s64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle(checker):
    """in this module `oracle` is the one-hop checker: the compiled reference itself when oracle/_ref is present"""
    return checker


def test_frames_match_oracle(hip, oracle):
    for name, data in _inputs(oracle):
        for codec in (0, 1):
            r, out = oracle.frame_compress(data, 5, codec)
            rg, og = hip.frame_compress(data, 5, codec)
            assert rg == r and (og[:r] == out[:r]).all(), (name, codec, rg, r)
            r2, o2 = hip.frame_decompress(out[:r], len(data))
            assert r2 == len(data) and (o2[:r2] == data).all(), (name, codec)
            r3, o3 = hip.frame_decompress(out[:r], len(data) + 100)      # larger destination
            assert r3 == len(data) and (o3[:r3] == data).all(), (name, codec)


def test_frame_block_sizes_golden_and_errors(hip, oracle, golden):
    data = oracle.probagen_batch(14, 1, 150000, 9)[0]
    for bsid in range(0, 7):
        for codec in (0, 1):
            r, out = oracle.frame_compress(data, bsid, codec)
            rg, og = hip.frame_compress(data, bsid, codec)
            assert rg == r and (og[:r] == out[:r]).all(), (bsid, codec)
            r2, o2 = hip.frame_decompress(out[:r], len(data))
            assert r2 == len(data) and (o2[:r2] == data).all(), (bsid, codec)
    if "frame_fse" in golden:                                              # frames written by the reference tool itself
        src = golden["frame_src"]
        for key, codec in (("frame_fse", 0), ("frame_huf", 1)):
            rg, og = hip.frame_compress(src, 5, codec)
            assert rg == len(golden[key]) and (og[:rg] == golden[key]).all(), key
            r2, o2 = hip.frame_decompress(golden[key], len(src))
            assert r2 == len(src) and (o2[:r2] == src).all(), key
    # error behaviour follows the oracle: corrupt checksum / magic / payload, truncation, small destination, bad id
    r, out = oracle.frame_compress(data, 5, 0)
    frame = out[:r]
    cases = []
    for pos in (r - 1, 0, 4, 5, 6, 7, 40, 2000, r - 3, r - 5):
        bad = frame.copy(); bad[pos] ^= 0x55; cases.append((bad, len(data)))
    cases += [(frame[:r - 4], len(data)), (frame[:9], len(data)), (frame[:5], len(data)), (frame, len(data) - 1), (frame, 1000)]
    rng = np.random.default_rng(3)
    for _ in range(6):
        bad = frame.copy(); idx = rng.integers(5, r, 3); bad[idx] = rng.integers(0, 256, 3); cases.append((bad, len(data)))
    cases += oversize_frames(oracle)                                        # announced size above the block size (bsid 0)
    for bad, cap in cases:
        ro, oo = oracle.frame_decompress(bad, cap)
        rg, og = hip.frame_decompress(bad, cap)
        assert rg == ro or (is_error(rg) and is_error(ro) and s64(rg) == s64(ro)), (rg, ro)
        if not is_error(ro):
            assert (og[:ro] == oo[:ro]).all()
    assert is_error(hip.frame_compress(data, 7, 0)[0])
    assert is_error(hip.frame_compress(data, 5, 0, cap=1000)[0])


def test_large_frames_take_the_pipelined_reader(hip, oracle):
    """frames of 2048+ full compressed blocks are decoded in pieces with the checksum streamed beside the device work: same bytes and
    verdicts as the oracle's reader -- intact frames, a damaged payload in the first / a middle / the last piece (the one-shot path
    takes over), a damaged checksum, a destination one byte short, a raw block in the middle (not the regular shape)"""
    rng = np.random.default_rng(11)
    data = oracle.probagen_batch(14, 1, 5000 * 1024, 21)[0]                    # 5000 blocks of 1 KB (block-size id 0), 4883 of 1 KB + ...
    for bsid, codec in ((0, 0), (0, 1), (1, 0)):
        r, out = oracle.frame_compress(data, bsid, codec)
        frame = out[:r]
        rg, og = hip.frame_compress(data, bsid, codec)
        assert rg == r and (og[:r] == frame).all(), (bsid, codec)
        r2, o2 = hip.frame_decompress(frame, len(data))
        assert r2 == len(data) and (o2[:r2] == data).all(), (bsid, codec)
        cases = [(frame, len(data) - 1), (frame, len(data) + 5)]
        for pos in (40, r // 2, r - 40, r - 1, r - 2):
            bad = frame.copy(); bad[pos] ^= 0x21; cases.append((bad, len(data)))
        for _ in range(3):
            bad = frame.copy(); idx = rng.integers(5, r - 3, 2); bad[idx] ^= 0x80; cases.append((bad, len(data)))
        for bad, cap in cases:
            ro, oo = oracle.frame_decompress(bad, cap)
            rg, og = hip.frame_decompress(bad, cap)
            assert rg == ro or (is_error(rg) and is_error(ro) and s64(rg) == s64(ro)), (bsid, codec, rg, ro)
            if not is_error(ro):
                assert (og[:ro] == oo[:ro]).all()
    # an incompressible block in the middle is stored raw: the frame is not "full compressed blocks only"
    mixed = data[:3000 * 1024].copy()
    mixed[1500 * 1024:1501 * 1024] = rng.integers(0, 256, 1024, dtype=np.uint8)
    r, out = oracle.frame_compress(mixed, 0, 0)
    r2, o2 = hip.frame_decompress(out[:r], len(mixed))
    assert r2 == len(mixed) and (o2[:r2] == mixed).all()


def test_frame_batch_calls_match_the_single_frame_calls(hip, oracle):
    """FSEHIP_frame_compress_batch / _decompress_batch: every frame of a batch gets the bytes and the result of the single-frame
    call (= the oracle's, pinned against the tool), whatever the pool size -- empty and tiny inputs, ragged tails, a frame large
    enough for the pipelined reader, raw and RLE blocks, and on the way back damaged frames, short destinations and non-frames"""
    rng = np.random.default_rng(5)
    srcs = [np.zeros(0, np.uint8), np.array([7], np.uint8), np.full(40000, 3, np.uint8),
            rng.integers(0, 256, 70000, dtype=np.uint8)]
    for i, (p, n) in enumerate(((14, 100000), (80, 33000), (2, 32768 * 3), (14, 2500 * 1024), (50, 1 << 20), (14, 777))):
        srcs.append(oracle.probagen_batch(p, 1, n, 100 + i)[0])
    for bsid, codec in ((5, 0), (5, 1), (0, 0), (2, 1)):
        want = [oracle.frame_compress(x, bsid, codec) for x in srcs]
        for nthreads in (1, 3, 0):
            got = hip.frame_compress_batch(srcs, bsid, codec, n_threads=nthreads)
            for (r, out), (rg, og) in zip(want, got):
                assert rg == r and (og[:r] == out[:r]).all(), (bsid, codec, nthreads)
        frames = [out[:r] for r, out in want]
        caps = [len(x) for x in srcs]
        # the way back: intact frames; then every second one damaged / cut short / given a short destination
        bad, bcaps = [], []
        for i, (f, c) in enumerate(zip(frames, caps)):
            f = f.copy()
            kind = i % 5
            if kind == 1 and len(f) > 12:
                f[int(rng.integers(5, len(f) - 3))] ^= 0x10
            elif kind == 2:
                f = f[:max(len(f) - 2, 0)]
            elif kind == 3:
                c = max(c - 1, 0)
            elif kind == 4:
                f[0] ^= 0xFF
            bad.append(f); bcaps.append(c)
        for fs, cs in ((frames, caps), (bad, bcaps)):
            wantd = [oracle.frame_decompress(f, c) for f, c in zip(fs, cs)]
            for nthreads in (1, 4, 0):
                gotd = hip.frame_decompress_batch(fs, cs, n_threads=nthreads)
                for i, ((ro, oo), (rg, og)) in enumerate(zip(wantd, gotd)):
                    assert rg == ro or (is_error(rg) and is_error(ro) and s64(rg) == s64(ro)), (bsid, codec, nthreads, i, rg, ro)
                    if not is_error(ro):
                        assert (og[:ro] == oo[:ro]).all(), (bsid, codec, nthreads, i)
    # a short destination for one frame of a compress batch fails that frame only
    caps = [int(hip.lib.FSEHIP_frame_compressBound(x.size, 5)) for x in srcs]
    caps[4] = 100
    got = hip.frame_compress_batch(srcs, 5, 0, caps=caps, n_threads=2)
    want = [oracle.frame_compress(x, 5, 0) for x in srcs]
    assert is_error(got[4][0])
    for i, ((r, out), (rg, og)) in enumerate(zip(want, got)):
        if i != 4:
            assert rg == r and (og[:r] == out[:r]).all()
    assert hip.frame_compress_batch([], 5, 0) == []
