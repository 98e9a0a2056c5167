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
