"""The .fse frame restatement (oracle/fse_oracle.c, programs/fileio.c:266-626) pinned against the reference's own
command-line tool compiled into oracle/_ref/fse_cli, and against the committed golden frames."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle.oracle import is_error

HERE = os.path.dirname(os.path.abspath(__file__))
CLI = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "fse_cli")


@pytest.fixture(scope="module")
def oracle(restatement):
    """here the restatement itself is under test (conftest.py: every other module's `oracle` is the compiled reference where present)"""
    return restatement


def _inputs(oracle):
    rng = np.random.default_rng(5)
    yield "empty", np.zeros(0, np.uint8)
    yield "one", np.array([7], np.uint8)
    yield "tiny", rng.integers(0, 4, 100, dtype=np.uint8)
    yield "p14_exact_block", oracle.probagen_batch(14, 1, 32768, 3)[0]
    yield "p14_ragged", oracle.probagen_batch(14, 1, 100000, 4)[0]
    yield "p80_multi", oracle.probagen_batch(80, 1, 3 * 32768 + 5, 5)[0]
    yield "noise_raw", rng.integers(0, 256, 70000, dtype=np.uint8)
    yield "rle_blocks", np.concatenate([np.full(32768, 9, np.uint8), np.full(32768, 200, np.uint8), np.full(123, 1, np.uint8)])
    yield "mixed", np.concatenate([oracle.probagen_batch(20, 1, 32768, 6)[0], rng.integers(0, 256, 32768, dtype=np.uint8),
                                   np.full(32768, 5, np.uint8), oracle.probagen_batch(2, 1, 4000, 7)[0]])


def _cli_compress(data, huf):
    with tempfile.TemporaryDirectory() as d:
        a, b, c = os.path.join(d, "in"), os.path.join(d, "out.fse"), os.path.join(d, "back")
        data.tofile(a)
        subprocess.run([CLI, "-fqq" + ("h" if huf else ""), a, b], check=True, capture_output=True)
        frame = np.fromfile(b, dtype=np.uint8)
        subprocess.run([CLI, "-dfqq", b, c], check=True, capture_output=True)
        back = np.fromfile(c, dtype=np.uint8)
    return frame, back


@pytest.mark.skipif(not os.path.exists(CLI), reason="reference CLI not built (oracle/_ref/fse_cli)")
def test_oracle_frames_match_reference_cli(oracle):
    for name, data in _inputs(oracle):
        for codec in (0, 1):
            frame, back = _cli_compress(data, codec == 1)
            assert (back == data).all(), name
            r, out = oracle.frame_compress(data, 5, codec)
            assert not is_error(r), (name, codec, r)
            assert r == len(frame) and (out[:r] == frame).all(), (name, codec, r, len(frame))
            r2, out2 = oracle.frame_decompress(frame, len(data))
            assert r2 == len(data) and (out2[:r2] == data).all(), (name, codec)


def test_oracle_frame_other_block_sizes_and_errors(oracle):
    data = oracle.probagen_batch(14, 1, 150000, 9)[0]
    for bsid in range(0, 7):
        for codec in (0, 1):
            r, out = oracle.frame_compress(data, bsid, codec)
            assert not is_error(r)
            r2, out2 = oracle.frame_decompress(out[:r], len(data))
            assert r2 == len(data) and (out2[:r2] == data).all(), (bsid, codec)
    r, out = oracle.frame_compress(data, 5, 0)
    bad = out[:r].copy(); bad[r - 1] ^= 1                                      # checksum
    assert is_error(oracle.frame_decompress(bad, len(data))[0])
    bad = out[:r].copy(); bad[0] ^= 1                                          # magic
    assert is_error(oracle.frame_decompress(bad, len(data))[0])
    assert is_error(oracle.frame_decompress(out[:r], len(data) - 1)[0])        # destination too small
    assert is_error(oracle.frame_decompress(out[:r - 4], len(data))[0])        # truncated
    assert is_error(oracle.frame_compress(data, 7, 0)[0])                      # block size id out of range
    for bad, cap in oversize_frames(oracle):                                   # announced size above the frame's block size
        assert (1 << 64) - oracle.frame_decompress(bad, cap)[0] == 4           # corruption_detected


def oversize_frames(oracle):
    """frames with block-size id 0 (1 KiB) whose first block announces a regenerated size of 0xFFFF / 1025 bytes: the
    reference tool would overrun its blockSize buffers (programs/fileio.c:509-510,570); we reject them"""
    data = oracle.probagen_batch(14, 1, 5000, 21)[0]
    out = []
    for codec in (0, 1):
        r, fr = oracle.frame_compress(data, 0, codec)
        fr = fr[:r]
        assert fr[5] == 0x20, "first block is a full compressed one"            # type 0, full-size flag
        for announced in (0xFFFF, 1025):
            bad = np.concatenate([fr[:5], np.array([0x00, announced >> 8, announced & 0xFF], np.uint8), fr[6:]])
            out.append((bad, 70000))
        raw = np.concatenate([fr[:5], np.array([0x40, 0x04, 0x01], np.uint8), data[:1025], fr[-3:]])   # raw block of 1025 bytes
        out.append((raw, 70000))
    return out


def test_golden_frames(oracle, golden):
    """frames written by the reference CLI, committed in tests/golden (made by tests/golden/make_golden.py)"""
    for key in ("frame_fse", "frame_huf"):
        if key not in golden:
            pytest.skip("golden file predates the frame fixtures")
        data = golden["frame_src"]
        frame = golden[key]
        r, out = oracle.frame_compress(data, 5, 0 if key == "frame_fse" else 1)
        assert r == len(frame) and (out[:r] == frame).all(), key
        r2, out2 = oracle.frame_decompress(frame, len(data))
        assert r2 == len(data) and (out2[:r2] == data).all(), key
    assert oracle.xxh32(np.frombuffer(b"", np.uint8)) == 0x02CC5D05 and oracle.xxh32(np.frombuffer(b"a", np.uint8)) == 0x550D7456
