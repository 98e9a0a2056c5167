"""CPU model of the wave-parallel formulations used by csrc/fse_prep.hip (normalisation with its fallback as reductions and
prefix sums, NCount header as per-symbol (value, nbBits) + scan) checked against the oracle / compiled reference.
Development aid: python scripts/sim/fse_glue_sim.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.oracle import Checker, is_error

RTB = [0, 473195, 504333, 520860, 550000, 700000, 750000, 830000]
M64 = (1 << 64) - 1


def normalize_par(count, total, maxsv, tl):
    """every symbol independently + reductions (sum, first-argmax, prefix sums) -- no loop-carried state"""
    n = maxsv + 1
    c = [int(x) for x in count[:n]]
    scale = 62 - tl
    step = (1 << 62) // total
    vstep = 1 << (scale - 20)
    low = total >> tl
    norm = [0] * n
    key = []                       # (proba, -s) for the arg max over the regular symbols
    for s in range(n):
        if c[s] == 0:
            norm[s] = 0
        elif c[s] <= low:
            norm[s] = -1
        else:
            p = ((c[s] * step) & M64) >> scale
            p &= 0xFFFF
            if p >= 0x8000: p -= 0x10000
            if p < 8:
                p += 1 if ((c[s] * step) & M64) - (p << scale) > vstep * RTB[p] else 0
            norm[s] = p
            key.append((p, -s))
    still = (1 << tl) - sum(abs(x) for x in norm)
    largest = 0; largestP = 0
    if key:
        p, ms = max(key)
        if p > 0: largest, largestP = -ms, p
    if -still >= (norm[largest] >> 1):
        return m2_par(c, total, maxsv, tl)
    norm[largest] += still
    return norm


def m2_par(c, total, maxsv, tl):
    n = maxsv + 1
    ts = 1 << tl
    low = total >> tl
    one = (total * 3) >> (tl + 1)
    kind = [0 if x == 0 else (-1 if x <= low else (1 if x <= one else 2)) for x in c]     # 2 = pending
    given = sum(1 for k in kind if k in (-1, 1))
    total -= sum(x for x, k in zip(c, kind) if k in (-1, 1))
    left = ts - given
    norm = [k if k != 2 else 0 for k in kind]
    if left == 0:
        return [k if k != 2 else -2 for k in kind]       # (the reference leaves NOT_YET_ASSIGNED = -2 behind)
    if total // left > one:
        one = (total * 3) // (left * 2)
        for s in range(n):
            if kind[s] == 2 and c[s] <= one:
                kind[s] = 1; norm[s] = 1; given += 1; total -= c[s]
        left = ts - given
    if given == n:
        best = max(range(n), key=lambda s: (c[s], -s))
        norm[best] += left
        return norm
    if total == 0:
        pos = [s for s in range(n) if norm[s] > 0]
        q, r = divmod(left, len(pos))
        for i, s in enumerate(pos):
            norm[s] += q + (1 if i < r else 0)
        return norm
    vlog = 62 - tl
    mid = (1 << (vlog - 1)) - 1
    rstep = ((1 << vlog) * left + mid) // total
    run = mid                      # exclusive prefix sum of count * rstep over the pending symbols
    for s in range(n):
        if kind[s] == 2:
            end = (run + c[s] * rstep) & M64
            w = ((end >> vlog) & 0xFFFFFFFF) - ((run >> vlog) & 0xFFFFFFFF)
            if w < 1: return None
            norm[s] = w
            run = end
    return norm


def ncount_par(norm, maxsv, tl):
    """per symbol: (value, nbBits) from the exclusive prefix sum of |norm|; zero runs coded at their first zero; bit offsets by a scan"""
    n = maxsv + 1
    a = [abs(int(x)) for x in norm[:n]]
    before = [0] * n
    for s in range(1, n): before[s] = before[s - 1] + a[s - 1]
    ts = 1 << tl
    pieces = [(tl - 5, 4)]
    for s in range(n):
        rem = ts + 1 - before[s]
        if rem <= 1: break
        if norm[s] == 0 and s > 0 and norm[s - 1] == 0:
            continue                                     # inside a zero run: coded by the run's first zero
        hb = rem.bit_length() - 1
        thr, nb = 1 << hb, hb + 1
        mx = 2 * thr - 1 - rem
        v = int(norm[s]) + 1
        if v >= thr: v += mx
        pieces.append((v, nb - (1 if v < mx else 0)))
        if norm[s] == 0:
            e = s + 1
            while e < n and norm[e] == 0: e += 1
            if e == n: break                             # (invalid distribution: the reference stops here)
            R = e - (s + 1)
            pieces += [(0xFFFF, 16)] * (R // 24)
            R %= 24
            pieces += [(3, 2)] * (R // 3)
            pieces.append((R % 3, 2))
    acc = 0; nbits = 0
    for v, nb in pieces:
        acc |= v << nbits; nbits += nb
    size = (nbits + 7) // 8
    return acc.to_bytes(size + 2, "little")[:size], nbits


def main():
    chk = Checker()
    rng = np.random.default_rng(5)
    ncases = nm2 = 0
    for trial in range(6000):
        kind = trial % 6
        n = int(rng.integers(2, 257))
        if kind == 0:
            cnt = rng.integers(0, 400, n)
        elif kind == 1:
            cnt = (rng.geometric(0.05, n) - 1) * rng.integers(0, 2, n)
        elif kind == 2:
            cnt = rng.integers(0, 3, n); cnt[rng.integers(0, n)] = rng.integers(1000, 60000)
        elif kind == 3:
            cnt = np.floor(32768 * 0.5 ** np.arange(n) * rng.uniform(0.5, 1.5, n)).astype(np.int64)
        elif kind == 4:
            cnt = rng.integers(0, 2, n) * rng.integers(1, 9, n); cnt[:3] += rng.integers(0, 5000, min(3, n))
        else:
            cnt = rng.integers(1, 40, n); cnt[rng.integers(0, n, 5)] = 0
        cnt = cnt.astype(np.uint32)
        if cnt[-1] == 0: cnt[-1] = 1
        total = int(cnt.sum())
        if total < 2 or int(cnt.max()) == total: continue
        maxsv = n - 1
        full = np.zeros(256, np.uint32); full[:n] = cnt
        for tlr in (0, 5, 7, 9, 11, 12):
            tl = chk.fse_optimal_tablelog(tlr if tlr else 11, total, maxsv)
            r, ref = chk.fse_normalize_count(tl, full, total, maxsv)
            if is_error(r): continue
            mine = normalize_par(full, total, maxsv, tl)
            assert mine is not None and (np.array(mine) == ref[:n]).all(), (trial, tl, mine, ref[:n])
            ncases += 1
            if sum(abs(x) for x in mine) != (1 << tl): continue
            h, out = chk.fse_write_ncount(600, ref, maxsv, tl)
            if is_error(h): continue
            b, nbits = ncount_par(ref, maxsv, tl)
            assert len(b) == h and bytes(out[:h]) == b, (trial, tl, h, len(b))
    print("ok", ncases)


if __name__ == "__main__":
    main()
