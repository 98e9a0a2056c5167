"""CPU model of the restated Huffman table build used by csrc/huf_prep.hip (two-queue merge on sorted keys, depths by parent
chasing, the height limit as moves of the boundaries between length classes, canonical values by per-length ranks, weights
header) checked against the compiled reference.  Development aid: python scripts/sim/huf_glue_sim.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.oracle import Checker, is_error

MAXTL = 12


def lengths_by_rank(counts_sorted):
    """counts_sorted: descending counts of the L present symbols.  Returns depth per rank (two-queue merge, ties -> internal)."""
    L = len(counts_sorted)
    par = [0] * (512)
    icnt = []
    li, ii = L - 1, 0
    NONE_LEAF, NONE_INT = 1 << 31, 1 << 30
    while len(icnt) < L - 1:
        ni = len(icnt); total = 0
        for _ in range(2):
            leaf = counts_sorted[li] if li >= 0 else NONE_LEAF
            inn = icnt[ii] if ii < ni else NONE_INT
            if leaf < inn:
                total += leaf; par[li] = ni; li -= 1
            else:
                total += inn; par[256 + ii] = ni; ii += 1
        icnt.append(total)
    root = L - 2
    depth = []
    for r in range(L):
        d, p = 1, par[r]
        while p != root:
            p = par[256 + p]; d += 1
        depth.append(d)
    return depth


def limit_height(nb, cnt, M):
    """nb: lengths by rank (non-decreasing), cnt: counts by rank (descending), M: limit.  Classes of equal length are contiguous;
    the repair moves class boundaries: `last[k]` = last rank of length M-k (None if empty)."""
    L = len(nb)
    largest = nb[L - 1]
    if largest <= M:
        return largest
    # debt of the truncation, in units of 2^-M
    debt = 0
    for r in range(L):
        if nb[r] > M:
            debt += (1 << (largest - M)) - (1 << (largest - nb[r])); nb[r] = M
    debt >>= (largest - M)
    last = [None] * (MAXTL + 2)
    for r in range(L):
        if nb[r] < M: last[M - nb[r]] = r                     # ranks ascend: the final write is the last of its class
    n = max([r for r in range(L) if nb[r] < M], default=-1)     # last rank shorter than M
    while debt > 0:
        k = debt.bit_length()                                  # a move out of class k pays 2^(k-1)
        while k > 1:
            hi, lo = last[k], last[k - 1]
            if hi is None: k -= 1; continue
            if lo is None: break
            if cnt[hi] <= 2 * cnt[lo]: break
            k -= 1
        while k <= MAXTL and last[k] is None: k += 1
        debt -= 1 << (k - 1)
        if last[k - 1] is None: last[k - 1] = last[k]
        nb[last[k]] += 1
        if last[k] == 0: last[k] = None
        else:
            last[k] -= 1
            if nb[last[k]] != M - k: last[k] = None
    while debt < 0:                                            # overpaid: the first ranks of length M go back to M-1
        if last[1] is None:
            while nb[n] == M: n -= 1
            nb[n + 1] -= 1; last[1] = n + 1
        else:
            nb[last[1] + 1] -= 1; last[1] += 1
        debt += 1
    return M


def build(count, maxsv, M):
    n = maxsv + 1
    keys = sorted([((int(count[s]) << 9) | (1 << 8) | (255 - s)) for s in range(n) if count[s]], reverse=True)
    cnt = [k >> 9 for k in keys]; sym = [255 - (k & 255) for k in keys]
    nb = lengths_by_rank(cnt)
    M2 = limit_height(nb, cnt, M)
    nbsym = [0] * n
    for r, s in enumerate(sym): nbsym[s] = nb[r]
    per = [0] * (MAXTL + 2)
    for s in range(n): per[nbsym[s]] += 1
    start = [0] * (MAXTL + 2); m = 0
    for L_ in range(M2, 0, -1):
        start[L_] = m; m = (m + per[L_]) >> 1
    seen = [0] * (MAXTL + 2); celt = [0] * 256
    for s in range(n):
        celt[s] = (start[nbsym[s]] + seen[nbsym[s]]) | (nbsym[s] << 16); seen[nbsym[s]] += 1
    return M2, celt


def main():
    chk = Checker(); rng = np.random.default_rng(3)
    ok = 0; limited = 0
    for trial in range(3000):
        n = int(rng.integers(2, 257)); kind = trial % 5
        if kind == 0: c = rng.integers(0, 500, n)
        elif kind == 1: c = np.floor(40000 * 0.93 ** np.arange(n) * rng.uniform(0.7, 1.3, n)).astype(np.int64)
        elif kind == 2: c = (rng.geometric(0.01, n)) * rng.integers(0, 2, n)
        elif kind == 3: c = np.array([int(1.6 ** (i % 24)) for i in range(n)])          # deep trees
        else: c = rng.integers(1, 4, n)
        c = c.astype(np.uint32)
        if c[-1] == 0: c[-1] = 1
        if (c > 0).sum() < 2: continue
        full = np.zeros(256, np.uint32); full[:n] = c
        total = int(full.sum())
        for Mreq in (11, 12, 8, 5):
            M = chk.fse_optimal_tablelog(Mreq, total, n - 1, 1)        # HUF_optimalTableLog: never below what the alphabet needs
            r, ref = chk.huf_build_ctable(full, n - 1, M)
            if is_error(r): continue
            M2, celt = build(full, n - 1, M)
            assert M2 == r and (np.array(celt, np.uint32)[:n] == ref[:n]).all(), (trial, M, M2, r)
            ok += 1
    print("ok", ok)


if __name__ == "__main__":
    main()


# ---- weights header (HUF_writeCTable): model of the wave formulation -------------------------------------------------
def tiny_ctable(norm, maxw, tl):
    """table of <= 64 states, one spread visit per lane: visit m lands on cell (m * step) mod size; the k-th kept visit belongs to
    the symbol whose cumulative range holds k; rank of a cell inside its symbol = kept cells of that symbol below it"""
    ts = 1 << tl; step = (ts >> 1) + (ts >> 3) + 3; mask = ts - 1
    lows = [s for s in range(maxw + 1) if norm[s] == -1]
    high = ts - 1 - len(lows)
    cell = [None] * ts
    for j, s in enumerate(lows): cell[ts - 1 - j] = s
    cum = []; run = 0
    for s in range(maxw + 1):
        cum.append(run); run += norm[s] if norm[s] > 0 else 0
    kept = [((m * step) & mask) <= high for m in range(ts)]
    k = 0
    for m in range(ts):
        u = (m * step) & mask
        if kept[m]:
            s = max(t for t in range(maxw + 1) if norm[t] > 0 and cum[t] <= k)
            cell[u] = s; k += 1
    first = []; run = 0
    for s in range(maxw + 1):
        first.append(run); run += 1 if norm[s] == -1 else norm[s]
    st = [0] * ts
    for u in range(ts):
        s = cell[u]; r = sum(1 for v in range(u) if cell[v] == s)
        st[first[s] + r] = ts + u
    tt = []
    total = 0
    for s in range(maxw + 1):
        nn = norm[s]
        if nn == 0: tt.append((0, ((tl + 1) << 16) - ts))
        elif nn in (-1, 1): tt.append((total - 1, (tl << 16) - ts)); total += 1
        else:
            mbo = tl - ((nn - 1).bit_length() - 1)
            tt.append((total - nn, (mbo << 16) - (nn << mbo))); total += nn
    return st, tt


def tans_encode(src, st, tt, tl, cap):
    n = len(src)
    if n <= 2 or cap <= 8: return 0, b""
    def init(sym):
        dfs, dnb = tt[sym]; nbo = (dnb + (1 << 15)) >> 16
        return st[(((nbo << 16) - dnb) >> nbo) + dfs]
    ch = [init(src[n - 1]), init(src[n - 2])]
    acc = 0; nbits = 0
    for j in range(2, n):
        dfs, dnb = tt[src[n - 1 - j]]; x = ch[j & 1]
        nb = (x + dnb) >> 16
        acc |= (x & ((1 << nb) - 1)) << nbits; nbits += nb
        ch[j & 1] = st[(x >> nb) + dfs]
    for c in ((ch[1], ch[0]) if n & 1 else (ch[0], ch[1])):
        acc |= (c & ((1 << tl) - 1)) << nbits; nbits += tl
    acc |= 1 << nbits; nbits += 1
    if (nbits >> 3) >= cap - 8: return 0, b""
    size = (nbits + 7) >> 3
    return size, acc.to_bytes(size, "little")


def write_ctable(celt, maxsv, hufflog, cap):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fse_glue_sim as F
    w = []
    for s in range(maxsv):
        nb = (celt[s] >> 16) & 0xFF
        w.append(hufflog + 1 - nb if nb else 0)
    hs = 0; body = b""
    n = len(w)
    if n > 1:
        cnt = [0] * 13
        for x in w: cnt[x] += 1
        maxw = max(i for i in range(13) if cnt[i]); top = max(cnt)
        if top == n: hs = 1
        elif top == 1: hs = 0
        else:
            tl = max(min(6, (n - 1).bit_length() - 1 - 2), min(n.bit_length(), maxw.bit_length() - 1 + 2)); tl = min(max(tl, 5), 12)
            full = np.zeros(256, np.uint32); full[:13] = cnt
            norm = F.normalize_par(full, n, maxw, tl)
            hb, nbits = F.ncount_par(norm, maxw, tl)
            st, tt = tiny_ctable(norm, maxw, tl)
            cs, cb = tans_encode(w, st, tt, tl, cap - 1 - len(hb))
            if cs: hs = len(hb) + cs; body = hb + cb
    if hs > 1 and hs < maxsv // 2:
        return hs + 1, bytes([hs]) + body
    if maxsv > 128: return None, b""
    if (maxsv + 1) // 2 + 1 > cap: return None, b""
    w = w + [0]
    out = bytes([128 + maxsv - 1]) + bytes([(w[i] << 4) + w[i + 1] for i in range(0, maxsv, 2)])
    return (maxsv + 1) // 2 + 1, out


def main2():
    chk = Checker(); rng = np.random.default_rng(4); ok = fse = 0
    for trial in range(2500):
        n = int(rng.integers(2, 257)); kind = trial % 4
        if kind == 0: c = rng.integers(0, 500, n)
        elif kind == 1: c = np.floor(40000 * 0.93 ** np.arange(n) * rng.uniform(0.7, 1.3, n)).astype(np.int64)
        elif kind == 2: c = np.floor(655 * 0.98 ** np.arange(n)).astype(np.int64) + 1
        else: c = rng.integers(1, 4, n)
        c = c.astype(np.uint32)
        if c[-1] == 0: c[-1] = 1
        if (c > 0).sum() < 2: continue
        full = np.zeros(256, np.uint32); full[:n] = c
        M = chk.fse_optimal_tablelog(11 if trial % 3 else 12, int(full.sum()), n - 1, 1)
        r, celt = chk.huf_build_ctable(full, n - 1, M)
        for cap in (300, 40, 12, 9, 5):
            h, ref = chk.huf_write_ctable(cap, celt, n - 1, r)
            mh, mine = write_ctable([int(x) for x in celt], n - 1, r, cap)
            if is_error(h):
                assert mh is None, (trial, cap, h, mh)
            else:
                assert mh == h and bytes(ref[:h]) == mine, (trial, cap, h, mh)
                ok += 1; fse += mine[0] < 128
    print("hdr ok", ok, "fse-coded", fse)


if __name__ == "__main__":
    main2()
