"""CPU model of k_fse_encode_wave's speculate / verify / repair scheme (csrc/fse_encode_wave.hip) on probagen blocks: what the map
"start state -> (bits emitted, end state)" of a super-range looks like, and how many repair rounds different policies need.
Development aid (uses the oracle for the generator and the reference's table builders only):

    python scripts/sim/repair_policies.py [P=80] [blocks=32] [warm factor=2]

Facts the kernel rests on, checked here on every super-range:
  * for a fixed run of symbols the lifted map  L(x) = bits(x) * 2^k + end(x)  is monotone non-decreasing in the start state x
    (every FSE_encodeSymbol, lib/fse.h:514-521, is a degree-1 monotone map of the circle of states: a state in the upper part of
    the symbol's interval emits one bit more and lands on the lower sub-states; compositions stay monotone);
  * it is a step function with few steps: the number of distinct (bits, end) values per super-range is printed per distribution
    (Proba80: 2 .. 8 over 1024 steps per chain, never 1 -- the reason a speculated start is wrong so often and stays wrong for long).
Policies compared (rounds = rounds of re-runs a wave of two blocks pays, the maximum over its four chains):
  current   a lane whose start differs from its predecessor's end re-runs from that end (rounds 2-5 of the build);
  cache     the same, but a lane keeps the sample it had before its last re-run and takes it back when the predecessor's end
            returns to it (look-ups ripple down the lanes between two rounds of runs)  -- in the kernel since round 6;
  bracket   cache + a start between two kept samples with equal L needs no run either (monotonicity)  -- adds nothing;
  helpers   cache (any number of samples) + bracket + the idle lanes of a round evaluate the successors of the running lanes at
            starts that split their largest unresolved arcs  -- 3.0 -> 2.6 rounds: not worth a sample store in LDS.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle.oracle import Oracle

LANES = 32


def tables(orc, src8, max_tl=11):
    _, msv, cnt = orc.hist_count(src8)
    tl = orc.fse_optimal_tablelog(max_tl, src8.size, msv)
    _, norm = orc.fse_normalize_count(tl, cnt, src8.size, msv)
    _, ct = orc.fse_build_ctable(norm, msv, tl)
    st = ct[1:1 + (1 << (tl - 1))].view(np.uint16).astype(np.int64)
    tt = ct[1 + (1 << (tl - 1)):].reshape(-1, 2)
    return tl, int((norm != 0).sum()), st, tt[:, 0].astype(np.int32).astype(np.int64), tt[:, 1].astype(np.int64)


def walk(x, js, src, st, dfs, dnb):
    """FSE_encodeSymbol over the symbols at distances js from the block end, vectorised over start states"""
    n = src.size
    bits = np.zeros_like(x)
    for j in js:
        s = src[n - 1 - j]
        nb = (x + dnb[s]) >> 16
        x = st[(x >> nb) + dfs[s]]
        bits = bits + nb
    return x, bits


def init_state(s, st, dfs, dnb):           # FSE_initCState2, lib/fse.h:503-512
    nb = (dnb[s] + (1 << 15)) >> 16
    return st[((((nb << 16) - dnb[s])) >> nb) + dfs[s]]


def block_maps(orc, src8, warm_factor):
    """per chain: full maps of every super-range (end state, lifted end, lifted mid) and the warmed-up guesses, as the kernel cuts them"""
    src = src8.astype(np.int64)
    tl, present, st, dfs, dnb = tables(orc, src8)
    T = 1 << tl
    n = src.size
    warm = int(warm_factor * T) // present
    warm = min(max((warm + 63) & ~63, 64), 4096)
    C = (((n - 2 + LANES - 1) // LANES) + 63) & ~63
    allx = np.arange(T, 2 * T)
    chains = []
    for c in range(2):
        E, LE, LM, g = [], [], [], []
        for k in range(LANES // 2):
            lo = 2 + 2 * k * C if k else 2
            mid, hi = min(n, 2 + (2 * k + 1) * C), min(n, 2 + (2 * k + 2) * C)
            if lo >= n:
                break
            x1, b1 = walk(allx.copy(), np.arange(lo + ((c - lo) % 2), mid, 2), src, st, dfs, dnb)
            x2, b2 = walk(x1, np.arange(mid + ((c - mid) % 2), hi, 2), src, st, dfs, dnb)
            E.append(x2); LE.append((b1 + b2) * 4 * T + x2); LM.append(b1 * 4 * T + x1)
            if lo <= 2 + warm:
                x0, jw = init_state(src[n - 1 - c], st, dfs, dnb), np.arange(2 + ((c - 2) % 2), lo, 2)
            else:
                x0, jw = T, np.arange(lo - warm + ((c - (lo - warm)) % 2), lo, 2)
            g.append(int(walk(np.array([x0]), jw, src, st, dfs, dnb)[0][0]))
        chains.append((E, LE, LM, g))
    return T, chains


def rounds(T, chains, cache_size, bracket, helpers):
    K = len(chains[0][0])
    end = lambda c, k, s: int(chains[c][0][k][s - T])
    cs = [[chains[c][3][k] for k in range(K)] for c in range(2)]
    ce = [[end(c, k, cs[c][k]) for k in range(K)] for c in range(2)]
    full = [[True] * K for _ in range(2)]
    samp = [[[cs[c][k]] for k in range(K)] for c in range(2)]         # starts evaluated, oldest first

    def resolve(c, k, s):
        S = samp[c][k]
        if s in S:
            return "full"
        lows, highs = [a for a in S if a < s], [a for a in S if a > s]
        if bracket and lows and highs:
            a, b = max(lows), min(highs)
            if chains[c][1][k][a - T] == chains[c][1][k][b - T]:
                return "full" if chains[c][2][k][a - T] == chains[c][2][k][b - T] else "end"
        return None

    r = 0
    while True:
        changed = True
        while changed:                                                  # look-ups ripple down the lanes for free
            changed = False
            for c in range(2):
                for k in range(1, K):
                    if cs[c][k] != ce[c][k - 1]:
                        s = ce[c][k - 1]
                        how = resolve(c, k, s)
                        if how:
                            cs[c][k], ce[c][k], full[c][k], changed = s, end(c, k, s), how == "full", True
        hard = [(c, k) for c in range(2) for k in range(1, K) if cs[c][k] != ce[c][k - 1]]
        pend = [(c, k) for c in range(2) for k in range(K) if not full[c][k] and (c, k) not in hard]
        if not hard and not pend:
            return r
        busy = set(hard) | set(pend)
        if helpers:
            idle = 2 * K - len(busy)
            targets = [(c, k + 1) for (c, k) in hard if k + 1 < K and (c, k + 1) not in busy]
            i = 0
            while idle > 0 and targets and i < 64:
                c, k = targets[i % len(targets)]; i += 1
                pts = sorted(samp[c][k]); pts.append(pts[0] + T)
                best = None
                for a, b in zip(pts[:-1], pts[1:]):
                    if b - a < 2 or (b < 2 * T and chains[c][1][k][a - T] == chains[c][1][k][b - T]):
                        continue
                    if best is None or b - a > best[1] - best[0]:
                        best = (a, b)
                if best:
                    s = (best[0] + best[1]) // 2
                    samp[c][k].append(s - T if s >= 2 * T else s); idle -= 1
        for c, k in hard:
            cs[c][k] = ce[c][k - 1]
            samp[c][k] = (samp[c][k] + [cs[c][k]])[-cache_size:] if not helpers else samp[c][k] + [cs[c][k]]
            full[c][k] = True
        for c, k in pend:
            full[c][k] = True; samp[c][k].append(cs[c][k])
        for c in range(2):
            for k in range(K):
                ce[c][k] = end(c, k, cs[c][k])
        r += 1


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    wf = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    orc = Oracle()
    blocks = orc.probagen_batch(P, nb)
    policies = {"current": (1, False, False), "cache": (2, False, False), "bracket": (2, True, False), "helpers": (99, True, True)}
    res = {p: [] for p in policies}
    values, states, mono = [], [], True
    for b in range(nb):
        T, chains = block_maps(orc, np.asarray(blocks[b]), wf)
        for c in range(2):
            for k in range(len(chains[c][0])):
                mono &= bool(np.all(np.diff(chains[c][1][k]) >= 0)) and bool(np.all(np.diff(chains[c][2][k]) >= 0))
                values.append(len(np.unique(chains[c][1][k]))); states.append(len(np.unique(chains[c][0][k])))
        for p, (cs, br, hp) in policies.items():
            res[p].append(rounds(T, chains, cs, br, hp))
    print("P%02d, %d blocks, warm factor %g: lifted map monotone on every super-range: %s" % (P, nb, wf, mono))
    print("  distinct (bits, end) values per super-range: min %d  mean %.1f  max %d;  distinct end states: min %d  mean %.1f  max %d"
          % (min(values), np.mean(values), max(values), min(states), np.mean(states), max(states)))
    for p in policies:
        a = np.array(res[p])
        pair = np.maximum(a[0::2], a[1::2]) if a.size > 1 else a
        print("  %-8s rounds per block: mean %.2f max %d;  per wave of two blocks: mean %.2f" % (p, a.mean(), a.max(), pair.mean()))


if __name__ == "__main__":
    main()
