"""cell-domain parse simulation: per chunk of C cells, far matches from hashed k-cell keys + dist-4 byte mask coding."""
import numpy as np, sys, heapq
from sim import huff_lengths

LEN_BASE = [3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258]
LEN_EXTRA = [0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0]
DIST_BASE = [1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577]
DIST_EXTRA = [0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13]
def len_sym(L):
    for i in range(28, -1, -1):
        if L >= LEN_BASE[i]: return i
def dist_sym(d):
    for i in range(29, -1, -1):
        if d >= DIST_BASE[i]: return i

def parse_member(codes, C, k, minfar, mode="latest"):
    """codes: array of cell codes (0..3) for one member (one row). returns (litlen_hist dict, dist_hist dict, extra_bits)"""
    n = len(codes)
    ll = {}; dd = {}; extra = 0
    def add(h, s): h[s] = h.get(s, 0) + 1
    # keys
    key = np.zeros(n, np.int64)
    for j in range(k):
        key[: n - k + 1] |= codes[j: n - k + 1 + j].astype(np.int64) << (2 * j)
    valid = np.zeros(n, bool); valid[: n - k + 1] = True
    # latest previous occurrence
    cand = np.full(n, -1, np.int64)
    if mode == "latest":
        last = {}
        for i in range(n - k + 1):
            kk = key[i]
            if kk in last: cand[i] = last[kk]
            last[kk] = i
    else:  # min/max tables over halves
        half = n // 2
        t1min = {}; t1max = {}; t2min = {}
        for i in range(n - k + 1):
            kk = key[i]
            if i < half:
                if kk not in t1min: t1min[kk] = i
                t1max[kk] = i
            elif kk not in t2min: t2min[kk] = i
        for i in range(n - k + 1):
            kk = key[i]
            if i < half:
                c = t1min.get(kk, -1)
                if 0 <= c < i: cand[i] = c
            else:
                c = t2min.get(kk, -1)
                if 0 <= c < i: cand[i] = c
                else:
                    c = t1max.get(kk, -1)
                    if c >= 0 and i - c <= 8192: cand[i] = c
    # byte arrays for mask coding
    a = codes & 1; b = codes >> 1
    byts = np.empty((n, 4), np.uint8); byts[:, 0] = 48 + a; byts[:, 1] = 47; byts[:, 2] = 48 + b; byts[:, 3] = 9
    byts = byts.reshape(-1)
    eq = np.zeros(4 * n, bool); eq[4:] = byts[4:] == byts[:-4]
    for c0 in range(0, n, C):
        c1 = min(n, c0 + C)
        covered = np.zeros(4 * (c1 - c0), bool)
        pos = c0
        while pos < c1:
            f = 0
            if valid[pos] and cand[pos] >= 0 and pos - cand[pos] <= 8192:
                s = cand[pos]
                f = 0
                while pos + f < c1 and codes[s + f] == codes[pos + f]: f += 1
            # dist-4 breaks within the far span
            if f >= minfar:
                br = (~eq[4 * pos: 4 * (pos + f)]).sum()
                if br >= 3:
                    L = 4 * f; d = 4 * (pos - cand[pos])
                    ls = len_sym(L); ds = dist_sym(d)
                    add(ll, 257 + ls); add(dd, ds); extra += LEN_EXTRA[ls] + DIST_EXTRA[ds]
                    covered[4 * (pos - c0): 4 * (pos + f - c0)] = True
                    pos += f
                    continue
            pos += 1
        # remaining bytes by dist-4 mask coding: runs of eq (not covered) >= 3 are matches
        m = eq[4 * c0: 4 * c1] & ~covered
        i = 0; nb = 4 * (c1 - c0)
        while i < nb:
            if covered[i]: i += 1; continue
            if m[i]:
                j = i
                while j < nb and m[j]: j += 1
                if j - i >= 3:
                    ls = len_sym(j - i); add(ll, 257 + ls); add(dd, 3); extra += LEN_EXTRA[ls]
                else:
                    for q in range(i, j): add(ll, int(byts[4 * c0 + q]))
                i = j
            else:
                add(ll, int(byts[4 * c0 + i])); i += 1
    return ll, dd, extra

def total_bits(ll, dd, extra):
    def hb(h):
        keys = sorted(h); f = [h[x] for x in keys]
        L = huff_lengths(f)
        return sum(f[i] * L[i] for i in range(len(f)))
    return hb(ll) + hb(dd) + extra

if __name__ == "__main__":
    rs = np.random.RandomState(2)
    n = 15360
    C = int(sys.argv[1]); mode = sys.argv[2]
    for p in [0.01, 0.05, 0.1, 0.2, 0.3, 0.495]:
        res = []
        for k, minfar in ((4, 4), (5, 5), (6, 6), (8, 8)):
            LL = {}; DD = {}; EX = 0
            for rep in range(2):
                codes = ((rs.random(n) < p).astype(np.uint8) | ((rs.random(n) < p).astype(np.uint8) << 1))
                ll, dd, ex = parse_member(codes, C, k, minfar, mode)
                for x, v in ll.items(): LL[x] = LL.get(x, 0) + v
                for x, v in dd.items(): DD[x] = DD.get(x, 0) + v
                EX += ex
            res.append(total_bits(LL, DD, EX) / (2 * n))
        print("p=%.3f C=%d %s  bits/cell k=4: %.3f k=5: %.3f k=6: %.3f k=8: %.3f" % ((p, C, mode) + tuple(res)))
