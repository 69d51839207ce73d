import numpy as np, sys
from sim import huff_lengths
from sim2 import len_sym, dist_sym, LEN_EXTRA, DIST_EXTRA, total_bits

def parse_member(codes, C, k, nseg, ncand_back, minbr=3):
    n = len(codes)
    ll = {}; dd = {}; extra = 0
    def add(h, s): h[s] = h.get(s, 0) + 1
    key = np.zeros(n, np.int64)
    for j in range(k):
        key[: n - k + 1] |= codes[j: n - k + 1 + j].astype(np.int64) << (2 * j)
    nvalid = n - k + 1
    seglen = (n + nseg - 1) // nseg
    qmin = [dict() for _ in range(nseg)]; qmax = [dict() for _ in range(nseg)]
    for i in range(nvalid):
        q = i // seglen; kk = key[i]
        if kk not in qmin[q]: qmin[q][kk] = i
        qmax[q][kk] = i
    a = codes & 1; b = codes >> 1
    byts = np.empty((n, 4), np.uint8); byts[:, 0] = 48 + a; byts[:, 1] = 47; byts[:, 2] = 48 + b; byts[:, 3] = 9
    byts = byts.reshape(-1)
    eq = np.zeros(4 * n, bool); eq[4:] = byts[4:] == byts[:-4]
    nfar = 0; farcells = 0
    for c0 in range(0, n, C):
        c1 = min(n, c0 + C)
        covered = np.zeros(4 * (c1 - c0), bool)
        pos = c0
        while pos < c1:
            f = 0; best = -1
            if pos < nvalid and codes[pos] != codes[pos - 1] if pos else True:
              if pos < nvalid:
                q = pos // seglen; kk = key[pos]
                cl = []
                c = qmin[q].get(kk, -1)
                if 0 <= c < pos: cl.append(c)
                for back in range(1, ncand_back + 1):
                    if q - back >= 0:
                        c = qmax[q - back].get(kk, -1)
                        if c >= 0: cl.append(c)
                for s in cl:
                    if pos - s > 8192: continue
                    g = 0
                    while pos + g < c1 and codes[s + g] == codes[pos + g]: g += 1
                    if g > f: f = g; best = s
            if f >= k:
                br = (~eq[4 * pos: 4 * (pos + f)]).sum()
                if br >= minbr:
                    L = 4 * f; d = 4 * (pos - best)
                    ls = len_sym(L); ds = dist_sym(d)
                    add(ll, 257 + ls); add(dd, ds); extra += LEN_EXTRA[ls] + DIST_EXTRA[ds]
                    covered[4 * (pos - c0): 4 * (pos + f - c0)] = True
                    pos += f; nfar += 1; farcells += f
                    continue
            pos += 1
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
                    for q2 in range(i, j): add(ll, int(byts[4 * c0 + q2]))
                i = j
            else:
                add(ll, int(byts[4 * c0 + i])); i += 1
    return ll, dd, extra

if __name__ == "__main__":
    rs = np.random.RandomState(2)
    n = 15360
    C = 64
    configs = [(5, 4, 1), (5, 8, 1), (5, 8, 2), (5, 16, 2), (6, 4, 1), (6, 8, 2), (6, 4, 2)]
    print("configs (k, nseg, back):", configs)
    for p in [0.05, 0.1, 0.2, 0.3, 0.495]:
        res = []
        codes = ((rs.random(n) < p).astype(np.uint8) | ((rs.random(n) < p).astype(np.uint8) << 1))
        for k, nseg, back in configs:
            ll, dd, ex = parse_member(codes, C, k, nseg, back)
            res.append(total_bits(ll, dd, ex) / n)
        print("p=%.3f " % p + " ".join("%.3f" % r for r in res))
