import numpy as np, heapq, sys
sys.path.insert(0, "/root/repo")
from dna_factory_amd import genome

def huff_lengths(freq, maxlen=15):
    items = [(f, i) for i, f in enumerate(freq) if f > 0]
    if len(items) == 1:
        return {items[0][1]: 1}
    h = [(f, [i]) for f, i in items]
    heapq.heapify(h)
    L = {i: 0 for _, i in items}
    cnt = 0
    while len(h) > 1:
        f1, s1 = heapq.heappop(h); f2, s2 = heapq.heappop(h)
        for i in s1 + s2: L[i] += 1
        heapq.heappush(h, (f1 + f2, s1 + s2))
    return L

def tokens_hist(text):
    """current encoder: distance 4, 64-byte chunks, runs >= 3 are matches. returns hist[512] """
    n = len(text) // 64 * 64
    t = text[:n]
    m = np.zeros(n, bool)
    m[4:] = t[4:] == t[:-4]
    m = m.reshape(-1, 64)
    tb = t.reshape(-1, 64)
    hist = np.zeros(512, np.int64)
    # runs per chunk
    pad = np.zeros((m.shape[0], 66), bool); pad[:, 1:65] = m
    d = np.diff(pad.astype(np.int8), axis=1)
    for r in range(m.shape[0]):
        s = np.flatnonzero(d[r] == 1); e = np.flatnonzero(d[r] == -1)
        L = e - s
        litmask = ~m[r].copy()
        for ss, ll in zip(s, L):
            if ll >= 3: hist[256 + ll - 3] += 1
            else: litmask[ss:ss + ll] = True
        np.add.at(hist, tb[r][litmask], 1)
    return hist

def gen_cells(p, ncells, rs):
    a = (rs.random(ncells) < p); b = (rs.random(ncells) < p)
    t = np.empty((ncells, 4), np.uint8)
    t[:, 0] = 48 + a; t[:, 1] = 47; t[:, 2] = 48 + b; t[:, 3] = 9
    return t.reshape(-1)

def cost(hist, L):
    return sum(hist[i] * L[i] for i in range(512) if hist[i])

if __name__ == "__main__":
    rs = np.random.RandomState(1)
    ps = [0.01, 0.02, 0.05, 0.1, 0.2, 0.3, 0.4, 0.495]
    nc = 15360 * 4
    hs = {p: tokens_hist(gen_cells(p, nc, rs)) for p in ps}
    for p in ps:
        L = huff_lengths(hs[p])
        ent = -2 * (p * np.log2(p) + (1 - p) * np.log2(1 - p))
        print(p, "own-table bits/cell %.3f (+1 bit/match dist incl: %.3f)  entropy %.3f" % (cost(hs[p], L) / nc, (cost(hs[p], L) + hs[p][256:].sum()) / nc, ent))
    import zlib
    for p in ps:
        t = gen_cells(p, 15360 * 8, rs).tobytes()
        out = []
        for lvl in (1, 6, 9):
            tot = 0
            for k in range(0, len(t), 61440):
                c = zlib.compressobj(lvl, zlib.DEFLATED, -15)
                tot += len(c.compress(t[k:k + 61440]) + c.flush())
            out.append(tot * 8 / (len(t) / 4))
        print(p, "zlib bits/cell  l1 %.3f l6 %.3f l9 %.3f" % tuple(out))
