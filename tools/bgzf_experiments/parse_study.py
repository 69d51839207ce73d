"""Where the genotype-domain encoder's size goes, against zlib: a CPU model of its token stream on one member of
15,360 cells of i.i.d. alleles (one row, no row boundary), priced with the entropy of the member's own symbol statistics
(what an optimal per-member Huffman table would cost, header aside), for variants of the match search:

  base      the kernel's rules: far matches start at cells that differ from their predecessor and have >= 3 differing
            alleles in the next 16 cells, >= 5 cells long, two candidates (earliest occurrence of the 5-cell key in the
            cell's own 5,120-cell segment, latest in the segment before), never across a 64-cell chunk boundary
  nochunk   matches may run across chunk boundaries
  allcand   every earlier occurrence of the 5-cell key within 8,192 cells is a candidate (longest wins, nearer on ties)
  both      nochunk + allcand
  lazy      both + one step of lazy evaluation (a match is dropped when the next startable cell has a longer one)
  min4      both, with a 4-cell key and matches from 4 cells on
  near4 / near8 / near16   both, with only the 4 / 8 / 16 nearest earlier occurrences of the key as candidates

Everything between far matches is what the kernel emits: a literal per differing allele, and for the bytes from one
event to the next one match at distance 4 (three bytes or more) or the separator's literal.  zlib (raw deflate of the
same 61,440 bytes) at levels 1 / 6 / 9 beside it.  usage: python tools/bgzf_experiments/parse_study.py [members per MAF]"""
import math
import sys
import zlib
from collections import Counter

import numpy as np

NCELL, SEG, CHUNK, WINDOW = 15360, 5120, 64, 8192
LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0] * 8 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193,
             12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0] + [k // 2 for k in range(2, 28)]


def len_sym(n):
    k = max(i for i, b in enumerate(LEN_BASE) if b <= n)
    return 257 + k, LEN_EXTRA[k]


def dist_sym(d):
    k = max(i for i, b in enumerate(DIST_BASE) if b <= d)
    return k, DIST_EXTRA[k]


def match_len(cells, a, b, limit):
    n = 0
    while n < limit and cells[a + n] == cells[b + n]:
        n += 1
    return n


def find_matches(cells, variant):
    key_cells = 4 if variant == "min4" else 5
    all_cand = variant != "base" and variant != "nochunk"
    chunked = variant in ("base", "allcand")
    nearest = int(variant[4:]) if variant.startswith("near") else 0
    n = len(cells)
    keys = [tuple(cells[j:j + key_cells]) for j in range(n - key_cells + 1)]
    brk = [j == 0 or cells[j] != cells[j - 1] for j in range(n)]
    nbrk = [0] * n  # differing alleles in the 16 cells from j
    diff = [2 if j == 0 else bin(cells[j] ^ cells[j - 1]).count("1") for j in range(n)]
    for j in range(n):
        nbrk[j] = sum(diff[j:j + 16])
    occ = {}       # key -> positions so far (ascending)
    seg_first = {}  # (segment, key) -> earliest position
    seg_last = {}   # (segment, key) -> latest position
    for j, k in enumerate(keys):
        seg_first.setdefault((j // SEG, k), j)
        seg_last[(j // SEG, k)] = j

    def best(j):
        if j + key_cells > n or not brk[j] or nbrk[j] < 3:
            return 0, 0
        room = (CHUNK - j % CHUNK) if chunked else min(64, n - j)
        if room < key_cells:
            return 0, 0
        k = keys[j]
        if all_cand:
            cands = [p for p in occ.get(k, ()) if j - p <= WINDOW]
            if nearest:
                cands = cands[-nearest:]
        else:
            q = j // SEG
            cands = [p for p in (seg_first.get((q, k)), seg_last.get((q - 1, k))) if p is not None and p < j]
        f, d = 0, 0
        for p in cands:
            m = match_len(cells, p, j, min(room, n - j))
            if m > f or (m == f and m and j - p < d):
                f, d = m, j - p
        if f < key_cells or sum(diff[j:j + min(f, 16)]) < 3:
            return 0, 0
        return f, d

    far = {}  # start cell -> (cells, distance in cells)
    filled = 0  # occ holds the positions below `filled`

    def fill(upto):
        nonlocal filled
        while filled < upto and filled <= n - key_cells:
            occ.setdefault(keys[filled], []).append(filled)
            filled += 1

    j = 0
    while j < n:
        fill(j)
        f, d = best(j)
        if f and variant == "lazy":
            j2 = j + 1
            while j2 < j + f and not brk[j2]:
                j2 += 1
            if j2 < j + f:
                fill(j2)
                f2, d2 = best(j2)
                if f2 > f:  # the cells in between stay literals
                    j, f, d = j2, f2, d2
        if f:
            far[j] = (f, d)
            j += f
        else:
            j += 1
    return far


def tokens(cells, far):
    """the kernel's token stream: symbols (literal / length) and distance symbols with their extra bits"""
    sym, dsym, extra = Counter(), Counter(), 0
    n = len(cells)
    text_at = 0  # bytes emitted so far

    def gap(upto):
        nonlocal text_at, extra
        L = upto - text_at
        if L <= 0:
            return
        if L < 3:  # one separator, or (behind a far match or a chunk boundary) an allele that repeats and its separator
            for b in range(text_at, upto):
                sym["/" if b % 4 == 1 else "\t" if b % 4 == 3 else str((cells[b // 4] >> ((b % 4) // 2)) & 1)] += 1
        else:
            while L > 0:
                piece = min(L, 258)
                if 0 < L - piece < 3:
                    piece -= 3
                s, e = len_sym(piece)
                sym[s] += 1
                extra += e
                dsym[dist_sym(4)[0]] += 1
                L -= piece
        text_at = upto

    j = 0
    while j < n:
        if j and j % CHUNK == 0:
            gap(4 * j)  # a chunk's tokens end with the chunk (one thread each)
        if j in far:
            f, d = far[j]
            gap(4 * j)
            s, e = len_sym(4 * f)
            sym[s] += 1
            extra += e
            ds, de = dist_sym(4 * d)
            dsym[ds] += 1
            extra += de
            text_at = 4 * (j + f)
            j += f
            continue
        prev = cells[j - 1] if j else 3 ^ cells[j]  # the member's first cell: both alleles are literals
        for h in (0, 1):  # allele a at byte 4 j, allele b at byte 4 j + 2
            if ((cells[j] ^ prev) >> h) & 1 or j == 0:
                gap(4 * j + 2 * h)
                sym["0" if not (cells[j] >> h) & 1 else "1"] += 1
                text_at = 4 * j + 2 * h + 1
        j += 1
    gap(4 * n)
    return sym, dsym, extra


def entropy_bits(counter):
    tot = sum(counter.values())
    return -sum(c * math.log2(c / tot) for c in counter.values()) if tot else 0.0


def cost(cells, variant):
    far = find_matches(cells, variant)
    sym, dsym, extra = tokens(cells, far)
    sym["eob"] += 1
    covered = sum(f for f, _ in far.values())
    return (entropy_bits(sym) + entropy_bits(dsym) + extra) / len(cells), covered / len(cells), len(far)


def text_of(cells):
    t = np.empty((len(cells), 4), np.uint8)
    t[:, 0] = 48 + (cells & 1)
    t[:, 1] = ord("/")
    t[:, 2] = 48 + (cells >> 1)
    t[:, 3] = 9
    return t.tobytes()


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    variants = ["base", "nochunk", "allcand", "both", "lazy", "min4", "near4", "near8", "near16"]
    rs = np.random.RandomState(11)
    print("bits per cell (share of cells inside far matches); MAF = p")
    for p in (0.03, 0.10, 0.25, 0.45):
        acc = {v: [0.0, 0.0] for v in variants}
        z = {1: 0.0, 6: 0.0, 9: 0.0}
        for _ in range(reps):
            a = (rs.rand(NCELL, 2) < p).astype(np.int64)
            cells = a[:, 0] | (a[:, 1] << 1)
            cl = [int(x) for x in cells]
            for v in variants:
                b, cov, _ = cost(cl, v)
                acc[v][0] += b / reps
                acc[v][1] += cov / reps
            txt = text_of(cells)
            for lvl in z:
                c = zlib.compressobj(lvl, zlib.DEFLATED, -15, 9)
                z[lvl] += (len(c.compress(txt)) + len(c.flush())) * 8 / NCELL / reps
        print("p=%.2f  " % p + "  ".join("%s %.3f (%.0f%%)" % (v, acc[v][0], 100 * acc[v][1]) for v in variants)
              + "   zlib -1 %.3f  -6 %.3f  -9 %.3f" % (z[1], z[6], z[9]), flush=True)


if __name__ == "__main__":
    main()
