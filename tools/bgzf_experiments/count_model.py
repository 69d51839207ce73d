"""Python model of gtz_tokens.hip.h: gtz_count_bits (mask arithmetic) against gtz_walk with a counting sink, on random event
configurations of a chunk (any j0 / j1, head cells, far matches, forced literals).  It found the one case the first version of the
kernel missed (the last alleles of a 64-cell chunk).  usage: python tools/bgzf_experiments/count_model.py  -> "bad 0" """
import random
def low_mask(n): return (1<<n)-1
def next_bit(x, frm):
    t = x >> frm << frm if frm < 128 else 0
    if frm >= 128 or t == 0: return 128
    return (t & -t).bit_length()-1
def gen(rnd):
    j0 = rnd.randint(0, 63); j1 = rnd.randint(j0+1, 64)
    if rnd.random() < 0.5: j0, j1 = 0, 64
    head = rnd.random() < 0.3
    V = rnd.getrandbits(128)
    dens = rnd.choice([0.02, 0.1, 0.3, 0.5, 0.8])
    x = 0
    for b in range(2*j0, 2*j1):
        if rnd.random() < dens: x |= 1 << b
    if head: x |= 3 << (2*j0)
    ev = x; far = 0; ends = 0; farl = []
    pos = j0
    # random far matches starting at break cells
    while pos + 5 <= j1:
        cellbrk = (x >> (2*pos)) & 3
        if cellbrk and rnd.random() < 0.3:
            f = rnd.randint(5, min(64, j1 - pos))
            m = (low_mask(2*(pos+f)) & ~low_mask(2*pos))
            ev &= ~m; far |= 1 << (2*pos)
            e = pos + f
            if e < 64: ends |= 1 << (2*e)
            farl.append((pos, f)); pos += f
        else: pos += 1
    valid = low_mask(2*j1) & ~low_mask(2*j0)
    starts = ends
    s0 = j0+1 if head else j0
    if s0 < 64: starts |= 1 << (2*s0)
    ev |= (ev >> 1) & starts & valid
    ev |= far
    return j0, j1, head, V, ev, far, ends, farl
def gapn(L, at, G): 
    return G[(L, (at & 2) >> 1)]
def walk(j0, j1, head, V, ev, far, ends, farl, G, litn, farbits):
    at = 4*j0; bits = 0
    if head and not ((far >> (2*j0)) & 1):
        code = (V >> (2*j0)) & 3
        bits += litn[code & 1] + litn[2] + litn[code >> 1] + litn[3]
        ev &= ~(3 << (2*j0)); at += 4
    T = ev; fi = 0; E = ends
    while T:
        b = (T & -T).bit_length()-1; T &= T-1
        byte = 2*b; L = byte - at
        if (far >> b) & 1:
            if E: e2 = (E & -E).bit_length()-1; E &= E-1
            else: e2 = 128
            bits += gapn(L, at, G) + farbits
            at = 2*e2
        else:
            bits += gapn(L, at, G) + litn[(V >> b) & 1]
            at = byte + 1
    bits += gapn(4*j1 - at, at, G)
    return bits
def fast(j0, j1, head, V, ev, far, ends, farl, G, litn, farbits):
    bits = farbits * len(farl); s0 = j0
    if head and not ((far >> (2*j0)) & 1):
        code = (V >> (2*j0)) & 3
        bits += litn[code & 1] + litn[2] + litn[code >> 1] + litn[3]
        ev &= ~(3 << (2*j0)); s0 = j0+1
    X = ev
    if j1 < 64: X |= 1 << (2*j1)
    end2 = 2*j1
    B = ev & ~far
    n1 = bin(B & V).count('1'); n0 = bin(B).count('1') - n1
    bits += n0*litn[0] + n1*litn[1]
    s1 = X >> 1; s2 = X >> 2
    a1 = B & s1
    EVEN = int('01'*64, 2)
    ns = bin(a1 & EVEN).count('1'); nt = bin(a1).count('1') - ns
    bits += ns*litn[2] + nt*litn[3]
    a2 = B & ~s1 & s2
    bits += bin(a2).count('1') * G[(3,0)]
    R = B & ~s1 & ~s2
    while R:
        p = (R & -R).bit_length()-1; R &= R-1
        nx = next_bit(X, p+1)
        L = 2*((nx if nx < 128 else end2) - p) - 1
        bits += G[(L, p & 1)]
    Z = ends
    while Z:
        z = (Z & -Z).bit_length()-1; Z &= Z-1
        nx = next_bit(X, z)
        bits += G[(2*((nx if nx < 128 else end2) - z), 0)]
    nx = next_bit(X, 2*s0)
    bits += G[(2*((nx if nx < 128 else end2) - 2*s0), 0)]
    return bits
rnd = random.Random(1)
G = {}
for L in range(0, 257):
    for h in (0,1):
        G[(L,h)] = 0 if L == 0 else (3+h if L == 1 else (1000 if L == 2 else 5 + (L % 7)))
litn = [2, 3, 3, 4]
bad = 0
for it in range(200000):
    c = gen(rnd)
    a = walk(*c, G, litn, 23); b = fast(*c, G, litn, 23)
    if a != b:
        bad += 1
        if bad < 5: print("MISMATCH", a, b, c[:3], hex(c[4]), hex(c[5]), hex(c[6]), c[7])
print("bad", bad)
