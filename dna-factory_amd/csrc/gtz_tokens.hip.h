// gtz_tokens.hip.h -- search and tokenisation of one 64-cell chunk for the genotype-domain BGZF encoder,
// written so that the lanes of a wave stay together: the chunk's state is a handful of bit masks in
// registers, loops run over events (breaks, tokens) rather than cells, and literals need no LDS access.
//
// Byte view of a chunk: 256 text bytes = 4 words of 64 bits, bit i of word w = byte 64 w + i.  Byte i of a
// cell-aligned word belongs to cell i >> 2; bytes 0 and 2 of a cell are the alleles, 1 is '/', 3 is '\t'.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dnaf {

// spread the 32 bits of v to the even... every 2-bit group of v to the low half of a nibble of the result
__device__ __forceinline__ uint64_t gtz_spread_pairs(uint32_t v)
{
    uint64_t t = v;
    t = (t | (t << 16)) & 0x0000FFFF0000FFFFull;
    t = (t | (t << 8)) & 0x00FF00FF00FF00FFull;
    t = (t | (t << 4)) & 0x0F0F0F0F0F0F0F0Full;
    t = (t | (t << 2)) & 0x3333333333333333ull;
    return t;
}

// bits [lo, hi) of a 256-bit mask, word w of it
__device__ __forceinline__ uint64_t gtz_range_word(uint32_t lo, uint32_t hi, uint32_t w)
{
    const uint32_t base = 64u * w;
    const uint32_t a = lo > base ? lo - base : 0u, b = hi > base ? hi - base : 0u;
    return gtz_low_mask(min(b, 64u)) & ~gtz_low_mask(min(a, 64u));
}

// Token masks of a chunk.  run3: bytes inside distance-4 matches (runs of >= 3 repeating bytes outside the
// LZ77 matches); plain: bytes that are tokens of their own or inside a distance-4 match (valid and not
// covered by an LZ77 match); far: first byte of each recorded LZ77 match.
struct GtzTokens {
    uint64_t run3[4], plain[4], far[4];
    uint32_t ext[4]; // ext[w]: bytes by which a distance-4 run that reaches the end of word w goes on in the words after it
};

__device__ __forceinline__ GtzTokens gtz_tokens(const GtzChunk& c, const uint32_t* farlist)
{
    GtzTokens t;
    uint64_t M[4];
    const uint32_t lo = 4u * c.j0, hi = 4u * c.j1;
    const uint32_t xs[4] = {(uint32_t)c.xlo, (uint32_t)(c.xlo >> 32), (uint32_t)c.xhi, (uint32_t)(c.xhi >> 32)};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint64_t s = gtz_spread_pairs(xs[w]);                                     // [xa, xb, 0, 0] per cell
        const uint64_t brk = (s & 0x1111111111111111ull) | ((s & 0x2222222222222222ull) << 1); // allele bytes that differ
        t.plain[w] = gtz_range_word(lo, hi, (uint32_t)w);
        M[w] = ~brk & t.plain[w];
        t.far[w] = 0;
    }
    if (c.head) { // the region's first cell: its '/' and '\t' have nothing to repeat either
#pragma unroll
        for (int w = 0; w < 4; ++w) M[w] &= ~gtz_range_word(lo, lo + 4u, (uint32_t)w);
    }
    for (uint32_t n = 0; n < GTZ_MAX_FAR; ++n) {
        const uint32_t e = farlist[n];
        if (!(e >> 6)) break;
        const uint32_t a = lo + 4u * (e & 63u), b = a + 4u * ((e >> 6) & 127u);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t r = gtz_range_word(a, b, (uint32_t)w);
            t.plain[w] &= ~r;
            M[w] &= ~r;
            t.far[w] |= gtz_range_word(a, a + 1u, (uint32_t)w);
        }
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint64_t below = w ? M[w - 1] : 0ull, above = w < 3 ? M[w + 1] : 0ull;
        const uint64_t up = (M[w] << 1) | (below >> 63), up2 = (M[w] << 2) | (below >> 62);
        const uint64_t dn = (M[w] >> 1) | (above << 63), dn2 = (M[w] >> 2) | (above << 62);
        t.run3[w] = M[w] & ((up & up2) | (up & dn) | (dn & dn2));
    }
    // worked out once here, not at every match of the two token walks
    t.ext[3] = 0;
#pragma unroll
    for (int w = 2; w >= 0; --w) {
        const uint32_t more = ~t.run3[w + 1] ? (uint32_t)__builtin_ctzll(~t.run3[w + 1]) : 64u;
        t.ext[w] = more + (more == 64u ? t.ext[w + 1] : 0u);
    }
    return t;
}

// Sinks receive literals as a kind (0 '0', 1 '1', 2 '/', 3 '\t') and matches as (length, distance in bytes).
struct GtzBitCount {
    uint32_t lit_bits;     // 4 x 8 bits: code length of each literal kind
    const uint32_t* len;   // LDS: match length -> code + extra bits, total bit count in the low 5 bits
    const uint32_t* dist;  // LDS: distance symbol -> code, its length in the low 5 bits
    uint32_t bits;
    __device__ __forceinline__ void lit(uint32_t kind) { bits += (lit_bits >> (8u * kind)) & 0xFFu; }
    // one table read per match (the distance symbol of the distance-4 matches folds to a constant at the call site);
    // working the length symbol out in registers instead cost 20 instructions per match
    __device__ __forceinline__ void match(uint32_t L, uint32_t d)
    {
        uint32_t sym, eb, ev;
        gtz_dist_symbol(d, &sym, &eb, &ev);
        bits += (len[L - 3u] & 31u) + (dist[sym] & 31u) + eb;
    }
};

struct GtzSymbolCount {
    uint32_t* hist; // LDS [GTZ_NSYM]
    __device__ __forceinline__ void lit(uint32_t kind)
    {
        const uint32_t b = kind == 0u ? '0' : kind == 1u ? '1' : kind == 2u ? '/' : '\t';
        atomicAdd(&hist[b], 1u);
    }
    __device__ __forceinline__ void match(uint32_t L, uint32_t d)
    {
        uint32_t sym, eb, ev;
        gtz_dist_symbol(d, &sym, &eb, &ev);
        atomicAdd(&hist[gtz_len_symbol(L)], 1u);
        atomicAdd(&hist[GTZ_DIST0 + sym], 1u);
    }
};

struct GtzBitEmit {
    uint32_t* image;       // LDS bit buffer (zero-initialised), written with atomic OR
    const uint32_t* len;   // LDS: match length -> code + extra bits
    const uint32_t* dist;  // LDS: distance symbol -> code
    uint32_t lit_entry[4]; // the four literal kinds' codes
    uint64_t acc;
    uint32_t nacc, word;
    __device__ __forceinline__ void put(uint32_t v, uint32_t n)
    {
        acc |= (uint64_t)v << nacc;
        nacc += n;
        if (nacc >= 32u) {
            atomicOr(&image[word], (uint32_t)acc);
            ++word;
            acc >>= 32;
            nacc -= 32u;
        }
    }
    __device__ __forceinline__ void entry(uint32_t e) { put(e >> 5, e & 31u); }
    __device__ __forceinline__ void lit(uint32_t kind)
    {
        const uint32_t e = kind == 0u ? lit_entry[0] : kind == 1u ? lit_entry[1] : kind == 2u ? lit_entry[2] : lit_entry[3];
        entry(e);
    }
    __device__ __forceinline__ void match(uint32_t L, uint32_t d)
    {
        uint32_t sym, eb, ev;
        gtz_dist_symbol(d, &sym, &eb, &ev);
        const uint32_t le = len[L - 3u], de = dist[sym];
        entry(le);
        entry(de);
        if (eb) put(ev, eb);
    }
    __device__ __forceinline__ void flush()
    {
        if (nacc) atomicOr(&image[word], (uint32_t)acc);
    }
};

// Deflate bits of the chunk's tokens without visiting the literals: their four kinds are counted with
// popcounts, only the matches are walked.
__device__ __forceinline__ uint32_t gtz_count_bits(const GtzChunk& c, const GtzTokens& t, const uint32_t* farlist, GtzBitCount& cnt)
{
    const uint32_t codes[4] = {(uint32_t)c.lo, (uint32_t)(c.lo >> 32), (uint32_t)c.hi, (uint32_t)(c.hi >> 32)};
    uint32_t n_zero = 0, n_one = 0, n_slash = 0, n_tab = 0, nfar = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint64_t r = t.run3[w];
        const uint64_t lit = t.plain[w] & ~r;
        const uint64_t sp = gtz_spread_pairs(codes[w]);
        const uint64_t ones = (sp & 0x1111111111111111ull) | ((sp & 0x2222222222222222ull) << 1); // allele bytes that are '1'
        n_one += (uint32_t)__builtin_popcountll(lit & ones);
        n_zero += (uint32_t)__builtin_popcountll(lit & 0x5555555555555555ull & ~ones);
        n_slash += (uint32_t)__builtin_popcountll(lit & 0x2222222222222222ull);
        n_tab += (uint32_t)__builtin_popcountll(lit & 0x8888888888888888ull);
        const uint64_t carry = w ? t.run3[w - 1] >> 63 : 0ull;
        uint64_t T = (r & ~((r << 1) | carry)) | t.far[w];
        while (T) {
            const uint32_t p = (uint32_t)__builtin_ctzll(T);
            T &= T - 1ull;
            if (t.far[w] & (1ull << p)) {
                const uint32_t e = farlist[nfar++];
                cnt.match(4u * ((e >> 6) & 127u), e >> 13);
            } else {
                const uint64_t stop = ~r & ~gtz_low_mask(p); // first byte at or after p that is not in the run
                const uint32_t L = stop ? (uint32_t)__builtin_ctzll(stop) - p : 64u - p + t.ext[w];
                cnt.match(L, 4u);
            }
        }
    }
    const uint32_t lb = cnt.lit_bits;
    cnt.bits += n_zero * (lb & 0xFFu) + n_one * ((lb >> 8) & 0xFFu) + n_slash * ((lb >> 16) & 0xFFu) + n_tab * (lb >> 24);
    return cnt.bits;
}

// Visit the chunk's tokens in byte order.
template <class Sink>
__device__ __forceinline__ void gtz_emit_tokens(const GtzChunk& c, const GtzTokens& t, const uint32_t* farlist, Sink& sink)
{
    const uint32_t codes[4] = {(uint32_t)c.lo, (uint32_t)(c.lo >> 32), (uint32_t)c.hi, (uint32_t)(c.hi >> 32)};
    uint32_t nfar = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint64_t r = t.run3[w];
        const uint64_t carry = w ? t.run3[w - 1] >> 63 : 0ull;
        const uint64_t mstart = r & ~((r << 1) | carry);
        uint64_t T = (t.plain[w] & ~r) | mstart | t.far[w];
        while (T) {
            const uint32_t p = (uint32_t)__builtin_ctzll(T);
            T &= T - 1ull;
            const uint64_t bit = 1ull << p;
            if (t.far[w] & bit) {
                const uint32_t e = farlist[nfar++];
                sink.match(4u * ((e >> 6) & 127u), e >> 13);
            } else if (r & bit) {
                // distance-4 match: runs on through the following words while their low bytes stay in the run
                const uint64_t stop = ~r & ~gtz_low_mask(p); // first byte at or after p that is not in the run
                const uint32_t L = stop ? (uint32_t)__builtin_ctzll(stop) - p : 64u - p + t.ext[w];
                sink.match(L, 4u);
            } else {
                const uint32_t which = p & 3u;
                const uint32_t allele = (codes[w] >> (p >> 1)) & 1u;
                sink.lit((which & 1u) ? 2u + (which >> 1) : allele);
            }
        }
    }
}

// LZ77 matches of one chunk (see gtz_kernels.hip.h).  Lanes step from break to break together; a break is
// only looked up when the 16 cells from it hold at least GTZ_MIN_BREAKS differing alleles.
__device__ __forceinline__ void gtz_find_matches(GtzShared* s, uint32_t i, uint32_t c64, const GtzChunk& c, uint32_t* farlist)
{
    const uint32_t gbase = 16u * s->gw[i] - 16u * s->kw0[i] + 64u * c64; // g of the chunk's cell 0
    const uint32_t grow = 16u * s->gw[i];                                // first g of this row
    const int32_t cs = s->cs[i];
    const uint64_t blo = (c.xlo | (c.xlo >> 1)) & 0x5555555555555555ull;
    const uint64_t bhi = (c.xhi | (c.xhi >> 1)) & 0x5555555555555555ull;
    if (c.j1 < GTZ_KEY_CELLS + c.j0) { farlist[0] = 0; return; }
    const uint32_t last = c.j1 - GTZ_KEY_CELLS; // last cell a match can start at
    uint32_t pos = c.j0, nfar = 0;
    for (;;) {
        const uint32_t j = gtz_next_break(blo, bhi, pos);
        if (j > last || nfar >= GTZ_MAX_FAR) break;
        // alleles that differ from the cell before, in the 16 cells from j
        const uint32_t xw = j < 32u ? (uint32_t)((c.xlo >> (2u * j)) | (j ? c.xhi << (64u - 2u * j) : 0ull)) : (uint32_t)(c.xhi >> (2u * j - 64u));
        if ((uint32_t)__builtin_popcount(xw) < GTZ_MIN_BREAKS) { pos = j + 1u; continue; }
        const uint32_t g = gbase + j;
        const uint32_t dw = c.window(j);
        const uint32_t key = dw & (GTZ_KEYS - 1u);
        const uint32_t q = min(g >> GTZ_SEG_SHIFT, GTZ_SEGS - 1u);
        // candidates: earliest in this quarter, latest / earliest in the one before, latest two back
        uint32_t cand[GTZ_CANDS];
        cand[0] = s->tab16[(2u * q) * GTZ_KEYS + key]; // 0xFFFF (none) is never below g
        cand[1] = q ? (uint32_t)s->tab16[(2u * (q - 1u) + 1u) * GTZ_KEYS + key] - 1u : 0xFFFFFFFFu;
        if (GTZ_CANDS > 2) cand[2] = q ? (uint32_t)s->tab16[(2u * (q - 1u)) * GTZ_KEYS + key] : 0xFFFFFFFFu;
        if (GTZ_CANDS > 3) cand[3] = q > 1u ? (uint32_t)s->tab16[(2u * (q - 2u) + 1u) * GTZ_KEYS + key] - 1u : 0xFFFFFFFFu;
        const uint32_t room = c.j1 - j;
        bool ok[GTZ_CANDS], other_row = false;
        uint32_t sg[GTZ_CANDS], n[GTZ_CANDS], lim[GTZ_CANDS], d[GTZ_CANDS];
#pragma unroll
        for (int t = 0; t < (int)GTZ_CANDS; ++t) {
            ok[t] = cand[t] < g;
            if (t == 2 && ok[2] && cand[2] == cand[1]) ok[2] = false; // one occurrence in that quarter: same position twice
            sg[t] = ok[t] ? cand[t] : g; // a disabled candidate reads its own position
            other_row |= ok[t] && sg[t] < grow;
        }
#pragma unroll
        for (int t = 0; t < (int)GTZ_CANDS; ++t) {
            const uint32_t xw0 = gtz_window(s->cells, sg[t]) ^ dw;
            n[t] = xw0 ? (uint32_t)__builtin_ctz(xw0) >> 1 : 16u;
            lim[t] = room;
            d[t] = 4u * (g - sg[t]);
        }
        if (other_row) { // a source in an earlier row of the member: its own byte distance and end
#pragma unroll
            for (int t = 0; t < (int)GTZ_CANDS; ++t) {
                if (!ok[t] || sg[t] >= grow) continue;
                uint32_t r = i;
                while (16u * s->gw[r] > sg[t]) --r;
                const uint32_t ks = sg[t] - 16u * s->gw[r] + 16u * s->kw0[r];
                d[t] = (uint32_t)(cs + 4 * (int32_t)(64u * c64 + j) - (s->cs[r] + 4 * (int32_t)ks));
                lim[t] = min(room, s->ke[r] - ks);
            }
        }
        uint32_t f = 0, dbytes = 0;
#pragma unroll
        for (int t = 0; t < (int)GTZ_CANDS; ++t) {
            // longer than one window: keep comparing (rare except in sparse rows)
            if (ok[t] && n[t] == 16u)
                while (n[t] < lim[t]) {
                    const uint32_t x = c.window(j + n[t]) ^ gtz_window(s->cells, sg[t] + n[t]);
                    if (x) { n[t] += (uint32_t)__builtin_ctz(x) >> 1; break; }
                    n[t] += 16u;
                }
            const uint32_t nt = (ok[t] && d[t] <= 32768u) ? min(n[t], lim[t]) : 0u;
            if (nt > f || (nt == f && nt && d[t] < dbytes)) { f = nt; dbytes = d[t]; } // ties: the nearer one
        }
        bool take = f >= GTZ_KEY_CELLS;
        if (take) {
            const uint32_t a = 2u * j, b = 2u * (j + f);
            const uint64_t mlo = gtz_low_mask(min(b, 64u)) & ~gtz_low_mask(min(a, 64u));
            const uint64_t mhi = gtz_low_mask(b > 64u ? b - 64u : 0u) & ~gtz_low_mask(a > 64u ? a - 64u : 0u);
            take = (uint32_t)(__builtin_popcountll(c.xlo & mlo) + __builtin_popcountll(c.xhi & mhi)) >= GTZ_MIN_BREAKS;
        }
        if (take) {
            farlist[nfar++] = (j - c.j0) | (f << 6) | (dbytes << 13);
            pos = j + f;
        } else pos = j + 1u;
    }
    if (nfar < GTZ_MAX_FAR) farlist[nfar] = 0; // terminator: a recorded match has cells != 0
}

} // namespace dnaf
