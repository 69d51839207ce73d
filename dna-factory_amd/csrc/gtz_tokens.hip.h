// gtz_tokens.hip.h -- search and tokenisation of one 64-cell chunk for the genotype-domain BGZF encoder.
//
// A chunk is 64 cells = 128 alleles = 256 text bytes "a/b\t".  Its state is a handful of bit masks in registers,
// ALLELE space: bit 2j of a 128-bit mask is allele a of cell j (text byte 4j), bit 2j + 1 allele b (byte 4j + 2).
// The cell codes (lo:hi) are the allele values in that very layout, and x = code ^ previous cell's code marks the
// alleles whose byte differs from the byte four back ("breaks").  Every other byte of the text -- '/' and '\t', and
// the alleles that do not break -- equals the byte four before it.
//
// Token stream of a chunk (deflate symbols): LZ77 matches found by the search ("far": whole cells copied from an
// earlier place of the member), and between them literals for the breaking alleles and, for the bytes from one
// event to the next (a "gap": every byte of it repeats the byte four back), one match at distance 4 when the gap
// has three bytes or more, literals when it has one or two.  So the stream is a list of EVENTS -- breaking alleles
// and far matches, in byte order -- and each event emits [gap before it][its own symbol]; the walk visits events,
// not bytes or tokens: one table read gives the gap's code (length code + extra bits + the code of distance 4, or
// the separator's literal), and gap and literal leave as ONE unit of at most 32 bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dnaf {

// 128-bit allele-space mask
struct GtzMask {
    uint64_t lo, hi;
};

// bits [a, b) of a 128-bit mask
__device__ __forceinline__ GtzMask gtz_range(uint32_t a, uint32_t b)
{
    GtzMask m;
    m.lo = gtz_low_mask(min(b, 64u)) & ~gtz_low_mask(min(a, 64u));
    m.hi = gtz_low_mask(b > 64u ? b - 64u : 0u) & ~gtz_low_mask(a > 64u ? a - 64u : 0u);
    return m;
}

// The far matches' distances, 16 bits each, in registers: a queue filled by the search (entry n at bits 16 n) and
// drained from the bottom by the walks (each walk takes a copy).  Keeping the matches out of LDS is what lets seven
// workgroups share a CU instead of four; the encoder is bound by latency, not by instruction issue.
struct GtzFarQueue {
    uint32_t w[GTZ_MAX_FAR / 2];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int r = 0; r < (int)(GTZ_MAX_FAR / 2); ++r) w[r] = 0;
    }
    __device__ __forceinline__ void push(uint32_t n, uint32_t d) // d <= 32768
    {
        const uint32_t v = d << (16u * (n & 1u));
#pragma unroll
        for (int r = 0; r < (int)(GTZ_MAX_FAR / 2); ++r) w[r] |= (n >> 1) == (uint32_t)r ? v : 0u;
    }
    __device__ __forceinline__ uint32_t pop()
    {
        const uint32_t d = w[0] & 0xFFFFu;
#pragma unroll
        for (int r = 0; r + 1 < (int)(GTZ_MAX_FAR / 2); ++r) w[r] = __builtin_amdgcn_alignbit(w[r + 1], w[r], 16);
        w[GTZ_MAX_FAR / 2 - 1] >>= 16;
        return d;
    }
};

// What the search leaves for the walks, all in registers: the alleles that are events, which of them begin a far
// match, where the far matches end (cell e at bit 2 e; a match that ends with the chunk's 64th cell has no bit), and
// the matches' distances in order.
struct GtzEvents {
    GtzMask ev;   // events: breaking alleles outside far matches, first alleles of far matches, forced literals
    GtzMask far;  // subset: first allele of each far match
    GtzMask ends; // bit 2 e for every cell e < 64 at which a far match ends
    GtzFarQueue dist;
    uint32_t far_bits; // deflate bits of the far matches' own symbols (length code + extra, distance code + extra), summed
};


// LZ77 matches of one chunk (see gtz_kernels.hip.h) and the chunk's event masks.  Lanes step from break to break
// together; a break is only looked up when the 16 cells from it hold at least GTZ_MIN_BREAKS differing alleles.
template <bool COUNT>
__device__ __forceinline__ GtzEvents gtz_find_matches(GtzShared* s, uint32_t i, uint32_t c64, const GtzChunk& c, bool search)
{
    GtzEvents E;
    E.dist.clear();
    E.far_bits = 0;
    const GtzMask valid = gtz_range(2u * c.j0, 2u * c.j1);
    E.ev.lo = c.xlo & valid.lo;
    E.ev.hi = c.xhi & valid.hi;
    E.far.lo = E.far.hi = 0;
    uint64_t ends_lo = 0, ends_hi = 0; // bit 2s for every cell s at which a far match ends
    uint32_t nfar = 0;
    if (search && c.j1 >= GTZ_KEY_CELLS + c.j0) {
        const uint32_t gbase = 16u * s->gw[i] - 16u * s->kw0[i] + 64u * c64; // g of the chunk's cell 0
        const int32_t delta_i = s->row[i].delta;
        // Cells that differ from their predecessor and can begin a match (one bit per cell, at its allele a), consumed from
        // the bottom.  The chunk's own cells are read back from LDS as 16-cell windows (one ds_read2 + one alignbit) rather
        // than shifted out of the registers (a 128-bit funnel shift by a variable: 15 instructions).
        const GtzMask startable = gtz_range(2u * c.j0, 2u * (c.j1 - GTZ_KEY_CELLS + 1u));
        uint64_t rlo = (c.xlo | (c.xlo >> 1)) & 0x5555555555555555ull & startable.lo;
        uint64_t rhi = (c.xhi | (c.xhi >> 1)) & 0x5555555555555555ull & startable.hi;
        while ((rlo | rhi) && nfar < GTZ_MAX_FAR) {
            const bool low = rlo != 0;
            const uint64_t rr = low ? rlo : rhi;
            const uint64_t bit = rr & (0ull - rr); // the cell's allele-a bit within its half
            const uint32_t j = (low ? 0u : 32u) + ((uint32_t)__builtin_ctzll(rr) >> 1);
            if (low) rlo ^= bit; else rhi ^= bit;
            const uint32_t g = gbase + j;
            const uint32_t dw = gtz_window(s->cells, g);
            // alleles that differ from the cell before, in the 16 cells from j
            uint32_t xw = dw ^ (g ? gtz_window(s->cells, g - 1u) : dw << 2);
            if (c.head && j == c.j0) xw |= 3u; // nothing before a row's first cell in the member
            if ((uint32_t)__builtin_popcount(xw) < GTZ_MIN_BREAKS) continue;
            const uint32_t key = dw & (GTZ_KEYS - 1u);
            const uint32_t q = gtz_segment(g);
            // candidates: earliest in this segment, latest / earliest in the one before, latest two back
            uint32_t cand[GTZ_CANDS];
            cand[0] = s->tab16[(2u * q) * GTZ_KEYS + key]; // 0xFFFF (none) is never below g
            cand[1] = q ? (uint32_t)s->tab16[(2u * (q - 1u) + 1u) * GTZ_KEYS + key] - 1u : 0xFFFFFFFFu;
#if DNAF_GTZ_CANDS > 2
            cand[2] = q ? (uint32_t)s->tab16[(2u * (q - 1u)) * GTZ_KEYS + key] : 0xFFFFFFFFu;
            if (cand[2] == cand[1]) cand[2] = 0xFFFFFFFFu; // one occurrence in that segment: same position twice
#endif
#if DNAF_GTZ_CANDS > 3
            cand[3] = q > 1u ? (uint32_t)s->tab16[(2u * (q - 2u) + 1u) * GTZ_KEYS + key] - 1u : 0xFFFFFFFFu;
#endif
#if DNAF_GTZ_CANDS > 4 // (experiments with short segments)
            cand[4] = q > 1u ? (uint32_t)s->tab16[(2u * (q - 2u)) * GTZ_KEYS + key] : 0xFFFFFFFFu;
            if (cand[4] == cand[3]) cand[4] = 0xFFFFFFFFu;
#endif
#if DNAF_GTZ_CANDS > 5
            cand[5] = q > 2u ? (uint32_t)s->tab16[(2u * (q - 3u) + 1u) * GTZ_KEYS + key] - 1u : 0xFFFFFFFFu;
#endif
            const uint32_t room = c.j1 - j;
            // Every candidate alike, in this row or an earlier one: the row of the source from the word map, its bounds
            // in the member and the byte distance from the rows' offsets (byte of position g = 4 g + delta[row]).  The
            // tables hold unverified positions (padding between rows, cells in front of the member): the bounds reject them.
            uint32_t f = 0, dbytes = 0, fsrc = 0, flim = 0;
#pragma unroll
            for (int t = 0; t < (int)GTZ_CANDS; ++t) {
                const bool before = cand[t] < g;
                const uint32_t sg = before ? cand[t] : g; // a disabled candidate reads its own position
                const GtzRowInfo ri = s->row[s->rowmap[sg >> 4]];
                const uint32_t x = gtz_window(s->cells, sg) ^ dw;
                const uint32_t n = x ? (uint32_t)__builtin_ctz(x) >> 1 : 16u;
                const uint32_t lim = min(room, ri.gend - sg); // (only used when sg < gend)
                const uint32_t d = 4u * (g - sg) + (uint32_t)(delta_i - ri.delta);
                const bool ok = before && sg >= ri.gfirst && sg < ri.gend && d <= 32768u;
                const uint32_t nt = ok ? min(n, lim) : 0u;
                if (nt > f || (nt == f && nt && d < dbytes)) { f = nt; dbytes = d; fsrc = sg; flim = lim; } // ties: the nearer one
            }
            // longer than one window: keep comparing (rare except in sparse rows); only the candidate chosen on the first 16 cells
            if (f == 16u)
                while (f < flim) {
                    const uint32_t x = gtz_window(s->cells, g + f) ^ gtz_window(s->cells, fsrc + f);
                    if (x) { f = min(flim, f + ((uint32_t)__builtin_ctz(x) >> 1)); break; }
                    f = min(flim, f + 16u);
                }
            // worth a match: five cells or more that hold three differing alleles or more (16 cells from j hold that many)
            if (f < GTZ_KEY_CELLS || (f < 16u && (uint32_t)__builtin_popcount(xw & ((1u << (2u * f)) - 1u)) < GTZ_MIN_BREAKS)) continue;
            E.dist.push(nfar++, dbytes);
            if (COUNT) { // what the match will cost, while f and the distance are at hand (the bit count needs no walk over them)
                uint32_t sym, eb, ev;
                gtz_dist_symbol(dbytes, &sym, &eb, &ev);
                E.far_bits += (s->len4()[f] & 31u) + (s->dist()[sym] & 31u) + eb;
            }
            // alleles [2 j, 2 (j + f)) leave the events and the cells that can begin a match; masks below 2 j from the cell's own
            // bit, below 2 (j + f) by one shift
            const uint64_t below_lo = low ? bit - 1ull : ~0ull, below_hi = low ? 0ull : bit - 1ull;
            const uint32_t e2 = 2u * (j + f);
            const uint64_t upto_lo = gtz_low_mask(min(e2, 64u)), upto_hi = gtz_low_mask(e2 > 64u ? e2 - 64u : 0u);
            E.ev.lo &= ~(upto_lo & ~below_lo);
            E.ev.hi &= ~(upto_hi & ~below_hi);
            rlo &= ~upto_lo;
            rhi &= ~upto_hi;
            if (low) E.far.lo |= bit; else E.far.hi |= bit;
            // the end mark is the bit just above `upto` (none when the match ends with the chunk's 64th cell)
            ends_lo |= upto_lo + 1ull; // (all ones + 1 = 0)
            if (upto_lo == ~0ull) ends_hi |= upto_hi + 1ull;
        }
    }
    // forced literals: allele a at the start of a run that allele b of the same cell cuts to two bytes
    uint64_t starts_lo = ends_lo, starts_hi = ends_hi;
    {
        const uint32_t s0 = c.head ? c.j0 + 1u : c.j0; // (a far match at the head cell has cleared that cell's events: harmless)
        if (s0 < 32u) starts_lo |= 1ull << (2u * s0); else if (s0 < 64u) starts_hi |= 1ull << (2u * s0 - 64u);
    }
    E.ev.lo |= (E.ev.lo >> 1) & starts_lo & valid.lo;
    E.ev.hi |= (E.ev.hi >> 1) & starts_hi & valid.hi;
    E.ev.lo |= E.far.lo;
    E.ev.hi |= E.far.hi;
    E.ends.lo = ends_lo;
    E.ends.hi = ends_hi;
    return E;
}

// ---- sinks of the walk ----
// gap(L, at): the L bytes from chunk byte `at` up to the next event all repeat the byte four back.  L is 0, 1
// ('/' when at & 3 == 1, '\t' when at & 3 == 3) or >= 3 (one match at distance 4).
// lit(kind): 0 '0', 1 '1', 2 '/', 3 '\t'.   far(L, d): LZ77 match of L bytes at distance d.

struct GtzBitCount {
    const uint32_t* gap;   // LDS [GTZ_GAP_STRIDE]: gap length -> code, bit count in the low 5 bits
    const uint32_t* len4;  // LDS: far match of f cells -> length code + extra bits
    const uint32_t* dist;  // LDS: distance symbol -> code
    uint32_t lit_n;        // 4 x 8 bits: code length of each literal kind
    uint32_t bits;
    uint32_t far_seen = 0; // bits of the far matches' own symbols within `bits`
    __device__ __forceinline__ uint32_t gap_bits(uint32_t L, uint32_t at) const { return (L == 1u && (at & 2u)) ? lit_n >> 24 : gap[L] & 31u; }
    __device__ __forceinline__ void gap_lit(uint32_t L, uint32_t at, uint32_t value) { bits += gap_bits(L, at) + ((lit_n >> (8u * value)) & 0xFFu); }
    __device__ __forceinline__ void gap_only(uint32_t L, uint32_t at) { bits += gap_bits(L, at); }
    __device__ __forceinline__ void lit(uint32_t kind) { bits += (lit_n >> (8u * kind)) & 0xFFu; }
    __device__ __forceinline__ void far(uint32_t f, uint32_t d)
    {
        uint32_t sym, eb, ev;
        gtz_dist_symbol(d, &sym, &eb, &ev);
        const uint32_t n = (len4[f] & 31u) + (dist[sym] & 31u) + eb;
        bits += n;
        far_seen += n;
    }
};

struct GtzSymbolCount {
    uint32_t* hist; // LDS [GTZ_NSYM]
    __device__ __forceinline__ void lit(uint32_t kind)
    {
        const uint32_t b = kind == 0u ? '0' : kind == 1u ? '1' : kind == 2u ? '/' : '\t';
        atomicAdd(&hist[b], 1u);
    }
    __device__ __forceinline__ void gap_only(uint32_t L, uint32_t at)
    {
        if (L >= 3u) { atomicAdd(&hist[gtz_len_symbol(L)], 1u); atomicAdd(&hist[GTZ_DIST0 + 3u], 1u); }
        else if (L == 1u) lit((at & 2u) ? 3u : 2u);
    }
    __device__ __forceinline__ void gap_lit(uint32_t L, uint32_t at, uint32_t value) { gap_only(L, at); lit(value); }
    __device__ __forceinline__ void far(uint32_t f, uint32_t d)
    {
        uint32_t sym, eb, ev;
        gtz_dist_symbol(d, &sym, &eb, &ev);
        atomicAdd(&hist[gtz_len_symbol(4u * f)], 1u);
        atomicAdd(&hist[GTZ_DIST0 + sym], 1u);
    }
};

struct GtzBitEmit {
    uint32_t* image;       // LDS bit buffer (zero-initialised), written with atomic OR
    const uint32_t* gap;
    const uint32_t* len4;
    const uint32_t* dist;
    uint32_t lit_entry[4]; // the four literal kinds' codes, (bits << 5) | count
    uint32_t pos;          // bit position in the image
    // v has n <= 32 bits
    __device__ __forceinline__ void put(uint32_t v, uint32_t n)
    {
        const uint64_t w = (uint64_t)v << (pos & 31u);
        uint32_t* at = image + (pos >> 5);
        atomicOr(at, (uint32_t)w);
        atomicOr(at + 1, (uint32_t)(w >> 32));
        pos += n;
    }
    __device__ __forceinline__ void entry(uint32_t e) { put(e >> 5, e & 31u); }
    // one byte behind an allele b is the tab's literal; everything else is in the table
    __device__ __forceinline__ uint32_t gap_entry(uint32_t L, uint32_t at) const { return (L == 1u && (at & 2u)) ? lit_entry[3] : gap[L]; }
    __device__ __forceinline__ void lit(uint32_t kind)
    {
        entry(kind == 0u ? lit_entry[0] : kind == 1u ? lit_entry[1] : kind == 2u ? lit_entry[2] : lit_entry[3]);
    }
    // gap and literal as one unit: the host guarantees that together they have at most 32 bits (GtzClassTables::unit_ok)
    __device__ __forceinline__ void gap_lit(uint32_t L, uint32_t at, uint32_t value)
    {
        const uint32_t g = gap_entry(L, at), l = value ? lit_entry[1] : lit_entry[0];
        const uint32_t gn = g & 31u;
        put((g >> 5) | ((l >> 5) << gn), gn + (l & 31u));
    }
    __device__ __forceinline__ void gap_only(uint32_t L, uint32_t at) { entry(gap_entry(L, at)); }
    __device__ __forceinline__ void far(uint32_t f, uint32_t d)
    {
        uint32_t sym, eb, ev;
        gtz_dist_symbol(d, &sym, &eb, &ev);
        entry(len4[f]);
        const uint32_t de = dist[sym], dn = de & 31u;
        put((de >> 5) | (ev << dn), dn + eb); // <= 15 + 13 bits
    }
};

// first set bit of a 128-bit mask at or above position `from` (128: none)
__device__ __forceinline__ uint32_t gtz_next_bit(uint64_t lo, uint64_t hi, uint32_t from)
{
    if (from < 64u) {
        const uint64_t t = lo & ~gtz_low_mask(from);
        if (t) return (uint32_t)__builtin_ctzll(t);
        return hi ? 64u + (uint32_t)__builtin_ctzll(hi) : 128u;
    }
    if (from >= 128u) return 128u;
    const uint64_t t = hi & ~gtz_low_mask(from - 64u);
    return t ? 64u + (uint32_t)__builtin_ctzll(t) : 128u;
}

// Deflate bits of a chunk's tokens WITHOUT walking its events (what gtz_walk with a counting sink returns, and must: the
// chunks' bit offsets in the member come from it).  Every event is followed by a gap that ends at the next event, so the
// gaps are counted by how far the next event is: breaking alleles with an event right behind them (one byte between:
// the separator's literal) or two alleles on (three bytes: the shortest match) are counted with shifted masks and
// popcounts -- in dense rows that is nearly all of them; only the alleles with a longer gap behind them, the ends of the
// far matches and the start of the chunk take a table read each.  The far matches' own symbols were priced by the search.
__device__ __forceinline__ uint32_t gtz_count_bits(const GtzChunk& c, const GtzEvents& E, const uint32_t* gap, uint32_t lit_n)
{
    GtzMask ev = E.ev;
    uint32_t bits = E.far_bits;
    uint32_t s0 = c.j0; // the cell at whose boundary the first gap begins
    if (c.head && !((c.j0 < 32u ? E.far.lo >> (2u * c.j0) : E.far.hi >> (2u * c.j0 - 64u)) & 1ull)) {
        const uint32_t code = c.code(c.j0);
        bits += ((lit_n >> (8u * (code & 1u))) & 0xFFu) + ((lit_n >> (8u * (code >> 1))) & 0xFFu) + ((lit_n >> 16) & 0xFFu) + (lit_n >> 24);
        const GtzMask m = gtz_range(2u * c.j0, 2u * c.j0 + 2u);
        ev.lo &= ~m.lo;
        ev.hi &= ~m.hi;
        s0 = c.j0 + 1u;
    }
    // events plus the chunk's end as a pseudo-event (allele 2 j1; past the mask when the chunk has all 64 cells)
    uint64_t xlo = ev.lo, xhi = ev.hi;
    if (c.j1 < 32u) xlo |= 1ull << (2u * c.j1); else if (c.j1 < 64u) xhi |= 1ull << (2u * c.j1 - 64u);
    const uint32_t end2 = 2u * c.j1; // where a gap ends when no event follows
    const uint64_t blo = ev.lo & ~E.far.lo, bhi = ev.hi & ~E.far.hi; // literal events: breaking alleles and forced ones
    // their own literals
    const uint32_t n1 = (uint32_t)(__builtin_popcountll(blo & c.lo) + __builtin_popcountll(bhi & c.hi));
    const uint32_t n0 = (uint32_t)(__builtin_popcountll(blo) + __builtin_popcountll(bhi)) - n1;
    bits += n0 * (lit_n & 0xFFu) + n1 * ((lit_n >> 8) & 0xFFu);
    // the next event one allele on: one byte between, '/' behind an allele a (even index), tab behind an allele b
    const uint64_t s1lo = (xlo >> 1) | (xhi << 63), s1hi = xhi >> 1;
    const uint64_t s2lo = (xlo >> 2) | (xhi << 62), s2hi = xhi >> 2;
    const uint64_t a1lo = blo & s1lo, a1hi = bhi & s1hi;
    const uint32_t n_slash = (uint32_t)(__builtin_popcountll(a1lo & 0x5555555555555555ull) + __builtin_popcountll(a1hi & 0x5555555555555555ull));
    const uint32_t n_tab = (uint32_t)(__builtin_popcountll(a1lo) + __builtin_popcountll(a1hi)) - n_slash;
    bits += n_slash * ((lit_n >> 16) & 0xFFu) + n_tab * (lit_n >> 24);
    // two alleles on: three bytes between
    const uint64_t a2lo = blo & ~s1lo & s2lo, a2hi = bhi & ~s1hi & s2hi;
    bits += (uint32_t)(__builtin_popcountll(a2lo) + __builtin_popcountll(a2hi)) * (gap[3] & 31u);
    // the rest of the literal events: a longer gap (or the chunk's end) behind them
    {
        uint64_t rlo = blo & ~s1lo & ~s2lo, rhi = bhi & ~s1hi & ~s2hi;
        while (rlo | rhi) {
            const bool low = rlo != 0;
            const uint32_t p = low ? (uint32_t)__builtin_ctzll(rlo) : 64u + (uint32_t)__builtin_ctzll(rhi);
            if (low) rlo &= rlo - 1ull; else rhi &= rhi - 1ull;
            const uint32_t nx = gtz_next_bit(xlo, xhi, p + 1u);
            const uint32_t L = 2u * ((nx < 128u ? nx : end2) - p) - 1u;
            // L >= 5 -- except behind the last alleles of a chunk that has all 64 cells, whose end is no bit of the mask:
            // three bytes, or one, the tab behind allele b of the last cell
            bits += (L == 1u && (p & 1u)) ? lit_n >> 24 : gap[L] & 31u;
        }
    }
    // the gaps behind the far matches (they begin on a cell boundary: an even number of bytes, never two)
    {
        uint64_t zlo = E.ends.lo, zhi = E.ends.hi;
        while (zlo | zhi) {
            const bool low = zlo != 0;
            const uint32_t z = low ? (uint32_t)__builtin_ctzll(zlo) : 64u + (uint32_t)__builtin_ctzll(zhi);
            if (low) zlo &= zlo - 1ull; else zhi &= zhi - 1ull;
            const uint32_t nx = gtz_next_bit(xlo, xhi, z);
            bits += gap[2u * ((nx < 128u ? nx : end2) - z)] & 31u;
        }
    }
    // and the gap at the chunk's start
    {
        const uint32_t nx = gtz_next_bit(xlo, xhi, 2u * s0);
        bits += gap[2u * ((nx < 128u ? nx : end2) - 2u * s0)] & 31u;
    }
#ifdef DNAF_GTZ_BREAK_COUNT
    if (c.j1 == 64u && ((ev.hi >> 63) & 1ull)) ++bits; // test builds: a wrong count in one rare case, to see the self-check fire
#endif
    return bits;
}

// Visit the chunk's events in byte order.
template <class Sink>
__device__ __forceinline__ void gtz_walk(const GtzChunk& c, const GtzEvents& E, Sink& sink)
{
    uint32_t at = 4u * c.j0; // chunk byte at which the current gap begins
    GtzMask ev = E.ev;
    GtzFarQueue dq = E.dist;
    uint64_t ends_lo = E.ends.lo, ends_hi = E.ends.hi;
    if (c.head && !((c.j0 < 32u ? E.far.lo >> (2u * c.j0) : E.far.hi >> (2u * c.j0 - 64u)) & 1ull)) {
        // the first cell of a row's region has nothing before it in the member to repeat: four literals
        const uint32_t code = c.code(c.j0);
        sink.lit(code & 1u);
        sink.lit(2u);
        sink.lit(code >> 1);
        sink.lit(3u);
        const GtzMask m = gtz_range(2u * c.j0, 2u * c.j0 + 2u);
        ev.lo &= ~m.lo;
        ev.hi &= ~m.hi;
        at += 4u;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint64_t T = h ? ev.hi : ev.lo;
        const uint64_t F = h ? E.far.hi : E.far.lo, V = h ? c.hi : c.lo;
        while (T) {
            const uint32_t b = (uint32_t)__builtin_ctzll(T);
            T &= T - 1ull;
            const uint32_t byte = 2u * (64u * (uint32_t)h + b);
            const uint32_t L = byte - at;
            if ((F >> b) & 1ull) {
                // the match ends at the first end mark not yet used (none left: at the chunk's 64th cell)
                uint32_t e2 = 128u;
                if (ends_lo) { e2 = (uint32_t)__builtin_ctzll(ends_lo); ends_lo &= ends_lo - 1ull; }
                else if (ends_hi) { e2 = 64u + (uint32_t)__builtin_ctzll(ends_hi); ends_hi &= ends_hi - 1ull; }
                sink.gap_only(L, at);
                sink.far((2u * e2 - byte) >> 2, dq.pop());
                at = 2u * e2;
            } else {
                sink.gap_lit(L, at, (uint32_t)(V >> b) & 1u);
                at = byte + 1u;
            }
        }
    }
    sink.gap_only(4u * c.j1 - at, at);
}

} // namespace dnaf
