// gtz_kernels.hip.h -- genotype-domain BGZF encoder (gfx950): BGZF members of the VCF text made from the
// 2-bit genotype matrix and the rows' fixed columns, without reading the text.
//
// A member (61,440 text bytes) that lies in biallelic autosome rows is a known function of the packed
// matrix: row r's text is its prefix followed by cells "a/b\t" (the last one ends in "\n"), 4 bytes per
// sample.  One workgroup encodes one member, one thread one ITEM of it -- a 64-cell chunk (256 text bytes) or
// the literal span between two rows' cell regions (last cell of a row, "\n", the next row's fixed columns);
// 256 threads where rows are long, 512 where short rows make more items (the host picks):
//
//   1. the rows that overlap the member are laid out in LDS ("g space": their 2-bit cells back to back,
//      word-aligned per row), 960 words instead of 61 KB of text; the spans' bytes are staged in LDS too;
//   2. key tables: earliest and latest position of every 5-cell key (10 bits) per segment of the g space,
//      each table written by one wave with plain stores in ascending / descending order -- no atomics;
//   3. search (gtz_tokens.hip.h): a thread steps from one "break" of its chunk (a cell that differs from
//      its predecessor) to the next, looks the key up, compares two candidates by XOR / count-trailing-zeros
//      on 16-cell windows read from LDS and takes an LZ77 ("far") match when it covers enough cells and
//      breaks.  The matches stay in registers (masks of their first alleles and ends, a queue of distances);
//   4. the chunk's deflate bits are counted by mask arithmetic, the counts scanned over the workgroup, and
//      one walk over the chunk's events (breaking alleles and far matches: [gap before it][own symbol], the
//      gap one table entry) writes the bits into an LDS image of the whole member (gzip header, deflate
//      block, CRC-32, ISIZE), which goes to the member's staging slot.  The member CRC comes from the cells
//      by linearity of the CRC (a table by pairs of cells), never from text bytes.
//
// Seven workgroups share a CU (22.8 KB of LDS, 72 registers): the kernel was bound by latency until the
// match lists left LDS; now its vector units are busy (profiles/r03_gtz_*).
// bgzf_scan_totals / _apply turn sizes into offsets and gtz_compact moves the members to their place in the
// stream.  Members the kernel does not take (other chromosome classes, multi-allelic rows, more than
// GTZ_MAX_ROWS rows, more items than threads, a body above the image size, a class whose codes do not fit
// the walk's 32-bit units) are left to the text-domain encoder (bgzf_kernels.hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgzf_kernels.hip.h"
#include "gtz_host.h"
#include "rowgen_kernels.hip.h"

namespace dnaf {

constexpr uint32_t GTZ_THREADS = 256;      // one item (64-cell chunk or literal span) per thread: members of long rows (up to 256 items)
constexpr uint32_t GTZ_THREADS_MANY = 512; // members of short rows: more rows, hence more spans and cut chunks
constexpr uint32_t GTZ_MAX_WAVES = GTZ_THREADS_MANY / 64;
#ifndef DNAF_GTZ_KEY_CELLS
#define DNAF_GTZ_KEY_CELLS 5
#endif
constexpr uint32_t GTZ_KEY_CELLS = DNAF_GTZ_KEY_CELLS; // cells of a table key = the shortest LZ77 match (experiments: 4)
constexpr uint32_t GTZ_KEYS = 1u << (2 * GTZ_KEY_CELLS);
#ifndef DNAF_GTZ_SEGS
#define DNAF_GTZ_SEGS 3
#endif
constexpr uint32_t GTZ_SEGS = DNAF_GTZ_SEGS;                 // segments of the member's g space, each with its own key tables
// 3 x 5120 cells (the last segment open-ended); experiments: 4 x 4096, 6 x 2560, 8 x 2048 -- with more, shorter segments the
// candidates of the segments before lie inside deflate's 32 KiB window (profiles/r03_parse_study.txt)
constexpr uint32_t GTZ_SEG_CELLS = GTZ_SEGS == 3 ? 5120 : GTZ_SEGS == 4 ? 4096 : GTZ_SEGS == 6 ? 2560 : 2048;
static_assert(GTZ_SEGS == 3 || GTZ_SEGS == 4 || GTZ_SEGS == 6 || GTZ_SEGS == 8, "segments");
__device__ __forceinline__ uint32_t gtz_segment(uint32_t g)
{
    if (GTZ_SEGS == 3) return (g >= GTZ_SEG_CELLS ? 1u : 0u) + (g >= 2u * GTZ_SEG_CELLS ? 1u : 0u);
    if (GTZ_SEGS == 6) return min(g / GTZ_SEG_CELLS, GTZ_SEGS - 1u);
    return min(g >> (GTZ_SEGS == 4 ? 12 : 11), GTZ_SEGS - 1u);
}
constexpr uint32_t GTZ_MAX_ROWS = 16;     // rows overlapping one member
constexpr uint32_t GTZ_MAX_FAR = 12;      // LZ77 matches recorded per chunk (64 cells / 5)
constexpr uint32_t GTZ_CELL_WORDS = BGZF_IN / 64 + 2 * GTZ_MAX_ROWS + 8;
constexpr uint32_t GTZ_STRIDE = 16384;    // staging slot of a member (global memory)
constexpr uint32_t GTZ_IMAGE_WORDS = 3840; // the member image in LDS: 15,360 bytes, as much as the search's tables it overlays (a
                                           // member of MAF-0.5 rows takes 8 KB; a larger one goes to the text-domain encoder)
constexpr uint32_t GTZ_MIN_BREAKS = 3;    // an LZ77 match must replace at least this many literal bytes
#ifndef DNAF_GTZ_CANDS
#define DNAF_GTZ_CANDS 2
#endif
constexpr uint32_t GTZ_CANDS = DNAF_GTZ_CANDS; // source candidates looked at per break (2..4)
constexpr uint32_t GTZ_SPAN_MAX = 72;     // bytes between two rows' cell regions: last cell + "\n" + fixed columns + a cut cell

struct GtzParams {
    const RowRec* recs;      // [table rows]
    const uint8_t* prefix;   // prefix blob
    const uint32_t* packed;  // [n_rows] rows of W words, Wp words apart: genotype matrix of this call
    uint64_t text_base;      // recs[row_begin].text_off
    uint64_t row_begin, n_rows;
    uint64_t n_bytes;        // text bytes of the call
    uint32_t n_blocks;
    uint32_t S, W, Wp;
    uint32_t sample_step;    // gtz_hist: every sample_step-th member
    const GtzTables* tables;
    uint32_t* member_row;    // [n_blocks] table row that holds the member's first byte (gtz_locate)
    uint32_t* hist;          // [GTZ_NCLASS][GTZ_NSYM]
    uint32_t* sizes;         // [n_blocks] member bytes (same array the text-domain encoder fills for the others)
    uint8_t* fast;           // [n_blocks] 1 = encoded here
    uint8_t* stage;          // [n_blocks][GTZ_STRIDE]
    const uint64_t* offsets; // [n_blocks + 1]
    uint8_t* out;
};

// first table row of every member: one thread per member, binary search over text offsets
__global__ __launch_bounds__(256) void gtz_locate(const GtzParams p)
{
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;
    if (m >= p.n_blocks) return;
    const uint64_t B = p.text_base + (uint64_t)m * BGZF_IN;
    uint64_t lo = p.row_begin, hi = p.row_begin + p.n_rows;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (p.recs[mid].text_off <= B) lo = mid;
        else hi = mid;
    }
    p.member_row[m] = (uint32_t)(lo - p.row_begin);
}

// a row of the member as the search sees it: g-space bounds of its cells inside the member, and the offset that turns
// a g-space position into a byte of the member (byte = 4 g + delta)
struct GtzRowInfo {
    int32_t delta;
    uint32_t gfirst, gend;
    uint32_t pad;
};

struct GtzShared {
    union {
        struct { // the search's view: key tables and the cells
            uint16_t tab16[(2 * GTZ_SEGS - 1) * GTZ_KEYS]; // [quarter][0 = earliest (0xFFFF: none), 1 = latest + 1 (0: none)][key]; the last quarter has no "latest"
            uint32_t cells[GTZ_CELL_WORDS + 4];
            uint8_t rowmap[GTZ_CELL_WORDS + 8]; // row (index into the arrays below) of every g-space word
        };
        uint32_t image[GTZ_IMAGE_WORDS];         // later the member image
    };
    GtzRowInfo row[GTZ_MAX_ROWS + 2];
#ifdef DNAF_GTZ_PAD_LDS
    uint32_t pad_lds[DNAF_GTZ_PAD_LDS]; // occupancy experiments
#endif
    uint32_t codes[128 + GTZ_GAP_STRIDE + GTZ_CHUNK_CELLS + 8 + 32]; // lit, gap, len4, dist code tables; the symbol histogram in gtz_encode<true>
    uint32_t crc_table[256];
    uint32_t crc_base[GTZ_CHUNK_CELLS + 1];
    uint32_t crc_nib[(GTZ_CHUNK_CELLS / 2) * 16]; // change of a chunk's CRC by the alleles of a pair of cells, [pairs before its end][4 alleles]
    // rows overlapping the member
    int32_t off[GTZ_MAX_ROWS + 2];     // row start relative to the member's first byte
    int32_t cs[GTZ_MAX_ROWS + 2];      // first cell's byte, relative to the member
    uint32_t prefix_off[GTZ_MAX_ROWS + 2];
    uint32_t kb[GTZ_MAX_ROWS + 2], ke[GTZ_MAX_ROWS + 2]; // cells wholly inside the member, without the row's last cell
    uint32_t gw[GTZ_MAX_ROWS + 2];     // first g-space word of the row
    uint32_t kw0[GTZ_MAX_ROWS + 2];    // packed word that g-space word gw[i] is a copy of
    uint32_t item_base[GTZ_MAX_ROWS + 2];
    uint32_t chunk_base[GTZ_MAX_ROWS + 2];
    int32_t span_lo[GTZ_MAX_ROWS + 2], span_hi[GTZ_MAX_ROWS + 2];
    uint32_t thr0[GTZ_MAX_ROWS + 2];
    uint8_t cls[GTZ_MAX_ROWS + 2], nall[GTZ_MAX_ROWS + 2];
    uint8_t span_bytes[(GTZ_MAX_ROWS + 2) * GTZ_SPAN_MAX]; // the literal spans' text
    uint32_t scratch[2 * GTZ_MAX_WAVES];
    uint32_t nr, n_items, ok, klass;
    __device__ __forceinline__ uint32_t* lit() { return codes; }
    __device__ __forceinline__ uint32_t* gap() { return codes + 128; }
    __device__ __forceinline__ uint32_t* len4() { return codes + 128 + GTZ_GAP_STRIDE; }
    __device__ __forceinline__ uint32_t* dist() { return codes + 128 + GTZ_GAP_STRIDE + GTZ_CHUNK_CELLS + 8; }
    __device__ __forceinline__ uint32_t* hist() { return codes; }
};
static_assert(GTZ_NSYM <= 128 + GTZ_GAP_STRIDE + GTZ_CHUNK_CELLS + 8 + 32, "the histogram reuses the code tables");

// 16 cells of the 2-bit stream starting at g
__device__ __forceinline__ uint32_t gtz_window(const uint32_t* cells, uint32_t g)
{
    const uint32_t w = g >> 4;
    const uint64_t v = ((uint64_t)cells[w + 1] << 32) | cells[w];
    return (uint32_t)(v >> (2u * (g & 15u)));
}

// The 64 cells of a chunk (cell j at bits 2j of lo:hi) and their difference from the cell before.
struct GtzChunk {
    uint64_t lo, hi;   // cell codes
    uint64_t xlo, xhi; // code ^ previous cell's code; 3 where there is no previous cell in the member
    uint32_t j0, j1;   // cells [j0, j1) of the 64-cell grid belong to the item
    bool head;         // cell j0 is the first cell of the row's region: nothing before it to refer to
    __device__ __forceinline__ uint32_t code(uint32_t j) const { return (uint32_t)((j < 32u ? lo >> (2u * j) : hi >> (2u * j - 64u)) & 3u); }
};

__device__ __forceinline__ uint64_t gtz_low_mask(uint32_t nbits) { return nbits >= 64u ? ~0ull : (1ull << nbits) - 1ull; }

__device__ __forceinline__ GtzChunk gtz_load_chunk(const GtzShared* s, uint32_t i, uint32_t c64, uint32_t k0, uint32_t k1)
{
    GtzChunk c;
    const int32_t w0 = (int32_t)s->gw[i] + 4 * (int32_t)c64 - (int32_t)s->kw0[i]; // g-space word of cell 64 c64
    const int32_t wlo = (int32_t)s->gw[i], whi = (int32_t)s->gw[i + 1];
    uint32_t w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = (w0 + q >= wlo && w0 + q < whi) ? s->cells[w0 + q] : 0u;
    const uint32_t prev = (w0 - 1 >= wlo) ? s->cells[w0 - 1] >> 30 : 0u;
    c.lo = ((uint64_t)w[1] << 32) | w[0];
    c.hi = ((uint64_t)w[3] << 32) | w[2];
    c.j0 = k0 - 64u * c64;
    c.j1 = k1 - 64u * c64;
    c.xlo = c.lo ^ ((c.lo << 2) | prev);
    c.xhi = c.hi ^ ((c.hi << 2) | (c.lo >> 62));
    c.head = k0 == s->kb[i];
    if (c.head) { // the region's first cell has no predecessor inside the member
        if (c.j0 < 32u) c.xlo |= 3ull << (2u * c.j0);
        else c.xhi |= 3ull << (2u * c.j0 - 64u);
    }
    return c;
}

__device__ __forceinline__ uint32_t gtz_len_symbol(uint32_t L)
{
    if (L == 258u) return 285u;
    const uint32_t v = L - 3u;
    if (v < 8u) return 257u + v;
    const uint32_t hb = 31u - (uint32_t)__builtin_clz(v);
    return 257u + 4u * (hb - 1u) + ((v >> (hb - 2u)) & 3u);
}

// distance symbol, extra-bit count and value of a distance 1..32768
__device__ __forceinline__ void gtz_dist_symbol(uint32_t d, uint32_t* sym, uint32_t* eb, uint32_t* ev)
{
    const uint32_t t = d - 1u;
    if (t < 4u) { *sym = t; *eb = 0; *ev = 0; return; }
    const uint32_t hb = 31u - (uint32_t)__builtin_clz(t);
    *sym = 2u * hb + ((t >> (hb - 1u)) & 1u);
    *eb = hb - 1u;
    *ev = t & ((1u << (hb - 1u)) - 1u);
}

struct GtzCount {
    GtzShared* s;
    uint32_t bits;
    __device__ __forceinline__ void lit(uint32_t b) { bits += s->lit()[b] & 31u; }
};

struct GtzHist {
    GtzShared* s;
    __device__ __forceinline__ void lit(uint32_t b) { atomicAdd(&s->hist()[b], 1u); }
};

struct GtzEmit {
    GtzShared* s;
    uint64_t acc;
    uint32_t nacc, word;
    __device__ __forceinline__ void put(uint32_t v, uint32_t n)
    {
        acc |= (uint64_t)v << nacc;
        nacc += n;
        if (nacc >= 32u) {
            atomicOr(&s->image[word], (uint32_t)acc);
            ++word;
            acc >>= 32;
            nacc -= 32u;
        }
    }
    __device__ __forceinline__ void entry(uint32_t e) { put(e >> 5, e & 31u); }
    __device__ __forceinline__ void lit(uint32_t b) { entry(s->lit()[b]); }
    __device__ __forceinline__ void flush()
    {
        if (nacc) atomicOr(&s->image[word], (uint32_t)acc);
    }
};

} // namespace dnaf
#include "gtz_tokens.hip.h"
namespace dnaf {

// byte of the virtual text at member offset pos, looked up from row i downwards
__device__ __forceinline__ uint32_t gtz_text_byte(const GtzParams& p, const GtzShared* s, uint32_t m_row, uint32_t i, int32_t pos)
{
    uint32_t j = min(i, s->nr - 1u);
    while (s->off[j] > pos) --j;
    const int32_t c = pos - s->cs[j];
    if (c < 0) return p.prefix[s->prefix_off[j] + (uint32_t)(pos - s->off[j])];
    const uint32_t k = (uint32_t)c >> 2, which = (uint32_t)c & 3u;
    if (which == 1u) return '/';
    if (which == 3u) return k == p.S - 1u ? '\n' : '\t';
    const uint32_t w = p.packed[(uint64_t)(m_row + j) * p.Wp + (k >> 4)];
    const uint32_t code = (w >> (2u * (k & 15u))) & 3u;
    return '0' + (which == 0u ? (code & 1u) : (code >> 1));
}

#ifdef DNAF_GTZ_PROFILE
// stage clock of gtz_encode<false>: cycles (s_memtime) summed over workgroups, per stage; debug builds only
__device__ unsigned long long gtz_prof[16];
#define GTZ_STAMP(k) do { if (!HIST && tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&gtz_prof[k], t_ - t_prev_); t_prev_ = t_; } } while (0)
#else
#define GTZ_STAMP(k) do { } while (0)
#endif

#ifndef DNAF_GTZ_EXP
#define DNAF_GTZ_EXP 0 // timing experiments (output invalid): 1 no search, 2 no emission walk, 3 no count walk, 4 no key tables, 5 no CRC
#endif
// The encoder is bound by latency (LDS round trips, dependent bit scans), not by instruction issue: seven workgroups per CU
// (LDS: 22.8 KB each) with 72 registers and a few spills beat five with 88 (profiles/r03_gtz_occupancy.txt).
#ifndef DNAF_GTZ_WAVES_PER_EU
#define DNAF_GTZ_WAVES_PER_EU 7
#endif
#if DNAF_GTZ_WAVES_PER_EU
#define GTZ_OCCUPANCY __attribute__((amdgpu_waves_per_eu(DNAF_GTZ_WAVES_PER_EU, 8)))
#else
#define GTZ_OCCUPANCY
#endif
template <bool HIST, uint32_t THREADS>
__global__ __launch_bounds__(THREADS) GTZ_OCCUPANCY void gtz_encode(const GtzParams p)
{
    constexpr uint32_t WAVES = THREADS / 64u;
#ifdef DNAF_GTZ_PROFILE
    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif
    __shared__ GtzShared sh;
    GtzShared* s = &sh;
    const uint32_t tid = threadIdx.x;
    const uint32_t m = HIST ? blockIdx.x * p.sample_step : blockIdx.x;
    if (m >= p.n_blocks) return;
    const uint32_t N = (uint32_t)min((uint64_t)BGZF_IN, p.n_bytes - (uint64_t)m * BGZF_IN);
    const uint32_t m_row = p.member_row[m]; // relative to row_begin
    const uint64_t B = p.text_base + (uint64_t)m * BGZF_IN;

    // ---- rows that overlap the member ----
    if (tid <= GTZ_MAX_ROWS) {
        const uint64_t r = (uint64_t)m_row + tid;
        int32_t off = 0x7FFFFFFF;
        if (r < p.n_rows) {
            const RowRec rec = p.recs[p.row_begin + r];
            const int64_t rel = (int64_t)(rec.text_off - B);
            if (rel < (int64_t)N) {
                off = (int32_t)rel;
                s->cs[tid] = off + (int32_t)rec.prefix_len;
                s->prefix_off[tid] = rec.prefix_off;
                s->cls[tid] = rec.cls;
                s->nall[tid] = rec.n_alleles;
                s->thr0[tid] = rec.thr[0];
            }
        }
        s->off[tid] = off;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t nr = 0, ok = 1;
        while (nr <= GTZ_MAX_ROWS && s->off[nr] != 0x7FFFFFFF) ++nr;
        if (nr > GTZ_MAX_ROWS || p.S < 2u) ok = 0;
        uint32_t items = 0, chunks = 0, words = 0, best_cells = 0, best_row = 0;
        int32_t prev_end = 0;
        for (uint32_t i = 0; i < nr && ok; ++i) {
            if (s->cls[i] != 0u || s->nall[i] > 2u) { ok = 0; break; }
            const int32_t cs = s->cs[i];
            uint32_t kb = cs >= 0 ? 0u : ((uint32_t)(-cs) + 3u) >> 2;
            uint32_t ke = (int32_t)N > cs ? (uint32_t)((int32_t)N - cs) >> 2 : 0u;
            ke = min(ke, p.S - 1u);
            if (ke < kb) ke = kb;
            s->kb[i] = kb;
            s->ke[i] = ke;
            const bool any = ke > kb;
            const uint32_t nch = any ? ((ke - 1u) >> 6) - (kb >> 6) + 1u : 0u;
            const uint32_t nw = any ? ((ke - 1u) >> 4) - (kb >> 4) + 1u : 0u;
            s->kw0[i] = kb >> 4;
            s->gw[i] = words;
            s->item_base[i] = items;
            s->chunk_base[i] = chunks;
            s->span_lo[i] = prev_end;
            s->span_hi[i] = any ? cs + 4 * (int32_t)kb : prev_end;
            if (any) prev_end = cs + 4 * (int32_t)ke;
            items += 1u + nch;
            chunks += nch;
            words += nw + 1u;
            if (ke - kb > best_cells) { best_cells = ke - kb; best_row = i; }
        }
        s->gw[nr] = words;
        s->item_base[nr] = items; // the closing span
        s->chunk_base[nr] = chunks;
        s->span_lo[nr] = prev_end;
        s->span_hi[nr] = (int32_t)N;
        items += 1u;
        if (items > THREADS || words + 1u > GTZ_CELL_WORDS) ok = 0; // (the host launches the 512-thread form where rows are short)
        for (uint32_t i = 0; i <= nr && ok; ++i)
            if (s->span_hi[i] - s->span_lo[i] > (int32_t)GTZ_SPAN_MAX) ok = 0;
        s->nr = nr;
        s->n_items = items;
        s->ok = ok;
        // table class from the alternative-allele frequency of the row that fills most of the member
        const float q = 1.0f - ((float)s->thr0[best_row] + 1.0f) * (1.0f / 4294967296.0f);
        s->klass = q < 0.02f ? 0u : q < 0.04f ? 1u : q < 0.08f ? 2u : q < 0.13f ? 3u : q < 0.2f ? 4u : q < 0.3f ? 5u : q < 0.4f ? 6u : 7u;
        if (!HIST && !p.tables->cls[s->klass].unit_ok) s->ok = 0; // this class's codes do not fit the walk's 32-bit units
    }
    __syncthreads();
    if (!s->ok) {
        if (!HIST && tid == 0) p.fast[m] = 0;
        return;
    }
    const uint32_t nr = s->nr, n_items = s->n_items, klass = s->klass;
    const GtzClassTables* ct = &p.tables->cls[klass];
    GTZ_STAMP(0); // rows located, layout decided

    // ---- g space, key tables, code tables ----
    for (uint32_t i = 0; i < nr; ++i) {
        const uint32_t nw = s->gw[i + 1] - s->gw[i];
        const uint32_t* src = p.packed + (uint64_t)(m_row + i) * p.Wp;
        for (uint32_t w = tid; w < nw; w += THREADS) {
            const uint32_t kw = s->kw0[i] + w;
            s->cells[s->gw[i] + w] = kw < p.W ? src[kw] : 0u;
            s->rowmap[s->gw[i] + w] = (uint8_t)i;
        }
    }
    if (tid < 4) { s->cells[s->gw[nr] + tid] = 0; s->rowmap[s->gw[nr] + tid] = (uint8_t)nr; }
    if (tid <= nr) {
        GtzRowInfo ri{0, 0, 0, 0}; // entry nr: the padding behind the last row, empty bounds
        if (tid < nr) {
            const int32_t g0 = 16 * (int32_t)s->gw[tid] - 16 * (int32_t)s->kw0[tid]; // g of the row's cell 0
            ri.delta = s->cs[tid] - 4 * g0;
            ri.gfirst = (uint32_t)(g0 + (int32_t)s->kb[tid]);
            ri.gend = (uint32_t)(g0 + (int32_t)s->ke[tid]);
        }
        s->row[tid] = ri;
    }
    // the text between the rows' cell regions (one global load per byte, all in flight together)
    for (uint32_t t = tid; t < (nr + 1u) * GTZ_SPAN_MAX; t += THREADS) {
        const uint32_t i = t / GTZ_SPAN_MAX;
        const int32_t pos = s->span_lo[i] + (int32_t)(t - i * GTZ_SPAN_MAX);
        if (pos < s->span_hi[i]) {
            const uint32_t b = gtz_text_byte(p, s, m_row, i, pos);
            s->span_bytes[t] = (uint8_t)b;
            if (!HIST && b >= 128u) s->ok = 0; // the literal table in LDS holds the ASCII half (VCF text of this build is ASCII)
        }
    }
    if (HIST) {
        for (uint32_t t = tid; t < GTZ_NSYM; t += THREADS) s->hist()[t] = 0;
        // self-check of the bit count (below): a made-up table of gap lengths, in the space the CRC table has in the other form
        // (entry 1 is the '/' literal, as in the real tables: 7 bits in the made-up literal lengths 3, 5, 7, 11)
        for (uint32_t t = tid; t < GTZ_GAP_STRIDE; t += THREADS) s->crc_nib[t] = t == 0u ? 0u : t == 1u ? 7u : (t * 37u + 11u) % 27u + 1u;
    } else {
        if (tid < 128) s->lit()[tid] = ct->lit[tid];
        for (uint32_t t = tid; t < GTZ_GAP_STRIDE; t += THREADS) s->gap()[t] = ct->gap[t];
        if (tid < GTZ_CHUNK_CELLS + 8) s->len4()[tid] = ct->len4[tid];
        if (tid < 32) s->dist()[tid] = ct->dist[tid];
        if (tid < 256) s->crc_table[tid] = p.tables->crc_table[tid];
        if (tid <= GTZ_CHUNK_CELLS) s->crc_base[tid] = p.tables->crc_base[tid];
        for (uint32_t t = tid; t < (GTZ_CHUNK_CELLS / 2) * 16; t += THREADS) s->crc_nib[t] = p.tables->crc_nib[t];
    }
    __syncthreads();
    if (!s->ok) { // a byte the literal table does not hold
        if (!HIST && tid == 0) p.fast[m] = 0;
        return;
    }
    GTZ_STAMP(1); // cells and code tables in LDS
    // Key tables: for every segment of the g space the earliest and the latest position of every 5-cell key, 16-bit
    // entries.  No atomics, no reads and no validity tests: plain stores in ascending ("latest") and descending ("earliest")
    // order; lanes of one step race, so both are approximate to the step (some occurrence near the end / the beginning);
    // the search verifies whatever it finds against the cells themselves and against the bounds of the source's row, so
    // any position that is entered is harmless.  (Entering only positions whose cell differs from the
    // cell before -- the only ones the search looks up -- was measured: slower and 1 % more bytes.)
    {
        for (uint32_t t = tid; t < (2 * GTZ_SEGS - 1) * GTZ_KEYS / 2; t += THREADS) { // [quarter][earliest, latest + 1][key], two entries per word
            const uint32_t which = (t / (GTZ_KEYS / 2u)) & 1u;
            reinterpret_cast<uint32_t*>(s->tab16)[t] = which ? 0u : 0xFFFFFFFFu;
        }
        __syncthreads();
        const uint32_t total = DNAF_GTZ_EXP == 4 ? 0u : 16u * s->gw[nr];
        // Every segment is swept by its own wave(s) in steps of 256 positions, four consecutive ones per lane (one window read
        // gives their keys), with plain stores only: ascending for "latest", then descending for "earliest" -- the last
        // writer of an entry is the highest / lowest position.  A table has one writer in program order, so both are exact
        // to the 256 positions of a step, whatever the waves' pace.
        {
            constexpr uint32_t WPS = WAVES / GTZ_SEGS ? WAVES / GTZ_SEGS : 1u; // waves per segment (256 threads, 3 segments: one, the fourth wave rests)
            const uint32_t wave_ = tid >> 6, lane_ = tid & 63u;
            const uint32_t sub = wave_ % WPS;
            for (uint32_t q = wave_ / WPS; q < GTZ_SEGS; q += WAVES / WPS) { // (more segments than waves: several each, experiments)
                const uint32_t seg_lo = q * GTZ_SEG_CELLS, seg_hi = q + 1u < GTZ_SEGS ? min(total, seg_lo + GTZ_SEG_CELLS) : total;
                uint16_t* tab = &s->tab16[(2u * q) * GTZ_KEYS];
                // "latest": ascending, plain stores; "earliest": descending, plain stores (the last writer is the lowest position)
                if (q + 1u < GTZ_SEGS)
                    for (uint32_t g0 = seg_lo + 256u * sub + 4u * lane_; g0 < seg_hi; g0 += 256u * WPS) {
                        const uint32_t w = gtz_window(s->cells, g0);
#pragma unroll
                        for (int k = 0; k < 4; ++k) tab[GTZ_KEYS + ((w >> (2 * k)) & (GTZ_KEYS - 1u))] = (uint16_t)(g0 + (uint32_t)k + 1u);
                    }
                if (seg_lo < seg_hi) { // (a member shorter than this segment's start has nothing here)
                    const uint32_t steps = (seg_hi - seg_lo + 256u * WPS - 1u) / (256u * WPS);
                    for (uint32_t st = steps; st-- > 0u;) {
                        const uint32_t g0 = seg_lo + 256u * (st * WPS + sub) + 4u * lane_;
                        if (g0 >= seg_hi) continue;
                        const uint32_t w = gtz_window(s->cells, g0);
#pragma unroll
                        for (int k = 3; k >= 0; --k) tab[(w >> (2 * k)) & (GTZ_KEYS - 1u)] = (uint16_t)(g0 + (uint32_t)k);
                    }
                }
            }
        }
        __syncthreads();
    }

    GTZ_STAMP(2); // key tables
    // code lengths of the four literals a cell is made of, one byte each
    const uint32_t lit_bits = HIST ? 0u : (s->lit()['0'] & 31u) | ((s->lit()['1'] & 31u) << 8) | ((s->lit()['/'] & 31u) << 16) | ((s->lit()['\t'] & 31u) << 24);
    // ---- pass 1: this thread's item -- a literal span or a 64-cell chunk: search, then count bits (or symbols); CRC ----
    uint32_t bits = 0, part = 0;
    uint32_t it_row = 0xFFFFFFFFu;
    bool it_span = false;
    GtzChunk ch{0, 0, 0, 0, 0, 0, false};
    GtzEvents evs;
    if (tid < n_items) {
        uint32_t i = 0;
        while (i < nr && s->item_base[i + 1] <= tid) ++i; // row (or nr: the closing span)
        it_row = i;
        const uint32_t local = tid - s->item_base[i];
        uint32_t crc = 0, nbytes = 0;
        int32_t end_byte = 0;
        if (local == 0u) {
            // literal span: bytes between two rows' cell regions
            const int32_t lo = s->span_lo[i], hi = s->span_hi[i];
            it_span = true;
            uint32_t c = 0xFFFFFFFFu;
            GtzCount cnt{s, 0};
            GtzHist hs{s};
            for (int32_t pos = lo; pos < hi; ++pos) {
                const uint32_t b = s->span_bytes[i * GTZ_SPAN_MAX + (uint32_t)(pos - lo)];
                if (HIST) hs.lit(b);
                else { cnt.lit(b); c = s->crc_table[(c ^ b) & 0xFFu] ^ (c >> 8); }
            }
            bits = cnt.bits;
            crc = ~c;
            nbytes = (uint32_t)(hi - lo);
            end_byte = hi;
        } else {
            const uint32_t kb = s->kb[i], ke = s->ke[i];
            const uint32_t c64 = (kb >> 6) + local - 1u;
            const uint32_t k0 = max(kb, 64u * c64), k1 = min(ke, 64u * c64 + 64u);
            ch = gtz_load_chunk(s, i, c64, k0, k1);
            evs = gtz_find_matches<!HIST>(s, i, c64, ch, DNAF_GTZ_EXP != 1);
            if (HIST) {
                GtzSymbolCount hs{s->hist()};
                gtz_walk(ch, evs, hs);
                // Self-check on the sampled members, in every run: the chunks' bit offsets come from gtz_count_bits, mask
                // arithmetic that must agree with the walk that writes the bits.  Both are run here with a made-up code-length
                // table (the real ones do not exist yet -- this kernel collects the histogram they are built from; far matches'
                // own symbols count 0 on both sides); a disagreement is counted in a spare slot of the histogram and the host
                // refuses to go on (dnaf_bgzf_compress).
                const uint32_t fake_lit = 0x0B070503u;
                GtzBitCount cnt{s->crc_nib, s->crc_nib, s->crc_nib, fake_lit, 0};
                gtz_walk(ch, evs, cnt);
                if (cnt.bits - cnt.far_seen != gtz_count_bits(ch, evs, s->crc_nib, fake_lit)) atomicAdd(&p.hist[klass * GTZ_NSYM + GTZ_NSYM - 1u], 1u);
            } else {
                bits = DNAF_GTZ_EXP == 3 ? 150u : gtz_count_bits(ch, evs, s->gap(), lit_bits);
#ifdef DNAF_GTZ_CHECK_COUNT
                { // debug builds: the mask arithmetic against the walk it replaces
                    GtzBitCount cnt{s->gap(), s->len4(), s->dist(), lit_bits, 0};
                    gtz_walk(ch, evs, cnt);
                    if (cnt.bits != bits) {
                        printf("gtz count mismatch: member %u item %u: %u (masks) vs %u (walk) j0 %u j1 %u head %d ev %016llx:%016llx far %016llx:%016llx\n", m, tid, bits,
                               cnt.bits, ch.j0, ch.j1, (int)ch.head, (unsigned long long)evs.ev.hi, (unsigned long long)evs.ev.lo, (unsigned long long)evs.far.hi,
                               (unsigned long long)evs.far.lo);
                        bits = cnt.bits;
                    }
                }
#endif
                // CRC of the chunk's bytes from its cells: CRC is affine, so it is the CRC of n cells "0/0\t" XOR one unit per '1'
                // allele, the unit depending on how far before the chunk's end the allele lies.  The cells are shifted so that
                // the chunk's last cell is cell 63, and the units come four alleles at a time from a table by pairs of cells:
                // 32 look-ups whatever the row (a loop over the set alleles took 12 instructions each, up to 90 of them in a
                // chunk of a MAF-0.45 row).
                const uint32_t n = k1 - k0;
                uint32_t cc = s->crc_base[n];
                if (DNAF_GTZ_EXP != 5) {
                    const GtzMask in = gtz_range(2u * ch.j0, 2u * ch.j1);
                    uint64_t vlo = ch.lo & in.lo, vhi = ch.hi & in.hi;
                    const uint32_t sh = 2u * (64u - ch.j1); // 0 .. 126
                    if (sh >= 64u) { vhi = vlo << (sh - 64u); vlo = 0; }
                    else if (sh) { vhi = (vhi << sh) | (vlo >> (64u - sh)); vlo <<= sh; }
                    const uint32_t w[4] = {(uint32_t)vlo, (uint32_t)(vlo >> 32), (uint32_t)vhi, (uint32_t)(vhi >> 32)};
#pragma unroll
                    for (uint32_t k = 0; k < 32u; ++k) cc ^= s->crc_nib[(31u - k) * 16u + ((w[k >> 3] >> (4u * (k & 7u))) & 15u)];
                }
                crc = cc;
                nbytes = 4u * n;
                end_byte = s->cs[i] + 4 * (int32_t)k1;
            }
        }
        if (!HIST && nbytes) {
            const uint32_t after = N - (uint32_t)end_byte;
            const uint32_t q = after >> 6, r = after & 63u;
            uint32_t shift = p.tables->crc_shift_chunk[q];
            if (r) shift = crc_mulmod(shift, p.tables->crc_shift_byte[r]);
            part = after ? crc_mulmod(shift, crc) : crc;
        }
    }
    if (HIST) {
        __syncthreads();
        for (uint32_t t = tid; t < GTZ_NSYM; t += THREADS)
            if (s->hist()[t]) atomicAdd(&p.hist[klass * GTZ_NSYM + t], s->hist()[t]);
        if (tid == 0) atomicAdd(&p.hist[klass * GTZ_NSYM + 256u], 1u); // end of block
        return;
    }

    // ---- offsets: exclusive scan of the items' bit counts ----
    GTZ_STAMP(3); // thread 0's own search + count + CRC (not the slowest thread's)
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint32_t incl = bits;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n0 = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += n0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part ^= __shfl_xor(part, d, 64);
    if (lane == 63u) s->scratch[wave] = incl;
    if (lane == 0u) s->scratch[GTZ_MAX_WAVES + wave] = part;
    __syncthreads(); // also: every search is done, the key tables and the cells are free
    GTZ_STAMP(4); // waiting for the slowest thread's pass 1
    uint32_t base = 0, total = 0, member_crc = 0;
    for (uint32_t w = 0; w < WAVES; ++w) {
        const uint32_t a = s->scratch[w];
        if (w < wave) base += a;
        total += a;
        member_crc ^= s->scratch[GTZ_MAX_WAVES + w];
    }
    const uint32_t hdr_bits = ct->hdr_nbits;
    const uint32_t body_bits = hdr_bits + total + (ct->eob & 31u);
    const uint32_t body = (body_bits + 7u) >> 3;
    const uint32_t size = body + BGZF_MEMBER_OVERHEAD;
    if (size + 8u > 4u * GTZ_IMAGE_WORDS || body >= N + 5u) { // does not fit the image, or a stored block would be smaller
        if (tid == 0) p.fast[m] = 0;
        return;
    }

    GTZ_STAMP(5);
    // ---- pass 2: the member image in LDS ----
    const uint32_t image_words = (size + 3u) >> 2;
    for (uint32_t t = tid; t < image_words + 1u; t += THREADS) s->image[t] = 0;
    __syncthreads();
    if (tid < 18) {
        const uint8_t h[18] = {0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 0x42, 0x43, 2, 0,
                               (uint8_t)((size - 1u) & 0xFFu), (uint8_t)((size - 1u) >> 8)};
        atomicOr(&s->image[tid >> 2], (uint32_t)h[tid] << (8u * (tid & 3u)));
    }
    if (tid < 8) {
        const uint32_t v = tid < 4 ? member_crc : N;
        const uint32_t at = 18u + body + tid;
        atomicOr(&s->image[at >> 2], ((v >> (8u * (tid & 3u))) & 0xFFu) << (8u * (at & 3u)));
    }
    constexpr uint32_t BIT0 = 18u * 8u; // deflate bits start behind the gzip header
    if (tid < (hdr_bits + 31u) / 32u) {
        const uint32_t v = ct->hdr[tid];
        atomicOr(&s->image[(BIT0 >> 5) + tid], v << (BIT0 & 31u));
        atomicOr(&s->image[(BIT0 >> 5) + tid + 1u], v >> (32u - (BIT0 & 31u)));
    }
    if (it_row != 0xFFFFFFFFu) {
        const uint32_t start = BIT0 + hdr_bits + base + incl - bits;
        if (it_span) {
            GtzEmit e{s, 0, start & 31u, start >> 5};
            const uint32_t i = it_row;
            for (int32_t pos = s->span_lo[i]; pos < s->span_hi[i]; ++pos) e.lit(s->span_bytes[i * GTZ_SPAN_MAX + (uint32_t)(pos - s->span_lo[i])]);
            e.flush();
        } else {
            GtzBitEmit e{s->image, s->gap(), s->len4(), s->dist(), {s->lit()['0'], s->lit()['1'], s->lit()['/'], s->lit()['\t']}, start};
            if (DNAF_GTZ_EXP != 2) gtz_walk(ch, evs, e);
        }
    }
    if (tid == 0) {
        const uint32_t start = BIT0 + hdr_bits + total;
        GtzEmit e{s, 0, start & 31u, start >> 5};
        e.entry(ct->eob);
        e.flush();
        p.sizes[m] = size;
        p.fast[m] = 1;
    }
    GTZ_STAMP(6); // thread 0's emission
    __syncthreads();
    GTZ_STAMP(7); // waiting for the slowest thread's emission
    uint4* dst = reinterpret_cast<uint4*>(p.stage + (uint64_t)m * GTZ_STRIDE);
    const uint4* src = reinterpret_cast<const uint4*>(s->image);
    for (uint32_t t = tid; t < (size + 15u) >> 4; t += THREADS) dst[t] = src[t];
    GTZ_STAMP(8); // copy to the staging slot
}

// staging slot -> the member's place in the stream (offsets from bgzf_scan)
__global__ __launch_bounds__(256) void gtz_compact(const GtzParams p)
{
    const uint32_t m = blockIdx.x;
    if (!p.fast[m]) return;
    const uint32_t size = p.sizes[m] & 0x3FFFFFFFu;
    const uint8_t* src = p.stage + (uint64_t)m * GTZ_STRIDE;
    uint8_t* dst = p.out + p.offsets[m];
    const uint32_t n16 = size >> 4;
    for (uint32_t t = threadIdx.x; t < n16; t += 256u) {
        const uint4 v = reinterpret_cast<const uint4*>(src)[t];
        __builtin_memcpy(dst + 16u * t, &v, 16);
    }
    for (uint32_t t = 16u * n16 + threadIdx.x; t < size; t += 256u) dst[t] = src[t];
}

} // namespace dnaf
