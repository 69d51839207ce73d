// rowgen_kernels.hip.h -- device side of the row generator (gfx950 / CDNA4).
//
// One wave (64 lanes) owns one contiguous segment of one SNP row.  A lane
// owns one *group* of 4 consecutive samples per iteration: two Philox4x32-10
// blocks give the 8 allele draws of the group (sample i, allele a <-> draw
// 2i+a, the lane mapping of pop_factory.py:477/:486/:492), the cell rules of
// pop_factory.py:479-499 turn them into 4 cells, and the lane stores 16
// bytes of VCF text (autosome) plus one byte of the 2-bit matrix.  Lanes of a
// wave therefore write 1 KiB of contiguous text per iteration.
//
// No MFMA: the path is integer VALU (Philox) + streaming stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dnaf {

// What a wave reads per row (32 B): the "SNP record" of the bytes model.
struct RowRec {
    uint32_t thr[3];     // integer thresholds: allele index = first k with draw <= thr[k]
    uint32_t prefix_off; // offset of the row's fixed columns in the prefix blob
    uint64_t text_off;   // row start in the table's text stream
    int32_t slot;        // forced-set slot of this SNP, -1 if not deleterious
    uint16_t prefix_len;
    uint8_t cls;         // DNAF_CLASS_*
    uint8_t n_alleles;
};
static_assert(sizeof(RowRec) == 32, "RowRec must stay 32 bytes");

struct GenParams {
    const RowRec* recs;       // [table rows]
    const uint8_t* prefix;    // prefix blob
    const uint8_t* male_nib;  // [G] bit k = sample 4g+k is male
    const uint32_t* xoff;     // [G+1] byte offset of group g's cells in an X row
    const uint8_t* forced;    // [n_slots][G] bit k = sample 4g+k carries the slot's SNP
    uint8_t* text;            // output text of this call (row_begin's first byte at text[0])
    uint8_t* packed;          // output [n_rows][4*W] bytes
    uint64_t text_base;       // recs[row_begin].text_off
    uint64_t row_begin;       // first table row of this call
    uint64_t n_rows;
    uint64_t rng_row0;        // RNG row of table row 0
    uint32_t key0, key1;      // Philox key = seed
    uint32_t S, G, W;         // samples, groups of 4, packed words per row
    uint32_t nseg, seg_groups; // segments per row, groups per segment (multiple of 64)
};

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(PHILOX_M0, c0), lo0 = PHILOX_M0 * c0;
        const uint32_t hi1 = __umulhi(PHILOX_M1, c2), lo1 = PHILOX_M1 * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    return make_uint4(c0, c1, c2, c3);
}

// SNPTuples.pick_allele_index (pop_factory.py:92-95) on integer thresholds:
// roll = u * 2^-32, `cum >= roll`  <=>  u <= floor(cum * 2^32).
template <bool MULTI>
__device__ __forceinline__ uint32_t pick(uint32_t u, uint32_t t0, uint32_t t1, uint32_t t2)
{
    if (!MULTI) return u > t0 ? 1u : 0u;
    return u <= t0 ? 0u : (u <= t1 ? 1u : (u <= t2 ? 2u : 3u));
}

template <int N>
__device__ __forceinline__ void store_unaligned(uint8_t* dst, const void* src)
{
    __builtin_memcpy(dst, src, N); // byte-aligned destination: hardware unaligned global stores
}

// One row segment, chromosome class known (wave-uniform).
template <bool TEXT, bool PACKED, bool MULTI>
__device__ __forceinline__ void row_segment(const GenParams& p, const RowRec& rec, uint64_t row_local,
                                            uint32_t g_begin, uint32_t g_end, uint32_t lane)
{
    const uint64_t rng_row = p.rng_row0 + p.row_begin + row_local;
    const uint32_t r_lo = (uint32_t)rng_row, r_hi = (uint32_t)(rng_row >> 32);
    const uint32_t t0 = rec.thr[0], t1 = rec.thr[1], t2 = rec.thr[2];
    const uint32_t cls = rec.cls;
    uint8_t* cells = p.text + (rec.text_off - p.text_base) + rec.prefix_len;
    uint8_t* prow = p.packed + row_local * (uint64_t)p.W * 4u;
    const uint8_t* frow = rec.slot >= 0 ? p.forced + (uint64_t)rec.slot * p.G : nullptr;
    const uint32_t S = p.S;
    const uint32_t g_stop = PACKED ? (g_end == p.G ? p.W * 4u : g_end) : g_end; // last segment also zeroes the pad bytes

    for (uint32_t g = g_begin + lane; g < g_stop; g += 64) {
        if (g >= p.G) { // pad bytes of the packed row
            if (PACKED) prow[g] = 0;
            continue;
        }
        const uint32_t nvalid = min(4u, S - 4u * g);
        const uint32_t male = p.male_nib[g];
        const uint32_t forced = frow ? frow[g] : 0u;
        const uint4 x = philox4x32_10(2u * g, 0u, r_lo, r_hi, p.key0, p.key1);
        const uint4 y = philox4x32_10(2u * g + 1u, 0u, r_lo, r_hi, p.key0, p.key1);
        uint32_t a[4] = {pick<MULTI>(x.x, t0, t1, t2), pick<MULTI>(x.z, t0, t1, t2), pick<MULTI>(y.x, t0, t1, t2),
                         pick<MULTI>(y.z, t0, t1, t2)};
        uint32_t b[4] = {pick<MULTI>(x.y, t0, t1, t2), pick<MULTI>(x.w, t0, t1, t2), pick<MULTI>(y.y, t0, t1, t2),
                         pick<MULTI>(y.w, t0, t1, t2)};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((forced >> k) & 1u) { a[k] = 1u; b[k] = 1u; } // pop_factory.py:495-499

        uint32_t pk = 0;
        if (cls == 0u) {
            // autosome: every cell "a/b" + separator = one dword
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t sep = (4u * g + k == S - 1u) ? 0x0A000000u : 0x09000000u;
                w[k] = 0x00302F30u + a[k] + (b[k] << 16) + sep;
                pk |= ((a[k] ? 1u : 0u) | (b[k] ? 2u : 0u)) << (2 * k);
            }
            if (TEXT) {
                uint8_t* dst = cells + 16ull * g;
                if (nvalid == 4u) store_unaligned<16>(dst, w);
                else
                    for (uint32_t k = 0; k < nvalid; ++k) store_unaligned<4>(dst + 4u * k, &w[k]);
            }
            if (nvalid < 4u) pk &= (1u << (2u * nvalid)) - 1u;
        } else if (cls == 1u) {
            // X: haploid "a" for males, diploid "a/b" for females (common/snp.py:102-109)
            uint8_t* dst = cells + p.xoff[g];
            for (uint32_t k = 0; k < nvalid; ++k) {
                const uint32_t last = (4u * g + k == S - 1u);
                if ((male >> k) & 1u) {
                    pk |= (a[k] ? 1u : 0u) << (2 * k);
                    if (TEXT) {
                        const uint16_t h = (uint16_t)((0x30u + a[k]) | ((last ? 0x0Au : 0x09u) << 8));
                        store_unaligned<2>(dst, &h);
                        dst += 2;
                    }
                } else {
                    pk |= ((a[k] ? 1u : 0u) | (b[k] ? 2u : 0u)) << (2 * k);
                    if (TEXT) {
                        const uint32_t w = 0x00302F30u + a[k] + (b[k] << 16) + (last ? 0x0A000000u : 0x09000000u);
                        store_unaligned<4>(dst, &w);
                        dst += 4;
                    }
                }
            }
        } else {
            // Y: '.' for females first (pop_factory.py:481-484), haploid for males; MT: haploid for everyone
            uint8_t* dst = cells + 8ull * g;
            for (uint32_t k = 0; k < nvalid; ++k) {
                const uint32_t last = (4u * g + k == S - 1u);
                uint32_t ch;
                if (cls == 2u && !((male >> k) & 1u)) ch = 0x2Eu; // '.'
                else {
                    ch = 0x30u + a[k];
                    pk |= (a[k] ? 1u : 0u) << (2 * k);
                }
                if (TEXT) {
                    const uint16_t h = (uint16_t)(ch | ((last ? 0x0Au : 0x09u) << 8));
                    store_unaligned<2>(dst + 2u * k, &h);
                }
            }
        }
        if (PACKED) prow[g] = (uint8_t)pk;
    }
}

template <bool TEXT, bool PACKED>
__global__ __launch_bounds__(256) void rowgen_kernel(const GenParams p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t row_local = item / p.nseg;
    const uint32_t seg = (uint32_t)(item % p.nseg);
    if (row_local >= p.n_rows) return;
    const RowRec rec = p.recs[p.row_begin + row_local];

    if (TEXT && seg == 0) {
        // the row's fixed columns (pop_factory.py:503-507), formatted on the host; S == 0: line = prefix + "\n"
        uint8_t* dst = p.text + (rec.text_off - p.text_base);
        const uint8_t* src = p.prefix + rec.prefix_off;
        for (uint32_t i = lane; i < rec.prefix_len; i += 64) dst[i] = src[i];
        if (p.S == 0 && lane == 0) dst[rec.prefix_len] = 0x0A;
    }
    const uint32_t g_begin = seg * p.seg_groups;
    if (g_begin >= p.G && !(PACKED && seg == 0)) return;
    const uint32_t g_end = min(p.G, g_begin + p.seg_groups);
    if (rec.n_alleles > 2) row_segment<TEXT, PACKED, true>(p, rec, row_local, g_begin, g_end, lane);
    else row_segment<TEXT, PACKED, false>(p, rec, row_local, g_begin, g_end, lane);
}

} // namespace dnaf
