// rowgen_kernels.hip.h -- device side of the row generator (gfx950 / CDNA4).
//
// One wave (64 lanes) owns one contiguous segment of one SNP row.  A lane
// owns one *group* of 4 consecutive samples per iteration: two Philox4x32-10
// blocks give the 8 allele draws of the group (sample i, allele a <-> draw
// 2i+a, the lane mapping of pop_factory.py:477/:486/:492), the cell rules of
// pop_factory.py:479-499 turn them into 4 cells, and the lane stores its
// cells' VCF text (16 bytes on an autosome) plus one byte of the 2-bit
// matrix.  The 64 lanes of a wave therefore write 1 KiB of contiguous text
// per iteration; row starts are byte-aligned only, and gfx950 runs the
// resulting unaligned dwordx4 stores at ~94 % of the aligned rate
// (profiles/r01_microbench.log), so no realignment pass is needed.
//
// The row's record is wave-uniform and lives in SGPRs; the per-sample tables
// (male / forced nibbles, X-row offsets) are read coalesced, one byte or
// dword per lane per iteration, and stay L2-resident.
//
// No MFMA: the path is integer VALU (Philox: 20 v_mad_u64_u32 + 20 v_bitop3_b32
// per block) + streaming stores.
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
    const RowRec* recs;        // [table rows]
    const uint8_t* prefix;     // prefix blob
    const uint8_t* male_nib;   // [G] bit k = sample 4g+k is male
    const uint32_t* xoff;      // [G+1] byte offset of group g's cells in an X row
    const uint8_t* forced;     // [n_slots][G] bit k = sample 4g+k carries the slot's SNP
    uint8_t* text;             // output text of this call (row_begin's first byte at text[0])
    uint8_t* packed;           // output [n_rows][4*W] bytes, row j at packed + j * packed_pitch
    uint64_t packed_pitch;     // bytes between packed rows (4*W for the ABI's dense matrix)
    uint64_t text_base;        // recs[row_begin].text_off
    uint64_t row_begin;        // first table row of this call
    uint64_t n_rows;
    uint64_t rng_row0;         // RNG row of table row 0
    uint32_t rng_div;          // table rows k * rng_div .. k * rng_div + rng_div - 1 share RNG row k (1: every row its own)
    uint32_t key0, key1;       // Philox key = seed
    uint32_t S, G, W;          // samples, groups of 4, packed words per row
    uint32_t nseg, seg_groups; // segments per row, groups per segment (multiple of 64)
    uint32_t rows_per_wave;    // rowrun_kernel: whole rows per wave
    uint32_t prefix_stride;    // rowmix_kernel: bytes per row in prefix_fixed (multiple of 16, <= 64); 0: no such copy
    const uint8_t* prefix_fixed; // [table rows][prefix_stride] the rows' fixed columns at a fixed stride
    uint32_t mix_gmagic;       // rowmix_kernel: min(floor(2^32 / G), 2^32 - 1), for q / G by one multiply-high (divmod_small)
    uint32_t mix_step_rows, mix_step_groups; // rowmix_kernel: a trip ahead = 64 groups = 64 / G rows + 64 mod G groups
    uint32_t mix_x_rows;       // rowmix_kernel: unforced biallelic X rows ride in the lane-spanning trips too (xgroup is set)
    const uint32_t* xgroup;    // [G] per group for X rows: cell offset | bytes << 16 | male nibble << 24 (dnaf_set_samples)
};

// RNG row (relative to rng_row0) of table row j.  rng_div > 1 reproduces the reference's forked workers, which all draw
// the parent's stream: with -n N, rows k N + 1 .. k N + N of a work chunk are made from the same 2 S uniforms
// (pop_factory.py:426-434).
__device__ __forceinline__ uint64_t rng_row_of(const GenParams& p, uint64_t j) { return p.rng_div > 1u ? j / p.rng_div : j; }

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

// Philox4x32-10; the two 32x32->64 products of a round are one v_mad_u64_u32 each.
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96); // a ^ b ^ c in one v_bitop3_b32
        c1 = (uint32_t)p1;
        c2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c3 = (uint32_t)p0;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    return make_uint4(c0, c1, c2, c3);
}

// The same block when the first round's product M0 * c0 is already known.  Callers that walk a row compute it
// as one multiply-add from a running counter: M0 * (c0 + d) = M0 * c0 + d * M0, with d * M0 a constant.
__device__ __forceinline__ uint4 philox4x32_10_p0(uint64_t p0_first, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                  uint32_t k1)
{
    uint32_t c0 = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = r == 0 ? p0_first : (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        c0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        c1 = (uint32_t)p1;
        c2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c3 = (uint32_t)p0;
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

// pointers into LDS carry their address space, so that accesses through a struct member stay ds_* instructions
#define LDS_PTR(T) T __attribute__((address_space(3)))*
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4))); // plain vector: loads through an LDS pointer

struct SegCtx {
    uint32_t t0, t1, t2;   // thresholds
    uint32_t r_lo, r_hi;   // RNG row
    uint32_t k0, k1;       // key
    uint32_t S, G;
    uint8_t* cells;        // first cell byte of the row
    uint8_t* prow;         // packed row
    const uint8_t* frow;   // forced nibbles of the row's slot or nullptr
    const uint8_t* male_nib;
    const uint32_t* xoff;
    const LDS_PTR(u32x4_t) lut; // the 16 text bytes of 4 biallelic diploid cells for each packed byte
    LDS_PTR(uint8_t) stage;     // this wave's packed bytes of the segment (PK_STAGE bytes), written out in 16-byte pieces
};
constexpr uint32_t PK_STAGE = 2048; // packed bytes a wave can stage = groups per segment at 32 trips

// The 8 allele indices of group g: a[k], b[k] = first / second allele of sample 4g+k,
// with the forced genotype 1/1 applied (pop_factory.py:485, :495-499).  General form (any allele count).
template <bool MULTI>
__device__ __forceinline__ void group_alleles(const SegCtx& c, uint32_t g, uint32_t& a0, uint32_t& a1, uint32_t& a2,
                                              uint32_t& a3, uint32_t& b0, uint32_t& b1, uint32_t& b2, uint32_t& b3)
{
    const uint4 x = philox4x32_10(2u * g, 0u, c.r_lo, c.r_hi, c.k0, c.k1);
    const uint4 y = philox4x32_10(2u * g + 1u, 0u, c.r_lo, c.r_hi, c.k0, c.k1);
    a0 = pick<MULTI>(x.x, c.t0, c.t1, c.t2); b0 = pick<MULTI>(x.y, c.t0, c.t1, c.t2);
    a1 = pick<MULTI>(x.z, c.t0, c.t1, c.t2); b1 = pick<MULTI>(x.w, c.t0, c.t1, c.t2);
    a2 = pick<MULTI>(y.x, c.t0, c.t1, c.t2); b2 = pick<MULTI>(y.y, c.t0, c.t1, c.t2);
    a3 = pick<MULTI>(y.z, c.t0, c.t1, c.t2); b3 = pick<MULTI>(y.w, c.t0, c.t1, c.t2);
    if (c.frow) { // wave-uniform
        const uint32_t f = c.frow[g];
        if (f & 1u) a0 = b0 = 1u;
        if (f & 2u) a1 = b1 = 1u;
        if (f & 4u) a2 = b2 = 1u;
        if (f & 8u) a3 = b3 = 1u;
    }
}

__device__ __forceinline__ uint32_t nz(uint32_t v) { return v ? 1u : 0u; }

#define DIPLOID_CELL(a, b, sep) (0x00302F30u + (a) + ((b) << 16) + ((sep) << 24)) /* "a/b" + separator */
#define HAPLOID_CELL(ch, sep) ((ch) | ((sep) << 8))                                /* "a" + separator   */

// Biallelic rows: the diploid cell "a/b\t" of one sample straight from its two draws.
#define CELL00 0x09302F30u /* "0/0\t" */
__device__ __forceinline__ uint32_t biallelic_cell(uint32_t ua, uint32_t ub, uint32_t t0)
{
    return (ub > t0 ? CELL00 + 0x10000u : CELL00) + (ua > t0 ? 1u : 0u);
}

// d[k] = "a/b\t" of sample 4g+k on a biallelic row, forced genotypes applied.
__device__ __forceinline__ void biallelic_cells(const SegCtx& c, uint32_t g, uint32_t& d0, uint32_t& d1, uint32_t& d2,
                                                uint32_t& d3)
{
    const uint4 x = philox4x32_10(2u * g, 0u, c.r_lo, c.r_hi, c.k0, c.k1);
    const uint4 y = philox4x32_10(2u * g + 1u, 0u, c.r_lo, c.r_hi, c.k0, c.k1);
    d0 = biallelic_cell(x.x, x.y, c.t0);
    d1 = biallelic_cell(x.z, x.w, c.t0);
    d2 = biallelic_cell(y.x, y.y, c.t0);
    d3 = biallelic_cell(y.z, y.w, c.t0);
    if (c.frow) { // wave-uniform; forced = "1/1", pop_factory.py:495-499
        const uint32_t f = c.frow[g];
        if (f & 1u) d0 = CELL00 + 0x10001u;
        if (f & 2u) d1 = CELL00 + 0x10001u;
        if (f & 4u) d2 = CELL00 + 0x10001u;
        if (f & 8u) d3 = CELL00 + 0x10001u;
    }
}

// pk = 2 pk + (u > t): the compare leaves its result in VCC and the add-with-carry shifts it in -- two
// instructions per draw (hipcc turns the C form into compare, select, shift and or).  t is wave-uniform.
__device__ __forceinline__ uint32_t shift_in_gt(uint32_t pk, uint32_t u, uint32_t t)
{
    asm("v_cmp_lt_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(pk) : "v"(u), "s"(t) : "vcc");
    return pk;
}
// the first draw of a byte: 0 + 0 + (u > t), no register to clear
__device__ __forceinline__ uint32_t first_gt(uint32_t u, uint32_t t)
{
    uint32_t pk;
    asm("v_cmp_lt_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, 0, 0, vcc" : "=v"(pk) : "v"(u), "s"(t) : "vcc");
    return pk;
}

// The packed byte of group g on a biallelic row straight from its 8 draws: bit 2k = first allele of sample 4g+k,
// bit 2k+1 = its second allele, forced genotypes applied.  One compare and one add-with-carry per draw; the
// text of the four cells is then one 16-byte LDS look-up (SegCtx::lut) instead of selects, ors and shifts.
// c0 = 2 g is the Philox counter of the group's first block (callers that walk a row keep it in a register);
// D = a constant number of groups ahead of c0's (the second trip of an unrolled pair): the four first-round
// products are multiply-adds of the one counter, M0 * (c0 + 2 D + j) = M0 * c0 + (2 D + j) * M0.
template <uint32_t D = 0>
__device__ __forceinline__ uint32_t biallelic_code(const SegCtx& c, uint32_t c0)
{
    const uint4 x = philox4x32_10_p0((uint64_t)PHILOX_M0 * c0 + (uint64_t)PHILOX_M0 * (2u * D), 0u, c.r_lo, c.r_hi, c.k0, c.k1);
    const uint4 y = philox4x32_10_p0((uint64_t)PHILOX_M0 * c0 + (uint64_t)PHILOX_M0 * (2u * D + 1u), 0u, c.r_lo, c.r_hi, c.k0, c.k1);
    uint32_t pk = first_gt(y.w, c.t0);
    pk = shift_in_gt(pk, y.z, c.t0);
    pk = shift_in_gt(pk, y.y, c.t0);
    pk = shift_in_gt(pk, y.x, c.t0);
    pk = shift_in_gt(pk, x.w, c.t0);
    pk = shift_in_gt(pk, x.z, c.t0);
    pk = shift_in_gt(pk, x.y, c.t0);
    pk = shift_in_gt(pk, x.x, c.t0);
    if (c.frow) { // wave-uniform; forced = "1/1", pop_factory.py:495-499
        uint32_t f = c.frow[(c0 >> 1) + D] & 0xFu;
        f = (f | (f << 2)) & 0x33u;
        f = (f | (f << 1)) & 0x55u;
        pk |= f * 3u;
    }
    return pk;
}

// 2-bit codes of 4 biallelic cells: the cells are CELL00 + a + (b << 16), so the shifted sum is
// linear in the allele bits: a_k lands on bit 2k and b_k on bit 16 + 2k of (sum - constant).
__device__ __forceinline__ uint32_t biallelic_spread(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3)
{
    const uint32_t K = CELL00 + (CELL00 << 2) + (CELL00 << 4) + (CELL00 << 6);
    return d0 + (d1 << 2) + (d2 << 4) + (d3 << 6) - K;
}

// Autosome row, groups that are complete and do not hold the row's last sample.
template <bool TEXT, bool PACKED, bool MULTI>
__device__ __forceinline__ void autosome_fast(const SegCtx& c, uint32_t g)
{
    uint32_t d0, d1, d2, d3, pk;
    if (!MULTI) {
        biallelic_cells(c, g, d0, d1, d2, d3);
        const uint32_t D = biallelic_spread(d0, d1, d2, d3);
        pk = D | (D >> 15);
    } else {
        uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
        group_alleles<true>(c, g, a0, a1, a2, a3, b0, b1, b2, b3);
        d0 = DIPLOID_CELL(a0, b0, 9u); d1 = DIPLOID_CELL(a1, b1, 9u);
        d2 = DIPLOID_CELL(a2, b2, 9u); d3 = DIPLOID_CELL(a3, b3, 9u);
        pk = nz(a0) | nz(b0) << 1 | nz(a1) << 2 | nz(b1) << 3 | nz(a2) << 4 | nz(b2) << 5 | nz(a3) << 6 | nz(b3) << 7;
    }
    if (TEXT) {
        const uint4 w = make_uint4(d0, d1, d2, d3);
        store_unaligned<16>(c.cells + 16ull * g, &w);
    }
    if (PACKED) c.prow[g] = (uint8_t)pk;
}

// X row (biallelic), complete groups without the row's last sample: males are haploid "a\t",
// females diploid "a/b\t" (common/snp.py:102-109), so a lane's 4 cells take 8..16 bytes starting at
// xoff[g].  The cells are shifted together in registers; the lane then stores the first 8 and the
// last 8 bytes of its run (the two stores overlap with identical bytes when the run is shorter than 16).
template <bool TEXT, bool PACKED>
__device__ __forceinline__ void x_fast(const SegCtx& c, uint32_t g)
{
    uint32_t d0, d1, d2, d3;
    biallelic_cells(c, g, d0, d1, d2, d3);
    const uint32_t m = c.male_nib[g];
    if (TEXT) {
        const bool m0 = m & 1u, m1 = m & 2u, m2 = m & 4u, m3 = m & 8u;
        // haploid cell = first character of the diploid cell + '\t'
        const uint32_t c0 = m0 ? (d0 & 0xFFu) | 0x0900u : d0, c1 = m1 ? (d1 & 0xFFu) | 0x0900u : d1;
        const uint32_t c2 = m2 ? (d2 & 0xFFu) | 0x0900u : d2, c3 = m3 ? (d3 & 0xFFu) | 0x0900u : d3;
        // R = cells back to back, built from the last cell: R = (R << width_k) | cell_k
        uint32_t r0 = c3, r1, r2, r3;
        r1 = m2 ? r0 >> 16 : r0;
        r0 = m2 ? (r0 << 16) | c2 : c2;
        r2 = m1 ? r1 >> 16 : r1;
        r1 = m1 ? __builtin_amdgcn_alignbit(r1, r0, 16) : r0;
        r0 = m1 ? (r0 << 16) | c1 : c1;
        r3 = m0 ? r2 >> 16 : r2;
        r2 = m0 ? __builtin_amdgcn_alignbit(r2, r1, 16) : r1;
        r1 = m0 ? __builtin_amdgcn_alignbit(r1, r0, 16) : r0;
        r0 = m0 ? (r0 << 16) | c0 : c0;
        const uint32_t s = 8u - 2u * (uint32_t)__builtin_popcount(m); // run length - 8, in bytes
        const uint32_t q = s >> 2, sh = (s & 2u) << 3;
        const uint32_t t0 = q == 0u ? r0 : (q == 1u ? r1 : r2);
        const uint32_t t1 = q == 0u ? r1 : (q == 1u ? r2 : r3);
        const uint32_t t2 = q == 0u ? r2 : (q == 1u ? r3 : 0u);
        uint8_t* dst = c.cells + c.xoff[g];
        const uint2 head = make_uint2(r0, r1);
        const uint2 tail = make_uint2(__builtin_amdgcn_alignbit(t1, t0, sh), __builtin_amdgcn_alignbit(t2, t1, sh));
        store_unaligned<8>(dst, &head);
        store_unaligned<8>(dst + s, &tail);
    }
    if (PACKED) {
        const uint32_t D = biallelic_spread(d0, d1, d2, d3);
        uint32_t sp = (m | (m << 2)) & 0x33u; // male bit k -> bit 2k
        sp = (sp | (sp << 1)) & 0x55u;
        c.prow[g] = (uint8_t)((D | (D >> 15)) & ~(sp << 1)); // haploid: no second allele
    }
}

// Y / MT row, complete groups without the row's last sample: 4 haploid cells = 8 bytes per lane.
template <bool TEXT, bool PACKED, bool MULTI, bool IS_Y>
__device__ __forceinline__ void haploid_fast(const SegCtx& c, uint32_t g)
{
    uint32_t a0, a1, a2, a3, b0, b1, b2, b3;
    group_alleles<MULTI>(c, g, a0, a1, a2, a3, b0, b1, b2, b3);
    uint32_t m = 0xFu;
    if (IS_Y) m = c.male_nib[g]; // '.' for females comes first (pop_factory.py:481-484)
    const uint32_t c0 = (m & 1u) ? 0x30u + a0 : 0x2Eu, c1 = (m & 2u) ? 0x30u + a1 : 0x2Eu;
    const uint32_t c2 = (m & 4u) ? 0x30u + a2 : 0x2Eu, c3 = (m & 8u) ? 0x30u + a3 : 0x2Eu;
    if (TEXT) {
        const uint2 w = make_uint2(HAPLOID_CELL(c0, 9u) | HAPLOID_CELL(c1, 9u) << 16,
                                   HAPLOID_CELL(c2, 9u) | HAPLOID_CELL(c3, 9u) << 16);
        store_unaligned<8>(c.cells + 8ull * g, &w);
    }
    if (PACKED)
        c.prow[g] = (uint8_t)((m & 1u ? nz(a0) : 0u) | (m & 2u ? nz(a1) : 0u) << 2 | (m & 4u ? nz(a2) : 0u) << 4 |
                              (m & 8u ? nz(a3) : 0u) << 6);
}

// The low N bytes of val at c.cells + off, restricted to the cell-stream offsets [lo, hi) (the bytes this wave owns)
template <int N>
__device__ __forceinline__ void store_clipped(uint8_t* cells, uint32_t off, uint32_t val, uint32_t lo, uint32_t hi)
{
    if (off >= lo && off + N <= hi) {
        if (N == 4) store_unaligned<4>(cells + off, &val);
        else { const uint16_t h = (uint16_t)val; store_unaligned<2>(cells + off, &h); }
        return;
    }
#pragma unroll
    for (int b = 0; b < N; ++b)
        if (off + b >= lo && off + b < hi) cells[off + b] = (uint8_t)(val >> (8 * b));
}

// Any group of any row class, cell by cell: used for the row's last group (newline, fewer than 4
// samples), the pad bytes of the packed row, multi-allelic X rows, and -- with CLIP -- for the few groups at
// the ends of a wave's line range, where only the text bytes at cell-stream offsets [lo, hi) are stored.
// STAGE: the packed byte goes to the wave's LDS ring (c.stage, index g mod PK_RING) instead of c.prow.
template <bool TEXT, bool PACKED, bool MULTI, bool CLIP = false, bool STAGE = false>
__device__ __forceinline__ void general_group(const SegCtx& c, uint32_t cls, uint32_t g, uint32_t lo = 0u, uint32_t hi = 0xFFFFFFFFu)
{
    if (g >= c.G) { // lanes past the last group only zero the pad bytes of the packed row
        if (PACKED) { if (STAGE) c.stage[g & (PK_STAGE - 1u)] = 0; else c.prow[g] = 0; }
        return;
    }
    uint32_t a[4], b[4];
    group_alleles<MULTI>(c, g, a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
    const uint32_t male = c.male_nib[g];
    uint32_t off = cls == 0u ? 16u * g : (cls == 1u ? c.xoff[g] : 8u * g);
    uint32_t pk = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = 4u * g + (uint32_t)k;
        if (i < c.S) {
            const uint32_t sep = (i == c.S - 1u) ? 0x0Au : 0x09u; // "\t".join(...) + "\n", pop_factory.py:508
            const bool is_male = (male >> k) & 1u;
            const bool missing = cls == 2u && !is_male;                           // pop_factory.py:481-484
            const bool haploid = cls >= 2u || (cls == 1u && is_male);             // common/snp.py:102-109
            if (missing) {
                if (TEXT) {
                    const uint16_t h = (uint16_t)HAPLOID_CELL(0x2Eu, sep);
                    if (CLIP) store_clipped<2>(c.cells, off, h, lo, hi); else store_unaligned<2>(c.cells + off, &h);
                    off += 2;
                }
            } else if (haploid) {
                pk |= nz(a[k]) << (2 * k);
                if (TEXT) {
                    const uint16_t h = (uint16_t)HAPLOID_CELL(0x30u + a[k], sep);
                    if (CLIP) store_clipped<2>(c.cells, off, h, lo, hi); else store_unaligned<2>(c.cells + off, &h);
                    off += 2;
                }
            } else {
                pk |= (nz(a[k]) | nz(b[k]) << 1) << (2 * k);
                if (TEXT) {
                    const uint32_t w = DIPLOID_CELL(a[k], b[k], sep);
                    if (CLIP) store_clipped<4>(c.cells, off, w, lo, hi); else store_unaligned<4>(c.cells + off, &w);
                    off += 4;
                }
            }
        }
    }
    if (PACKED) { if (STAGE) c.stage[g & (PK_STAGE - 1u)] = (uint8_t)pk; else c.prow[g] = (uint8_t)pk; }
}

template <bool TEXT, bool PACKED, bool MULTI>
__device__ __forceinline__ void row_segment(const SegCtx& c, uint32_t cls, uint32_t g_begin, uint32_t g_end,
                                            uint32_t g_stop, uint32_t lane)
{
    // groups [g_begin, g_fast) are complete and do not hold the row's last sample: class-specialised loops
    const uint32_t g_last = c.S ? (c.S - 1u) / 4u : 0u;
    const uint32_t g_fast = min(g_end, g_last);
    uint32_t gb = g_begin;
    if (cls == 0u && !MULTI) {
        // biallelic autosome (almost every row): the row's last group rides in the same pass -- its lane draws
        // the same two Philox blocks and only stores differently -- so short rows need one trip, not two
        // The packed bytes of the segment are staged in LDS and leave as whole 16-byte stores after the loop instead
        // of one byte store per lane and trip (64 B per wave, half a cache line): 3 % faster.
        const bool staged = PACKED && g_end - g_begin <= PK_STAGE;
        // whole trips of complete groups first: no lane is out of range and none holds the row's last sample
        // (counter, text pointer and staging pointer advance by constants: no per-trip address arithmetic from g)
        {
            uint32_t c0 = 2u * (gb + lane);
            uint8_t* tp = c.cells + 16ull * (gb + lane);
            LDS_PTR(uint8_t) sp = c.stage + (gb + lane - g_begin);
            // Two trips per pass, software-pipelined: the cells looked up in one pass are stored in the next, so the
            // LDS read flies under the following Philox rounds (the stores address the previous pass through tp).
            if (gb + 128u <= g_fast) {
                u32x4_t w0 = {0, 0, 0, 0}, w1 = {0, 0, 0, 0};
                auto pass = [&](bool first) {
                    const uint32_t pk0 = biallelic_code<0>(c, c0), pk1 = biallelic_code<64>(c, c0);
                    if (TEXT) {
                        if (!first) { store_unaligned<16>(tp - 2048, &w0); store_unaligned<16>(tp - 1024, &w1); }
                        w0 = c.lut[pk0];
                        w1 = c.lut[pk1];
                    }
                    if (PACKED) {
                        if (staged) { sp[0] = (uint8_t)pk0; sp[64] = (uint8_t)pk1; }
                        else { c.prow[c0 >> 1] = (uint8_t)pk0; c.prow[(c0 >> 1) + 64u] = (uint8_t)pk1; }
                    }
                    gb += 128u; c0 += 256u; tp += 2048; sp += 128;
                };
                pass(true);
                while (gb + 128u <= g_fast) pass(false);
                if (TEXT) { store_unaligned<16>(tp - 2048, &w0); store_unaligned<16>(tp - 1024, &w1); }
            }
            for (; gb + 64u <= g_fast; gb += 64u, c0 += 128u, tp += 1024, sp += 64) {
                const uint32_t pk = biallelic_code(c, c0);
                if (TEXT) {
                    const u32x4_t w = c.lut[pk];
                    store_unaligned<16>(tp, &w);
                }
                if (PACKED) {
                    if (staged) *sp = (uint8_t)pk;
                    else c.prow[c0 >> 1] = (uint8_t)pk;
                }
            }
        }
        for (; gb < g_end; gb += 64u) {
            const uint32_t g = gb + lane;
            if (g < g_end) {
                uint32_t pk = biallelic_code(c, 2u * g);
                u32x4_t w = {0, 0, 0, 0};
                if (TEXT) w = c.lut[pk];
                const uint32_t d0 = w.x, d1 = w.y, d2 = w.z, d3 = w.w;
                if (g < g_fast) {
                    if (TEXT) store_unaligned<16>(c.cells + 16ull * g, &w);
                } else { // the group with the row's last sample: newline instead of the last tab, maybe < 4 samples
                    const uint32_t nvalid = c.S - 4u * g;
                    const uint32_t fix = 0x0A000000u - 0x09000000u;
                    if (TEXT) {
                        uint8_t* dst = c.cells + 16ull * g;
                        const uint32_t w0 = d0 + (nvalid == 1u ? fix : 0u), w1 = d1 + (nvalid == 2u ? fix : 0u);
                        const uint32_t w2 = d2 + (nvalid == 3u ? fix : 0u), w3 = d3 + (nvalid == 4u ? fix : 0u);
                        store_unaligned<4>(dst, &w0);
                        if (nvalid > 1u) store_unaligned<4>(dst + 4, &w1);
                        if (nvalid > 2u) store_unaligned<4>(dst + 8, &w2);
                        if (nvalid > 3u) store_unaligned<4>(dst + 12, &w3);
                    }
                    pk &= (1u << (2u * nvalid)) - 1u; // codes of samples past the end stay 0
                }
                if (PACKED) {
                    if (staged) c.stage[g - g_begin] = (uint8_t)pk;
                    else c.prow[g] = (uint8_t)pk;
                }
            }
        }
        if (staged) {
            const uint32_t n = g_end - g_begin;
            uint8_t* dst = c.prow + g_begin;
            for (uint32_t o = 16u * lane; o + 16u <= n; o += 1024u) {
                const u32x4_t v = *reinterpret_cast<const LDS_PTR(u32x4_t)>(c.stage + o);
                store_unaligned<16>(dst + o, &v);
            }
            if (lane < (n & 15u)) dst[(n & ~15u) + lane] = c.stage[(n & ~15u) + lane];
        }
        if (PACKED && g_stop > g_end && g_end + lane < g_stop) c.prow[g_end + lane] = 0; // pad bytes (at most 3)
        return;
    }
    if (cls == 0u) {
        for (; gb < g_fast; gb += 64u)
            if (gb + lane < g_fast) autosome_fast<TEXT, PACKED, MULTI>(c, gb + lane);
    } else if (cls == 1u && !MULTI) {
        for (; gb < g_fast; gb += 64u)
            if (gb + lane < g_fast) x_fast<TEXT, PACKED>(c, gb + lane);
    } else if (cls == 2u) {
        for (; gb < g_fast; gb += 64u)
            if (gb + lane < g_fast) haploid_fast<TEXT, PACKED, MULTI, true>(c, gb + lane);
    } else if (cls == 3u) {
        for (; gb < g_fast; gb += 64u)
            if (gb + lane < g_fast) haploid_fast<TEXT, PACKED, MULTI, false>(c, gb + lane);
    }
    // what is left: the row's last group and the packed pad bytes (or a whole multi-allelic X segment)
    for (gb = (cls == 1u && MULTI) ? g_begin : max(g_begin, g_fast); gb < g_stop; gb += 64u)
        if (gb + lane < g_stop) general_group<TEXT, PACKED, MULTI>(c, cls, gb + lane);
}

template <bool TEXT, bool PACKED>
__global__ __launch_bounds__(256) void rowgen_kernel(const GenParams p)
{
    // text of 4 biallelic diploid cells by packed byte: entry pk = "a0/b0\ta1/b1\ta2/b2\ta3/b3\t"
    __shared__ u32x4_t cell_lut[256];
    __shared__ __attribute__((aligned(16))) uint8_t pk_stage[4][PK_STAGE];
    if (TEXT) {
        const uint32_t t = threadIdx.x;
        const u32x4_t e = {CELL00 + (t & 1u) + ((t & 2u) << 15), CELL00 + ((t >> 2) & 1u) + ((t & 8u) << 13),
                           CELL00 + ((t >> 4) & 1u) + ((t & 32u) << 11), CELL00 + ((t >> 6) & 1u) + ((t & 128u) << 9)};
        cell_lut[t] = e;
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    // wave-uniform work item -> scalar registers
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    const uint64_t row_local = item / p.nseg;
    const uint32_t seg = (uint32_t)(item % p.nseg);
    if (row_local >= p.n_rows) return;
    const RowRec rec = p.recs[p.row_begin + row_local];
    uint8_t* row_text = p.text + (rec.text_off - p.text_base);

    if (TEXT && seg == 0) {
        // the row's fixed columns (pop_factory.py:503-507), formatted on the host; S == 0: line = prefix + "\n"
        const uint8_t* src = p.prefix + rec.prefix_off;
        for (uint32_t i = lane; i < rec.prefix_len; i += 64) row_text[i] = src[i];
        if (p.S == 0 && lane == 0) row_text[rec.prefix_len] = 0x0A;
    }
    const uint32_t g_begin = seg * p.seg_groups;
    if (g_begin >= p.G && !(PACKED && seg == 0)) return;
    const uint32_t g_end = min(p.G, g_begin + p.seg_groups);
    const uint32_t g_stop = (PACKED && g_end == p.G) ? p.W * 4u : g_end; // the last segment also zeroes the pad bytes

    const uint64_t rng_row = p.rng_row0 + rng_row_of(p, p.row_begin + row_local);
    SegCtx c;
    c.t0 = rec.thr[0]; c.t1 = rec.thr[1]; c.t2 = rec.thr[2];
    c.r_lo = (uint32_t)rng_row; c.r_hi = (uint32_t)(rng_row >> 32);
    c.k0 = p.key0; c.k1 = p.key1;
    c.S = p.S; c.G = p.G;
    c.cells = row_text + rec.prefix_len;
    c.prow = p.packed + row_local * p.packed_pitch;
    c.frow = rec.slot >= 0 ? p.forced + (uint64_t)rec.slot * p.G : nullptr;
    c.male_nib = p.male_nib;
    c.lut = (const LDS_PTR(u32x4_t))cell_lut;
    c.stage = (LDS_PTR(uint8_t))pk_stage[threadIdx.x >> 6];
    c.xoff = p.xoff;
    if (rec.n_alleles > 2) row_segment<TEXT, PACKED, true>(c, rec.cls, g_begin, g_end, g_stop, lane);
    else row_segment<TEXT, PACKED, false>(c, rec.cls, g_begin, g_end, g_stop, lane);
}

// Case / control allele counts from the packed matrix (the check behind BASELINE config 5: per-SNP
// alternative-allele frequency of cases vs controls).  One wave per row: popcounts of the row's words
// split by the control mask (both bits of a control sample's code set in ctrl2), wave-reduced.
__global__ __launch_bounds__(256) void allele_count_kernel(const uint32_t* packed, const uint32_t* ctrl2, uint32_t W, uint32_t Wp,
                                                           uint64_t n_rows, uint32_t* counts)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t row = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint32_t* p = packed + row * Wp; // rows are Wp words apart
    uint32_t nc = 0, nk = 0;
    for (uint32_t w = lane; w < W; w += 64u) {
        const uint32_t v = p[w], m = ctrl2[w];
        nc += (uint32_t)__builtin_popcount(v & m);
        nk += (uint32_t)__builtin_popcount(v & ~m);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        nc += __shfl_xor(nc, d, 64);
        nk += __shfl_xor(nk, d, 64);
    }
    if (lane == 0) {
        counts[2 * row] = nc;
        counts[2 * row + 1] = nk;
    }
}

} // namespace dnaf
