// rowrun_kernels.hip.h -- the row generator in its line-owning form (gfx950 / CDNA4).
//
// Same rows, same bytes as rowgen_kernels.hip.h (whose per-group code it reuses); what changes is WHO stores
// which bytes.  Measured on MI355X (profiles/r02_placement/): a 128-byte line of the output that two waves each
// write a part of reaches HBM as two partial writes (the halves are written tens of microseconds apart and the
// first has left the L2 by then), and such lines -- not the byte misalignment of the 16-byte lane stores -- are
// what held the text stream at 5.3 TB/s instead of 5.8 and made the kernel's speed depend on where the packed
// matrix happened to be allocated.  So here every line of the text stream has exactly one writer:
//
//   * a wave owns a run of `rows_per_wave` whole rows j0 .. j1-1 and stores the LINES
//     [ceil128(T(j0)), ceil128(T(j1)))  of the text stream, T(j) = address of row j's first byte:
//     it leaves the first bytes of row j0 (up to the next line boundary) to the wave before it, and after its
//     last row it also produces the first bytes of row j1 -- that row's fixed columns and, if the line boundary
//     lies beyond them, its first few cells (one clipped generic pass over <= 16 groups, < 2 % extra work at
//     20k samples);
//   * the packed matrix is written with a row pitch that is a multiple of 128 bytes (GenParams::packed_pitch;
//     dense when rows are short), a wave stages its row's bytes in a 2 KiB LDS ring and stores them 1 KiB
//     (8 whole lines) at a time.
//
// The cell rules, the RNG lane mapping and the text format are those of pop_factory.py:471-513, see
// rowgen_kernels.hip.h.  Launch bounds: every row must be at least 256 bytes long and have >= 256 groups
// (the host falls back to rowgen_kernel otherwise).
#pragma once
#include "rowgen_kernels.hip.h"

#ifndef DNAF_RUN_TRIPS
#define DNAF_RUN_TRIPS 2
#endif
#ifndef DNAF_RUN_BOUNDS
// No register limits: the kernel is bound by its vector-instruction count, not by latency, and capping it at 64 VGPRs /
// 80 SGPRs (8 waves per SIMD) makes hipcc re-materialise scalars in the hot loop (measured: 1 % slower than 6 workgroups per CU)
#define DNAF_RUN_BOUNDS __launch_bounds__(256)
#endif

namespace dnaf {

// 16 bytes at cell-stream offset off, of which only the bytes at offsets >= lo are this wave's
__device__ __forceinline__ void store16_from(uint8_t* cells, uint32_t off, const u32x4_t& w, uint32_t lo)
{
    if (off >= lo) { store_unaligned<16>(cells + off, &w); return; }
    if (off + 16u <= lo) return;
    store_clipped<4>(cells, off, w.x, lo, 0xFFFFFFFFu);
    store_clipped<4>(cells, off + 4u, w.y, lo, 0xFFFFFFFFu);
    store_clipped<4>(cells, off + 8u, w.z, lo, 0xFFFFFFFFu);
    store_clipped<4>(cells, off + 12u, w.w, lo, 0xFFFFFFFFu);
}

// 16 bytes at cell-stream offset off, of which only the bytes at offsets < hi are this wave's
__device__ __forceinline__ void store16_upto(uint8_t* cells, uint32_t off, const u32x4_t& w, uint32_t hi)
{
    if (off + 16u <= hi) { store_unaligned<16>(cells + off, &w); return; }
    if (off >= hi) return;
    store_clipped<4>(cells, off, w.x, 0u, hi);
    store_clipped<4>(cells, off + 4u, w.y, 0u, hi);
    store_clipped<4>(cells, off + 8u, w.z, 0u, hi);
    store_clipped<4>(cells, off + 12u, w.w, 0u, hi);
}

// 1 KiB of the wave's packed ring -> the row's chunk m (8 whole lines; prow is 128-byte aligned)
__device__ __forceinline__ void flush_chunk(const SegCtx& c, uint32_t m, uint32_t lane)
{
    const u32x4_t v = *reinterpret_cast<const LDS_PTR(u32x4_t)>(c.stage + ((m & 1u) * 1024u + 16u * lane));
    *reinterpret_cast<u32x4_t*>(c.prow + 1024ull * m + 16u * lane) = v;
}

// What the Philox blocks of one row have in common (wave-uniform, scalar registers).  With the counter
// (c0, 0, r_lo, r_hi) only c0 differs between lanes, so in rounds 0-2 several inputs of the round function are the
// same for the whole wave; they are folded here once per row, and every vector instruction of those rounds then has
// a single scalar operand (the compiler, left alone, keeps copies of them in vector registers).
struct RowConst {
    uint32_t x0;     // round 0: c2 <- hi(M0 c0) ^ x0,           x0 = r_hi ^ k1
    uint32_t x1;     // round 1: c0 <- hi(M1 c2) ^ x1,           x1 = lo(M1 r_lo) ^ (k0 + W0)
    uint32_t x2;     // round 1: c2 <- lo(M0 c0 of round 0) ^ x2, x2 = hi(M0 u) ^ (k1 + W1),  u = hi(M1 r_lo) ^ k0
    uint32_t x3;     // round 2: c2 <- hi(M0 c0) ^ x3,           x3 = lo(M0 u) ^ (k1 + 2 W1)
    uint32_t k0, k1; // the key
};

__device__ __forceinline__ RowConst make_row_const(const SegCtx& c)
{
    RowConst rc;
    const uint64_t p1 = (uint64_t)PHILOX_M1 * c.r_lo;
    const uint32_t u = (uint32_t)(p1 >> 32) ^ c.k0;
    const uint64_t pu = (uint64_t)PHILOX_M0 * u;
    rc.x0 = c.r_hi ^ c.k1;
    rc.x1 = (uint32_t)p1 ^ (c.k0 + PHILOX_W0);
    rc.x2 = (uint32_t)(pu >> 32) ^ (c.k1 + PHILOX_W1);
    rc.x3 = (uint32_t)pu ^ (c.k1 + 2u * PHILOX_W1);
    rc.k0 = c.k0; rc.k1 = c.k1;
    return rc;
}

// The packed byte of one group of 4 samples -- two Philox4x32-10 blocks, counters (2g, 0, r_lo, r_hi) and (2g+1, ...), and
// the 8 compares against the row's threshold -- as one hand-written instruction sequence: 90 vector instructions,
//   1 + 36 per block: v_lshl_add_u64 (first-round product = lane part + wave part, see below), round 0: 1 x v_xor,
//   round 1: v_mad_u64_u32 + 2 x v_xor, round 2: 2 x v_mad_u64_u32 + v_bitop3 + v_xor, rounds 3-9: 2 + 2 each;
//   16 for the compares: v_cmp_lt_u32 + v_addc_co_u32 (pk = 2 pk + carry) per draw.
// hipcc's own code for the same arithmetic needs 10 % more (copies of wave-uniform values in vector registers because
// an instruction may read only one scalar register, 64-bit adds done twice, a counter and a pointer advanced per lane),
// and the kernel is bound by exactly this count (SQ_ACTIVE_INST_VALU x 4 cycles = 100 % of the SIMD cycles).
//   P    = M0 * 2 * lane, the lane's share of the first-round product (constant for the whole kernel)
//   offa = M0 * 2 * gb, offb = offa + M0: the wave's share for the group's two blocks (scalar registers)
// Temporaries are fixed registers (v40 .. v67): the products need aligned pairs whose halves are named one by one.  The two
// blocks are interleaved instruction by instruction: two independent dependency chains per wave (the kernel runs at three
// waves per SIMD, see DNAF_RUN_LDS_PAD_DEFAULT).
struct RowKeys { uint32_t k02, k0[7], k1[7]; }; // k0 + r W0 for rounds 2 .. 9, k1 + r W1 for rounds 3 .. 9

__device__ __forceinline__ RowKeys make_row_keys(uint32_t k0, uint32_t k1)
{
    RowKeys k;
    k.k02 = k0 + 2u * PHILOX_W0;
#pragma unroll
    for (int r = 3; r < 10; ++r) { k.k0[r - 3] = k0 + (uint32_t)r * PHILOX_W0; k.k1[r - 3] = k1 + (uint32_t)r * PHILOX_W1; }
    return k;
}

#define DNAF_GROUP_ASM DNAF_GROUP_HEAD("0") DNAF_GROUP_BODY
// first-round products of the two blocks: (P << shift) + the wave's share
#define DNAF_GROUP_HEAD(shift) \
        "v_lshl_add_u64 v[40:41], %[P], " shift ", %[offb]\n\t" \
        "v_lshl_add_u64 v[54:55], %[P], " shift ", %[offa]\n\t"
#define DNAF_GROUP_BODY \
        "v_xor_b32 v46, %[x0], v41\n\t" \
        "v_xor_b32 v60, %[x0], v55\n\t" \
        "v_mad_u64_u32 v[52:53], vcc, v46, %[m1], 0\n\t" \
        "v_mad_u64_u32 v[66:67], vcc, v60, %[m1], 0\n\t" \
        "v_xor_b32 v44, %[x1], v53\n\t" \
        "v_xor_b32 v58, %[x1], v67\n\t" \
        "v_xor_b32 v46, %[x2], v40\n\t" \
        "v_xor_b32 v60, %[x2], v54\n\t" \
        "v_mad_u64_u32 v[40:41], vcc, v44, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[54:55], vcc, v58, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[48:49], vcc, v46, %[m1], 0\n\t" \
        "v_mad_u64_u32 v[62:63], vcc, v60, %[m1], 0\n\t" \
        "v_bitop3_b32 v44, v49, v52, %[k02] bitop3:0x96\n\t" \
        "v_bitop3_b32 v58, v63, v66, %[k02] bitop3:0x96\n\t" \
        "v_xor_b32 v46, %[x3], v41\n\t" \
        "v_xor_b32 v60, %[x3], v55\n\t" \
        "v_mad_u64_u32 v[50:51], vcc, v44, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[64:65], vcc, v58, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[52:53], vcc, v46, %[m1], 0\n\t" \
        "v_mad_u64_u32 v[66:67], vcc, v60, %[m1], 0\n\t" \
        "v_bitop3_b32 v44, v53, v48, %[k03] bitop3:0x96\n\t" \
        "v_bitop3_b32 v58, v67, v62, %[k03] bitop3:0x96\n\t" \
        "v_bitop3_b32 v46, v51, v40, %[k13] bitop3:0x96\n\t" \
        "v_bitop3_b32 v60, v65, v54, %[k13] bitop3:0x96\n\t" \
        "v_mad_u64_u32 v[40:41], vcc, v44, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[54:55], vcc, v58, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[48:49], vcc, v46, %[m1], 0\n\t" \
        "v_mad_u64_u32 v[62:63], vcc, v60, %[m1], 0\n\t" \
        "v_bitop3_b32 v44, v49, v52, %[k04] bitop3:0x96\n\t" \
        "v_bitop3_b32 v58, v63, v66, %[k04] bitop3:0x96\n\t" \
        "v_bitop3_b32 v46, v41, v50, %[k14] bitop3:0x96\n\t" \
        "v_bitop3_b32 v60, v55, v64, %[k14] bitop3:0x96\n\t" \
        "v_mad_u64_u32 v[50:51], vcc, v44, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[64:65], vcc, v58, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[52:53], vcc, v46, %[m1], 0\n\t" \
        "v_mad_u64_u32 v[66:67], vcc, v60, %[m1], 0\n\t" \
        "v_bitop3_b32 v44, v53, v48, %[k05] bitop3:0x96\n\t" \
        "v_bitop3_b32 v58, v67, v62, %[k05] bitop3:0x96\n\t" \
        "v_bitop3_b32 v46, v51, v40, %[k15] bitop3:0x96\n\t" \
        "v_bitop3_b32 v60, v65, v54, %[k15] bitop3:0x96\n\t" \
        "v_mad_u64_u32 v[40:41], vcc, v44, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[54:55], vcc, v58, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[48:49], vcc, v46, %[m1], 0\n\t" \
        "v_mad_u64_u32 v[62:63], vcc, v60, %[m1], 0\n\t" \
        "v_bitop3_b32 v44, v49, v52, %[k06] bitop3:0x96\n\t" \
        "v_bitop3_b32 v58, v63, v66, %[k06] bitop3:0x96\n\t" \
        "v_bitop3_b32 v46, v41, v50, %[k16] bitop3:0x96\n\t" \
        "v_bitop3_b32 v60, v55, v64, %[k16] bitop3:0x96\n\t" \
        "v_mad_u64_u32 v[50:51], vcc, v44, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[64:65], vcc, v58, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[52:53], vcc, v46, %[m1], 0\n\t" \
        "v_mad_u64_u32 v[66:67], vcc, v60, %[m1], 0\n\t" \
        "v_bitop3_b32 v44, v53, v48, %[k07] bitop3:0x96\n\t" \
        "v_bitop3_b32 v58, v67, v62, %[k07] bitop3:0x96\n\t" \
        "v_bitop3_b32 v46, v51, v40, %[k17] bitop3:0x96\n\t" \
        "v_bitop3_b32 v60, v65, v54, %[k17] bitop3:0x96\n\t" \
        "v_mad_u64_u32 v[40:41], vcc, v44, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[54:55], vcc, v58, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[48:49], vcc, v46, %[m1], 0\n\t" \
        "v_mad_u64_u32 v[62:63], vcc, v60, %[m1], 0\n\t" \
        "v_bitop3_b32 v44, v49, v52, %[k08] bitop3:0x96\n\t" \
        "v_bitop3_b32 v58, v63, v66, %[k08] bitop3:0x96\n\t" \
        "v_bitop3_b32 v46, v41, v50, %[k18] bitop3:0x96\n\t" \
        "v_bitop3_b32 v60, v55, v64, %[k18] bitop3:0x96\n\t" \
        "v_mad_u64_u32 v[50:51], vcc, v44, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[64:65], vcc, v58, %[m0], 0\n\t" \
        "v_mad_u64_u32 v[52:53], vcc, v46, %[m1], 0\n\t" \
        "v_mad_u64_u32 v[66:67], vcc, v60, %[m1], 0\n\t" \
        "v_bitop3_b32 v44, v53, v48, %[k09] bitop3:0x96\n\t" \
        "v_bitop3_b32 v58, v67, v62, %[k09] bitop3:0x96\n\t" \
        "v_bitop3_b32 v46, v51, v40, %[k19] bitop3:0x96\n\t" \
        "v_bitop3_b32 v60, v65, v54, %[k19] bitop3:0x96\n\t" \
        "v_cmp_lt_u32 vcc, %[t0], v50\n\t" \
        "v_addc_co_u32 %[pk], vcc, 0, 0, vcc\n\t" \
        "v_cmp_lt_u32 vcc, %[t0], v46\n\t" \
        "v_addc_co_u32 %[pk], vcc, %[pk], %[pk], vcc\n\t" \
        "v_cmp_lt_u32 vcc, %[t0], v52\n\t" \
        "v_addc_co_u32 %[pk], vcc, %[pk], %[pk], vcc\n\t" \
        "v_cmp_lt_u32 vcc, %[t0], v44\n\t" \
        "v_addc_co_u32 %[pk], vcc, %[pk], %[pk], vcc\n\t" \
        "v_cmp_lt_u32 vcc, %[t0], v64\n\t" \
        "v_addc_co_u32 %[pk], vcc, %[pk], %[pk], vcc\n\t" \
        "v_cmp_lt_u32 vcc, %[t0], v60\n\t" \
        "v_addc_co_u32 %[pk], vcc, %[pk], %[pk], vcc\n\t" \
        "v_cmp_lt_u32 vcc, %[t0], v66\n\t" \
        "v_addc_co_u32 %[pk], vcc, %[pk], %[pk], vcc\n\t" \
        "v_cmp_lt_u32 vcc, %[t0], v58\n\t" \
        "v_addc_co_u32 %[pk], vcc, %[pk], %[pk], vcc\n\t"

#define DNAF_GROUP_KEYS(k) \
    [k02] "s"((k).k02), [k03] "s"((k).k0[0]), [k04] "s"((k).k0[1]), [k05] "s"((k).k0[2]), [k06] "s"((k).k0[3]), [k07] "s"((k).k0[4]), \
    [k08] "s"((k).k0[5]), [k09] "s"((k).k0[6]), [k13] "s"((k).k1[0]), [k14] "s"((k).k1[1]), [k15] "s"((k).k1[2]), [k16] "s"((k).k1[3]), \
    [k17] "s"((k).k1[4]), [k18] "s"((k).k1[5]), [k19] "s"((k).k1[6])
#define DNAF_GROUP_CLOBBERS "v40", "v41", "v44", "v46", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v58", "v60", "v62", "v63", "v64", "v65", "v66", "v67", "vcc"

__device__ __forceinline__ uint32_t group_code(uint64_t P, uint64_t offa, uint64_t offb, const RowConst& rc, const RowKeys& k, uint32_t t0)
{
    uint32_t pk;
    asm(DNAF_GROUP_ASM
        : [pk] "=&v"(pk)
        : [P] "v"(P), [offa] "s"(offa), [offb] "s"(offb), [m0] "s"(PHILOX_M0), [m1] "s"(PHILOX_M1), [x0] "s"(rc.x0), [x1] "s"(rc.x1),
          [x2] "s"(rc.x2), [x3] "s"(rc.x3), [t0] "s"(t0), DNAF_GROUP_KEYS(k)
        : DNAF_GROUP_CLOBBERS);
    return pk;
}

// The same sequence with the row's folded constants and threshold in vector registers: lanes of one trip may then belong
// to two different rows (the last groups of a wave's row and the first groups of the next wave's row, run_row_biallelic).
// P = M0 * 2 * g of the lane's own group.
__device__ __forceinline__ uint32_t group_code_lanes(uint64_t P, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t t0, const RowKeys& k)
{
    uint32_t pk;
    asm(DNAF_GROUP_ASM
        : [pk] "=&v"(pk)
        : [P] "v"(P), [offa] "s"((uint64_t)0), [offb] "s"((uint64_t)PHILOX_M0), [m0] "s"(PHILOX_M0), [m1] "s"(PHILOX_M1), [x0] "v"(x0), [x1] "v"(x1),
          [x2] "v"(x2), [x3] "v"(x3), [t0] "v"(t0), DNAF_GROUP_KEYS(k)
        : DNAF_GROUP_CLOBBERS);
    return pk;
}

// group_code_lanes with P = M0 * g (the doubling is the shift of the sequence's first instruction: one instruction less
// for a caller that has g, not 2 g, in a register)
__device__ __forceinline__ uint32_t group_code_lanes_g(uint64_t P, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t t0, const RowKeys& k)
{
    uint32_t pk;
    asm(DNAF_GROUP_HEAD("1") DNAF_GROUP_BODY
        : [pk] "=&v"(pk)
        : [P] "v"(P), [offa] "s"((uint64_t)0), [offb] "s"((uint64_t)PHILOX_M0), [m0] "s"(PHILOX_M0), [m1] "s"(PHILOX_M1), [x0] "v"(x0), [x1] "v"(x1),
          [x2] "v"(x2), [x3] "v"(x3), [t0] "v"(t0), DNAF_GROUP_KEYS(k)
        : DNAF_GROUP_CLOBBERS);
    return pk;
}

// forced cells of group g ("1/1", pop_factory.py:495-499) on a packed byte; c.frow != nullptr
__device__ __forceinline__ uint32_t force_code(const SegCtx& c, uint32_t pk, uint32_t g)
{
    uint32_t f = c.frow[g] & 0xFu;
    f = (f | (f << 2)) & 0x33u;
    f = (f | (f << 1)) & 0x55u;
    return pk | f * 3u;
}

// a wave-uniform global address, told to the compiler as such (scalar registers, global address space), so that the
// text stores take the form  global_store_dwordx4 v_lane_offset, data, s[base:base+1] offset:imm  and the base advances
// on the scalar unit
typedef uint8_t __attribute__((address_space(1))) * gptr_t;
__device__ __forceinline__ gptr_t uniform_ptr(uint8_t* p)
{
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (gptr_t)(((uint64_t)hi << 32) | lo);
}
typedef u32x4_t __attribute__((aligned(1))) u32x4_u_t;
__device__ __forceinline__ void store16_global(gptr_t dst, const u32x4_t& w)
{
    *reinterpret_cast<u32x4_u_t __attribute__((address_space(1)))*>(dst) = w;
}

// The first bytes of the NEXT wave's first row when that row is an unforced biallelic autosome row too: its first nh
// groups ride in the last, partly filled trip of this wave's row (run_row_biallelic), or in one short trip of their own.
struct HeadJob {
    uint32_t nh;      // groups of the next row that begin below hc (0: nothing to do here)
    uint32_t hc;      // bytes of the next row's cell stream that complete this wave's last line
    uint8_t* cells;   // the next row's first cell byte
    RowConst rc;      // its folded constants
    uint32_t t0;      // and threshold
};

// One trip whose lanes [0, n1) take groups g1 + lane of the wave's own row and whose lanes [n1, n1 + h.nh) take groups
// lane - n1 of the next row: one pass of the Philox sequence for both.  n1 may be 0 (head alone).
template <bool TEXT, bool PACKED>
__device__ __forceinline__ void mixed_trip(const SegCtx& c, const RowConst& rc, const RowKeys& keys, uint32_t g1, uint32_t n1, const HeadJob& h,
                                           uint32_t lane)
{
    const bool own = lane < n1, nxt = !own && lane < n1 + h.nh;
    const uint32_t g = own ? g1 + lane : lane - n1;
    const uint64_t P = (uint64_t)PHILOX_M0 * (2u * g);
    const uint32_t pk0 = group_code_lanes(P, own ? rc.x0 : h.rc.x0, own ? rc.x1 : h.rc.x1, own ? rc.x2 : h.rc.x2, own ? rc.x3 : h.rc.x3,
                                          own ? c.t0 : h.t0, keys);
    uint32_t pk = pk0;
    u32x4_t w = {0, 0, 0, 0};
    if (TEXT) w = c.lut[pk0 & 0xFFu];
    if (own) {
        const uint32_t g_fast = (c.S - 1u) / 4u;
        if (g < g_fast) {
            if (TEXT) store_unaligned<16>(c.cells + 16ull * g, &w);
        } else { // the group with the row's last sample: newline instead of the last tab, maybe < 4 samples
            const uint32_t nvalid = c.S - 4u * g;
            const uint32_t fix = 0x0A000000u - 0x09000000u;
            if (TEXT) {
                uint8_t* dst = c.cells + 16ull * g;
                const uint32_t w0 = w.x + (nvalid == 1u ? fix : 0u), w1 = w.y + (nvalid == 2u ? fix : 0u);
                const uint32_t w2 = w.z + (nvalid == 3u ? fix : 0u), w3 = w.w + (nvalid == 4u ? fix : 0u);
                store_unaligned<4>(dst, &w0);
                if (nvalid > 1u) store_unaligned<4>(dst + 4, &w1);
                if (nvalid > 2u) store_unaligned<4>(dst + 8, &w2);
                if (nvalid > 3u) store_unaligned<4>(dst + 12, &w3);
            }
            pk &= (1u << (2u * nvalid)) - 1u; // codes of samples past the end stay 0
        }
        if (PACKED) c.stage[g & (PK_STAGE - 1u)] = (uint8_t)pk;
    } else if (nxt) {
        if (TEXT) store16_upto(h.cells, 16u * g, w, h.hc);
    }
}

// One biallelic autosome row (almost every row).  skipc = leading bytes of the cell stream owned by the wave
// before this one; head = what this wave writes of the next wave's first row (nh = 0: nothing, or done elsewhere).
// CHUNKED: padded pitch, the packed bytes leave through the ring in 1 KiB chunks; otherwise the row
// has at most PK_STAGE groups and its packed bytes are staged whole (dense pitch, 16-byte stores at the end).
// Returns true when the head was written here.
template <bool TEXT, bool PACKED, bool CHUNKED>
__device__ __forceinline__ bool run_row_biallelic(const SegCtx& c, uint32_t skipc, uint32_t g_stop, const HeadJob& head, uint32_t lane)
{
    const uint32_t g_end = c.G, g_fast = (c.S - 1u) / 4u; // groups [0, g_fast) are complete and do not hold the last sample
    uint32_t gb = 0, flushed = 0;
    bool head_done = false;
    const RowConst rc = make_row_const(c);
    const RowKeys keys = make_row_keys(c.k0, c.k1);
    // first-round products M0 * c0, c0 = 2 (gb + lane) [+ 1]: the lane's part never changes, the wave's part lives in
    // scalar registers; the text address likewise is a wave-uniform base plus the lane's fixed offset
    const uint64_t pl = (uint64_t)PHILOX_M0 * (2u * lane);
    const uint64_t M = PHILOX_M0, M128 = (uint64_t)PHILOX_M0 * 128u;
    constexpr uint32_t T = DNAF_RUN_TRIPS; // whole trips per pass = KiB per block of text stores
    uint64_t off = 0;                      // M0 * 2 * gb, the wave's share of the first-round products
    const uint32_t tl = 16u * lane;
    // one whole trip at gb, stored at once (clip > 0: only the bytes from cell-stream offset clip on)
    auto single = [&](uint32_t clip) {
        uint32_t pk = group_code(pl, off, off + M, rc, keys, c.t0);
        if (c.frow) pk = force_code(c, pk, gb + lane);
        if (TEXT) {
            const u32x4_t w = c.lut[pk];
            if (clip) store16_from(c.cells, 16u * (gb + lane), w, clip);
            else store_unaligned<16>(c.cells + 16ull * (gb + lane), &w);
        }
        if (PACKED) c.stage[(gb + lane) & (PK_STAGE - 1u)] = (uint8_t)pk;
        gb += 64u; off += M128;
        if (PACKED && CHUNKED && (gb & 1023u) == 0u) { flush_chunk(c, flushed, lane); ++flushed; }
    };
    if (skipc) {
        // the run's first row: its first trip is clipped at the line boundary; T trips on their own keep the passes below
        // on multiples of 64 T groups (a 1024-group chunk of packed bytes is then never straddled)
        single(skipc);
        for (uint32_t t = 1; t < T && gb + 64u <= g_fast; ++t) single(0u);
    }
    {
        gptr_t tb = uniform_ptr(c.cells + 16ull * gb);
        LDS_PTR(uint8_t) sp = c.stage + ((gb + lane) & (PK_STAGE - 1u));
        // T trips per pass, software-pipelined: the cells looked up in one pass are stored in the next, so the LDS reads fly
        // under the following Philox rounds.  The T KiB of a pass leave in back-to-back stores: a 128-byte line that two
        // stores each cover a part of is written twice, and when the two parts are a pass apart in time (microseconds) the
        // first has left the L2 by then -- HBM sees two partial writes.  The longer the blocks, the fewer such lines
        // (tools/placement_study.hip, k_text_own with s_sleep: 1 KiB blocks cost 13 % of the store rate).
        if (gb + 64u * T <= g_fast) {
            u32x4_t w[T];
#pragma unroll
            for (uint32_t t = 0; t < T; ++t) w[t] = u32x4_t{0, 0, 0, 0};
            auto pass = [&](bool first) {
                uint32_t pk[T];
#pragma unroll
                for (uint32_t t = 0; t < T; ++t) pk[t] = group_code(pl, off + t * M128, off + t * M128 + M, rc, keys, c.t0);
                if (c.frow) { // wave-uniform
#pragma unroll
                    for (uint32_t t = 0; t < T; ++t) pk[t] = force_code(c, pk[t], gb + 64u * t + lane);
                }
                if (TEXT) {
                    if (!first) {
#pragma unroll
                        for (uint32_t t = 0; t < T; ++t) store16_global(tb - 1024 * (int)(T - t) + tl, w[t]);
                    }
#pragma unroll
                    for (uint32_t t = 0; t < T; ++t) w[t] = c.lut[pk[t]];
                }
                if (PACKED) {
#pragma unroll
                    for (uint32_t t = 0; t < T; ++t) sp[64u * t] = (uint8_t)pk[t];
                }
                gb += 64u * T; off += T * M128; tb += 1024 * T; sp += 64 * T;
                if (PACKED && (gb & 1023u) == 0u) { // wave-uniform
                    if (CHUNKED) { flush_chunk(c, flushed, lane); ++flushed; }
                    if ((gb & (PK_STAGE - 1u)) == 0u) sp -= PK_STAGE;
                }
            };
            pass(true);
            while (gb + 64u * T <= g_fast) pass(false);
            if (TEXT) {
#pragma unroll
                for (uint32_t t = 0; t < T; ++t) store16_global(tb - 1024 * (int)(T - t) + tl, w[t]);
            }
        }
        // the whole trips that are left (fewer than T): one more block
        {
            u32x4_t w[T];
            uint32_t n = 0;
#pragma unroll
            for (uint32_t t = 0; t + 1 < T; ++t) {
                if (gb + 64u <= g_fast) { // wave-uniform
                    uint32_t pk = group_code(pl, off, off + M, rc, keys, c.t0);
                    if (c.frow) pk = force_code(c, pk, gb + lane);
                    if (TEXT) w[t] = c.lut[pk];
                    if (PACKED) c.stage[(gb + lane) & (PK_STAGE - 1u)] = (uint8_t)pk;
                    gb += 64u; off += M128; n = t + 1;
                    if (PACKED && CHUNKED && (gb & 1023u) == 0u) { flush_chunk(c, flushed, lane); ++flushed; }
                }
            }
            if (TEXT) {
#pragma unroll
                for (uint32_t t = 0; t + 1 < T; ++t)
                    if (t < n) store_unaligned<16>(c.cells + 16ull * (gb - 64u * (n - t) + lane), &w[t]);
            }
        }
    }
    // the last, partly filled trip; its last group holds the row's last sample (newline, maybe < 4 samples).  Its idle
    // lanes take the first groups of the next wave's row when there is room (and no forced cells to apply)
    {
        const uint32_t ntail = g_end - gb; // 1 .. 64
        if (!c.frow && (head.nh == 0u || ntail + head.nh <= 64u)) {
            mixed_trip<TEXT, PACKED>(c, rc, keys, gb, ntail, head, lane);
            head_done = head.nh != 0u;
        } else {
            const uint32_t g = gb + lane;
            if (g < g_end) {
                uint32_t pk = group_code(pl, (uint64_t)PHILOX_M0 * (2u * gb), (uint64_t)PHILOX_M0 * (2u * gb) + M, rc, keys, c.t0);
                if (c.frow) pk = force_code(c, pk, g);
                u32x4_t w = {0, 0, 0, 0};
                if (TEXT) w = c.lut[pk];
                if (g < g_fast) {
                    if (TEXT) store_unaligned<16>(c.cells + 16ull * g, &w);
                } else {
                    const uint32_t nvalid = c.S - 4u * g;
                    const uint32_t fix = 0x0A000000u - 0x09000000u;
                    if (TEXT) {
                        uint8_t* dst = c.cells + 16ull * g;
                        const uint32_t w0 = w.x + (nvalid == 1u ? fix : 0u), w1 = w.y + (nvalid == 2u ? fix : 0u);
                        const uint32_t w2 = w.z + (nvalid == 3u ? fix : 0u), w3 = w.w + (nvalid == 4u ? fix : 0u);
                        store_unaligned<4>(dst, &w0);
                        if (nvalid > 1u) store_unaligned<4>(dst + 4, &w1);
                        if (nvalid > 2u) store_unaligned<4>(dst + 8, &w2);
                        if (nvalid > 3u) store_unaligned<4>(dst + 12, &w3);
                    }
                    pk &= (1u << (2u * nvalid)) - 1u; // codes of samples past the end stay 0
                }
                if (PACKED) c.stage[g & (PK_STAGE - 1u)] = (uint8_t)pk;
            }
        }
    }
    if (PACKED) {
        // pad bytes of the row (up to the pitch when it is padded: the last line leaves whole), then what the ring still holds
        for (uint32_t g = g_end + lane; g < g_stop; g += 64u) {
            if (CHUNKED) c.stage[g & (PK_STAGE - 1u)] = 0;
            else c.prow[g] = 0;
        }
        const uint32_t f0 = CHUNKED ? 1024u * flushed : 0u, n = (CHUNKED ? g_stop : g_end) - f0;
        uint8_t* dst = c.prow + f0;
        for (uint32_t o = 16u * lane; o + 16u <= n; o += 1024u) {
            const u32x4_t v = *reinterpret_cast<const LDS_PTR(u32x4_t)>(c.stage + ((f0 + o) & (PK_STAGE - 1u)));
            store_unaligned<16>(dst + o, &v);
        }
        if (lane < (n & 15u)) dst[(n & ~15u) + lane] = c.stage[(f0 + (n & ~15u) + lane) & (PK_STAGE - 1u)];
    }
    return head_done;
}

// One complete group of a biallelic X row from its packed byte (the hand-written Philox group gives all eight alleles; a
// male's second allele is drawn and dropped, pop_factory.py:477, :486-494): the diploid text of the four cells comes from
// the LDS table, the males' cells are cut to "a\t" and the cells shifted together in registers (as x_fast does from its own
// draws); the lane stores the first 8 and the last 8 bytes of its 8..16-byte run.
template <bool TEXT, bool PACKED>
__device__ __forceinline__ void x_group_from_code(const SegCtx& c, uint32_t g, uint32_t pk)
{
    const uint32_t m = c.male_nib[g];
    if (TEXT) {
        const u32x4_t w = c.lut[pk & 0xFFu];
        const bool m0 = m & 1u, m1 = m & 2u, m2 = m & 4u, m3 = m & 8u;
        const uint32_t c0 = m0 ? (w.x & 0xFFu) | 0x0900u : w.x, c1 = m1 ? (w.y & 0xFFu) | 0x0900u : w.y;
        const uint32_t c2 = m2 ? (w.z & 0xFFu) | 0x0900u : w.z, c3 = m3 ? (w.w & 0xFFu) | 0x0900u : w.w;
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
        const uint32_t sl = 8u - 2u * (uint32_t)__builtin_popcount(m); // run length - 8, in bytes
        const uint32_t q = sl >> 2, sh = (sl & 2u) << 3;
        const uint32_t t0 = q == 0u ? r0 : (q == 1u ? r1 : r2);
        const uint32_t t1 = q == 0u ? r1 : (q == 1u ? r2 : r3);
        const uint32_t t2 = q == 0u ? r2 : (q == 1u ? r3 : 0u);
        uint8_t* dst = c.cells + c.xoff[g];
        const uint2 head = make_uint2(r0, r1);
        const uint2 tail = make_uint2(__builtin_amdgcn_alignbit(t1, t0, sh), __builtin_amdgcn_alignbit(t2, t1, sh));
        store_unaligned<8>(dst, &head);
        store_unaligned<8>(dst + sl, &tail);
    }
    if (PACKED) {
        uint32_t sp = (m | (m << 2)) & 0x33u; // male bit k -> bit 2k
        sp = (sp | (sp << 1)) & 0x55u;
        c.prow[g] = (uint8_t)(pk & ~(sp << 1)); // haploid: no second allele
    }
}

// The 4 cells of a complete Y / MT group from its packed byte (the second alleles are drawn and dropped like a male's on
// X): "a\t" for a haploid sample, ".\t" for a female on Y (pop_factory.py:481-484); 8 bytes per lane.
template <bool TEXT, bool PACKED, bool IS_Y>
__device__ __forceinline__ void haploid_group_from_code(const SegCtx& c, uint32_t g, uint32_t pk)
{
    const uint32_t m = IS_Y ? (uint32_t)c.male_nib[g] : 0xFu;
    if (TEXT) {
        const u32x4_t w = c.lut[pk & 0xFFu];
        const uint32_t c0 = (m & 1u) ? (w.x & 0xFFu) | 0x0900u : 0x092Eu, c1 = (m & 2u) ? (w.y & 0xFFu) | 0x0900u : 0x092Eu;
        const uint32_t c2 = (m & 4u) ? (w.z & 0xFFu) | 0x0900u : 0x092Eu, c3 = (m & 8u) ? (w.w & 0xFFu) | 0x0900u : 0x092Eu;
        const uint2 d = make_uint2(c0 | (c1 << 16), c2 | (c3 << 16));
        store_unaligned<8>(c.cells + 8ull * g, &d);
    }
    if (PACKED) {
        uint32_t sp = (m | (m << 2)) & 0x33u; // male bit k -> bit 2k
        sp = (sp | (sp << 1)) & 0x55u;
        c.prow[g] = (uint8_t)(pk & sp); // the first allele of the haploid samples
    }
}

// Number of leading groups of a row whose cells begin below cell-stream offset `bytes` (<= 127): those are the
// groups a line boundary `bytes` into the cell stream can cut.  At most 16 (a group is at least 8 bytes).
__device__ __forceinline__ uint32_t groups_below(const SegCtx& c, uint32_t cls, uint32_t bytes, uint32_t lane)
{
    if (cls == 0u) return (bytes + 15u) / 16u;
    if (cls != 1u) return (bytes + 7u) / 8u;
    const bool below = lane < 16u && lane < c.G && c.xoff[lane] < bytes;
    return (uint32_t)__builtin_popcountll(__ballot(below));
}

// Any other row (X, Y, MT, multi-allelic): the class loops of rowgen_kernels.hip.h, started behind the groups that
// the line boundary cuts; those go through the clipped generic path.  Packed bytes are stored directly.
template <bool TEXT, bool PACKED, bool MULTI>
__device__ __forceinline__ void run_row_other(const SegCtx& c, uint32_t cls, uint32_t skipc, uint32_t g_stop, uint32_t lane)
{
    const uint32_t g_end = c.G, g_fast = (c.S - 1u) / 4u;
    uint32_t g_first = 0;
    if (skipc) {
        g_first = groups_below(c, cls, skipc, lane);
        if (lane < g_first) general_group<TEXT, PACKED, MULTI, true>(c, cls, lane, skipc, 0xFFFFFFFFu);
    }
    uint32_t gb = g_first;
    if (cls == 0u) { // multi-allelic autosome
        for (; gb < g_fast; gb += 64u)
            if (gb + lane < g_fast) autosome_fast<TEXT, PACKED, MULTI>(c, gb + lane);
    } else if (cls == 1u && !MULTI) {
        // X rows are one row in twenty: the same hand-written Philox group as the autosome rows (90 instructions instead of
        // the compiler's 110 + the compares), its packed byte turned into text through the LDS table
        const RowConst rc = make_row_const(c);
        const RowKeys keys = make_row_keys(c.k0, c.k1);
        const uint64_t pl = (uint64_t)PHILOX_M0 * (2u * lane);
        for (; gb < g_fast; gb += 64u) {
            const uint64_t off = (uint64_t)PHILOX_M0 * (2u * gb);
            uint32_t pk = group_code(pl, off, off + PHILOX_M0, rc, keys, c.t0);
            if (gb + lane < g_fast) {
                if (c.frow) pk = force_code(c, pk, gb + lane);
                x_group_from_code<TEXT, PACKED>(c, gb + lane, pk);
            }
        }
    } else if (!MULTI) { // Y (cls 2) / MT rows: the hand-written Philox group as well
        const RowConst rc = make_row_const(c);
        const RowKeys keys = make_row_keys(c.k0, c.k1);
        const uint64_t pl = (uint64_t)PHILOX_M0 * (2u * lane);
        for (; gb < g_fast; gb += 64u) {
            const uint64_t off = (uint64_t)PHILOX_M0 * (2u * gb);
            uint32_t pk = group_code(pl, off, off + PHILOX_M0, rc, keys, c.t0);
            if (gb + lane < g_fast) {
                if (c.frow) pk = force_code(c, pk, gb + lane); // (a carrier's "1/1": both alleles set; a female on Y stays '.')
                if (cls == 2u) haploid_group_from_code<TEXT, PACKED, true>(c, gb + lane, pk);
                else haploid_group_from_code<TEXT, PACKED, false>(c, gb + lane, pk);
            }
        }
    } else if (cls == 2u) {
        for (; gb < g_fast; gb += 64u)
            if (gb + lane < g_fast) haploid_fast<TEXT, PACKED, MULTI, true>(c, gb + lane);
    } else if (cls == 3u) {
        for (; gb < g_fast; gb += 64u)
            if (gb + lane < g_fast) haploid_fast<TEXT, PACKED, MULTI, false>(c, gb + lane);
    }
    // what is left: the row's last group and the packed pad bytes (or all of a multi-allelic X row)
    for (gb = (cls == 1u && MULTI) ? g_first : max(g_first, g_fast); gb < g_stop; gb += 64u)
        if (gb + lane < g_stop) general_group<TEXT, PACKED, MULTI>(c, cls, gb + lane);
    (void)g_end;
}

__device__ __forceinline__ void make_ctx(SegCtx& c, const GenParams& p, const RowRec& rec, uint64_t row_local, uint8_t* row_text)
{
    const uint64_t rng_row = p.rng_row0 + rng_row_of(p, p.row_begin + row_local);
    c.t0 = rec.thr[0]; c.t1 = rec.thr[1]; c.t2 = rec.thr[2];
    c.r_lo = (uint32_t)rng_row; c.r_hi = (uint32_t)(rng_row >> 32);
    c.k0 = p.key0; c.k1 = p.key1;
    c.S = p.S; c.G = p.G;
    c.cells = row_text + rec.prefix_len;
    c.prow = p.packed + row_local * p.packed_pitch;
    c.frow = rec.slot >= 0 ? p.forced + (uint64_t)rec.slot * p.G : nullptr;
    c.male_nib = p.male_nib;
    c.xoff = p.xoff;
}

template <bool TEXT, bool PACKED>
__global__ DNAF_RUN_BOUNDS void rowrun_kernel(const GenParams p)
{
    __shared__ u32x4_t cell_lut[256];
    __shared__ __attribute__((aligned(16))) uint8_t pk_stage[4][PK_STAGE];
    if (TEXT) {
        const uint32_t t = threadIdx.x;
        const u32x4_t e = {CELL00 + (t & 1u) + ((t & 2u) << 15), CELL00 + ((t >> 2) & 1u) + ((t & 8u) << 13),
                           CELL00 + ((t >> 4) & 1u) + ((t & 32u) << 11), CELL00 + ((t >> 6) & 1u) + ((t & 128u) << 9)};
        cell_lut[t] = e;
        __syncthreads();
    }
    const uint32_t lane_id = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t run = (uint64_t)blockIdx.x * 4u + wave;
    const uint64_t j0 = run * p.rows_per_wave;
    if (j0 >= p.n_rows) return;
    const uint64_t j1 = min(j0 + (uint64_t)p.rows_per_wave, p.n_rows);
    const bool chunked = (p.packed_pitch & 127u) == 0u; // padded pitch
    const uint32_t g_stop = PACKED ? (chunked ? (uint32_t)p.packed_pitch : p.W * 4u) : p.G;

    SegCtx c;
    c.lut = (const LDS_PTR(u32x4_t))cell_lut;
    c.stage = (LDS_PTR(uint8_t))pk_stage[threadIdx.x >> 6];

    // What this wave writes of the next wave's first row: the bytes up to that row's first line boundary (they complete
    // this wave's last line).  They are written at the END of the run, next to the rest of that line.
    uint32_t head_bytes = 0;
    HeadJob head;
    head.nh = 0; head.hc = 0; head.cells = nullptr; head.t0 = 0;
    head.rc.x0 = head.rc.x1 = head.rc.x2 = head.rc.x3 = head.rc.k0 = head.rc.k1 = 0;
    if (TEXT && j1 < p.n_rows) {
        const RowRec rec2 = p.recs[p.row_begin + j1];
        uint8_t* row_text2 = p.text + (rec2.text_off - p.text_base);
        head_bytes = __builtin_amdgcn_readfirstlane((128u - ((uint32_t)(uintptr_t)row_text2 & 127u)) & 127u);
        if (head_bytes > rec2.prefix_len && rec2.cls == 0u && rec2.n_alleles == 2 && rec2.slot < 0) {
            // an unforced biallelic autosome row: its first groups can ride in this wave's last trip
            SegCtx c2;
            make_ctx(c2, p, rec2, j1, row_text2);
            head.hc = head_bytes - rec2.prefix_len;
            head.nh = (head.hc + 15u) / 16u;
            head.cells = c2.cells;
            head.rc = make_row_const(c2);
            head.t0 = c2.t0;
        }
    }
    bool head_done = false;
    for (uint64_t j = j0; j < j1; ++j) {
        // an opaque copy of the lane number per row: nothing derived from it is hoisted out of the row loop and kept
        // in registers across all of a row's code paths (that cost half the kernel's occupancy)
        uint32_t lane = lane_id;
        asm volatile("" : "+v"(lane));
        const RowRec rec = p.recs[p.row_begin + j];
        uint8_t* row_text = p.text + (rec.text_off - p.text_base);
        // the run's first row gives its bytes below the next line boundary to the wave before it (TEXT only)
        uint32_t skip = 0;
        if (TEXT && j == j0 && j0 != 0) skip = __builtin_amdgcn_readfirstlane((128u - ((uint32_t)(uintptr_t)row_text & 127u)) & 127u);
        if (TEXT) {
            const uint8_t* src = p.prefix + rec.prefix_off;
            for (uint32_t i = skip + lane; i < rec.prefix_len; i += 64) row_text[i] = src[i];
        }
        make_ctx(c, p, rec, j, row_text);
        const uint32_t skipc = skip > rec.prefix_len ? skip - rec.prefix_len : 0u;
        HeadJob none;
        none.nh = 0;
        const bool last = j + 1 == j1;
        if (rec.n_alleles > 2) run_row_other<TEXT, PACKED, true>(c, rec.cls, skipc, g_stop, lane);
        else if (rec.cls != 0u) run_row_other<TEXT, PACKED, false>(c, rec.cls, skipc, g_stop, lane);
        else if (chunked) head_done = run_row_biallelic<TEXT, PACKED, true>(c, skipc, g_stop, last ? head : none, lane);
        else head_done = run_row_biallelic<TEXT, PACKED, false>(c, skipc, g_stop, last ? head : none, lane);
    }
    if (TEXT && head_bytes) {
        uint32_t lane = lane_id;
        asm volatile("" : "+v"(lane));
        const RowRec rec = p.recs[p.row_begin + j1];
        uint8_t* row_text = p.text + (rec.text_off - p.text_base);
        const uint8_t* src = p.prefix + rec.prefix_off;
        const uint32_t lim = head_bytes < rec.prefix_len ? head_bytes : rec.prefix_len;
        for (uint32_t i = lane; i < lim; i += 64) row_text[i] = src[i];
        if (head_bytes > rec.prefix_len && !head_done) {
            make_ctx(c, p, rec, j1, row_text);
            if (head.nh) { // the same short cut in a trip of its own (the wave's last row was of another kind, or had no room)
                mixed_trip<true, false>(c, head.rc, make_row_keys(c.k0, c.k1), 0u, 0u, head, lane);
            } else {       // any other kind of row: cell by cell
                const uint32_t hc = head_bytes - rec.prefix_len;
                const uint32_t n = groups_below(c, rec.cls, hc, lane);
                if (lane < n) {
                    if (rec.n_alleles > 2) general_group<true, false, true, true>(c, rec.cls, lane, 0u, hc);
                    else general_group<true, false, false, true>(c, rec.cls, lane, 0u, hc);
                }
            }
        }
    }
}

} // namespace dnaf
