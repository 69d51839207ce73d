// rowmix_kernels.hip.h -- the row generator for SHORT rows (gfx950 / CDNA4): the lanes of a trip span rows.
//
// rowgen_kernel / rowrun_kernel give a trip's 64 lanes to 64 consecutive groups of ONE row.  A row of a few
// hundred samples has fewer groups than that (S = 200: 50), so most of the work would be the per-row fixed part --
// a partly filled trip, the row's scalar set-up, its fixed columns (0.13 of the HBM peak at S = 100, 0.21 at 200).
// Here a wave owns a run of up to 64 whole rows and treats the groups of consecutive ordinary rows (biallelic
// autosome, no forced cells: almost every row) as ONE sequence q = 0, 1, 2, ...: lane l of trip t takes
// q = 64 t + l, i.e. row q / G, group q mod G.  What differs between rows -- the four folded Philox constants, the
// threshold, where the row's cells start -- is computed once per row by one lane each and kept in a small LDS table;
// the Philox group sequence runs with these in vector registers (group_code_lanes_g, rowrun_kernels.hip.h).  The fixed
// columns of a row are stored right behind the cells of the trip in which the row begins (the cells leave one trip late,
// so the columns wait a trip too): a line is written in one burst of stores, not in parts a trip apart.  Runs of X rows and
// of Y / MT rows go through trips of the same kind, with the cells' places and widths from a per-group word (xg_tab).  Any
// other row of the run (multi-allelic, forced cells) is written by the row-at-a-time code of rowrun_kernels.hip.h behind them.
//
// Same cell rules, RNG lane mapping and text format as everywhere (pop_factory.py:471-513, rowgen_kernels.hip.h).
// Used wherever the packed rows are dense (W < 128 words, S <= 2,032): 0.39-0.42 instead of 0.13 of the HBM peak at S = 100,
// 0.45-0.49 / 0.21 at 200, 0.53 / 0.30 at 400, 0.52-0.56 / 0.43 at 1,000, 0.58-0.60 / 0.52 at 2,000 (profiles/r03_shapes.txt;
// second figure: a wave per row segment, from 1,021 samples a wave per run of whole rows).  A trip is 103 vector
// instructions: the 90 of the Philox group and 13 of bookkeeping (the lane's next row and group, one LDS address, table
// index, store addresses, newline); a wave's start-up (its rows' records, then the row table) is a tenth of its life.
#pragma once
#include "rowrun_kernels.hip.h"
#include <type_traits>

namespace dnaf {

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

struct MixRow {        // what a lane needs of its row: one 32-byte entry of the wave's LDS table
    uint32_t x0, x1, x2, x3;
    uint32_t t0, cells_rel, pad0, pad1; // cells_rel: the row's first cell byte, relative to the run's first text byte
};

// q / d and q mod d for q < 2^22, d < 2^16 with m = floor(2^32 / d): one multiply-high, one correction step
__device__ __forceinline__ void divmod_small(uint32_t q, uint32_t d, uint32_t m, uint32_t& quot, uint32_t& rem)
{
    uint32_t r = __umulhi(q, m);
    uint32_t g = q - r * d;
    if (g >= d) { g -= d; ++r; }
    quot = r;
    rem = g;
}

#ifndef DNAF_MIX_PFX_BYTES
#define DNAF_MIX_PFX_BYTES 3072
#endif
#ifndef DNAF_MIX_BOUNDS
#define DNAF_MIX_BOUNDS __launch_bounds__(256)
#endif
#ifndef DNAF_MIX_EXP
#define DNAF_MIX_EXP 0
#endif
constexpr uint32_t MIX_PFX_BYTES = DNAF_MIX_PFX_BYTES; // the fixed columns of a run's rows, cached in LDS (64 rows x 48 bytes, or 48 x 64: the host caps the run)
constexpr uint32_t MIX_MAX_GROUPS = 509; // the host never sends rows of this many groups or more here (dense pitch: S <= 2,032)
constexpr uint32_t MIX_STAGE = 768;      // packed bytes of one row handled a row at a time (< MIX_MAX_GROUPS + pad)

// prefix_fixed[j] = the fixed columns of row j, zero-padded to `stride` bytes (a multiple of 16): one thread per row
__global__ __launch_bounds__(256) void prefix_fix_kernel(const RowRec* recs, const uint8_t* prefix, uint8_t* out, uint64_t n, uint32_t stride)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (j >= n) return;
    const uint8_t* src = prefix + recs[j].prefix_off;
    const uint32_t len = recs[j].prefix_len;
    uint8_t* dst = out + j * stride;
    for (uint32_t i = 0; i < stride; ++i) dst[i] = i < len ? src[i] : (uint8_t)0;
}

#ifdef DNAF_MIX_PROFILE
// stage clock of rowmix_kernel: cycles (s_memtime) summed over waves; debug builds only (tools/mix_stage_profile.py)
__device__ unsigned long long mix_prof[16];
#define MIX_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane_id == 0) atomicAdd(&mix_prof[k], t_ - t_prev_); t_prev_ = t_; } while (0)
#else
#define MIX_STAMP(k) do { } while (0)
#endif

// FULL4: the sample count is a multiple of 4 (every group whole): the row's last group needs only its newline
// XMIX: runs of unforced biallelic X rows, Y rows and MT rows go through lane-spanning trips of their own (GenParams::mix_x_rows);
// otherwise the kernel holds no code for that (the further sets of trips cost 1.5 % at 2,000 samples even when they never run)
template <bool TEXT, bool PACKED, bool FULL4, bool XMIX>
__global__ DNAF_MIX_BOUNDS void rowmix_kernel(const GenParams p)
{
#ifdef DNAF_MIX_PROFILE
    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif
    __shared__ u32x4_t cell_lut[256];
    __shared__ __attribute__((aligned(16))) uint8_t pk_stage[4][MIX_STAGE];
    __shared__ __attribute__((aligned(16))) u32x4_t rows_all[4][128]; // per row of the run, 32 bytes (MixRow): the four folded Philox
                                                                     // constants; its threshold and where its cells start (cells_rel)
    __shared__ __attribute__((aligned(16))) uint8_t pfx_all[4][MIX_PFX_BYTES];
#ifdef DNAF_MIX_LDS_PAD
    __shared__ uint8_t lds_pad[DNAF_MIX_LDS_PAD]; // occupancy experiments: fewer workgroups per CU
    if (p.n_rows == ~0ull) lds_pad[threadIdx.x] = 1;
#endif
    // per group of 4 samples, for X rows (the same for every X row): byte offset of the group's cells in the row's cell
    // stream (bits 0-15), how many bytes they take (16-20: 2 per haploid male cell, 4 per diploid one; the row's last group
    // only its samples), which of the four are male (24-27; Y rows use this part only).  Made on the host (dnaf_set_samples).
    __shared__ uint32_t xg_tab[XMIX ? MIX_MAX_GROUPS + 3 : 1];
    if (XMIX)
        for (uint32_t t = threadIdx.x; t < p.G; t += 256u) xg_tab[t] = p.xgroup[t];
    if (TEXT) {
        const uint32_t t = threadIdx.x;
        const u32x4_t e = {CELL00 + (t & 1u) + ((t & 2u) << 15), CELL00 + ((t >> 2) & 1u) + ((t & 8u) << 13),
                           CELL00 + ((t >> 4) & 1u) + ((t & 32u) << 11), CELL00 + ((t >> 6) & 1u) + ((t & 128u) << 9)};
        cell_lut[t] = e;
    }
    __syncthreads();
    const uint32_t lane_id = threadIdx.x & 63u;
    MIX_STAMP(0); // cell table + barrier
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t run = (uint64_t)blockIdx.x * 4u + wave;
    const uint32_t R = p.rows_per_wave; // <= 64
    const uint64_t j0 = run * R;
    if (j0 >= p.n_rows) return;
    const uint32_t nrun = (uint32_t)min((uint64_t)R, p.n_rows - j0);
    const uint32_t G = p.G, S = p.S;
    const uint32_t g_stop = PACKED ? p.W * 4u : G;
    LDS_PTR(u32x4_t) rtab = (LDS_PTR(u32x4_t))rows_all[threadIdx.x >> 6]; // one address per lane and trip: entry 2 row, its second half 16 bytes on

    SegCtx c;
    c.lut = (const LDS_PTR(u32x4_t))cell_lut;
    c.stage = (LDS_PTR(uint8_t))pk_stage[threadIdx.x >> 6];

    // one lane per row of the run: its record, whether it is an ordinary row, its constants -> LDS
    const uint64_t text0 = p.recs[p.row_begin + j0].text_off; // first text byte of the run
    uint8_t* run_text = p.text + (text0 - p.text_base);
    uint8_t* packed_run = p.packed + j0 * p.packed_pitch;
    const uint32_t pitch32 = (uint32_t)p.packed_pitch;
    RowRec myrec = p.recs[p.row_begin + j0 + (lane_id < nrun ? lane_id : 0u)];
    // The fixed columns of the run's rows go into LDS once, so that no trip waits on a global load for the few dozen bytes in
    // front of a row (that wait, once per row, was what bound the kernel at 200 samples).  Where they are in the prefix blob
    // is written in the rows' records -- a load that depends on a load, and a wave's life is short (a third of it was the
    // start-up, tools/mix_stage_profile.py) -- so the library keeps a second copy of them at a fixed stride (prefix_fixed):
    // lane = row asks for its row's bytes together with its record.  Without that copy: the run's piece of the blob, below.
    const uint32_t pstride = p.prefix_stride;
    u32x4_t pv[4] = {u32x4_t{0, 0, 0, 0}, u32x4_t{0, 0, 0, 0}, u32x4_t{0, 0, 0, 0}, u32x4_t{0, 0, 0, 0}};
    if (TEXT && pstride) { // wave-uniform
        const uint8_t* src = p.prefix_fixed + (p.row_begin + j0 + (lane_id < nrun ? lane_id : 0u)) * pstride;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            if (16u * k < pstride) pv[k] = *reinterpret_cast<const u32x4_t*>(src + 16u * k);
    }
    const bool ordinary = lane_id < nrun && myrec.cls == 0u && myrec.n_alleles == 2 && myrec.slot < 0;
    const uint32_t my_rel = (uint32_t)(myrec.text_off - text0), my_plen = myrec.prefix_len, my_poff = myrec.prefix_off;
    {
        SegCtx cr;
        const uint64_t rng_row = p.rng_row0 + rng_row_of(p, p.row_begin + j0 + lane_id);
        cr.r_lo = (uint32_t)rng_row; cr.r_hi = (uint32_t)(rng_row >> 32);
        cr.k0 = p.key0; cr.k1 = p.key1;
        const RowConst rc = make_row_const(cr);
        MixRow e;
        e.x0 = rc.x0; e.x1 = rc.x1; e.x2 = rc.x2; e.x3 = rc.x3;
        e.t0 = myrec.thr[0]; e.cells_rel = my_rel + myrec.prefix_len; e.pad0 = e.pad1 = 0;
        if (lane_id < nrun) {
            rtab[2u * lane_id] = u32x4_t{e.x0, e.x1, e.x2, e.x3};
            *reinterpret_cast<LDS_PTR(u32x2_t)>(rtab + 2u * lane_id + 1u) = u32x2_t{e.t0, e.cells_rel};
        }
    }
    MIX_STAMP(1); // records loaded, row table written
    const uint64_t ord_mask = __ballot(ordinary);
    // unforced biallelic X rows: the same trips with the cells' widths and places from xg_tab (X rows come in long runs: a
    // sorted table has them in one block, 5 % of a genome's rows)
    const uint64_t x_mask = XMIX ? __ballot(lane_id < nrun && myrec.cls == 1u && myrec.n_alleles == 2 && myrec.slot < 0) : 0ull;
    // and Y / MT rows: four haploid cells of two bytes per group, '.' for a female on Y (pop_factory.py:481-484)
    const uint64_t y_mask = XMIX ? __ballot(lane_id < nrun && myrec.cls == 2u && myrec.n_alleles == 2 && myrec.slot < 0) : 0ull;
    const uint64_t mt_mask = XMIX ? __ballot(lane_id < nrun && myrec.cls == 3u && myrec.n_alleles == 2 && myrec.slot < 0) : 0ull;
    const uint32_t pfx0 = __builtin_amdgcn_readlane(my_poff, 0);
    const uint32_t pfx_bytes = __builtin_amdgcn_readlane(my_poff + my_plen, nrun - 1u) - pfx0;
    const bool pfx_cached = TEXT && (pstride != 0u || pfx_bytes <= MIX_PFX_BYTES) && DNAF_MIX_EXP != 1;
    LDS_PTR(uint8_t) pfx = (LDS_PTR(uint8_t))pfx_all[threadIdx.x >> 6];
    if (TEXT && pstride) {
        if (lane_id < nrun) {
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k)
                if (16u * k < pstride) *reinterpret_cast<LDS_PTR(u32x4_t)>(pfx + lane_id * pstride + 16u * k) = pv[k];
        }
    } else if (pfx_cached)
        for (uint32_t i = lane_id; i < pfx_bytes; i += 64u) pfx[i] = p.prefix[pfx0 + i];
#ifdef DNAF_MIX_PROFILE
    if (pfx_cached) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    MIX_STAMP(2); // fixed columns in LDS
    const RowKeys keys = make_row_keys(p.key0, p.key1);
    const uint32_t gmagic = p.mix_gmagic;
    const uint32_t nvalid_last = FULL4 ? 4u : S - 4u * (G - 1u); // samples in a row's last group, 1 .. 4
    const uint32_t step_groups = p.mix_step_groups; // a trip ahead: 64 groups = step_rows rows + step_groups (divided on the host)
    uint32_t neg_step = 0u - (p.mix_step_rows + 1u); // in a vector register for the loop's life: v_subb reads its one scalar, vcc, already
    asm volatile("" : "+v"(neg_step));
    const uint32_t fix = 0x0A000000u - 0x09000000u; // '\n' instead of the last '\t'

    // kind 0: biallelic autosome rows; 1: X rows; 2: Y rows (all_haploid false: '.' for females) and MT rows (true)
    auto run_stretches = [&](auto kind_tag, const uint64_t kind_mask, const bool all_haploid) {
    constexpr int KIND = decltype(kind_tag)::value;
    constexpr bool XROW = KIND == 1, HAPLOID = KIND == 2;
    uint32_t r = 0;
    while (r < nrun) {
        uint32_t lane = lane_id; // an opaque copy per stretch: keeps lane-derived values out of registers across the loop
        asm volatile("" : "+v"(lane));
        if (!((kind_mask >> r) & 1ull)) { // any other kind of row: in another pass, or after the stretches, below
            ++r;
            continue;
        }
        // the stretch of rows of this kind [r, e)
        const uint64_t rest = ~(kind_mask >> r);
        const uint32_t len = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
        const uint32_t e = min(nrun, r + len);
        const uint32_t n = (e - r) * G; // groups of the stretch
        uint32_t next_row = r, next_q = 0; // the next row whose first group has not been reached yet
        // Software pipeline over the trips: the row-table entries of trip t+1 are read before trip t's arithmetic, and the
        // text of trip t (a look-up in the LDS cell table) is stored after trip t+1's arithmetic -- no trip waits on LDS.
        auto coords = [&](uint32_t qb, uint32_t& row, uint32_t& g, bool& live) {
            const uint32_t q = qb + lane;
            live = q < n;
            divmod_small(q, G, gmagic, row, g); // lanes past the stretch get coordinates too: they are stepped from there
            row += r;
        };
        if (TEXT && !pfx_cached) {
            // no copy of the fixed columns in LDS (they are longer than the cache): all of the stretch's, ahead of its trips.
            // Kept out of the trip loop: a global load in there makes the compiler wait for the loop's STORES too (vmcnt
            // counts both), once per trip.
            for (uint32_t rr = r; rr < e; ++rr) {
                const uint32_t plen = __builtin_amdgcn_readlane(my_plen, rr), poff = __builtin_amdgcn_readlane(my_poff, rr);
                const uint32_t rel = __builtin_amdgcn_readlane(my_rel, rr);
                for (uint32_t i = lane; i < plen; i += 64) run_text[rel + i] = p.prefix[poff + i];
            }
        }
        // No load is in flight past this point.  The waves' loads above and in the other branch of this loop would otherwise
        // leave the compiler's scoreboard with "maybe pending" registers at the head of the trip loop, and it would wait for
        // vmcnt inside the loop -- where that counter also holds the loop's own stores, i.e. every trip would wait for the
        // stores of the trip before it (measured: the vector units 67 % busy).  vmcnt(0), other counters untouched.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        uint32_t row0, g0;
        bool live0;
        coords(0u, row0, g0, live0);
        u32x4_t xs0 = rtab[2u * row0];
        u32x2_t ts0 = *reinterpret_cast<LDS_PTR(u32x2_t)>(rtab + 2u * row0 + 1u);
        uint32_t xe0 = KIND ? xg_tab[g0 < G ? g0 : 0u] : 0u; // X / Y rows: the lane's group in xg_tab
        bool have_prev = false, live_prev = false, last_prev = false;
        u32x4_t w_prev = {0, 0, 0, 0};
        uint32_t off_prev = 0, xe_prev = 0;
        auto store_cells = [&](u32x4_t w, uint32_t off, bool live, bool last) {
            if (!live) return;
            uint8_t* dst = run_text + off;
            if (FULL4) { // every group is whole, the row's last one ends in a newline
                w.w += last ? fix : 0u;
                store_unaligned<16>(dst, &w);
            } else if (!last) store_unaligned<16>(dst, &w);
            else { // the group with the row's last sample: newline instead of the last tab, fewer than 4 samples
                const uint32_t w0 = w.x + (nvalid_last == 1u ? fix : 0u), w1 = w.y + (nvalid_last == 2u ? fix : 0u);
                const uint32_t w2 = w.z + (nvalid_last == 3u ? fix : 0u);
                store_unaligned<4>(dst, &w0);
                if (nvalid_last > 1u) store_unaligned<4>(dst + 4, &w1);
                if (nvalid_last > 2u) store_unaligned<4>(dst + 8, &w2);
            }
        };
        // X row: the diploid text of the four cells from the table, the males' cells cut to "a\t" and the cells shifted together
        // in registers (x_group_from_code, rowrun_kernels.hip.h); the lane stores the first 8 and the last 8 bytes of its run
        auto store_cells_x = [&](const u32x4_t& w, uint32_t off, uint32_t xe, bool live, bool last) {
            if (!live) return;
            const uint32_t m = xe >> 24, len = (xe >> 16) & 31u;
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
            uint8_t* dst = run_text + off;
            if (!FULL4 && last) { // the row's last byte (the tab of its last cell, an odd byte of r0 .. r3) becomes the newline
                const uint32_t one = (len & 2u) ? 0x100u : 0x1000000u, wi = (len - 1u) >> 2;
                r0 += wi == 0u ? one : 0u; r1 += wi == 1u ? one : 0u; r2 += wi == 2u ? one : 0u; r3 += wi == 3u ? one : 0u;
            }
            if (FULL4 || !last || len >= 8u) { // 8 .. 16 bytes
                const uint32_t sl = len - 8u;
                const uint32_t q = sl >> 2, sh = (sl & 2u) << 3;
                const uint32_t t0 = q == 0u ? r0 : (q == 1u ? r1 : r2);
                const uint32_t t1 = q == 0u ? r1 : (q == 1u ? r2 : r3);
                const uint32_t t2 = q == 0u ? r2 : (q == 1u ? r3 : 0u);
                uint2 head = make_uint2(r0, r1);
                uint2 tail = make_uint2(__builtin_amdgcn_alignbit(t1, t0, sh), __builtin_amdgcn_alignbit(t2, t1, sh));
                if (FULL4 && last) { // the row's last byte is a newline; with 8 bytes in all the two stores are the same bytes
                    tail.y += fix;
                    if (sl == 0u) head.y += fix;
                }
                store_unaligned<8>(dst, &head);
                store_unaligned<8>(dst + sl, &tail);
            } else { // the row's last group when it has fewer than 8 bytes (one to three samples): 2, 4 or 6 bytes
                const uint16_t h0 = (uint16_t)r0, h1 = (uint16_t)(r0 >> 16), h2 = (uint16_t)r1;
                store_unaligned<2>(dst, &h0);
                if (len >= 4u) store_unaligned<2>(dst + 2, &h1);
                if (len >= 6u) store_unaligned<2>(dst + 4, &h2);
            }
        };
        // Y / MT row: four cells of two bytes, "a\t" for a haploid sample (its first allele), ".\t" for a female on Y
        auto store_cells_h = [&](const u32x4_t& w, uint32_t off, uint32_t xe, bool live, bool last) {
            if (!live) return;
            const uint32_t m = all_haploid ? 0xFu : xe >> 24;
            const uint32_t c0 = (m & 1u) ? (w.x & 0xFFu) | 0x0900u : 0x092Eu, c1 = (m & 2u) ? (w.y & 0xFFu) | 0x0900u : 0x092Eu;
            const uint32_t c2 = (m & 4u) ? (w.z & 0xFFu) | 0x0900u : 0x092Eu, c3 = (m & 8u) ? (w.w & 0xFFu) | 0x0900u : 0x092Eu;
            uint8_t* dst = run_text + off;
            if (FULL4 || !last) {
                uint2 d = make_uint2(c0 | (c1 << 16), c2 | (c3 << 16));
                if (FULL4 && last) d.y += fix;
                store_unaligned<8>(dst, &d);
            } else { // the row's last group: one to three samples, the last one's tab is the newline
                const uint16_t h0 = (uint16_t)(c0 + (nvalid_last == 1u ? 0x100u : 0u)), h1 = (uint16_t)(c1 + (nvalid_last == 2u ? 0x100u : 0u));
                const uint16_t h2 = (uint16_t)(c2 + 0x100u);
                store_unaligned<2>(dst, &h0);
                if (nvalid_last > 1u) store_unaligned<2>(dst + 2, &h1);
                if (nvalid_last > 2u) store_unaligned<2>(dst + 4, &h2);
            }
        };
        auto store_prev = [&]() {
            if (XROW) store_cells_x(w_prev, off_prev, xe_prev, live_prev, last_prev);
            else if (HAPLOID) store_cells_h(w_prev, off_prev, xe_prev, live_prev, last_prev);
            else store_cells(w_prev, off_prev, live_prev, last_prev);
        };
        // The fixed columns (pop_factory.py:503-507, formatted on the host) of the rows whose first group lies below group
        // `upto` of the stretch: stored right behind the cells of the trip in which those rows begin -- the cells leave one
        // trip late (software pipeline), so the columns wait a trip too and the line they share with those cells is written
        // within one burst of stores, not in two parts a trip apart.
        auto copy_columns = [&](uint32_t upto) {
            while (DNAF_MIX_EXP != 2 && next_row < e && next_q < upto) {
                const uint32_t plen = __builtin_amdgcn_readlane(my_plen, next_row);
                const uint32_t rel = __builtin_amdgcn_readlane(my_rel, next_row);
                const uint32_t at = pstride ? next_row * pstride : __builtin_amdgcn_readlane(my_poff, next_row) - pfx0;
                if (pstride) { // (wave-uniform) at most pstride <= 64 bytes: one predicated byte store, no loop to set up
                    if (lane < plen) run_text[rel + lane] = pfx[at + lane];
                } else {
#pragma clang loop unroll(disable)
                    for (uint32_t i = lane; i < plen; i += 64) run_text[rel + i] = pfx[at + i];
                }
                ++next_row;
                next_q += G;
            }
        };
        // one trip at qb with the current coordinates / table entries (row0, g0, live0, xs0, ts0); those of the trip after it
        // are produced into (row1, ...).  Called alternately with the two register sets, so nothing is copied between trips.
        auto trip = [&](uint32_t qb, uint32_t row0, uint32_t g0, bool live0, const u32x4_t& xs0, const u32x2_t& ts0, uint32_t xe0,
                        uint32_t& row1, uint32_t& g1, bool& live1, u32x4_t& xs1, u32x2_t& ts1, uint32_t& xe1) {
            if (qb + 64u < n) { // wave-uniform
                // one trip on: no division, the lane's group moves by 64 mod G and its row by 64 / G (+ 1 when the group wraps);
                // four instructions: the borrow of gs - G selects the group and is subtracted from row0 + 64 / G + 1
                uint32_t gs, gt;
                asm("v_add_u32 %[gs], %[sg], %[g0]\n\t"
                    "v_subrev_co_u32 %[gt], vcc, %[G], %[gs]\n\t"
                    "v_cndmask_b32 %[g1], %[gt], %[gs], vcc\n\t"
                    "v_subb_co_u32 %[row1], vcc, %[row0], %[ns], vcc"
                    : [gs] "=&v"(gs), [gt] "=&v"(gt), [g1] "=&v"(g1), [row1] "=&v"(row1)
                    : [sg] "s"(step_groups), [G] "s"(G), [g0] "v"(g0), [row0] "v"(row0), [ns] "v"(neg_step)
                    : "vcc");
                live1 = row1 < e; // the groups of the stretch's rows are all live, those of "rows" past it are not
                xs1 = rtab[2u * row1];
                ts1 = *reinterpret_cast<LDS_PTR(u32x2_t)>(rtab + 2u * row1 + 1u);
                if (KIND) xe1 = xg_tab[g1];
            }
            const uint64_t P = (uint64_t)PHILOX_M0 * g0;
            uint32_t pk = group_code_lanes_g(P, xs0.x, xs0.y, xs0.z, xs0.w, ts0.x, keys);
            const bool last = g0 == G - 1u;
            if (TEXT) {
                if (have_prev) store_prev();
                if (pfx_cached) copy_columns(qb);
                w_prev = c.lut[pk & 0xFFu];
                off_prev = XROW ? ts0.y + (xe0 & 0xFFFFu) : ts0.y + (HAPLOID ? 8u : 16u) * g0;
                xe_prev = xe0;
                live_prev = live0;
                last_prev = last;
                have_prev = true;
            }
            if (PACKED && live0) {
                if (KIND) { // male bit k -> bit 2k
                    const uint32_t m = (HAPLOID && all_haploid) ? 0xFu : xe0 >> 24;
                    uint32_t sp = (m | (m << 2)) & 0x33u;
                    sp = (sp | (sp << 1)) & 0x55u;
                    if (XROW) pk &= ~(sp << 1); // haploid males: no second allele
                    else pk &= sp;              // Y / MT: the first allele of the haploid samples, nothing for a female on Y
                }
                if (last) pk &= (1u << (2u * nvalid_last)) - 1u; // codes of samples past the end stay 0
                packed_run[row0 * pitch32 + g0] = (uint8_t)pk;   // 32-bit offsets from the run's first packed byte
            }
        };
        {
            uint32_t row1 = row0, g1 = g0, xe1 = xe0;
            bool live1 = false;
            u32x4_t xs1 = xs0;
            u32x2_t ts1 = ts0;
            for (uint32_t qb = 0; qb < n; qb += 128u) {
                trip(qb, row0, g0, live0, xs0, ts0, xe0, row1, g1, live1, xs1, ts1, xe1);
                if (qb + 64u < n) trip(qb + 64u, row1, g1, live1, xs1, ts1, xe1, row0, g0, live0, xs0, ts0, xe0);
            }
        }
        if (TEXT && have_prev) store_prev();
        if (TEXT && pfx_cached) copy_columns(n);
        if (PACKED && g_stop > G) { // pad bytes of the stretch's packed rows (at most 3 per row)
            const uint32_t npad = g_stop - G;
            for (uint32_t i = lane; i < (e - r) * npad; i += 64u) {
                const uint32_t rr = i / npad, k = i - rr * npad;
                packed_run[(r + rr) * pitch32 + G + k] = 0;
            }
        }
        r = e;
    }
    };
    run_stretches(std::integral_constant<int, 0>{}, ord_mask, false);
    if (XMIX && x_mask) run_stretches(std::integral_constant<int, 1>{}, x_mask, false); // (wave-uniform)
    if (XMIX && y_mask) run_stretches(std::integral_constant<int, 2>{}, y_mask, false);
    if (XMIX && mt_mask) run_stretches(std::integral_constant<int, 2>{}, mt_mask, true);
    // The rows of any other kind (multi-allelic, forced cells; without XMIX also X / Y / MT), a row at a time by the code of
    // rowrun_kernels.hip.h, whole, with this wave's own first and last bytes.  In a loop of their own behind the stretches
    // rather than between them: as a branch inside the loop above, this rare path's live values pushed the trip loop's
    // scalars out of their registers (it reloaded them with v_readlane every trip).
    for (uint64_t left = ~(ord_mask | x_mask | y_mask | mt_mask) & (nrun >= 64u ? ~0ull : (1ull << nrun) - 1ull); left; left &= left - 1ull) {
        const uint32_t ro = (uint32_t)__builtin_ctzll(left);
        uint32_t lane = lane_id;
        asm volatile("" : "+v"(lane));
        const RowRec rec = p.recs[p.row_begin + j0 + ro];
        uint8_t* row_text = p.text + (rec.text_off - p.text_base);
        if (TEXT) {
            const uint8_t* src = p.prefix + rec.prefix_off;
            for (uint32_t i = lane; i < rec.prefix_len; i += 64) row_text[i] = src[i];
        }
        make_ctx(c, p, rec, j0 + ro, row_text);
        const HeadJob none{};
        if (rec.n_alleles > 2) run_row_other<TEXT, PACKED, true>(c, rec.cls, 0u, g_stop, lane);
        else if (rec.cls != 0u) run_row_other<TEXT, PACKED, false>(c, rec.cls, 0u, g_stop, lane);
        else (void)run_row_biallelic<TEXT, PACKED, false>(c, 0u, g_stop, none, lane); // forced cells
    }
    MIX_STAMP(3); // the rows
#ifdef DNAF_MIX_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MIX_STAMP(4); // stores drained
    if (lane_id == 0) atomicAdd(&mix_prof[8], 1ull);
#endif
}

} // namespace dnaf
