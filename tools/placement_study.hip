// placement_study.hip -- why does the row kernel's time depend on where the packed matrix was allocated?
//
// Stand-alone driver of dnaf::rowgen_kernel (the product kernel, included as is) on a synthetic all-autosome
// table at the bench shape (S = 20,000, 250,000 rows), with full control over where the two output buffers live:
//   lottery   K separate hipMalloc'ed matrices (all kept), the same launch timed on each
//   arena     one big allocation, the matrix placed at chosen offsets inside it
//   pitch     the matrix written at other row pitches (5000 B = the ABI's dense rows, 5120 = whole 128-B lines ...)
//   split     text only / packed only on the slowest and the fastest placement
// Every timed launch prints one line "D<n> ..." with its dispatch number, so that a rocprofv3 --pmc run of this
// program (per-dispatch counters) can be joined with the times.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o placement_study placement_study.hip
#include "../dna-factory_amd/csrc/rowgen_kernels.hip.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

using dnaf::GenParams;
using dnaf::RowRec;

// pure aligned streaming stores over [dst, dst + bytes): the store rate of a physical region by itself
__global__ __launch_bounds__(256) void k_fill(uint4* dst, uint64_t n16)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = v;
}

// the row kernel's two store streams without its arithmetic: one wave per row segment writes the segment's text
// (1 KiB per trip, byte-aligned rows) and then its piece of the packed row in 16-byte stores
__global__ __launch_bounds__(256) void k_two_streams(uint8_t* text, uint8_t* packed, uint64_t row_bytes, uint32_t pfx, uint64_t pitch,
                                                     uint64_t n_rows, uint32_t nseg, uint32_t seg_groups, uint32_t G, uint32_t what)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t row = item / nseg;
    const uint32_t seg = (uint32_t)(item % nseg);
    if (row >= n_rows) return;
    const uint32_t g0 = seg * seg_groups, g1 = g0 + seg_groups < G ? g0 + seg_groups : G;
    uint32_t v[4] = {0x09302F30u, 0x09302F31u, 0x09312F30u, 0x09312F31u};
    if (what & 1u) {
        uint8_t* p = text + row * row_bytes + pfx + 16ull * (g0 + lane);
        for (uint32_t g = g0; g + 64u <= g1; g += 64u, p += 1024) {
            v[0] += lane;
            __builtin_memcpy(p, v, 16);
        }
    }
    if (what & 2u) {
        uint8_t* q = packed + row * pitch + g0;
        const uint32_t n = g1 - g0;
        for (uint32_t o = 16u * lane; o + 16u <= n; o += 1024u) __builtin_memcpy(q + o, v, 16);
    }
}

// variants of the packed stream of k_two_streams (the text stream is unchanged):
//  0 as the row kernel: each wave stores its piece after its text            1 piece stored BEFORE the text
//  2 the piece in nt stores     3 the piece in sc1 (write-through) stores    4 one wave (the row's last segment) stores the whole packed row
//  5 pieces stored by 4-byte stores (64 lanes x 4 B = 256 B per instruction) 6 piece spread over the trips: 64 B after every text KiB
//  7 the whole row's packed bytes by the first segment's wave, before its text
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int V>
__global__ __launch_bounds__(256) void k_two_var(uint8_t* text, uint8_t* packed, uint64_t row_bytes, uint32_t pfx, uint64_t pitch,
                                                 uint64_t n_rows, uint32_t nseg, uint32_t seg_groups, uint32_t G)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t row = item / nseg;
    const uint32_t seg = (uint32_t)(item % nseg);
    if (row >= n_rows) return;
    const uint32_t g0 = seg * seg_groups, g1 = g0 + seg_groups < G ? g0 + seg_groups : G;
    u32x4_a1 v = {0x09302F30u, 0x09302F31u, 0x09312F30u, 0x09312F31u};
    uint8_t* q = packed + row * pitch + g0;
    const uint32_t n = g1 - g0;
    auto piece = [&](uint8_t* dst, uint32_t len) {
        if (V == 5) { for (uint32_t o = 4u * lane; o + 4u <= len; o += 256u) __builtin_memcpy(dst + o, &v, 4); return; }
        for (uint32_t o = 16u * lane; o + 16u <= len; o += 1024u) {
            if (V == 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_a1*>(dst + o));
            else if (V == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst + o), "v"(v) : "memory");
            else __builtin_memcpy(dst + o, &v, 16);
        }
    };
    if (V == 1) piece(q, n);
    if (V == 7 && seg == 0) piece(packed + row * pitch, G);
    uint8_t* p = text + row * row_bytes + pfx + 16ull * (g0 + lane);
    uint32_t po = 0;
    for (uint32_t g = g0; g + 64u <= g1; g += 64u, p += 1024) {
        v.x += lane;
        __builtin_memcpy(p, &v, 16);
        if (V == 6) { if (lane < 4 && po + 64u <= n) __builtin_memcpy(q + po + 16u * lane, &v, 16); po += 64u; }
    }
    if (V == 0 || V == 2 || V == 3 || V == 5) piece(q, n);
    if (V == 4 && seg == nseg - 1) piece(packed + row * pitch, G);
}

// text stream alignment study: rows of row_bytes (a multiple of 128), cells start at byte `mis` of the row, so `mis` is the
// misalignment of every 1 KiB wave store; seg_groups a multiple of 8 keeps segment boundaries on 128-byte lines when mis = 0.
// packed: 0 none, 1 dense pieces (pitch 5000), 2 pitch 5120 with line-aligned pieces
__global__ __launch_bounds__(256) void k_text_align(uint8_t* text, uint8_t* packed, uint64_t row_bytes, uint32_t mis, uint64_t n_rows,
                                                    uint32_t nseg, uint32_t seg_groups, uint32_t G, uint32_t pk_mode)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t row = item / nseg;
    const uint32_t seg = (uint32_t)(item % nseg);
    if (row >= n_rows) return;
    const uint32_t g0 = seg * seg_groups, g1 = g0 + seg_groups < G ? g0 + seg_groups : G;
    uint32_t v[4] = {0x09302F30u, 0x09302F31u, 0x09312F30u, 0x09312F31u};
    uint8_t* p = text + row * row_bytes + mis + 16ull * (g0 + lane);
    for (uint32_t g = g0; g + 64u <= g1; g += 64u, p += 1024) {
        v[0] += lane;
        __builtin_memcpy(p, v, 16);
    }
    if (pk_mode) {
        uint8_t* q = packed + row * (pk_mode == 2 ? 5120ull : 5000ull) + g0;
        const uint32_t n = g1 - g0;
        for (uint32_t o = 16u * lane; o + 16u <= n; o += 1024u) __builtin_memcpy(q + o, v, 16);
    }
}

// The text stream through a per-wave LDS ring: every trip's 1 KiB goes into the ring at its byte phase (unaligned
// ds_write_b128), and leaves as the 128-byte-aligned KiB that is complete by then (aligned ds_read_b128 + aligned
// global store).  The first and the last trip of a segment are also stored directly (unaligned), which covers the head
// before the first aligned line and the tail after the last one.  Same bytes as k_text_align with the same arguments.
__global__ __launch_bounds__(256) void k_text_ring(uint8_t* text, uint8_t* packed, uint64_t row_bytes, uint32_t mis, uint64_t n_rows,
                                                   uint32_t nseg, uint32_t seg_groups, uint32_t G, uint32_t pk_mode)
{
    __shared__ __attribute__((aligned(128))) uint8_t ring_all[4][2048 + 256];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t row = item / nseg;
    const uint32_t seg = (uint32_t)(item % nseg);
    if (row >= n_rows) return;
    uint8_t* ring = ring_all[threadIdx.x >> 6];
    const uint32_t g0 = seg * seg_groups, g1 = g0 + seg_groups < G ? g0 + seg_groups : G;
    u32x4_a1 v = {0x09302F30u, 0x09302F31u, 0x09312F30u, 0x09312F31u};
    uint8_t* cseg = text + row * row_bytes + mis + 16ull * g0;         // first byte of the segment's cells
    const uint32_t r = (uint32_t)((uintptr_t)cseg & 127u);              // its phase in a 128-byte line
    const uint32_t a = (128u - r) & 127u;                               // stream offset of the first aligned line
    uint8_t* p = cseg + 16u * lane;
    uint8_t* wdst = cseg + a + 16u * lane;                              // aligned window 1 starts at stream offset a
    const uint32_t ntrips = (g1 - g0) / 64u;
    for (uint32_t k = 0; k < ntrips; ++k, p += 1024) {
        v.x += lane;
        // stream offset t of this lane's 16 bytes = 1024 k + 16 lane; ring offset = t mod 2048 (+ a copy of the first
        // `a` bytes of the even slots behind the ring, so that windows that wrap read straight on)
        const uint32_t t = (k & 1u) * 1024u + 16u * lane;
        *reinterpret_cast<u32x4_a1*>(ring + t) = v;
        if (!(k & 1u) && lane < 8u) *reinterpret_cast<u32x4_a1*>(ring + 2048u + 16u * lane) = v;
        if (k == 0 || k + 1 == ntrips) __builtin_memcpy(p, &v, 16);      // head / tail: direct, unaligned
        if (k >= 1) {
            // window k = stream offsets [a + 1024 (k-1), a + 1024 k)
            const uint32_t o = ((k - 1u) & 1u) * 1024u + a + 16u * lane;
            const u32x4_a1 w = *reinterpret_cast<const u32x4_a1*>(ring + o);
            __builtin_memcpy(wdst, &w, 16);
            wdst += 1024;
        }
    }
    if (pk_mode) {
        uint8_t* q = packed + row * (pk_mode == 2 ? 5120ull : 5000ull) + g0;
        const uint32_t n = g1 - g0;
        for (uint32_t o = 16u * lane; o + 16u <= n; o += 1024u) __builtin_memcpy(q + o, &v, 16);
    }
}

// The same with every LDS access aligned: a trip's 1 KiB goes into the ring at its natural place (aligned ds_write_b128); the
// aligned output window is shifted by a = 16 qa + 4 da + ba bytes against it: each lane reads the 5 dwords around its 16
// bytes (dword-aligned reads) and funnel-shifts them by ba bytes (4 x v_alignbyte_b32).
__global__ __launch_bounds__(256) void k_text_ring2(uint8_t* text, uint8_t* packed, uint64_t row_bytes, uint32_t mis, uint64_t n_rows,
                                                    uint32_t nseg, uint32_t seg_groups, uint32_t G, uint32_t pk_mode)
{
    __shared__ __attribute__((aligned(128))) uint8_t ring_all[4][2048 + 128 + 128];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t row = item / nseg;
    const uint32_t seg = (uint32_t)(item % nseg);
    if (row >= n_rows) return;
    uint8_t* ring = ring_all[threadIdx.x >> 6];
    const uint32_t g0 = seg * seg_groups, g1 = g0 + seg_groups < G ? g0 + seg_groups : G;
    u32x4_a1 v = {0x09302F30u, 0x09302F31u, 0x09312F30u, 0x09312F31u};
    uint8_t* cseg = text + row * row_bytes + mis + 16ull * g0;
    const uint32_t r = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)cseg & 127u));
    const uint32_t a = (128u - r) & 127u;
    const uint32_t ba = a & 3u;
    uint8_t* p = cseg + 16u * lane;
    uint8_t* wdst = cseg + a + 16u * lane;
    const uint32_t ntrips = (g1 - g0) / 64u;
    const uint32_t mirror = lane < 8u ? 2048u + 16u * lane : 2048u + 128u + 16u * (lane & 7u); // lanes >= 8: a dump area
    const uint32_t rd_e = (a & ~3u) + 16u * lane, rd_o = 1024u + rd_e; // window after an odd / even trip starts in the even / odd slot
    for (uint32_t k = 0; k < ntrips; ++k, p += 1024) {
        v.x += lane;
        *reinterpret_cast<u32x4*>(ring + (k & 1u) * 1024u + 16u * lane) = v;
        if (!(k & 1u)) *reinterpret_cast<u32x4*>(ring + mirror) = v;
        if (k == 0 || k + 1 == ntrips) __builtin_memcpy(p, &v, 16);
        if (k >= 1) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(ring + ((k & 1u) ? rd_e : rd_o));
            const uint32_t d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3], d4 = src[4];
            u32x4 w;
            w.x = __builtin_amdgcn_alignbyte(d1, d0, ba);
            w.y = __builtin_amdgcn_alignbyte(d2, d1, ba);
            w.z = __builtin_amdgcn_alignbyte(d3, d2, ba);
            w.w = __builtin_amdgcn_alignbyte(d4, d3, ba);
            *reinterpret_cast<u32x4*>(wdst) = w;
            wdst += 1024;
        }
    }
    if (pk_mode) {
        uint8_t* q = packed + row * (pk_mode == 2 ? 5120ull : 5000ull) + g0;
        const uint32_t n = g1 - g0;
        for (uint32_t o = 16u * lane; o + 16u <= n; o += 1024u) __builtin_memcpy(q + o, &v, 16);
    }
}

// ring2 with a minimal head and tail (no whole trip is stored twice).  Every LDS access aligned: a trip's 1 KiB goes into the ring at its natural place (aligned ds_write_b128); the
// aligned output window is shifted by a = 16 qa + 4 da + ba bytes against it: each lane reads the 5 dwords around its 16
// bytes (dword-aligned reads) and funnel-shifts them by ba bytes (4 x v_alignbyte_b32).
__global__ __launch_bounds__(256) void k_text_ring3(uint8_t* text, uint8_t* packed, uint64_t row_bytes, uint32_t mis, uint64_t n_rows,
                                                    uint32_t nseg, uint32_t seg_groups, uint32_t G, uint32_t pk_mode)
{
    __shared__ __attribute__((aligned(128))) uint8_t ring_all[4][2048 + 128 + 128];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t row = item / nseg;
    const uint32_t seg = (uint32_t)(item % nseg);
    if (row >= n_rows) return;
    uint8_t* ring = ring_all[threadIdx.x >> 6];
    const uint32_t g0 = seg * seg_groups, g1 = g0 + seg_groups < G ? g0 + seg_groups : G;
    u32x4_a1 v = {0x09302F30u, 0x09302F31u, 0x09312F30u, 0x09312F31u};
    uint8_t* cseg = text + row * row_bytes + mis + 16ull * g0;
    const uint32_t r = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)cseg & 127u));
    const uint32_t a = (128u - r) & 127u;
    const uint32_t ba = a & 3u;
    uint8_t* p = cseg + 16u * lane;
    uint8_t* wdst = cseg + a + 16u * lane;
    const uint32_t ntrips = (g1 - g0) / 64u;
    const uint32_t mirror = lane < 8u ? 2048u + 16u * lane : 2048u + 128u + 16u * (lane & 7u); // lanes >= 8: a dump area
    const uint32_t rd_e = (a & ~3u) + 16u * lane, rd_o = 1024u + rd_e; // window after an odd / even trip starts in the even / odd slot
    for (uint32_t k = 0; k < ntrips; ++k, p += 1024) {
        v.x += lane;
        *reinterpret_cast<u32x4*>(ring + (k & 1u) * 1024u + 16u * lane) = v;
        if (!(k & 1u)) *reinterpret_cast<u32x4*>(ring + mirror) = v;
        if (k == 0 && 16u * lane < a) __builtin_memcpy(p, &v, 16);           // head: the bytes before the first aligned line
        if (k >= 1) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(ring + ((k & 1u) ? rd_e : rd_o));
            const uint32_t d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3], d4 = src[4];
            u32x4 w;
            w.x = __builtin_amdgcn_alignbyte(d1, d0, ba);
            w.y = __builtin_amdgcn_alignbyte(d2, d1, ba);
            w.z = __builtin_amdgcn_alignbyte(d3, d2, ba);
            w.w = __builtin_amdgcn_alignbyte(d4, d3, ba);
            *reinterpret_cast<u32x4*>(wdst) = w;
            wdst += 1024;
        }
    }
    if (ntrips) {
        // what is left in the ring: stream offsets [a + 1024 (K-1), 1024 K) -- whole aligned 16-byte pieces, then the last < 16 bytes
        if (16u * lane + 16u + a <= 1024u) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(ring + (((ntrips - 1u) & 1u) ? rd_o : rd_e));
            const uint32_t d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3], d4 = src[4];
            u32x4 w;
            w.x = __builtin_amdgcn_alignbyte(d1, d0, ba);
            w.y = __builtin_amdgcn_alignbyte(d2, d1, ba);
            w.z = __builtin_amdgcn_alignbyte(d3, d2, ba);
            w.w = __builtin_amdgcn_alignbyte(d4, d3, ba);
            *reinterpret_cast<u32x4*>(wdst) = w;
        }
        if ((a & 15u) && lane == 63u) __builtin_memcpy(p - 1024, &v, 16);
    }
    if (pk_mode) {
        uint8_t* q = packed + row * (pk_mode == 2 ? 5120ull : 5000ull) + g0;
        const uint32_t n = g1 - g0;
        for (uint32_t o = 16u * lane; o + 16u <= n; o += 1024u) __builtin_memcpy(q + o, &v, 16);
    }
}

// Who writes the lines that two rows share?  One wave per row, rows of row_bytes (any value: every row has its own phase).
//  MODE 0: wave j stores the bytes of row j, [R_j, R_j+1), in 16-byte pieces at the row's own phase (the row kernel today)
//  MODE 1: wave j stores the LINES [ceil128(R_j), ceil128(R_j+1)) in aligned 16-byte pieces
//  MODE 2: the same range as MODE 1 but in 16-byte pieces at the row's phase (R_j mod 16), the < 16 bytes at either end by byte stores
//  SLEEP: s_sleep between a wave's stores, so that they are as far apart in time as the row kernel's (one trip = 1.4 us)
template <int MODE, int SLEEP = 0>
__global__ __launch_bounds__(256) void k_text_own(uint8_t* text, uint64_t row_bytes, uint64_t n_rows)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const uint64_t r0 = row * row_bytes, r1 = r0 + row_bytes;
    uint64_t b = r0, e = r1;
    if (MODE) { b = (r0 + 127) & ~127ull; e = (r1 + 127) & ~127ull; }
    u32x4_a1 v = {0x09302F30u, 0x09302F31u, 0x09312F30u, 0x09312F31u};
    uint64_t s = b;
    if (MODE == 2) {
        s = b + ((r0 - b) & 15u);                       // first piece at the row's phase at or after b
        if (lane < (uint32_t)(s - b)) text[b + lane] = (uint8_t)v.x;
    }
    uint64_t o = s + 16ull * lane;
    for (; o + 16 <= e; o += 1024) {
        v.x += lane;
        __builtin_memcpy(text + o, &v, 16);
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
    }
    // the last < 16 bytes: the lane whose piece would cross e stores them byte by byte
    if (o < e) for (uint64_t q = o; q < e; ++q) text[q] = (uint8_t)v.y;
}

// k_text_own<2> (a wave per row, owned lines) with the text realigned through a per-wave LDS ring: every trip's KiB is
// written to the ring at its natural, 16-byte aligned place and read back SHIFTED (one byte-unaligned ds_read_b128 per lane),
// so that every global store instruction covers 8 whole 128-byte lines.  No vector ALU work for the shift.
template <int SLEEP>
__global__ __launch_bounds__(256) void k_text_own_ring(uint8_t* text, uint64_t row_bytes, uint64_t n_rows)
{
    __shared__ __attribute__((aligned(128))) uint8_t ring_all[4][2048 + 128 + 128];
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    uint8_t* ring = ring_all[threadIdx.x >> 6];
    const uint64_t r0 = row * row_bytes, r1 = r0 + row_bytes;
    const uint64_t b = (r0 + 127) & ~127ull, e = (r1 + 127) & ~127ull;   // the lines this wave owns
    u32x4_a1 v = {0x09302F30u, 0x09302F31u, 0x09312F30u, 0x09312F31u};
    // the wave's data stream starts at s (first piece at the row's phase at or after b), as in k_text_own<2>
    const uint64_t s = b + ((r0 - b) & 15u);
    if (lane < (uint32_t)(s - b)) text[b + lane] = (uint8_t)v.x;
    // stream offset t: trip k, lane i holds bytes [1024 k + 16 i, +16); output windows are aligned: window k covers stream
    // offsets [a + 1024 (k-1), a + 1024 k) with a = (128 - (s & 127)) & 127 -- but here s - b < 16, so the first line is b itself:
    const uint32_t a = (uint32_t)((128u - (uint32_t)(s & 127u)) & 127u);
    const uint32_t mirror = lane < 8u ? 2048u + 16u * lane : 2048u + 128u + 16u * (lane & 7u);
    const uint32_t rd_e = a + 16u * lane, rd_o = 1024u + rd_e;
    uint64_t o = s + 16ull * lane;          // address of this lane's natural piece
    uint64_t wdst = s + a + 16ull * lane;   // address of this lane's piece of the next aligned window
    uint32_t k = 0;
    for (; o + 16 <= e + 1024; o += 1024, ++k) { // one trip beyond the data flushes the ring's last window
        const bool have = o + 16 <= e;
        v.x += lane;
        if (have) *reinterpret_cast<u32x4*>(ring + (k & 1u) * 1024u + 16u * lane) = v;
        if (have && !(k & 1u)) *reinterpret_cast<u32x4*>(ring + mirror) = v;
        if (k == 0) { if (16u * lane < a) __builtin_memcpy(text + o, &v, 16); } // head: the bytes before the first aligned line
        else if (wdst + 16 <= e) {
            const u32x4_a1 w = *reinterpret_cast<const u32x4_a1*>(ring + ((k & 1u) ? rd_e : rd_o));
            *reinterpret_cast<u32x4*>(text + wdst) = w;
            wdst += 1024;
        }
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
    }
}

static GenParams P;
static hipStream_t stream;
static hipEvent_t ev0, ev1;
static int n_dispatch = 0;
static int reps = 3;

static float launch_once(uint32_t what)
{
    const uint64_t items = P.n_rows * P.nseg, blocks = (items + 3) / 4;
    CK(hipEventRecord(ev0, stream));
    if (what == 3) hipLaunchKernelGGL((dnaf::rowgen_kernel<true, true>), dim3((uint32_t)blocks), dim3(256), 0, stream, P);
    else if (what == 1) hipLaunchKernelGGL((dnaf::rowgen_kernel<true, false>), dim3((uint32_t)blocks), dim3(256), 0, stream, P);
    else hipLaunchKernelGGL((dnaf::rowgen_kernel<false, true>), dim3((uint32_t)blocks), dim3(256), 0, stream, P);
    CK(hipGetLastError());
    CK(hipEventRecord(ev1, stream));
    CK(hipEventSynchronize(ev1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, ev0, ev1));
    ++n_dispatch;
    return ms;
}

// median of `reps` launches after one warm-up; prints every launch
static float timed(const char* tag, uint8_t* text, uint8_t* packed, uint64_t pitch, uint32_t what = 3)
{
    P.text = text; P.packed = packed; P.packed_pitch = pitch;
    std::vector<float> v;
    for (int i = 0; i <= reps; ++i) {
        const int d = n_dispatch;
        const float ms = launch_once(what);
        printf("D%d %s packed=%p pitch=%llu what=%u %s %.3f ms\n", d, tag, (void*)packed, (unsigned long long)pitch, what, i ? "timed" : "warm", ms);
        if (i) v.push_back(ms);
    }
    std::sort(v.begin(), v.end());
    const float med = v[v.size() / 2];
    printf("== %s packed=%p pitch=%llu what=%u median %.3f ms\n", tag, (void*)packed, (unsigned long long)pitch, what, med);
    fflush(stdout);
    return med;
}

int main(int argc, char** argv)
{
    const uint32_t S = 20000, G = (S + 3) / 4, W = (S + 15) / 16;
    uint64_t N = 250000;
    int K = 8;
    std::string mode = "all";
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--rows") && i + 1 < argc) N = strtoull(argv[++i], nullptr, 10);
        else if (!strcmp(argv[i], "--k") && i + 1 < argc) K = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--reps") && i + 1 < argc) reps = atoi(argv[++i]);
        else mode = argv[i];
    }
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&stream));
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));

    // synthetic table: autosome rows, MAF bins 0.01 .. 0.495, prefixes of 30..41 bytes (rows are byte-aligned only)
    std::vector<RowRec> recs(N);
    std::vector<uint8_t> prefix;
    uint64_t off = 0;
    uint32_t lcg = 12345;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return lcg >> 8; };
    for (uint64_t j = 0; j < N; ++j) {
        RowRec& r = recs[j];
        const double maf = 0.005 * (2 + rnd() % 98);
        r.thr[0] = (uint32_t)((1.0 - maf) * 4294967296.0);
        r.thr[1] = r.thr[2] = 0xFFFFFFFFu;
        r.prefix_len = (uint16_t)(30 + rnd() % 12);
        r.prefix_off = (uint32_t)prefix.size();
        for (uint32_t k = 0; k < r.prefix_len; ++k) prefix.push_back(k + 1 == r.prefix_len ? '\t' : (uint8_t)('0' + k % 10));
        r.text_off = off;
        r.slot = -1;
        r.cls = 0;
        r.n_alleles = 2;
        off += r.prefix_len + 4ull * S;
    }
    const uint64_t text_bytes = off;
    std::vector<uint8_t> male(G);
    for (auto& m : male) m = (uint8_t)(rnd() & 15u);
    std::vector<uint32_t> xoff(G + 1, 0);
    RowRec* d_recs; uint8_t *d_prefix, *d_male; uint32_t* d_xoff;
    CK(hipMalloc((void**)&d_recs, N * sizeof(RowRec)));
    CK(hipMalloc((void**)&d_prefix, prefix.size()));
    CK(hipMalloc((void**)&d_male, G));
    CK(hipMalloc((void**)&d_xoff, (G + 1) * 4));
    CK(hipMemcpy(d_recs, recs.data(), N * sizeof(RowRec), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_prefix, prefix.data(), prefix.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_male, male.data(), G, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_xoff, xoff.data(), (G + 1) * 4, hipMemcpyHostToDevice));
    memset(&P, 0, sizeof P);
    P.recs = d_recs; P.prefix = d_prefix; P.male_nib = d_male; P.xoff = d_xoff; P.forced = nullptr;
    P.text_base = 0; P.row_begin = 0; P.n_rows = N; P.rng_row0 = 0; P.key0 = 12345; P.key1 = 0;
    P.S = S; P.G = G; P.W = W;
    P.nseg = (G + 64u * 32u - 1) / (64u * 32u);
    P.seg_groups = ((G + P.nseg - 1) / P.nseg + 63u) / 64u * 64u;
    const uint64_t dense = 4ull * W, max_pitch = 8192;
    const uint64_t pbytes = N * max_pitch + 4096; // room for every pitch tried
    printf("rows %llu, text %.3f GB, packed (dense) %.3f GB, nseg %u seg_groups %u\n", (unsigned long long)N, text_bytes / 1e9,
           N * dense / 1e9, P.nseg, P.seg_groups);

    uint8_t* d_text;
    CK(hipMalloc((void**)&d_text, text_bytes + 64));
    printf("text=%p\n", (void*)d_text);
    CK(hipMemsetAsync(d_text, 0, text_bytes, stream));

    const bool all = mode == "all";
    std::vector<uint8_t*> bufs;
    std::vector<float> tms;
    if (all || mode == "lottery" || mode == "pitch" || mode == "split") {
        // K separate allocations of the matrix, all alive at once: K different placements
        for (int k = 0; k < K; ++k) {
            uint8_t* p;
            CK(hipMalloc((void**)&p, pbytes));
            CK(hipMemsetAsync(p, 0, pbytes, stream));
            bufs.push_back(p);
        }
        timed("warmup", d_text, bufs[0], dense);
        timed("warmup", d_text, bufs[0], dense);
        for (int pass = 0; pass < 2; ++pass)
            for (int k = 0; k < K; ++k) {
                char tag[64];
                snprintf(tag, sizeof tag, "lottery pass%d buf%d", pass, k);
                const float ms = timed(tag, d_text, bufs[k], dense);
                if (pass == 1) tms.push_back(ms);
            }
        const int slow = (int)(std::max_element(tms.begin(), tms.end()) - tms.begin());
        const int fast = (int)(std::min_element(tms.begin(), tms.end()) - tms.begin());
        printf("slowest buf%d %.3f ms, fastest buf%d %.3f ms\n", slow, tms[slow], fast, tms[fast]);
        if (all || mode == "pitch") {
            const uint64_t pitches[] = {5000, 5008, 5056, 5120, 5184, 5248, 6144, 8192, 5000};
            for (int which : {slow, fast})
                for (uint64_t pitch : pitches) {
                    char tag[64];
                    snprintf(tag, sizeof tag, "pitch %s buf%d", which == slow ? "slow" : "fast", which);
                    timed(tag, d_text, bufs[which], pitch);
                }
            // the dense matrix at byte offsets inside the slow / fast allocation
            const uint64_t shifts[] = {64, 128, 256, 1024, 4096, 65536, 1ull << 20, 2ull << 20, 64ull << 20, 0};
            for (int which : {slow, fast})
                for (uint64_t sh : shifts) {
                    char tag[64];
                    snprintf(tag, sizeof tag, "shift%llu %s buf%d", (unsigned long long)sh, which == slow ? "slow" : "fast", which);
                    timed(tag, d_text, bufs[which] + sh, dense);
                }
        }
        if (all || mode == "split") {
            for (int which : {slow, fast}) {
                char tag[64];
                snprintf(tag, sizeof tag, "split %s buf%d", which == slow ? "slow" : "fast", which);
                timed(tag, d_text, bufs[which], dense, 1);
                timed(tag, d_text, bufs[which], dense, 2);
                timed(tag, d_text, bufs[which], dense, 3);
            }
        }
        for (uint8_t* p : bufs) CK(hipFree(p));
        bufs.clear();
    }
    if (all || mode == "arena") {
        // one allocation, the dense matrix at offsets k * step inside it
        const uint64_t step = ((N * dense + (2ull << 20) - 1) >> 21) << 21; // next multiple of 2 MiB
        uint8_t* arena;
        CK(hipMalloc((void**)&arena, step * (uint64_t)K + (64ull << 20)));
        CK(hipMemsetAsync(arena, 0, step * (uint64_t)K + (64ull << 20), stream));
        printf("arena=%p step=%llu\n", (void*)arena, (unsigned long long)step);
        timed("warmup", d_text, arena, dense);
        for (int pass = 0; pass < 2; ++pass)
            for (int k = 0; k < K; ++k) {
                char tag[64];
                snprintf(tag, sizeof tag, "arena pass%d slot%d", pass, k);
                timed(tag, d_text, arena + step * (uint64_t)k, dense);
            }
        CK(hipFree(arena));
    }
    if (mode == "map") {
        // per placement: the row kernel, the bare two-stream store pattern, and the plain store rate of the region
        for (int k = 0; k < K; ++k) {
            uint8_t* p;
            CK(hipMalloc((void**)&p, pbytes));
            CK(hipMemsetAsync(p, 0, pbytes, stream));
            bufs.push_back(p);
        }
        timed("warmup", d_text, bufs[0], dense);
        timed("warmup", d_text, bufs[0], dense);
        auto ev_time = [&](auto&& launch) {
            std::vector<float> v;
            for (int i = 0; i <= reps; ++i) {
                CK(hipEventRecord(ev0, stream));
                launch();
                CK(hipGetLastError());
                CK(hipEventRecord(ev1, stream));
                CK(hipEventSynchronize(ev1));
                float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
                if (i) v.push_back(ms);
            }
            std::sort(v.begin(), v.end());
            return v[v.size() / 2];
        };
        const uint64_t row_bytes = 4ull * S + 30; // the shortest row: N of them fit the text buffer
        const uint64_t items = N * P.nseg, blocks = (items + 3) / 4;
        for (int pass = 0; pass < 2; ++pass)
            for (int k = 0; k < K; ++k) {
                char tag[64];
                snprintf(tag, sizeof tag, "map pass%d buf%d", pass, k);
                const float t_row = timed(tag, d_text, bufs[k], dense);
                const float t_two = ev_time([&] { hipLaunchKernelGGL(k_two_streams, dim3((uint32_t)blocks), dim3(256), 0, stream, d_text, bufs[k], row_bytes, 30u, dense, N, P.nseg, P.seg_groups, G, 3u); });
                const float t_txt = ev_time([&] { hipLaunchKernelGGL(k_two_streams, dim3((uint32_t)blocks), dim3(256), 0, stream, d_text, bufs[k], row_bytes, 30u, dense, N, P.nseg, P.seg_groups, G, 1u); });
                const float t_pk = ev_time([&] { hipLaunchKernelGGL(k_two_streams, dim3((uint32_t)blocks), dim3(256), 0, stream, d_text, bufs[k], row_bytes, 30u, dense, N, P.nseg, P.seg_groups, G, 2u); });
                const uint64_t fill = N * dense / 16;
                const float t_fill = ev_time([&] { hipLaunchKernelGGL(k_fill, dim3(256 * 8), dim3(256), 0, stream, (uint4*)bufs[k], fill); });
                printf("MAP pass%d buf%d %p rowgen %.3f two_streams %.3f text_only %.3f packed_only %.3f fill %.3f ms (%.0f GB/s)\n", pass, k, (void*)bufs[k],
                       t_row, t_two, t_txt, t_pk, t_fill, fill * 16 / t_fill / 1e6);
                fflush(stdout);
            }
        // the plain store rate of the text buffer, 2 GB at a time
        for (uint64_t o = 0; o + (2ull << 30) <= text_bytes; o += 2ull << 30) {
            const float t = ev_time([&] { hipLaunchKernelGGL(k_fill, dim3(256 * 8), dim3(256), 0, stream, (uint4*)(d_text + o), (2ull << 30) / 16); });
            printf("TEXTFILL +%lluGB %.3f ms (%.0f GB/s)\n", (unsigned long long)(o >> 30), t, (2ull << 30) / t / 1e6);
        }
        for (uint8_t* p : bufs) CK(hipFree(p));
        CK(hipFree(d_text));
        return 0;
    }
    if (mode == "variants") {
        for (int k = 0; k < K; ++k) {
            uint8_t* p;
            CK(hipMalloc((void**)&p, pbytes));
            CK(hipMemsetAsync(p, 0, pbytes, stream));
            bufs.push_back(p);
        }
        timed("warmup", d_text, bufs[0], dense);
        timed("warmup", d_text, bufs[0], dense);
        auto ev_time = [&](auto&& launch) {
            std::vector<float> v;
            for (int i = 0; i <= reps; ++i) {
                CK(hipEventRecord(ev0, stream));
                launch();
                CK(hipGetLastError());
                CK(hipEventRecord(ev1, stream));
                CK(hipEventSynchronize(ev1));
                float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
                if (i) v.push_back(ms);
            }
            std::sort(v.begin(), v.end());
            return v[v.size() / 2];
        };
        const uint64_t row_bytes = 4ull * S + 30; // the shortest row: N of them fit the text buffer
        const uint64_t items = N * P.nseg, blocks = (items + 3) / 4;
        const dim3 grid((uint32_t)blocks), block(256);
        for (int k = 0; k < K; ++k) {
            char tag[64];
            snprintf(tag, sizeof tag, "var buf%d", k);
            const float t_row = timed(tag, d_text, bufs[k], dense);
            float t[8];
#define TV(V) t[V] = ev_time([&] { hipLaunchKernelGGL(k_two_var<V>, grid, block, 0, stream, d_text, bufs[k], row_bytes, 30u, dense, N, P.nseg, P.seg_groups, G); });
            TV(0) TV(1) TV(2) TV(3) TV(4) TV(5) TV(6) TV(7)
            const float t5120 = ev_time([&] { hipLaunchKernelGGL(k_two_var<0>, grid, block, 0, stream, d_text, bufs[k], row_bytes, 30u, (uint64_t)5120, N, P.nseg, P.seg_groups, G); });
            const float t8192 = ev_time([&] { hipLaunchKernelGGL(k_two_var<0>, grid, block, 0, stream, d_text, bufs[k], row_bytes, 30u, (uint64_t)8192, N, P.nseg, P.seg_groups, G); });
            printf("VAR buf%d %p rowgen %.3f | v0 %.3f before %.3f nt %.3f sc1 %.3f lastwave %.3f dword %.3f spread %.3f firstwave %.3f | pitch5120 %.3f pitch8192 %.3f\n", k,
                   (void*)bufs[k], t_row, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t5120, t8192);
            fflush(stdout);
        }
        for (uint8_t* p : bufs) CK(hipFree(p));
        CK(hipFree(d_text));
        return 0;
    }
    if (mode == "align") {
        for (int k = 0; k < K; ++k) {
            uint8_t* p;
            CK(hipMalloc((void**)&p, pbytes));
            CK(hipMemsetAsync(p, 0, pbytes, stream));
            bufs.push_back(p);
        }
        timed("warmup", d_text, bufs[0], dense);
        timed("warmup", d_text, bufs[0], dense);
        auto ev_time = [&](auto&& launch) {
            std::vector<float> v;
            for (int i = 0; i <= reps; ++i) {
                CK(hipEventRecord(ev0, stream));
                launch();
                CK(hipGetLastError());
                CK(hipEventRecord(ev1, stream));
                CK(hipEventSynchronize(ev1));
                float ms; CK(hipEventElapsedTime(&ms, ev0, ev1));
                if (i) v.push_back(ms);
            }
            std::sort(v.begin(), v.end());
            return v[v.size() / 2];
        };
        // rows of 80,000 cell bytes in a pitch of 79,872 + 256 = 80,128 (a multiple of 128); 249,000 rows fit the text buffer
        const uint64_t row_bytes = 80128, NR = 249000;
        if (NR * row_bytes + 256 > text_bytes) { printf("text buffer too small\n"); return 1; }
        {   // the ring kernel must write the very bytes of the direct kernel
            const uint32_t sg = 1728, ns = (G + sg - 1) / sg;
            const dim3 grid((uint32_t)((NR * ns + 3) / 4)), block(256);
            const size_t nchk = 256u << 20;
            std::vector<uint8_t> h0(nchk), h1(nchk);
            for (uint32_t mis : {37u, 0u, 100u, 3u, 126u, 16u, 112u, 127u, 1u}) {
                CK(hipMemsetAsync(d_text, 0, text_bytes, stream));
                hipLaunchKernelGGL(k_text_align, grid, block, 0, stream, d_text, bufs[0], row_bytes, mis, NR, ns, sg, G, 0u);
                CK(hipMemcpyAsync(h0.data(), d_text + (8ull << 30), nchk, hipMemcpyDeviceToHost, stream));
                CK(hipMemsetAsync(d_text, 0, text_bytes, stream));
                hipLaunchKernelGGL(k_text_ring, grid, block, 0, stream, d_text, bufs[0], row_bytes, mis, NR, ns, sg, G, 0u);
                CK(hipMemcpyAsync(h1.data(), d_text + (8ull << 30), nchk, hipMemcpyDeviceToHost, stream));
                CK(hipStreamSynchronize(stream));
                size_t bad = 0, first = 0;
                for (size_t i = 0; i < nchk; ++i) if (h0[i] != h1[i]) { if (!bad) first = i; ++bad; }
                printf("CHECK ring vs direct, mis %u: %zu differing bytes of %zu (first at %zu)\n", mis, bad, nchk, first);
                CK(hipMemsetAsync(d_text, 0, text_bytes, stream));
                hipLaunchKernelGGL(k_text_ring2, grid, block, 0, stream, d_text, bufs[0], row_bytes, mis, NR, ns, sg, G, 0u);
                CK(hipMemcpyAsync(h1.data(), d_text + (8ull << 30), nchk, hipMemcpyDeviceToHost, stream));
                CK(hipStreamSynchronize(stream));
                bad = 0;
                for (size_t i = 0; i < nchk; ++i) if (h0[i] != h1[i]) { if (!bad) first = i; ++bad; }
                printf("CHECK ring2 vs direct, mis %u: %zu differing bytes of %zu (first at %zu)\n", mis, bad, nchk, first);
                CK(hipMemsetAsync(d_text, 0, text_bytes, stream));
                hipLaunchKernelGGL(k_text_ring3, grid, block, 0, stream, d_text, bufs[0], row_bytes, mis, NR, ns, sg, G, 0u);
                CK(hipMemcpyAsync(h1.data(), d_text + (8ull << 30), nchk, hipMemcpyDeviceToHost, stream));
                CK(hipStreamSynchronize(stream));
                bad = 0;
                for (size_t i = 0; i < nchk; ++i) if (h0[i] != h1[i]) { if (!bad) first = i; ++bad; }
                printf("CHECK ring3 vs direct, mis %u: %zu differing bytes of %zu (first at %zu)\n", mis, bad, nchk, first);
            }
        }
        for (int rep = 0; rep < 3; ++rep) {
            // 80,037-byte rows, one wave each: who owns the shared lines
            const uint64_t rb = 80037, nr = 249900;
            const dim3 grid((uint32_t)((nr + 3) / 4)), block(256);
            const float t0 = ev_time([&] { hipLaunchKernelGGL(k_text_own<0>, grid, block, 0, stream, d_text, rb, nr); });
            const float t1 = ev_time([&] { hipLaunchKernelGGL(k_text_own<1>, grid, block, 0, stream, d_text, rb, nr); });
            const float t2 = ev_time([&] { hipLaunchKernelGGL(k_text_own<2>, grid, block, 0, stream, d_text, rb, nr); });
            const uint64_t rb2 = 80128;
            const float t3 = ev_time([&] { hipLaunchKernelGGL(k_text_own<0>, dim3((uint32_t)((249000 + 3) / 4)), block, 0, stream, d_text, rb2, (uint64_t)249000); });
            printf("OWN rows of 80037 B, a wave per row: row bytes at row phase %.3f | owned lines, aligned pieces %.3f | owned lines, row-phase pieces %.3f | rows of 80128 B %.3f ms\n", t0, t1, t2, t3);
            const float s0 = ev_time([&] { hipLaunchKernelGGL((k_text_own<0, 40>), grid, block, 0, stream, d_text, rb, nr); });
            const float s1 = ev_time([&] { hipLaunchKernelGGL((k_text_own<1, 40>), grid, block, 0, stream, d_text, rb, nr); });
            const float s2 = ev_time([&] { hipLaunchKernelGGL((k_text_own<2, 40>), grid, block, 0, stream, d_text, rb, nr); });
            const float u1 = ev_time([&] { hipLaunchKernelGGL((k_text_own<1, 20>), grid, block, 0, stream, d_text, rb, nr); });
            const float u2 = ev_time([&] { hipLaunchKernelGGL((k_text_own<2, 20>), grid, block, 0, stream, d_text, rb, nr); });
            printf("OWN with s_sleep 40 between trips: row phase %.3f | aligned pieces %.3f | owned lines row-phase pieces %.3f ;  s_sleep 20: aligned %.3f row-phase %.3f ms\n", s0, s1, s2, u1, u2);
            const float g0 = ev_time([&] { hipLaunchKernelGGL((k_text_own_ring<0>), grid, block, 0, stream, d_text, rb, nr); });
            const float g20 = ev_time([&] { hipLaunchKernelGGL((k_text_own_ring<20>), grid, block, 0, stream, d_text, rb, nr); });
            const float g40 = ev_time([&] { hipLaunchKernelGGL((k_text_own_ring<40>), grid, block, 0, stream, d_text, rb, nr); });
            printf("OWN ring (aligned lines through LDS, unaligned ds_read_b128): no sleep %.3f | s_sleep 20 %.3f | s_sleep 40 %.3f ms\n", g0, g20, g40);
            if (rep == 0) {
                const size_t nchk = 128u << 20;
                std::vector<uint8_t> h0(nchk), h1(nchk);
                CK(hipMemsetAsync(d_text, 0, text_bytes, stream));
                hipLaunchKernelGGL((k_text_own<2, 0>), grid, block, 0, stream, d_text, rb, nr);
                CK(hipMemcpyAsync(h0.data(), d_text + (4ull << 30), nchk, hipMemcpyDeviceToHost, stream));
                CK(hipMemsetAsync(d_text, 0, text_bytes, stream));
                hipLaunchKernelGGL((k_text_own_ring<0>), grid, block, 0, stream, d_text, rb, nr);
                CK(hipMemcpyAsync(h1.data(), d_text + (4ull << 30), nchk, hipMemcpyDeviceToHost, stream));
                CK(hipStreamSynchronize(stream));
                size_t bad = 0, first = 0, holes0 = 0, holes1 = 0;
                for (size_t i = 0; i < nchk; ++i) { if (!h0[i]) ++holes0; if (!h1[i]) ++holes1; if ((h0[i] != 0) != (h1[i] != 0)) { if (!bad) first = i; ++bad; } }
                printf("OWN ring coverage check: %zu bytes written by one kernel only (first at %zu); unwritten %zu vs %zu\n", bad, first, holes0, holes1);
            }
        }
        for (int k = 0; k < K; ++k) {
            char tag[64];
            snprintf(tag, sizeof tag, "align buf%d", k);
            const float t_row = timed(tag, d_text, bufs[k], dense);
            printf("ALIGN buf%d rowgen %.3f |", k, t_row);
            // segments of 27 trips (1728 groups, as the row kernel) and of 28 trips (1792 groups = 14 lines of packed bytes)
            for (uint32_t sg : {1728u, 1792u}) {
                const uint32_t ns = (G + sg - 1) / sg;
                const dim3 grid((uint32_t)((NR * ns + 3) / 4)), block(256);
                for (uint32_t pk : {0u, 1u, 2u})
                    for (uint32_t mis : {0u, 16u, 64u, 37u}) {
                        const float t = ev_time([&] { hipLaunchKernelGGL(k_text_align, grid, block, 0, stream, d_text, bufs[k], row_bytes, mis, NR, ns, sg, G, pk); });
                        printf(" sg%u pk%u mis%u %.3f", sg, pk, mis, t);
                        if (mis == 37u || mis == 0u) {
                            const float t2 = ev_time([&] { hipLaunchKernelGGL(k_text_ring, grid, block, 0, stream, d_text, bufs[k], row_bytes, mis, NR, ns, sg, G, pk); });
                            printf(" ring %.3f", t2);
                            const float t3 = ev_time([&] { hipLaunchKernelGGL(k_text_ring2, grid, block, 0, stream, d_text, bufs[k], row_bytes, mis, NR, ns, sg, G, pk); });
                            printf(" ring2 %.3f", t3);
                            const float t4 = ev_time([&] { hipLaunchKernelGGL(k_text_ring3, grid, block, 0, stream, d_text, bufs[k], row_bytes, mis, NR, ns, sg, G, pk); });
                            printf(" ring3 %.3f", t4);
                        }
                    }
                printf(" |");
            }
            printf("\n");
            fflush(stdout);
        }
        for (uint8_t* p : bufs) CK(hipFree(p));
        CK(hipFree(d_text));
        return 0;
    }
    if (mode == "textarena") {
        // both outputs in ONE allocation: text first, the matrix right behind it (2 MiB aligned)
        CK(hipFree(d_text));
        const uint64_t toff = ((text_bytes + (2ull << 20)) >> 21) << 21;
        uint8_t* both;
        CK(hipMalloc((void**)&both, toff + pbytes));
        printf("both=%p packed at +%llu\n", (void*)both, (unsigned long long)toff);
        timed("warmup", both, both + toff, dense);
        for (int i = 0; i < 4; ++i) timed("one-allocation", both, both + toff, dense);
        CK(hipFree(both));
        return 0;
    }
    CK(hipFree(d_text));
    printf("done, %d dispatches\n", n_dispatch);
    return 0;
}
