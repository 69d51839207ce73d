// bgzf_kernels.hip.h -- GPU BGZF (blocked gzip) encoder for the VCF text stream (gfx950).
//
// One 1024-thread workgroup encodes one BGZF member from 61,440 input bytes; 960 of its threads own
// a 64-byte chunk each.  Tokens are literals and matches at distance 4 only: byte i matches when it
// equals byte i-4 of the same block (GT cells repeat with period 4; haploid cells with period 2).
// Runs of matching bytes never cross a thread's chunk, so a chunk's 64 match bits live in one 64-bit
// register and tokenisation is bit arithmetic: runs of >= 3 ones are matches, everything else is a
// literal.  The block is staged in LDS once (coalesced 16-byte global loads) in a padded image that
// makes the per-thread chunk reads bank-conflict free; token walks fetch literal bytes and code-table
// entries four tokens at a time so their LDS latencies overlap.
//
//   bgzf_hist    token histogram over a sample of blocks          -> Huffman tables (host, bgzf_host.h)
//   bgzf_size    exact compressed size of every block             -> exclusive scan -> output offsets
//   bgzf_encode  header + deflate bits (assembled in LDS) + CRC-32 + ISIZE, written at the block's offset
//
// A block whose deflate body would not fit the LDS buffer (or not beat a stored block) is emitted as
// a stored block.  Output is checked by inflating it on the host (tests/test_bgzf.py): any conforming
// gzip reader must return the input bytes and accept CRC-32 / ISIZE.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgzf_host.h"

namespace dnaf {

constexpr uint32_t BGZF_THREADS = 1024;
constexpr uint32_t BGZF_WAVES = BGZF_THREADS / 64;
constexpr uint32_t BGZF_ACTIVE = BGZF_IN / BGZF_CHUNK; // 960 threads own a chunk
constexpr uint32_t BGZF_CHUNK_WORDS = BGZF_CHUNK / 4;  // 16
constexpr uint32_t BGZF_OUT_SMALL = 12288;             // LDS bytes for a member's deflate body: two workgroups per CU
constexpr uint32_t BGZF_OUT_LARGE = 32768;             // ... for the members that compress worse (one workgroup per CU)
constexpr uint32_t BGZF_MEMBER_OVERHEAD = 18 + 8;      // gzip header with BC field + CRC32 + ISIZE
static_assert(BGZF_CHUNK == 64, "a chunk's match bits must fit one 64-bit register");
static_assert(BGZF_ACTIVE * BGZF_CHUNK == BGZF_IN && BGZF_ACTIVE <= BGZF_THREADS, "chunking must tile the block");

struct BgzfParams {
    const uint8_t* text;     // input stream (16-byte aligned base, readable 64 bytes past n_bytes)
    uint64_t n_bytes;        // its length
    uint32_t n_blocks;       // ceil(n_bytes / BGZF_IN)
    uint32_t sample_step;    // bgzf_hist: encode every sample_step-th block
    const BgzfTables* tables;
    uint32_t* hist;          // [256 literals][256 match lengths][1 eob]
    uint32_t* sizes;         // [n_blocks] member size in bytes; bit 31 = stored block, bit 30 = needs the large buffer
    uint16_t* thread_bits;   // [n_blocks][BGZF_THREADS] deflate bits each thread's chunk encodes to (bgzf_size -> bgzf_encode)
    uint32_t* crcs;          // [n_blocks] CRC-32 of each block's input (bgzf_size -> bgzf_encode)
    uint32_t* large_list;    // [0] = number of members that need the large buffer, [1..] their block indices
    const uint64_t* offsets; // [n_blocks+1] exclusive scan of sizes
    uint8_t* out;
    const uint8_t* fast;     // [n_blocks] or null: 1 = the genotype-domain encoder (gtz_kernels.hip.h) made this member
    const uint32_t* todo;    // null: every block; else [0] = count, [1..] = the blocks bgzf_size / bgzf_encode<SMALL> are launched over
};

__device__ __forceinline__ uint32_t block_len(const BgzfParams& p, uint32_t b)
{
    const uint64_t left = p.n_bytes - (uint64_t)b * BGZF_IN;
    return left < BGZF_IN ? (uint32_t)left : BGZF_IN;
}

// LDS image of the block: dword i lives at index i + i/16, i.e. every 64-byte chunk is followed by one
// pad dword.  Thread t's k-th dword is at 17 t + k: bank (17 t + k) mod 32, distinct over 32 lanes.
constexpr uint32_t BGZF_LDS_IN_WORDS = BGZF_IN / 4 + BGZF_ACTIVE + 8;
__device__ __forceinline__ uint32_t lds_word(uint32_t i) { return i + (i >> 4); }

// Copy the block into LDS: coalesced 16-byte global loads (BGZF_IN is a multiple of 16 and so is every
// block's start offset in the stream), dword stores into the padded image.
__device__ __forceinline__ void stage_block(const uint8_t* blk, uint32_t N, uint32_t* in_lds)
{
    const uint4* src = reinterpret_cast<const uint4*>(blk);
    for (uint32_t i = threadIdx.x; i < (N + 15u) / 16u; i += BGZF_THREADS) {
        const uint4 v = src[i];
        const uint32_t w = lds_word(4u * i); // the 4 dwords of a 16-byte group never straddle a chunk
        in_lds[w] = v.x; in_lds[w + 1] = v.y; in_lds[w + 2] = v.z; in_lds[w + 3] = v.w;
    }
}

// The tokens of one chunk as bit sets (bit p = byte p of the chunk).
struct ChunkTokens {
    uint64_t lit;   // bytes emitted as literals
    uint64_t mstart; // first byte of each match
    uint64_t mend;   // last byte of each match (k-th end belongs to k-th start)
};

// Match bits of one thread's chunk (bit p = byte c0+p equals byte c0+p-4) from the staged block, then
// the token sets.  Optionally accumulates the chunk's CRC-32 (standard: init ~0, final xor ~0).
template <bool WITH_CRC>
__device__ __forceinline__ ChunkTokens chunk_tokens(const uint32_t* in_lds, uint32_t tid, uint32_t len,
                                                    const uint32_t (*crc_tab)[256], uint32_t* crc_out)
{
    uint32_t crc = 0xFFFFFFFFu;
    const uint32_t* d = in_lds + tid * (BGZF_CHUNK_WORDS + 1u);
    uint32_t prev = tid ? d[-2] : 0u; // last dword of the previous chunk (d[-1] is that chunk's pad)
    uint64_t M = 0;
#pragma unroll
    for (uint32_t k = 0; k < BGZF_CHUNK_WORDS; ++k) {
        if (4u * k < len) {
            const uint32_t cur = d[k];
            const uint32_t x = cur ^ prev;
            const uint32_t y = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x; // bit 7 of each byte: byte != 0
            const uint32_t e = ~y & 0x80808080u;
            const uint32_t nib = ((e >> 7) & 1u) | ((e >> 14) & 2u) | ((e >> 21) & 4u) | ((e >> 28) & 8u);
            M |= (uint64_t)nib << (4u * k);
            if (WITH_CRC && 4u * k + 4u <= len) {
                const uint32_t v = crc ^ cur;
                crc = crc_tab[3][v & 0xFFu] ^ crc_tab[2][(v >> 8) & 0xFFu] ^ crc_tab[1][(v >> 16) & 0xFFu] ^ crc_tab[0][v >> 24];
            }
            prev = cur;
        }
    }
    if (tid == 0) M &= ~0xFull; // no history before the block's first 4 bytes
    const uint64_t valid = len >= 64u ? ~0ull : ((1ull << len) - 1ull);
    M &= valid; // only the stream's last block has a short chunk
    if (WITH_CRC) {
        if (len & 3u) {
            const uint8_t* bytes = reinterpret_cast<const uint8_t*>(d);
            for (uint32_t i = len & ~3u; i < len; ++i) crc = crc_tab[0][(crc ^ bytes[i]) & 0xFFu] ^ (crc >> 8);
        }
        *crc_out = ~crc;
    }
    // runs of ones shorter than 3 cannot be matches (deflate's minimum match length)
    const uint64_t up = M << 1, dn = M >> 1;
    const uint64_t r1 = M & ~up & ~dn;             // isolated ones
    const uint64_t a = M & dn & ~(M >> 2) & ~up;   // low bit of a run of exactly two
    const uint64_t m3 = M & ~(r1 | a | (a << 1));  // ones in runs of >= 3
    ChunkTokens t;
    t.lit = ~m3 & valid;
    t.mstart = m3 & ~(m3 << 1);
    t.mend = m3 & ~(m3 >> 1);
    return t;
}

// Visit the chunk's tokens in byte order, four at a time: positions come from registers, so the four
// literal-byte reads, then the four table reads, are issued back to back.  sink(entry) receives
// (bits << 5 | nbits) from the code table tab[0..255] = literals, tab[256 + L - 3] = match length L.
template <class Sink>
__device__ __forceinline__ void walk_tokens(ChunkTokens t, const uint8_t* chunk, const uint32_t* tab, Sink& sink)
{
    uint64_t T = t.lit | t.mstart, E = t.mend;
    while (T) {
        uint32_t idx[4];
        bool has[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            has[u] = T != 0;
            const uint32_t p = has[u] ? (uint32_t)__builtin_ctzll(T) : 0u;
            T &= T - 1; // stays 0 once empty
            const bool is_match = has[u] && ((t.mstart >> p) & 1ull);
            uint32_t L = 3;
            if (is_match) {
                L = (uint32_t)__builtin_ctzll(E) - p + 1u;
                E &= E - 1;
            }
            idx[u] = is_match ? 256u + L - 3u : p; // for literals: resolved to the byte value below
            has[u] = has[u];
            if (!is_match) idx[u] |= 0x80000000u;
        }
        uint32_t byte[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) byte[u] = chunk[idx[u] & 63u];
        uint32_t entry[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) entry[u] = tab[(idx[u] >> 31) ? byte[u] : idx[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (has[u]) sink.token(entry[u], (idx[u] >> 31) ? byte[u] : idx[u]);
    }
}

struct CountSink {
    uint32_t bits;
    __device__ __forceinline__ void token(uint32_t entry, uint32_t) { bits += entry & 31u; }
};

struct HistSink {
    uint32_t* h; // LDS histogram [512]: literal byte, or 256 + L - 3
    __device__ __forceinline__ void token(uint32_t, uint32_t symbol) { atomicAdd(&h[symbol], 1u); }
};

struct EmitSink {
    uint32_t* out; // LDS bit buffer (zero-initialised), written with atomic OR
    uint64_t acc;  // pending bits, LSB first
    uint32_t nacc; // number of pending bits
    uint32_t word; // index of the LDS word the pending bits start in
    __device__ __forceinline__ void token(uint32_t entry, uint32_t)
    {
        acc |= (uint64_t)(entry >> 5) << nacc;
        nacc += entry & 31u;
        if (nacc >= 32u) {
            atomicOr(&out[word], (uint32_t)acc);
            ++word;
            acc >>= 32;
            nacc -= 32u;
        }
    }
    __device__ __forceinline__ void flush()
    {
        if (nacc) atomicOr(&out[word], (uint32_t)acc);
    }
};

// exclusive scan of one value per thread over the workgroup; returns the thread's prefix and the
// total in *total.  scratch: BGZF_WAVES words of LDS.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* scratch, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += n;
    }
    __syncthreads();
    if (lane == 63u) scratch[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (uint32_t w = 0; w < BGZF_WAVES; ++w) {
        const uint32_t s = scratch[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total = tot;
    return base + incl - v;
}

__global__ __launch_bounds__(BGZF_THREADS) void bgzf_hist(const BgzfParams p)
{
    __shared__ uint32_t in_lds[BGZF_LDS_IN_WORDS];
    __shared__ uint32_t tab[512];
    __shared__ uint32_t h[513];
    const uint32_t b = blockIdx.x * p.sample_step, tid = threadIdx.x;
    if (b >= p.n_blocks) return;
    if (tid < 513) h[tid] = 0;
    if (tid < 512) tab[tid] = 0; // entries are not needed for counting symbols
    const uint32_t N = block_len(p, b);
    stage_block(p.text + (uint64_t)b * BGZF_IN, N, in_lds);
    __syncthreads();
    const uint32_t c0 = tid * BGZF_CHUNK;
    if (c0 < N) {
        const uint32_t len = min(BGZF_CHUNK, N - c0);
        const ChunkTokens t = chunk_tokens<false>(in_lds, tid, len, nullptr, nullptr);
        HistSink s{h};
        walk_tokens(t, reinterpret_cast<const uint8_t*>(in_lds + tid * (BGZF_CHUNK_WORDS + 1u)), tab, s);
    }
    if (tid == 0) atomicAdd(&h[512], 1u);
    __syncthreads();
    if (tid < 513 && h[tid]) atomicAdd(&p.hist[tid], h[tid]);
}

__device__ __forceinline__ uint32_t crc_mulmod(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, r = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & m) r ^= b;
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return r;
}

__device__ __forceinline__ void load_code_table(const BgzfParams& p, uint32_t* tab)
{
    if (threadIdx.x < 256) tab[threadIdx.x] = p.tables->lit[threadIdx.x];
    else if (threadIdx.x < 512) tab[threadIdx.x] = p.tables->match[threadIdx.x - 256];
}

__global__ __launch_bounds__(BGZF_THREADS) void bgzf_size(const BgzfParams p)
{
    __shared__ uint32_t in_lds[BGZF_LDS_IN_WORDS];
    __shared__ uint32_t tab[512];
    __shared__ uint32_t crc_tab[4][256];
    __shared__ uint32_t scratch[BGZF_WAVES];
    const uint32_t b = p.todo ? p.todo[1u + blockIdx.x] : blockIdx.x, tid = threadIdx.x;
    if (p.fast && p.fast[b]) return;
    load_code_table(p, tab);
    crc_tab[tid >> 8][tid & 255u] = p.tables->crc_table[tid >> 8][tid & 255u];
    const uint32_t N = block_len(p, b);
    stage_block(p.text + (uint64_t)b * BGZF_IN, N, in_lds);
    __syncthreads();
    const uint32_t c0 = tid * BGZF_CHUNK;
    uint32_t bits = 0, part = 0;
    if (c0 < N) {
        const uint32_t len = min(BGZF_CHUNK, N - c0);
        uint32_t crc;
        const ChunkTokens t = chunk_tokens<true>(in_lds, tid, len, crc_tab, &crc);
        CountSink s{0};
        walk_tokens(t, reinterpret_cast<const uint8_t*>(in_lds + tid * (BGZF_CHUNK_WORDS + 1u)), tab, s);
        bits = s.bits;
        // CRC-32 of the block: combine the chunk CRCs, crc(A||B) = crc(A) * x^(8|B|) + crc(B)
        const uint32_t after = N - (c0 + len); // bytes behind this chunk
        const uint32_t q = after / BGZF_CHUNK, r = after - q * BGZF_CHUNK;
        uint32_t sh = p.tables->crc_shift_chunk[q];
        if (r) sh = crc_mulmod(sh, p.tables->crc_shift_byte[r]);
        part = after ? crc_mulmod(sh, crc) : crc;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part ^= __shfl_xor(part, d, 64);
    if ((tid & 63u) == 0) scratch[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        uint32_t block_crc = 0;
        for (uint32_t w = 0; w < BGZF_WAVES; ++w) block_crc ^= scratch[w];
        p.crcs[b] = block_crc;
    }
    p.thread_bits[(uint64_t)b * BGZF_THREADS + tid] = (uint16_t)bits; // <= 64 literals of <= 15 bits
    uint32_t total;
    block_exclusive_scan(bits, scratch, &total);
    if (tid == 0) {
        const uint32_t body_bits = p.tables->hdr_nbits + total + (p.tables->eob & 31u);
        const uint32_t body = (body_bits + 7u) >> 3;
        const uint32_t stored = N + 5u;
        if (body > BGZF_OUT_LARGE || body >= stored) p.sizes[b] = (stored + BGZF_MEMBER_OVERHEAD) | 0x80000000u;
        else p.sizes[b] = (body + BGZF_MEMBER_OVERHEAD) | (body > BGZF_OUT_SMALL ? 0x40000000u : 0u);
    }
}

// ---- exclusive scans over the members of a call (hundreds of thousands): two launches each, every workgroup takes
// SCAN_BLOCK consecutive members: first the workgroups' totals, then each workgroup adds up the totals in front of it
// and scans its own members.  (One workgroup walking all of them took 0.9 ms per tile.) ----
constexpr uint32_t SCAN_THREADS = 256, SCAN_PER_THREAD = 8, SCAN_BLOCK = SCAN_THREADS * SCAN_PER_THREAD;

// sum of v over the workgroup (256 threads), returned to every thread; *excl = sum over the threads before this one
__device__ __forceinline__ uint64_t scan_block_u64(uint64_t v, uint64_t* excl, uint64_t* lds /* [4] */)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += o;
    }
    __syncthreads(); // (lds may still be read from a call before)
    if (lane == 63u) lds[wave] = incl;
    __syncthreads();
    uint64_t base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < SCAN_THREADS / 64u; ++w) {
        const uint64_t t = lds[w];
        if (w < wave) base += t;
        total += t;
    }
    *excl = base + incl - v;
    return total;
}

// member sizes (low 30 bits; bit 30 = "large": encoded by the large-buffer kernel) -> per-workgroup totals
__global__ __launch_bounds__(SCAN_THREADS) void bgzf_scan_totals(const uint32_t* sizes, uint32_t n, uint64_t* block_bytes, uint32_t* block_large)
{
    __shared__ uint64_t lds[4];
    const uint32_t i0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    uint64_t bytes = 0, large = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k)
        if (i0 + k < n) { const uint32_t sz = sizes[i0 + k]; bytes += sz & 0x3FFFFFFFu; large += (sz >> 30) & 1u; }
    uint64_t excl;
    const uint64_t total_bytes = scan_block_u64(bytes, &excl, lds);
    const uint64_t total_large = scan_block_u64(large, &excl, lds);
    if (threadIdx.x == 0) { block_bytes[blockIdx.x] = total_bytes; block_large[blockIdx.x] = (uint32_t)total_large; }
}

// offsets[i] = bytes of the members before i, offsets[n] = all bytes; large_list[0] = how many large members, [1..] which
__global__ __launch_bounds__(SCAN_THREADS) void bgzf_scan_apply(const uint32_t* sizes, uint32_t n, const uint64_t* block_bytes, const uint32_t* block_large,
                                                                uint64_t* offsets, uint32_t* large_list)
{
    __shared__ uint64_t lds[4];
    uint64_t before_bytes = 0, before_large = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += SCAN_THREADS) { before_bytes += block_bytes[b]; before_large += block_large[b]; }
    uint64_t unused;
    const uint64_t base_bytes = scan_block_u64(before_bytes, &unused, lds);
    const uint64_t base_large = scan_block_u64(before_large, &unused, lds);
    const uint32_t i0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    uint32_t sz[SCAN_PER_THREAD];
    uint64_t bytes = 0, large = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k) {
        sz[k] = i0 + k < n ? sizes[i0 + k] : 0u;
        bytes += sz[k] & 0x3FFFFFFFu;
        large += (sz[k] >> 30) & 1u;
    }
    uint64_t excl_bytes, excl_large;
    const uint64_t total_bytes = scan_block_u64(bytes, &excl_bytes, lds);
    const uint64_t total_large = scan_block_u64(large, &excl_large, lds);
    uint64_t run = base_bytes + excl_bytes;
    uint32_t lrun = (uint32_t)(base_large + excl_large);
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k)
        if (i0 + k < n) {
            offsets[i0 + k] = run;
            run += sz[k] & 0x3FFFFFFFu;
            if ((sz[k] >> 30) & 1u) large_list[1u + lrun++] = i0 + k;
        }
    if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 0) {
        offsets[n] = base_bytes + total_bytes;
        large_list[0] = (uint32_t)(base_large + total_large);
    }
}

// the members with flag[b] == 0, in order: list[0] = how many, list[1..] = which.  Same two steps.
__global__ __launch_bounds__(SCAN_THREADS) void bgzf_unset_totals(const uint8_t* flag, uint32_t n, uint32_t* block_count)
{
    __shared__ uint64_t lds[4];
    const uint32_t i0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    uint64_t v = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k)
        if (i0 + k < n && !flag[i0 + k]) ++v;
    uint64_t excl;
    const uint64_t total = scan_block_u64(v, &excl, lds);
    if (threadIdx.x == 0) block_count[blockIdx.x] = (uint32_t)total;
}

__global__ __launch_bounds__(SCAN_THREADS) void bgzf_unset_apply(const uint8_t* flag, uint32_t n, const uint32_t* block_count, uint32_t* list)
{
    __shared__ uint64_t lds[4];
    uint64_t before = 0;
    for (uint32_t b = threadIdx.x; b < blockIdx.x; b += SCAN_THREADS) before += block_count[b];
    uint64_t unused;
    const uint64_t base = scan_block_u64(before, &unused, lds);
    const uint32_t i0 = blockIdx.x * SCAN_BLOCK + threadIdx.x * SCAN_PER_THREAD;
    bool un[SCAN_PER_THREAD];
    uint64_t v = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k) {
        un[k] = i0 + k < n && !flag[i0 + k];
        v += un[k] ? 1u : 0u;
    }
    uint64_t excl;
    const uint64_t total = scan_block_u64(v, &excl, lds);
    uint32_t at = (uint32_t)(base + excl);
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k)
        if (un[k]) list[1u + at++] = i0 + k;
    if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 0) list[0] = (uint32_t)(base + total);
}

// OUT_BYTES = BGZF_OUT_SMALL: every member whose body fits 8 KB, and the stored ones (LDS < 80 KB: two
// workgroups per CU).  OUT_BYTES = BGZF_OUT_LARGE: the rest.  Both are launched over all blocks; a
// workgroup whose block belongs to the other kernel exits at once.
template <uint32_t OUT_BYTES>
__global__ __launch_bounds__(BGZF_THREADS, OUT_BYTES == BGZF_OUT_SMALL ? 8 : 4) void bgzf_encode(const BgzfParams p)
{
    __shared__ uint32_t in_lds[BGZF_LDS_IN_WORDS];
    __shared__ uint32_t outbuf[OUT_BYTES / 4];
    __shared__ uint32_t tab[512];
    __shared__ uint32_t scratch[BGZF_WAVES];
    const uint32_t tid = threadIdx.x;
    const uint32_t b = OUT_BYTES == BGZF_OUT_LARGE ? p.large_list[1u + blockIdx.x] : p.todo ? p.todo[1u + blockIdx.x] : blockIdx.x;
    if (p.fast && p.fast[b]) return;
    const uint32_t N = block_len(p, b);
    const uint8_t* blk = p.text + (uint64_t)b * BGZF_IN;
    uint8_t* dst = p.out + p.offsets[b];
    const uint32_t size = p.sizes[b] & 0x3FFFFFFFu;
    const bool stored = p.sizes[b] >> 31;
    if (((p.sizes[b] >> 30) & 1u) != (OUT_BYTES == BGZF_OUT_LARGE ? 1u : 0u)) return; // the other kernel's block

    load_code_table(p, tab);
    for (uint32_t i = tid; i < (size - BGZF_MEMBER_OVERHEAD + 3u) / 4u && i < OUT_BYTES / 4; i += BGZF_THREADS) outbuf[i] = 0;
    stage_block(blk, N, in_lds);
    __syncthreads();

    const uint32_t c0 = tid * BGZF_CHUNK;
    const uint32_t len = c0 < N ? min(BGZF_CHUNK, N - c0) : 0u;
    const uint8_t* chunk = reinterpret_cast<const uint8_t*>(in_lds + tid * (BGZF_CHUNK_WORDS + 1u));
    ChunkTokens t{0, 0, 0};
    uint32_t bits = 0;
    if (len && !stored) {
        t = chunk_tokens<false>(in_lds, tid, len, nullptr, nullptr);
        bits = p.thread_bits[(uint64_t)b * BGZF_THREADS + tid]; // counted by bgzf_size
    }
    uint32_t total;
    const uint32_t prefix = block_exclusive_scan(bits, scratch, &total);
    const uint32_t hdr_bits = p.tables->hdr_nbits;

    if (!stored) {
        if (tid < (hdr_bits + 31u) / 32u) atomicOr(&outbuf[tid], p.tables->hdr[tid]);
        if (len) {
            const uint32_t start = hdr_bits + prefix;
            EmitSink e{outbuf, 0, start & 31u, start >> 5};
            walk_tokens(t, chunk, tab, e);
            e.flush();
        }
        if (tid == 0) {
            const uint32_t start = hdr_bits + total;
            EmitSink e{outbuf, 0, start & 31u, start >> 5};
            e.token(p.tables->eob, 256u);
            e.flush();
        }
    }
    __syncthreads(); // outbuf is complete
    const uint32_t block_crc = p.crcs[b];

    // gzip member header (RFC 1952) with the BGZF extra field: BSIZE = member size - 1
    if (tid < 18) {
        const uint8_t h[18] = {0x1F, 0x8B, 8, 4, 0, 0, 0, 0, 0, 0xFF, 6, 0, 0x42, 0x43, 2, 0,
                               (uint8_t)((size - 1u) & 0xFFu), (uint8_t)((size - 1u) >> 8)};
        dst[tid] = h[tid];
    }
    uint8_t* body = dst + 18;
    uint32_t body_bytes;
    if (stored) {
        body_bytes = N + 5u;
        if (tid == 0) {
            body[0] = 1; // BFINAL=1, BTYPE=00
            body[1] = (uint8_t)(N & 0xFFu); body[2] = (uint8_t)(N >> 8);
            body[3] = (uint8_t)(~N & 0xFFu); body[4] = (uint8_t)((~N >> 8) & 0xFFu);
        }
        const uint8_t* src = reinterpret_cast<const uint8_t*>(in_lds);
        for (uint32_t i = tid; i < N; i += BGZF_THREADS) body[5 + i] = src[4u * lds_word(i >> 2) + (i & 3u)];
    } else {
        body_bytes = size - BGZF_MEMBER_OVERHEAD;
        const uint8_t* src = reinterpret_cast<const uint8_t*>(outbuf);
        for (uint32_t i = tid; i < body_bytes; i += BGZF_THREADS) body[i] = src[i];
    }
    if (tid < 8) {
        const uint32_t v = tid < 4 ? block_crc : N;
        body[body_bytes + tid] = (uint8_t)(v >> (8u * (tid & 3u)));
    }
}

} // namespace dnaf
