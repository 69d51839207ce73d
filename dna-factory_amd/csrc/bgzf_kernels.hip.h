// bgzf_kernels.hip.h -- GPU BGZF (blocked gzip) encoder for the VCF text stream (gfx950).
//
// One 256-thread workgroup encodes one BGZF member from 64,512 input bytes; a thread owns a
// 252-byte chunk.  Tokens are literals and matches at distance 4 only: byte i matches when it equals
// byte i-4 of the same block (GT cells repeat with period 4; haploid cells with period 2).  Runs of
// matching bytes never cross a thread's chunk, so tokenisation needs only the dword in front of the
// chunk, and a run is a single match token (<= 252 <= 258).
//
//   bgzf_hist    token histogram over a sample of blocks          -> Huffman tables (host, bgzf_host.h)
//   bgzf_size    exact compressed size of every block             -> exclusive scan -> output offsets
//   bgzf_encode  header + deflate bits (assembled in LDS) + CRC-32 + ISIZE, written at the block's offset
//
// A block whose deflate body would not fit the LDS buffer (or not beat a stored block) is emitted as
// a stored block.  Output is checked by inflating it on the host (tests/test_bgzf_gpu.py): any
// conforming gzip reader must return the input bytes and accept CRC-32 / ISIZE.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bgzf_host.h"

namespace dnaf {

constexpr uint32_t BGZF_THREADS = 256;
constexpr uint32_t BGZF_LDS_OUT = 32768;              // bytes of LDS for one block's deflate body
constexpr uint32_t BGZF_MASK_WORDS = 9;               // 252 match bits + a zero sentinel word
constexpr uint32_t BGZF_MEMBER_OVERHEAD = 18 + 8;     // gzip header with BC field + CRC32 + ISIZE

struct BgzfParams {
    const uint8_t* text;     // input stream
    uint64_t n_bytes;        // its length
    uint32_t n_blocks;       // ceil(n_bytes / BGZF_IN)
    uint32_t sample_step;    // bgzf_hist: encode every sample_step-th block
    const BgzfTables* tables;
    uint32_t* hist;          // [256 literals][256 match lengths][1 eob]
    uint32_t* sizes;         // [n_blocks] member size in bytes; bit 31 set = stored block
    const uint64_t* offsets; // [n_blocks+1] exclusive scan of sizes
    uint8_t* out;
};

__device__ __forceinline__ uint32_t load_u32(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// Match bits of one thread's chunk -> mk[0..8] in LDS (bit p = byte c0+p equals byte c0+p-4).
// Optionally accumulates the chunk's CRC-32 (standard: init ~0, final xor ~0).
template <bool WITH_CRC>
__device__ __forceinline__ uint32_t chunk_mask(const uint8_t* blk, uint32_t c0, uint32_t len, uint32_t* mk,
                                               const uint32_t (*crc_tab)[256])
{
    uint32_t crc = 0xFFFFFFFFu;
    uint32_t prev = c0 ? load_u32(blk + c0 - 4) : 0u;
    uint32_t word = 0;
#pragma unroll
    for (uint32_t k = 0; k < BGZF_CHUNK / 4; ++k) {
        uint32_t nib = 0;
        if (4u * k < len) { // bytes past the end of the stream are never read beyond the buffer's slack
            const uint32_t cur = load_u32(blk + c0 + 4u * k);
            const uint32_t x = cur ^ prev;
            const uint32_t y = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x; // bit 7 of each byte: byte != 0
            const uint32_t e = ~y & 0x80808080u;
            nib = ((e >> 7) & 1u) | ((e >> 14) & 2u) | ((e >> 21) & 4u) | ((e >> 28) & 8u);
            if (WITH_CRC && 4u * k + 4u <= len) {
                const uint32_t v = crc ^ cur;
                crc = crc_tab[3][v & 0xFFu] ^ crc_tab[2][(v >> 8) & 0xFFu] ^ crc_tab[1][(v >> 16) & 0xFFu] ^ crc_tab[0][v >> 24];
            }
            prev = cur;
        }
        if (c0 == 0 && k == 0) nib = 0; // no history before the block's first 4 bytes
        word |= nib << (4u * (k & 7u));
        if ((k & 7u) == 7u || k == BGZF_CHUNK / 4 - 1) {
            mk[k >> 3] = word;
            word = 0;
        }
    }
    mk[8] = 0;
    // clear bits at and past `len` (only the stream's last block has short chunks)
    if (len < BGZF_CHUNK) {
        for (uint32_t w = len >> 5; w < 8; ++w) {
            const uint32_t keep = (w == (len >> 5)) ? ((1u << (len & 31u)) - 1u) : 0u;
            mk[w] &= keep;
        }
        if (WITH_CRC)
            for (uint32_t i = len & ~3u; i < len; ++i) crc = crc_tab[0][(crc ^ blk[c0 + i]) & 0xFFu] ^ (crc >> 8);
    }
    return ~crc;
}

// Visit the tokens of a chunk in order: v.literal(byte offset in chunk) / v.match(length).
template <class V>
__device__ __forceinline__ void walk_tokens(const uint32_t* mk, uint32_t len, V& v)
{
    uint32_t pos = 0;
    while (pos < len) {
        const uint32_t wi = pos >> 5, sh = pos & 31u;
        const uint32_t w = (uint32_t)((((uint64_t)mk[wi + 1] << 32) | mk[wi]) >> sh); // 32 valid bits from pos
        if (w & 1u) {
            uint32_t run = (~w) ? (uint32_t)__builtin_ctz(~w) : 32u;
            while ((run & 31u) == 0u && pos + run < len) { // the run fills the window: keep counting
                const uint32_t p2 = pos + run;
                const uint32_t w2 = (uint32_t)((((uint64_t)mk[(p2 >> 5) + 1] << 32) | mk[p2 >> 5]) >> (p2 & 31u));
                const uint32_t c = (~w2) ? (uint32_t)__builtin_ctz(~w2) : 32u;
                run += c;
                if (c < 32u) break;
            }
            if (run > len - pos) run = len - pos;
            if (run >= 3u) v.match(run);
            else
                for (uint32_t j = 0; j < run; ++j) v.literal(pos + j);
            pos += run;
        } else {
            uint32_t z = w ? (uint32_t)__builtin_ctz(w) : 32u;
            if (z > len - pos) z = len - pos;
            for (uint32_t j = 0; j < z; ++j) v.literal(pos + j);
            pos += z;
        }
    }
}

struct CountVisitor {
    const uint8_t* chunk;
    const uint32_t* lit_tab;
    const uint32_t* match_tab;
    uint32_t bits;
    __device__ __forceinline__ void literal(uint32_t p) { bits += lit_tab[chunk[p]] & 31u; }
    __device__ __forceinline__ void match(uint32_t L) { bits += match_tab[L - 3u] & 31u; }
};

struct HistVisitor {
    const uint8_t* chunk;
    uint32_t* h; // LDS histogram
    __device__ __forceinline__ void literal(uint32_t p) { atomicAdd(&h[chunk[p]], 1u); }
    __device__ __forceinline__ void match(uint32_t L) { atomicAdd(&h[256u + L - 3u], 1u); }
};

struct EmitVisitor {
    const uint8_t* chunk;
    const uint32_t* lit_tab;
    const uint32_t* match_tab;
    uint32_t* out;      // LDS bit buffer (zero-initialised), written with atomic OR
    uint64_t acc;       // pending bits, LSB first
    uint32_t nacc;      // number of pending bits
    uint32_t word;      // index of the LDS word the pending bits start in
    __device__ __forceinline__ void put(uint32_t entry)
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
    __device__ __forceinline__ void literal(uint32_t p) { put(lit_tab[chunk[p]]); }
    __device__ __forceinline__ void match(uint32_t L) { put(match_tab[L - 3u]); }
    __device__ __forceinline__ void flush()
    {
        if (nacc) atomicOr(&out[word], (uint32_t)acc);
    }
};

__device__ __forceinline__ uint32_t block_len(const BgzfParams& p, uint32_t b)
{
    const uint64_t start = (uint64_t)b * BGZF_IN;
    const uint64_t left = p.n_bytes - start;
    return left < BGZF_IN ? (uint32_t)left : BGZF_IN;
}

// exclusive scan of one value per thread over the 256-thread workgroup; returns the thread's prefix and
// the total in *total.  scratch: 8 words of LDS.
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* scratch, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += n;
    }
    if (lane == 63u) scratch[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += scratch[w];
    *total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(BGZF_THREADS) void bgzf_hist(const BgzfParams p)
{
    __shared__ uint32_t mask[BGZF_THREADS * BGZF_MASK_WORDS];
    __shared__ uint32_t h[513];
    const uint32_t b = blockIdx.x * p.sample_step;
    if (b >= p.n_blocks) return;
    for (uint32_t i = threadIdx.x; i < 513; i += BGZF_THREADS) h[i] = 0;
    __syncthreads();
    const uint32_t N = block_len(p, b);
    const uint8_t* blk = p.text + (uint64_t)b * BGZF_IN;
    const uint32_t c0 = threadIdx.x * BGZF_CHUNK;
    if (c0 < N) {
        const uint32_t len = min(BGZF_CHUNK, N - c0);
        uint32_t* mk = mask + threadIdx.x * BGZF_MASK_WORDS;
        chunk_mask<false>(blk, c0, len, mk, nullptr);
        HistVisitor v{blk + c0, h};
        walk_tokens(mk, len, v);
    }
    if (threadIdx.x == 0) atomicAdd(&h[512], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 513; i += BGZF_THREADS)
        if (h[i]) atomicAdd(&p.hist[i], h[i]);
}

__global__ __launch_bounds__(BGZF_THREADS) void bgzf_size(const BgzfParams p)
{
    __shared__ uint32_t mask[BGZF_THREADS * BGZF_MASK_WORDS];
    __shared__ uint32_t lit[256], match[256];
    __shared__ uint32_t scratch[8];
    const uint32_t b = blockIdx.x;
    lit[threadIdx.x] = p.tables->lit[threadIdx.x];
    match[threadIdx.x] = p.tables->match[threadIdx.x];
    __syncthreads();
    const uint32_t N = block_len(p, b);
    const uint8_t* blk = p.text + (uint64_t)b * BGZF_IN;
    const uint32_t c0 = threadIdx.x * BGZF_CHUNK;
    uint32_t bits = 0;
    if (c0 < N) {
        const uint32_t len = min(BGZF_CHUNK, N - c0);
        uint32_t* mk = mask + threadIdx.x * BGZF_MASK_WORDS;
        chunk_mask<false>(blk, c0, len, mk, nullptr);
        CountVisitor v{blk + c0, lit, match, 0};
        walk_tokens(mk, len, v);
        bits = v.bits;
    }
    uint32_t total;
    block_exclusive_scan(bits, scratch, &total);
    if (threadIdx.x == 0) {
        const uint32_t body_bits = p.tables->hdr_nbits + total + (p.tables->eob & 31u);
        const uint32_t body = (body_bits + 7u) >> 3;
        const uint32_t stored = N + 5u;
        if (body > BGZF_LDS_OUT || body >= stored) p.sizes[b] = (stored + BGZF_MEMBER_OVERHEAD) | 0x80000000u;
        else p.sizes[b] = body + BGZF_MEMBER_OVERHEAD;
    }
}

// exclusive scan of the member sizes: one workgroup, sequential over slices (n_blocks is a few 10^5)
__global__ __launch_bounds__(1024) void bgzf_scan(const uint32_t* sizes, uint32_t n, uint64_t* offsets)
{
    __shared__ uint64_t part[1024];
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint64_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += sizes[i] & 0x7FFFFFFFu;
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t run = 0;
        for (uint32_t i = 0; i < 1024; ++i) { const uint64_t t = part[i]; part[i] = run; run += t; }
        offsets[n] = run;
    }
    __syncthreads();
    uint64_t run = part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) { offsets[i] = run; run += sizes[i] & 0x7FFFFFFFu; }
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

__global__ __launch_bounds__(BGZF_THREADS) void bgzf_encode(const BgzfParams p)
{
    __shared__ uint32_t outbuf[BGZF_LDS_OUT / 4];
    __shared__ uint32_t mask[BGZF_THREADS * BGZF_MASK_WORDS];
    __shared__ uint32_t lit[256], match[256];
    __shared__ uint32_t crc_tab[4][256];
    __shared__ uint32_t scratch[8];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    const uint32_t N = block_len(p, b);
    const uint8_t* blk = p.text + (uint64_t)b * BGZF_IN;
    uint8_t* dst = p.out + p.offsets[b];
    const uint32_t size = p.sizes[b] & 0x7FFFFFFFu;
    const bool stored = p.sizes[b] >> 31;

    lit[tid] = p.tables->lit[tid];
    match[tid] = p.tables->match[tid];
#pragma unroll
    for (int k = 0; k < 4; ++k) crc_tab[k][tid] = p.tables->crc_table[k][tid];
    for (uint32_t i = tid; i < BGZF_LDS_OUT / 4; i += BGZF_THREADS) outbuf[i] = 0;
    __syncthreads();

    const uint32_t c0 = tid * BGZF_CHUNK;
    const uint32_t len = c0 < N ? min(BGZF_CHUNK, N - c0) : 0u;
    uint32_t* mk = mask + tid * BGZF_MASK_WORDS;
    uint32_t bits = 0, crc = 0;
    if (len) {
        crc = chunk_mask<true>(blk, c0, len, mk, crc_tab);
        if (!stored) {
            CountVisitor v{blk + c0, lit, match, 0};
            walk_tokens(mk, len, v);
            bits = v.bits;
        }
    }
    uint32_t total;
    const uint32_t prefix = block_exclusive_scan(bits, scratch, &total);
    const uint32_t hdr_bits = p.tables->hdr_nbits;

    if (!stored) {
        for (uint32_t i = tid; i < (hdr_bits + 31u) / 32u; i += BGZF_THREADS) atomicOr(&outbuf[i], p.tables->hdr[i]);
        if (len) {
            const uint32_t start = hdr_bits + prefix;
            EmitVisitor e{blk + c0, lit, match, outbuf, 0, start & 31u, start >> 5};
            walk_tokens(mk, len, e);
            e.flush();
        }
        if (tid == 0) {
            const uint32_t start = hdr_bits + total;
            EmitVisitor e{blk, lit, match, outbuf, 0, start & 31u, start >> 5};
            e.put(p.tables->eob);
            e.flush();
        }
    }
    // CRC-32 of the block: combine the chunk CRCs, crc(A||B) = crc(A) * x^(8|B|) + crc(B)
    {
        uint32_t part = 0;
        if (len) {
            const uint32_t after = N - (c0 + len);       // bytes behind this chunk
            const uint32_t q = after / BGZF_CHUNK, r = after - q * BGZF_CHUNK;
            uint32_t sh = p.tables->crc_shift_chunk[q];
            if (r) sh = crc_mulmod(sh, p.tables->crc_shift_byte[r]);
            part = after ? crc_mulmod(sh, crc) : crc;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part ^= __shfl_xor(part, d, 64);
        __syncthreads();
        if ((tid & 63u) == 0) scratch[tid >> 6] = part;
    }
    __syncthreads();
    const uint32_t block_crc = scratch[0] ^ scratch[1] ^ scratch[2] ^ scratch[3];

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
        for (uint32_t i = tid; i < N; i += BGZF_THREADS) body[5 + i] = blk[i];
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
