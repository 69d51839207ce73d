// mt_kernels.hip.h -- numpy-compatible MT19937 mode on the GPU (DNAF_RNG_MT19937).
//
// The reference's rows come from one sequential MT19937 stream: row r uses output words
// [4 S r, 4 S (r+1)) after the worker's entry state, two words per float64 roll
// ((a >> 5) * 2^26 + (b >> 6)) / 2^53 (numpy legacy random_sample; pop_factory.py:477).  Here
// every wave owns R consecutive rows, with R chosen so that 4 S R is a multiple of 624: every
// wave then starts on a block boundary of the stream (shifted by the entry position), and the
// block-aligned generator states of all waves are obtained by jump-ahead (mt_jump_kernel: the
// polynomial of mt_jump.h evaluated as a correlation over the word sequence, in LDS).
//
// mt_rowgen_kernel: one wave per workgroup.  The wave keeps the generator state (624 words) and a
// ring of tempered output words in LDS, regenerates a block in ten batches of 64 consecutive words,
// and consumes 256 words per step: lane l takes the four words (two rolls) of sample 64 k + l.  Cell
// rules, text layout and packed codes are those of rowgen_kernels.hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rowgen_kernels.hip.h"

namespace dnaf {

constexpr uint32_t MTN = 624, MTM = 397;
constexpr uint32_t MT_SEQ_WORDS = 19937 + 624 + 8; // words of sequence a jump correlates over
constexpr uint32_t MT_RING = 2 * 624;              // ring of generator words: the block being consumed and the one before / after it

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// dst[j] = window J words after src[map ? map[j] : j], for the polynomial g = x^(J-1) mod phi:
// dst[j][q] = XOR over set bits i of g of x[i + q + 1].  One workgroup per jump.  The polynomial arrives as the list of
// its set bits (mt_poly_list: word 0 = number of 16-entry groups, entries from word 16 on, each the BYTE offset 4 i, the
// last group padded with offsets of words that are zero): the positions are wave-uniform, so they come through scalar
// loads sixteen at a time and every term costs one address add, one LDS read and one xor.  (Scanning the bits of the
// coefficient words in the kernel cost ~8 scalar instructions per term, on the one scalar unit a CU's SIMDs share:
// the kernel was bound by that, 11.2 ms for 6410 jumps.)
constexpr uint32_t MT_POLY_LIST_HEAD = 16;                                     // words before the first entry
constexpr uint32_t MT_POLY_LIST_MAX = MT_POLY_LIST_HEAD + ((19937 + 15) / 16) * 16; // words of a full list
constexpr uint32_t MT_PAD_WORD = MT_SEQ_WORDS;                                 // x[MT_PAD_WORD + q + 1] = 0 for q < 624
__global__ __launch_bounds__(640) void mt_jump_kernel(const uint32_t* src, const uint32_t* map, uint32_t* dst,
                                                      const uint32_t* __restrict__ plist)
{
    __shared__ uint32_t x[MT_SEQ_WORDS + MTN + 1];
    const uint32_t j = blockIdx.x, t = threadIdx.x;
    const uint32_t* s = src + (uint64_t)(map ? map[j] : j) * MTN;
    if (t < MTN) x[t] = s[t];
    for (uint32_t k = MT_SEQ_WORDS + t; k < MT_SEQ_WORDS + MTN + 1u; k += 640u) x[k] = 0;
    __syncthreads();
    // the sequence continues by the recurrence; word k needs k-624, k-623, k-227: 227 words per step
    for (uint32_t k0 = MTN; k0 < MT_SEQ_WORDS; k0 += 227u) {
        const uint32_t k = k0 + t;
        if (t < 227u && k < MT_SEQ_WORDS) x[k] = x[k - 227u] ^ mt_twist(x[k - MTN], x[k - MTN + 1u]);
        __syncthreads();
    }
    if (t < MTN) {
        const uint32_t groups = plist[0];
        const uint32_t* e = plist + MT_POLY_LIST_HEAD;
        const uint8_t* xt = reinterpret_cast<const uint8_t*>(x + t + 1u);
        uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
        for (uint32_t gk = 0; gk < groups; ++gk, e += 16) {
            uint32_t v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const uint32_t*>(xt + e[u]);
#pragma unroll
            for (int u = 0; u < 16; u += 4) { acc0 ^= v[u]; acc1 ^= v[u + 1]; acc2 ^= v[u + 2]; acc3 ^= v[u + 3]; }
        }
        dst[(uint64_t)j * MTN + t] = (acc0 ^ acc1) ^ (acc2 ^ acc3);
    }
}

struct MtGenParams {
    GenParams g;              // tables and outputs as in the Philox path (keys unused)
    const uint64_t* thr53;    // [table rows][3] thresholds on the 53-bit roll numerator
    const uint32_t* states;   // [waves][624] block-aligned windows, wave w of this launch
    uint32_t start_off;       // offset of the first output word inside a wave's first block (0..623)
    uint32_t rows_per_wave;   // R
    uint64_t wave0_row;       // stream row (local to the call) of wave 0's first row
    // stream row k of the call is drawn for table row row_phase + k * row_stride (local to row_begin): stride 1, phase 0
    // ordinarily; with rng_div = N the call is launched N times, phase 0 .. N-1, over the same states -- the reference's
    // N forked workers drawing the same stream for their stripes (pop_factory.py:426-434)
    uint32_t row_stride, row_phase;
    uint64_t n_stream_rows;   // stream rows of this launch
};

// SNPTuples.pick_allele_index on the 53-bit numerator: roll = r / 2^53, `cum >= roll` <=> r <= floor(cum * 2^53)
__device__ __forceinline__ uint32_t pick53(uint64_t r, uint64_t t0, uint64_t t1, uint64_t t2)
{
    return r <= t0 ? 0u : (r <= t1 ? 1u : (r <= t2 ? 2u : 3u));
}

__device__ __forceinline__ uint64_t roll53(uint32_t a, uint32_t b) { return ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6); }

// Constants of the recurrence and the tempering kept in SGPRs behind an opaque asm, so that every and-xor / select pair
// is one v_bitop3_b32 with a scalar operand (as literals they cost a separate v_and each: VOP3 takes no literal on gfx9).
struct MtConsts { uint32_t upper, matrix, c1, c2; };
__device__ __forceinline__ MtConsts mt_consts()
{
    MtConsts k{0x80000000u, 0x9908B0DFu, 0x9D2C5680u, 0xEFC60000u};
    asm volatile("" : "+s"(k.upper), "+s"(k.matrix), "+s"(k.c1), "+s"(k.c2));
    return k;
}
// (a & c) ^ b in one instruction; the compiler fuses this pair only in some places
__device__ __forceinline__ uint32_t and_xor(uint32_t a, uint32_t c, uint32_t b)
{
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x6c" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
__device__ __forceinline__ uint32_t mt_twist_k(uint32_t s0, uint32_t s1, uint32_t sm, const MtConsts& k)
{
    uint32_t y;                                                          // top bit of s0, the rest of s1
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(y) : "s"(k.upper), "v"(s0), "v"(s1));
    const uint32_t mag = (uint32_t)((int32_t)(s1 << 31) >> 31);          // all ones when s1 is odd: v_bfe_i32
    return and_xor(mag, k.matrix, sm) ^ (y >> 1);
}
__device__ __forceinline__ uint32_t mt_temper_k(uint32_t y, const MtConsts& k)
{
    y ^= y >> 11;
    y = and_xor(y << 7, k.c1, y);
    y = and_xor(y << 15, k.c2, y);
    y ^= y >> 18;
    return y;
}

// orders this wave's LDS traffic for the compiler (the hardware keeps one wave's LDS operations in order by itself)
__device__ __forceinline__ void mt_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One wave: the next block of 624 generator words into the ring slot `cur`, from the block before it in slot `prev` -- the
// ring of (untempered) words the consumer reads IS the generator state, there is no second copy: per new word three LDS
// reads and one write.  Ten batches of 64 consecutive words (the last one 48).  Word i needs the previous block's words i
// and i+1 (for i = 623: the NEW word 0) and its word i+397, or from i = 227 on the new word i-227, made at least three
// batches earlier.  All addresses are `lane + constant` on one of the two slot bases, except in the batch where i+397
// wraps and for word 623's neighbour (one select each).
__device__ __forceinline__ void mt_wave_next_block(const uint32_t* prev, uint32_t* cur, uint32_t lane, const MtConsts& k)
{
    const uint32_t* pp = prev + lane;
    uint32_t* cp = cur + lane;
    const uint32_t* at_i1_9 = lane == 47u ? cur : pp + 577u;                        // batch 9: word i + 1
    const uint32_t* at_im3 = lane < 35u ? pp + (192u + MTM) : cur + (lane - 35u);   // batch 3: word i + 397
    // The lanes talk to each other through the ring, which the compiler cannot see (to one thread all these addresses are
    // distinct): a wave barrier per batch keeps the memory operations of different batches in program order.  The reads of
    // batch b+1 depend on nothing batch b writes, so they are issued before batch b's arithmetic and land behind it.
    uint32_t s0 = pp[0], s1 = pp[1], sm = pp[MTM];
#pragma unroll
    for (uint32_t b = 0; b < 10u; ++b) {
        mt_wave_sync();
        uint32_t n0 = 0, n1 = 0, nm = 0;
        if (b < 8u || (b == 8u && lane < 48u)) {
            const uint32_t c = b + 1u;
            n0 = pp[64u * c];
            n1 = c < 9u ? pp[64u * c + 1u] : *at_i1_9;
            nm = c < 3u ? pp[64u * c + MTM] : (c == 3u ? *at_im3 : cp[64u * c - (MTN - MTM)]);
        }
        if (b < 9u || lane < 48u) cp[64u * b] = mt_twist_k(s0, s1, sm, k);
        s0 = n0; s1 = n1; sm = nm;
    }
    mt_wave_sync();
}

// the 53-bit numerator r = hi * 2^26 + lo (hi = a >> 5, lo = b >> 6) as the 64-bit number hi * 2^32 + lo: the order of two
// such numbers is the order of the numerators, so `r > t` is one 64-bit compare and nothing has to be assembled
__device__ __forceinline__ uint64_t mt_roll_key(uint32_t a, uint32_t b) { return ((uint64_t)(a >> 5) << 32) | (uint64_t)(b >> 6); }
__device__ __forceinline__ uint64_t mt_thr_key(uint64_t t) { return ((t >> 26) << 32) | (t & 0x3FFFFFFull); }

// One lane = one SAMPLE here (the Philox kernels give a lane four): a step of the row loop takes 64 samples = 256 consecutive
// words of the stream, lane l reads the aligned quad of words of sample sb + l from the ring -- consecutive lanes, consecutive
// quads: no LDS bank conflicts (with four samples per lane the 64-byte lane stride made the ring reads 4-way conflicted and
// the per-lane packing cost more instructions than generating the numbers) -- and the comparisons come out as wave masks, from
// which the packed bytes are cut directly.
//
// Ring: two slots of 624 untempered generator words, which are at once the generator's state: a block is written to the
// slot that does not hold the newest block, from the newest block, and only when fewer than 256 words are left to consume
// -- those lie in the newest block.  The first block (the wave's entry state) is used from word start_off on, so the
// consumer's quads start at indices = start_off (mod 4): the ring is laid into its storage shifted by (-start_off) mod 4
// words, which makes every quad 16-byte aligned, and the first three words are mirrored behind the end for the one quad
// per turn that straddles it.  5 KB of LDS per wave.
template <bool TEXT, bool PACKED>
__global__ __launch_bounds__(64) void mt_rowgen_kernel(const MtGenParams mp)
{
    __shared__ __attribute__((aligned(16))) uint32_t ring_store[4 + MT_RING + 4];
    const GenParams& p = mp.g;
    const uint32_t lane = threadIdx.x;
    const uint64_t wave = blockIdx.x;
    const uint64_t row_lo = mp.wave0_row + wave * mp.rows_per_wave; // local row index of this wave's first row
    if (row_lo >= mp.n_stream_rows) return;
    const uint64_t row_hi = min(mp.n_stream_rows, row_lo + mp.rows_per_wave);
    const MtConsts K = mt_consts();
    uint32_t* ring = ring_store + ((4u - (mp.start_off & 3u)) & 3u);

    // the wave's entry state is the first block, in slot 0; ring bookkeeping in words, all wave-uniform
    for (uint32_t i = lane; i < MTN; i += 64u) ring[i] = mp.states[wave * MTN + i];
    uint32_t head = MTN, tail = mp.start_off, avail = MTN - mp.start_off;
    mt_wave_sync();

    const uint32_t S = p.S;
    const uint32_t row_words = 4u * S;
    const uint32_t row_pk_bytes = p.W * 4u;
    for (uint64_t stream_row = row_lo; stream_row < row_hi; ++stream_row) {
        const uint64_t row_local = mp.row_phase + stream_row * mp.row_stride; // the table row these words are drawn for
        const RowRec rec = p.recs[p.row_begin + row_local];
        const uint64_t* thr = mp.thr53 + (p.row_begin + row_local) * 3u;
        const uint64_t T0 = mt_thr_key(thr[0]), T1 = mt_thr_key(thr[1]), T2 = mt_thr_key(thr[2]);
        uint8_t* row_text = p.text + (rec.text_off - p.text_base);
        if (TEXT) {
            const uint8_t* src = p.prefix + rec.prefix_off;
            for (uint32_t i = lane; i < rec.prefix_len; i += 64u) row_text[i] = src[i];
            if (S == 0 && lane == 0) row_text[rec.prefix_len] = 0x0A;
        }
        uint8_t* cells = row_text + rec.prefix_len;
        uint8_t* prow = p.packed + row_local * p.packed_pitch;
        const uint8_t* frow = rec.slot >= 0 ? p.forced + (uint64_t)rec.slot * p.G : nullptr;
        const uint32_t cls = rec.cls;
        const bool multi = rec.n_alleles > 2;
        const bool plain = cls == 0u && !multi && !frow; // biallelic autosomal row without forced cells: the short step
        uint32_t consumed = 0; // words of this row consumed so far
        for (uint32_t sb = 0; sb < S; sb += 64u) {
            const uint32_t need = min(256u, row_words - consumed);
            while (avail < need) { // wave-uniform
                mt_wave_next_block(ring + (MTN - head), ring + head, lane, K); // head is 0 or 624: the other slot holds the newest block
                if (head == 0u && lane < 3u) ring[MT_RING + lane] = ring[lane];
                head = MTN - head;
                avail += MTN;
                mt_wave_sync();
            }
            uint32_t q = tail + 4u * lane;
            q = q >= MT_RING ? q - MT_RING : q;
            // 16-byte aligned by the ring's shift; lanes past the row's end read words that are not used.  The ring holds the
            // generator's words as they are: a roll is hi * 2^26 + lo with hi = temper(a) >> 5 and lo = temper(b) >> 6, and
            // unless hi EQUALS the high part of a threshold (2^-27 per draw) the low part decides nothing -- so the step below
            // tempers only the a-words (tempering is 8 of the 13 instructions a word costs) and hands the rare tie, like
            // every row that is not plain, to the general step, which tempers all four.
            const uint4 raw = *reinterpret_cast<const uint4*>(__builtin_assume_aligned(ring + q, 16));
            bool done = false;
            if (plain && sb + 64u < S) {
                // all 64 samples live, none of them the row's last
                const uint32_t ha = mt_temper_k(raw.x, K) >> 5, hb = mt_temper_k(raw.z, K) >> 5, th = (uint32_t)(T0 >> 32);
                if (__ballot(ha == th || hb == th) == 0ull) { // wave-uniform; otherwise the general step decides with the low parts
                    done = true;
                    const uint64_t A = __ballot(ha > th), B = __ballot(hb > th);
                    if (TEXT) {
                        const uint32_t cell = DIPLOID_CELL(0u, 0u, 0x09u) + (ha > th ? 1u : 0u) + (hb > th ? 0x10000u : 0u);
                        store_unaligned<4>(cells + 4ull * sb + 4u * lane, &cell);
                    }
                    if (PACKED) {
                        // even bits from A, odd bits from B: s_bitreplicate doubles every bit, the masks pick the copy
                        uint64_t a0, a1, b0, b1;
                        asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(a0) : "s"((uint32_t)A));
                        asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(a1) : "s"((uint32_t)(A >> 32)));
                        asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(b0) : "s"((uint32_t)B));
                        asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(b1) : "s"((uint32_t)(B >> 32)));
                        const uint64_t p0 = (a0 & 0x5555555555555555ull) | (b0 & 0xAAAAAAAAAAAAAAAAull);
                        const uint64_t p1 = (a1 & 0x5555555555555555ull) | (b1 & 0xAAAAAAAAAAAAAAAAull);
                        if (lane == 0u) {
                            const uint4 pk = make_uint4((uint32_t)p0, (uint32_t)(p0 >> 32), (uint32_t)p1, (uint32_t)(p1 >> 32));
                            store_unaligned<16>(prow + (sb >> 2), &pk);
                        }
                    }
                }
            }
            if (!done) {
                const uint4 w = make_uint4(mt_temper_k(raw.x, K), mt_temper_k(raw.y, K), mt_temper_k(raw.z, K), mt_temper_k(raw.w, K));
                const uint64_t va = mt_roll_key(w.x, w.y), vb = mt_roll_key(w.z, w.w);
                const uint32_t i = sb + lane; // this lane's sample
                const bool live = i < S;
                // SNPTuples.pick_allele_index (pop_factory.py:92-95): first index whose cumulative frequency is >= the roll
                uint32_t a = va > T0 ? 1u : 0u, b = vb > T0 ? 1u : 0u;
                if (multi) { // wave-uniform
                    a += (va > T1 ? 1u : 0u) + (va > T2 ? 1u : 0u);
                    b += (vb > T1 ? 1u : 0u) + (vb > T2 ? 1u : 0u);
                }
                const uint32_t g = i >> 2, k = i & 3u;
                if (frow && live && ((frow[g] >> k) & 1u)) a = b = 1u; // forced genotype 1/1, pop_factory.py:495-499
                bool missing = false, haploid = false;
                uint32_t male_nib = 0;
                if (cls != 0u) { // wave-uniform
                    male_nib = live ? p.male_nib[g] : 0u;
                    const bool is_male = (male_nib >> k) & 1u;
                    missing = cls == 2u && !is_male;              // pop_factory.py:481-484
                    haploid = cls >= 2u || (cls == 1u && is_male); // common/snp.py:102-109
                }
                if (PACKED) {
                    // 2-bit codes of the 64 samples as two wave masks -> 16 bytes; lane j < 16 cuts byte j (samples 4j .. 4j+3)
                    const uint64_t A = __ballot(live && !missing && a != 0u), B = __ballot(live && !missing && !haploid && b != 0u);
                    uint32_t x = (uint32_t)(A >> (4u * (lane & 15u))) & 15u, y = (uint32_t)(B >> (4u * (lane & 15u))) & 15u;
                    x = (x | (x << 2)) & 0x33u; x = (x | (x << 1)) & 0x55u;
                    y = (y | (y << 2)) & 0x33u; y = (y | (y << 1)) & 0x55u;
                    const uint32_t first = sb >> 2; // the step's first byte of the row; bytes up to the row's padded length are stored
                    if (lane < 16u && first + lane < row_pk_bytes) prow[first + lane] = (uint8_t)(x | (y << 1));
                }
                if (TEXT && live) {
                    const uint32_t sep = (i == S - 1u) ? 0x0Au : 0x09u; // "\t".join(...) + "\n", pop_factory.py:508
                    if (cls == 0u) {
                        const uint32_t cell = DIPLOID_CELL(a, b, sep);
                        store_unaligned<4>(cells + 4ull * i, &cell);
                    } else {
                        // X: 2 bytes for a male, 4 for a female; Y / MT: 2 bytes each
                        const uint32_t before = male_nib & ((1u << k) - 1u);
                        const uint64_t off = cls == 1u ? (uint64_t)p.xoff[g] + 4u * k - 2u * (uint32_t)__builtin_popcount(before) : 2ull * i;
                        if (missing) { const uint16_t h = (uint16_t)HAPLOID_CELL(0x2Eu, sep); store_unaligned<2>(cells + off, &h); }
                        else if (haploid) { const uint16_t h = (uint16_t)HAPLOID_CELL(0x30u + a, sep); store_unaligned<2>(cells + off, &h); }
                        else { const uint32_t cell = DIPLOID_CELL(a, b, sep); store_unaligned<4>(cells + off, &cell); }
                    }
                }
            }
            mt_wave_sync();
            tail += need;
            if (tail >= MT_RING) tail -= MT_RING;
            avail -= need;
            consumed += need;
        }
        if (PACKED) // bytes of the packed row that no step reached (S = 0, or a row length that is a multiple of 64 samples plus pad)
            for (uint32_t gq = ((S + 63u) / 64u) * 16u + lane; gq < row_pk_bytes; gq += 64u) prow[gq] = 0;
    }
}

} // namespace dnaf
