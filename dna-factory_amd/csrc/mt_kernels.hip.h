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
// ring of tempered output words in LDS, regenerates blocks in the three dependent phases of the
// recurrence (227 / 227 / 170 words, 64 lanes wide), and consumes 16 words per lane per iteration:
// the 8 rolls of a group of 4 samples, exactly the lane mapping of the Philox kernel.  Cell rules,
// text layout and packed codes are those of rowgen_kernels.hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rowgen_kernels.hip.h"

namespace dnaf {

constexpr uint32_t MTN = 624, MTM = 397;
constexpr uint32_t MT_SEQ_WORDS = 19937 + 624 + 8; // words of sequence a jump correlates over
constexpr uint32_t MT_RING = 1872;                 // 3 blocks of tempered words (multiple of 16)

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

// dst[j] = window J words after src[map ? map[j] : j], for the polynomial g = x^(J-1) mod phi given as
// 312 x u64 coefficient words: dst[j][q] = XOR over set bits i of x[i + q + 1].  One workgroup per jump.
__global__ __launch_bounds__(640) void mt_jump_kernel(const uint32_t* src, const uint32_t* map, uint32_t* dst,
                                                      const uint64_t* poly)
{
    __shared__ uint32_t x[MT_SEQ_WORDS];
    const uint32_t j = blockIdx.x, t = threadIdx.x;
    const uint32_t* s = src + (uint64_t)(map ? map[j] : j) * MTN;
    if (t < MTN) x[t] = s[t];
    __syncthreads();
    // the sequence continues by the recurrence; word k needs k-624, k-623, k-227: 227 words per step
    for (uint32_t k0 = MTN; k0 < MT_SEQ_WORDS; k0 += 227u) {
        const uint32_t k = k0 + t;
        if (t < 227u && k < MT_SEQ_WORDS) x[k] = x[k - 227u] ^ mt_twist(x[k - MTN], x[k - MTN + 1u]);
        __syncthreads();
    }
    if (t < MTN) {
        uint32_t acc = 0;
        const uint32_t* xt = x + t + 1u;
        for (uint32_t w = 0; w < 312u; ++w) {
            uint64_t bits = poly[w]; // wave-uniform: the positions are scalar, the loads are independent
            const uint32_t* xw = xt + 64u * w;
            while (bits) {
                uint32_t idx[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    idx[u] = bits ? (uint32_t)__builtin_ctzll(bits) : 0xFFFFFFFFu;
                    bits &= bits - 1; // stays 0 once empty
                }
                uint32_t val[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) val[u] = idx[u] != 0xFFFFFFFFu ? xw[idx[u]] : 0u;
#pragma unroll
                for (int u = 0; u < 8; ++u) acc ^= val[u];
            }
        }
        dst[(uint64_t)j * MTN + t] = acc;
    }
}

struct MtGenParams {
    GenParams g;              // tables and outputs as in the Philox path (keys unused)
    const uint64_t* thr53;    // [table rows][3] thresholds on the 53-bit roll numerator
    const uint32_t* states;   // [waves][624] block-aligned windows, wave w of this launch
    uint32_t start_off;       // offset of the first output word inside a wave's first block (0..623)
    uint32_t rows_per_wave;   // R
    uint64_t wave0_row;       // table row (local to row_begin) of wave 0's first row may be negative? no: >= 0
};

// SNPTuples.pick_allele_index on the 53-bit numerator: roll = r / 2^53, `cum >= roll` <=> r <= floor(cum * 2^53)
__device__ __forceinline__ uint32_t pick53(uint64_t r, uint64_t t0, uint64_t t1, uint64_t t2)
{
    return r <= t0 ? 0u : (r <= t1 ? 1u : (r <= t2 ? 2u : 3u));
}

__device__ __forceinline__ uint64_t roll53(uint32_t a, uint32_t b) { return ((uint64_t)(a >> 5) << 26) | (uint64_t)(b >> 6); }

// One wave: the next block.  The state is regenerated in place in the three dependent phases of the
// recurrence (words 0..226 need only old words; 227..453 need the new words 0..226; 454..622 need the
// new words 227..395, and word 623 the new words 396 and 0).  Inside a phase nothing depends on anything
// else written in that phase, so all of a phase's LDS reads are issued before its writes: three latency
// exposures per block instead of ten.  Each new word is also tempered into the output ring at
// ring[(head + i) mod MT_RING].  The code is branch-free: lanes past the end of a phase read a valid
// word and write to a spare slot behind each array.
template <uint32_t LO, uint32_t HI>
__device__ __forceinline__ void mt_wave_phase(uint32_t* st, uint32_t* ring, uint32_t head, uint32_t lane)
{
    constexpr uint32_t STEPS = (HI - LO + 63u) / 64u;
    constexpr bool OLD_M = HI <= MTN - MTM; // word i + 397 is still the old one
    uint32_t v[STEPS];
#pragma unroll
    for (uint32_t s = 0; s < STEPS; ++s) {
        const uint32_t i = min(LO + 64u * s + lane, HI - 1u);
        const uint32_t i1 = HI == MTN ? 0u : i + 1u; // only the phase of word 623 wraps
        const uint32_t im = OLD_M ? i + MTM : i - (MTN - MTM);
        v[s] = st[im] ^ mt_twist(st[i], st[i1]);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (uint32_t s = 0; s < STEPS; ++s) {
        const uint32_t i = LO + 64u * s + lane;
        const bool ok = i < HI;
        uint32_t h = head + i;
        h = h >= MT_RING ? h - MT_RING : h;
        st[ok ? i : MTN] = v[s];                 // st[MTN] and ring[MT_RING] are spare slots
        ring[ok ? h : MT_RING] = mt_temper(v[s]);
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void mt_wave_next_block(uint32_t* st, uint32_t* ring, uint32_t head, uint32_t lane)
{
    mt_wave_phase<0, 227>(st, ring, head, lane);
    mt_wave_phase<227, 454>(st, ring, head, lane);
    mt_wave_phase<454, 623>(st, ring, head, lane);
    mt_wave_phase<623, 624>(st, ring, head, lane); // needs the new word 0
}

template <bool TEXT, bool PACKED>
__global__ __launch_bounds__(64) void mt_rowgen_kernel(const MtGenParams mp)
{
    __shared__ uint32_t st[MTN + 1];
    __shared__ __attribute__((aligned(16))) uint32_t ring[MT_RING + 4];
    const GenParams& p = mp.g;
    const uint32_t lane = threadIdx.x;
    const uint64_t wave = blockIdx.x;
    const uint64_t row_lo = mp.wave0_row + wave * mp.rows_per_wave; // local row index of this wave's first row
    if (row_lo >= p.n_rows) return;
    const uint64_t row_hi = min(p.n_rows, row_lo + mp.rows_per_wave);

    for (uint32_t i = lane; i < MTN; i += 64u) st[i] = mp.states[wave * MTN + i];
    __builtin_amdgcn_wave_barrier();
    // ring bookkeeping in words; all wave-uniform
    uint32_t head = 0, tail = 0, avail = 0;
    // the first block is used from start_off on
    for (uint32_t i = mp.start_off + lane; i < MTN; i += 64u) ring[i - mp.start_off] = mt_temper(st[i]);
    head = avail = MTN - mp.start_off;
    __builtin_amdgcn_wave_barrier();

    const uint32_t S = p.S, G = p.G;
    const uint32_t row_words = 4u * S;
    for (uint64_t row_local = row_lo; row_local < row_hi; ++row_local) {
        const RowRec rec = p.recs[p.row_begin + row_local];
        const uint64_t* thr = mp.thr53 + (p.row_begin + row_local) * 3u;
        const uint64_t t0 = thr[0], t1 = thr[1], t2 = thr[2];
        uint8_t* row_text = p.text + (rec.text_off - p.text_base);
        if (TEXT) {
            const uint8_t* src = p.prefix + rec.prefix_off;
            for (uint32_t i = lane; i < rec.prefix_len; i += 64u) row_text[i] = src[i];
            if (S == 0 && lane == 0) row_text[rec.prefix_len] = 0x0A;
        }
        uint8_t* cells = row_text + rec.prefix_len;
        uint8_t* prow = p.packed + row_local * p.packed_pitch;
        const uint8_t* frow = rec.slot >= 0 ? p.forced + (uint64_t)rec.slot * G : nullptr;
        const uint32_t cls = rec.cls;
        uint32_t consumed = 0; // words of this row consumed so far
        for (uint32_t gb = 0; gb < G; gb += 64u) {
            const uint32_t need = min(1024u, row_words - consumed);
            while (avail < need) { // wave-uniform
                mt_wave_next_block(st, ring, head, lane);
                head += MTN;
                if (head >= MT_RING) head -= MT_RING;
                avail += MTN;
                __builtin_amdgcn_wave_barrier();
            }
            const uint32_t g = gb + lane;
            if (g < G) {
                // the group's 16 words = 4 aligned quads (tail is a multiple of 4 words); a quad never straddles
                // the ring end.  Quads past the row's last sample may hold stale words: they are never used.
                uint32_t q0 = tail + 16u * lane, q1 = q0 + 4u, q2 = q0 + 8u, q3 = q0 + 12u;
                q0 = q0 >= MT_RING ? q0 - MT_RING : q0; q1 = q1 >= MT_RING ? q1 - MT_RING : q1;
                q2 = q2 >= MT_RING ? q2 - MT_RING : q2; q3 = q3 >= MT_RING ? q3 - MT_RING : q3;
                const uint4 w0 = *reinterpret_cast<const uint4*>(ring + q0), w1 = *reinterpret_cast<const uint4*>(ring + q1);
                const uint4 w2 = *reinterpret_cast<const uint4*>(ring + q2), w3 = *reinterpret_cast<const uint4*>(ring + q3);
                uint32_t a[4], b[4];
                a[0] = pick53(roll53(w0.x, w0.y), t0, t1, t2); b[0] = pick53(roll53(w0.z, w0.w), t0, t1, t2);
                a[1] = pick53(roll53(w1.x, w1.y), t0, t1, t2); b[1] = pick53(roll53(w1.z, w1.w), t0, t1, t2);
                a[2] = pick53(roll53(w2.x, w2.y), t0, t1, t2); b[2] = pick53(roll53(w2.z, w2.w), t0, t1, t2);
                a[3] = pick53(roll53(w3.x, w3.y), t0, t1, t2); b[3] = pick53(roll53(w3.z, w3.w), t0, t1, t2);
                if (frow) { // forced genotype 1/1, pop_factory.py:495-499
                    const uint32_t f = frow[g];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((f >> k) & 1u) a[k] = b[k] = 1u;
                }
                uint32_t pk = 0;
                if (cls == 0u && 4u * g + 4u < S) { // complete autosome group that does not hold the row's last sample
                    if (TEXT) {
                        const uint4 w = make_uint4(DIPLOID_CELL(a[0], b[0], 9u), DIPLOID_CELL(a[1], b[1], 9u),
                                                   DIPLOID_CELL(a[2], b[2], 9u), DIPLOID_CELL(a[3], b[3], 9u));
                        store_unaligned<16>(cells + 16ull * g, &w);
                    }
                    pk = nz(a[0]) | nz(b[0]) << 1 | nz(a[1]) << 2 | nz(b[1]) << 3 | nz(a[2]) << 4 | nz(b[2]) << 5 |
                         nz(a[3]) << 6 | nz(b[3]) << 7;
                } else {
                    const uint32_t male = p.male_nib[g]; // only the sex-chromosome rules need it
                    uint8_t* dst = cells + (cls == 0u ? 16ull * g : (cls == 1u ? (uint64_t)p.xoff[g] : 8ull * g));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t i = 4u * g + (uint32_t)k;
                        if (i < S) {
                            const uint32_t sep = (i == S - 1u) ? 0x0Au : 0x09u;
                            const bool is_male = (male >> k) & 1u;
                            const bool missing = cls == 2u && !is_male;              // pop_factory.py:481-484
                            const bool haploid = cls >= 2u || (cls == 1u && is_male); // common/snp.py:102-109
                            if (missing) {
                                if (TEXT) { const uint16_t h = (uint16_t)HAPLOID_CELL(0x2Eu, sep); store_unaligned<2>(dst, &h); dst += 2; }
                            } else if (haploid) {
                                pk |= nz(a[k]) << (2 * k);
                                if (TEXT) { const uint16_t h = (uint16_t)HAPLOID_CELL(0x30u + a[k], sep); store_unaligned<2>(dst, &h); dst += 2; }
                            } else {
                                pk |= (nz(a[k]) | nz(b[k]) << 1) << (2 * k);
                                if (TEXT) { const uint32_t w = DIPLOID_CELL(a[k], b[k], sep); store_unaligned<4>(dst, &w); dst += 4; }
                            }
                        }
                    }
                }
                if (PACKED) prow[g] = (uint8_t)pk;
            }
            __builtin_amdgcn_wave_barrier();
            tail += need;
            if (tail >= MT_RING) tail -= MT_RING;
            avail -= need;
            consumed += need;
        }
        if (PACKED)
            for (uint32_t g = G + lane; g < p.W * 4u; g += 64u) prow[g] = 0; // pad bytes of the packed row
    }
}

} // namespace dnaf
