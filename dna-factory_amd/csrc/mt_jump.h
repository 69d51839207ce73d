// mt_jump.h -- host side of the numpy-compatible MT19937 mode: jump-ahead polynomials.
//
// The reference draws every row from numpy's global MT19937 (pop_factory.py:477 after :235), one
// sequential stream.  To let thousands of waves start anywhere in that stream, the generator state
// is advanced by J steps with the polynomial method (Haramoto, Matsumoto, Nishimura, Panneton,
// L'Ecuyer 2008): if phi is the characteristic polynomial of the one-step transition F and
// g(x) = x^J mod phi(x) = sum c_i x^i, then F^J = sum c_i F^i, and F^i applied to a state is just the
// window of the word sequence i steps later.  So word q of the jumped window is the XOR, over the set
// coefficients c_i, of sequence word i + q  -- a correlation the GPU evaluates (mt_kernels.hip.h).
//
// This file computes phi (Berlekamp-Massey on an output bit sequence) and g (square-and-multiply-by-x
// modulo phi); polynomials over GF(2) are bit vectors, bit i = coefficient of x^i.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

namespace dnaf {

constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_DEG = 19937;                     // degree of the characteristic polynomial
constexpr int MT_POLY_WORDS = (MT_DEG + 64) / 64; // 312 words hold degrees 0..19967

struct MtPoly {
    uint64_t w[MT_POLY_WORDS];
};

inline void mt_regenerate(uint32_t* k)
{
    for (int i = 0; i < MT_N; ++i) {
        const uint32_t y = (k[i] & 0x80000000u) | (k[(i + 1) % MT_N] & 0x7fffffffu);
        k[i] = k[(i + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}

inline void mt_init_genrand(uint32_t seed, uint32_t* k)
{
    for (int i = 0; i < MT_N; ++i) {
        k[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1u;
    }
}

// phi(x): characteristic polynomial of the one-step (one word) transition, by Berlekamp-Massey on
// 2*19937 + 64 bits of the sequence "top bit of the untempered word".  Any non-zero output bit
// sequence has phi as its minimal polynomial because phi is irreducible.
inline void mt_characteristic_polynomial(MtPoly* phi)
{
    const int N = 2 * MT_DEG + 64;
    std::vector<uint8_t> s((size_t)N);
    {
        uint32_t k[MT_N];
        mt_init_genrand(5489u, k);
        int pos = MT_N;
        for (int n = 0; n < N; ++n) {
            if (pos >= MT_N) { mt_regenerate(k); pos = 0; }
            s[(size_t)n] = (uint8_t)(k[pos++] >> 31);
        }
    }
    const int W = (MT_DEG + 2 + 63) / 64 + 1;
    std::vector<uint64_t> C((size_t)W, 0), B((size_t)W, 0), T((size_t)W, 0), S((size_t)W, 0);
    C[0] = B[0] = 1;
    int L = 0, m = 1;
    for (int n = 0; n < N; ++n) {
        // S: bit i = s[n - i]
        uint64_t carry = s[(size_t)n];
        for (int w = 0; w < W; ++w) { const uint64_t nc = S[(size_t)w] >> 63; S[(size_t)w] = (S[(size_t)w] << 1) | carry; carry = nc; }
        uint64_t acc = 0;
        for (int w = 0; w <= L / 64 && w < W; ++w) acc ^= C[(size_t)w] & S[(size_t)w];
        const int d = __builtin_parityll(acc);
        if (!d) { ++m; continue; }
        const bool grow = 2 * L <= n;
        if (grow) T = C;
        // C ^= B << m
        const int ws = m / 64, bs = m % 64;
        for (int w = W - 1; w >= ws; --w) {
            uint64_t v = B[(size_t)(w - ws)] << bs;
            if (bs && w - ws - 1 >= 0) v |= B[(size_t)(w - ws - 1)] >> (64 - bs);
            C[(size_t)w] ^= v;
        }
        if (grow) { L = n + 1 - L; B = T; m = 1; }
        else ++m;
    }
    // connection polynomial C(x) = 1 + c1 x + ... + cL x^L, s[n] = sum c_i s[n-i];  phi(x) = x^L C(1/x)
    memset(phi, 0, sizeof *phi);
    for (int i = 0; i <= L; ++i)
        if ((C[(size_t)(i / 64)] >> (i % 64)) & 1ull) {
            const int e = L - i;
            phi->w[e / 64] |= 1ull << (e % 64);
        }
}

struct MtJumper {
    MtPoly phi;
    // phi without its leading term, for the multiply-by-x step
    bool ready = false;
    void init()
    {
        if (ready) return;
        mt_characteristic_polynomial(&phi);
        ready = true;
    }
    // r(x) = r(x)^2 mod phi
    void square_mod(MtPoly* r) const
    {
        // bit-spreading table, built once (thread-safe static initialisation)
        struct Spread {
            uint16_t v[256];
            Spread()
            {
                for (int b = 0; b < 256; ++b) {
                    uint16_t x = 0;
                    for (int i = 0; i < 8; ++i) x |= (uint16_t)(((b >> i) & 1) << (2 * i));
                    v[b] = x;
                }
            }
        };
        static const Spread spread_table;
        const uint16_t* spread = spread_table.v;
        uint64_t t[2 * MT_POLY_WORDS];
        for (int w = 0; w < MT_POLY_WORDS; ++w) {
            const uint64_t v = r->w[w];
            uint64_t lo = 0, hi = 0;
            for (int b = 0; b < 4; ++b) {
                lo |= (uint64_t)spread[(v >> (8 * b)) & 0xFF] << (16 * b);
                hi |= (uint64_t)spread[(v >> (32 + 8 * b)) & 0xFF] << (16 * b);
            }
            t[2 * w] = lo;
            t[2 * w + 1] = hi;
        }
        // reduce degrees 2*19936 .. 19937 with phi (whose top term x^19937 clears the bit)
        for (int i = 2 * (MT_DEG - 1); i >= MT_DEG; --i) {
            if (!((t[i / 64] >> (i % 64)) & 1ull)) continue;
            const int sh = i - MT_DEG, ws = sh / 64, bs = sh % 64;
            for (int w = 0; w < MT_POLY_WORDS; ++w) {
                const uint64_t v = phi.w[w];
                t[w + ws] ^= v << bs;
                if (bs) t[w + ws + 1] ^= v >> (64 - bs);
            }
        }
        memcpy(r->w, t, sizeof r->w);
    }
    // r(x) = x * r(x) mod phi
    void mulx_mod(MtPoly* r) const
    {
        uint64_t carry = 0;
        for (int w = 0; w < MT_POLY_WORDS; ++w) {
            const uint64_t nc = r->w[w] >> 63;
            r->w[w] = (r->w[w] << 1) | carry;
            carry = nc;
        }
        if ((r->w[MT_DEG / 64] >> (MT_DEG % 64)) & 1ull)
            for (int w = 0; w < MT_POLY_WORDS; ++w) r->w[w] ^= phi.w[w];
    }
    // g(x) = x^J mod phi(x), left-to-right square and multiply-by-x
    void xpow_mod(uint64_t J, MtPoly* g) const
    {
        memset(g, 0, sizeof *g);
        g->w[0] = 1;
        if (J == 0) return;
        for (int b = 63 - __builtin_clzll(J); b >= 0; --b) {
            square_mod(g);
            if ((J >> b) & 1ull) mulx_mod(g);
        }
    }
};

// Reference evaluation on the host (tests and tiny jumps): out[q] = XOR_{c_i = 1} x[i + q + 1] for
// q = 0..623, where x[0..623] = base window and x continues with the recurrence.  With g = x^(J-1)
// mod phi this is the window J words after the base (every term has sequence index >= 1, so the 31
// low bits of x[0], which are not part of the generator state, never enter).
inline void mt_jump_window_host(const uint32_t* base, const MtPoly& g, uint32_t* out)
{
    std::vector<uint32_t> x((size_t)MT_DEG + MT_N + 2);
    memcpy(x.data(), base, MT_N * sizeof(uint32_t));
    for (size_t k = MT_N; k < x.size(); ++k) {
        const uint32_t y = (x[k - MT_N] & 0x80000000u) | (x[k - MT_N + 1] & 0x7fffffffu);
        x[k] = x[k - MT_N + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    memset(out, 0, MT_N * sizeof(uint32_t));
    for (int i = 0; i < MT_DEG; ++i)
        if ((g.w[i / 64] >> (i % 64)) & 1ull)
            for (int q = 0; q < MT_N; ++q) out[q] ^= x[(size_t)(i + q + 1)];
}

} // namespace dnaf
