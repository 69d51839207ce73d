// bgzf_host.h -- host side of the GPU BGZF encoder: Huffman code construction for a
// deflate "dynamic" block, its header bit string, and the CRC-32 combination constants.
//
// The reference writes population.vcf.gz through Bio.bgzf.BgzfWriter (pop_factory.py:403, :449), i.e.
// zlib on the host.  Here the text never leaves the GPU uncompressed: every 61,440 input bytes become
// one BGZF member encoded by one workgroup (bgzf_kernels.hip.h).  The token alphabet is deliberately
// small -- literals and matches at distance 4 only (VCF GT cells repeat with period 4, or 2 for
// haploid cells, which also has period 4) -- so one Huffman table per tile, built here from a sampled
// histogram, serves every block.
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <queue>
#include <vector>

namespace dnaf {

constexpr uint32_t BGZF_IN = 61440;   // input bytes per BGZF block = 960 threads x 64 bytes
constexpr uint32_t BGZF_CHUNK = 64;   // bytes tokenised by one thread
constexpr uint32_t BGZF_NLIT = 286;   // literal/length symbols 0..285
constexpr uint32_t BGZF_HDR_WORDS = 48;

// What the device needs to encode: entries are (bit pattern << 5) | bit count, patterns already in
// the LSB-first order of a deflate stream.
struct BgzfTables {
    uint32_t lit[256];            // literal byte -> code
    uint32_t match[256];          // match length L (index L - 3) -> length code + extra bits + the 1-bit distance code
    uint32_t eob;                 // end-of-block code
    uint32_t hdr_nbits;           // bits of the block header (BFINAL, BTYPE, code descriptions)
    uint32_t hdr[BGZF_HDR_WORDS]; // header bits, LSB first
    uint32_t crc_table[4][256];   // slicing-by-4 CRC-32 tables
    uint32_t crc_shift_chunk[1024]; // x^(8*64*j) mod P, j = 0..1023 (zlib's reflected convention)
    uint32_t crc_shift_byte[256];  // x^(8*r) mod P, r = 0..255
};

inline uint32_t bit_reverse(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

// deflate length symbol / extra bits of a match length 3..258 (RFC 1951, 3.2.5)
inline void length_symbol(uint32_t len, uint32_t* sym, uint32_t* extra_bits, uint32_t* extra_val)
{
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t ebits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    int s = 28;
    while (s > 0 && base[s] > len) --s;
    *sym = 257u + (uint32_t)s;
    *extra_bits = ebits[s];
    *extra_val = len - base[s];
}

// Huffman code lengths (<= max_len) for freq[0..n); every symbol with freq > 0 gets a code.
inline void huffman_lengths(const std::vector<uint64_t>& freq_in, int max_len, std::vector<uint8_t>& len_out)
{
    const size_t n = freq_in.size();
    std::vector<uint64_t> freq(freq_in);
    len_out.assign(n, 0);
    for (;;) {
        struct Node { uint64_t w; int left, right; };
        std::vector<Node> nodes;
        typedef std::pair<uint64_t, int> QE;
        std::priority_queue<QE, std::vector<QE>, std::greater<QE>> q;
        for (size_t i = 0; i < n; ++i)
            if (freq[i]) { nodes.push_back({freq[i], -1, (int)i}); q.push({freq[i], (int)nodes.size() - 1}); }
        if (nodes.size() == 1) { len_out[(size_t)nodes[0].right] = 1; return; }
        while (q.size() > 1) {
            QE a = q.top(); q.pop();
            QE b = q.top(); q.pop();
            nodes.push_back({a.first + b.first, a.second, b.second});
            q.push({a.first + b.first, (int)nodes.size() - 1});
        }
        // depth of every leaf
        std::vector<std::pair<int, int>> stack{{q.top().second, 0}};
        int deepest = 0;
        while (!stack.empty()) {
            auto [id, d] = stack.back();
            stack.pop_back();
            if (nodes[(size_t)id].left < 0) { len_out[(size_t)nodes[(size_t)id].right] = (uint8_t)d; deepest = std::max(deepest, d); }
            else { stack.push_back({nodes[(size_t)id].left, d + 1}); stack.push_back({nodes[(size_t)id].right, d + 1}); }
        }
        if (deepest <= max_len) return;
        for (size_t i = 0; i < n; ++i)  // flatten the distribution and try again
            if (freq[i]) freq[i] = (freq[i] >> 1) | 1;
    }
}

// canonical codes (RFC 1951, 3.2.2), returned bit-reversed for LSB-first emission
inline void canonical_codes(const std::vector<uint8_t>& len, std::vector<uint32_t>& code)
{
    uint32_t bl_count[16] = {0}, next[16] = {0};
    for (uint8_t l : len) bl_count[l]++;
    bl_count[0] = 0;
    uint32_t c = 0;
    for (int b = 1; b < 16; ++b) { c = (c + bl_count[b - 1]) << 1; next[b] = c; }
    code.assign(len.size(), 0);
    for (size_t i = 0; i < len.size(); ++i)
        if (len[i]) code[i] = bit_reverse(next[len[i]]++, len[i]);
}

struct BitWriter {
    std::vector<uint32_t> words;
    uint32_t nbits = 0;
    void put(uint32_t v, uint32_t n)
    {
        for (uint32_t i = 0; i < n; ++i, ++nbits) {
            if ((nbits >> 5) >= words.size()) words.push_back(0);
            words[nbits >> 5] |= ((v >> i) & 1u) << (nbits & 31);
        }
    }
};

// zlib crc32.c conventions: polynomials reflected, x^0 at bit 31
constexpr uint32_t CRC_POLY = 0xEDB88320u;
inline uint32_t crc_multmodp(uint32_t a, uint32_t b)
{
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
// x^(8*n) mod P
inline uint32_t crc_x8n(uint64_t n)
{
    uint32_t p = 1u << 31;       // x^0
    uint32_t sq = 1u << 23;      // x^8
    while (n) {
        if (n & 1) p = crc_multmodp(sq, p);
        sq = crc_multmodp(sq, sq);
        n >>= 1;
    }
    return p;
}

// Build every table from a token histogram: lit_hist[256] literal counts, match_hist[256] counts of
// match lengths 3..258, n_eob end-of-block count.  Every symbol gets a code (counts are smoothed by 1),
// so the tables encode any input, not only data distributed like the sample.
inline void bgzf_build_tables(const uint64_t* lit_hist, const uint64_t* match_hist, uint64_t n_eob, BgzfTables* t)
{
    std::vector<uint64_t> freq(BGZF_NLIT, 1);
    for (int i = 0; i < 256; ++i) freq[(size_t)i] += lit_hist[i];
    freq[256] += n_eob;
    for (uint32_t L = 3; L <= 258; ++L) {
        uint32_t sym, eb, ev;
        length_symbol(L, &sym, &eb, &ev);
        freq[sym] += match_hist[L - 3];
    }
    std::vector<uint8_t> len;
    huffman_lengths(freq, 15, len);
    std::vector<uint32_t> code;
    canonical_codes(len, code);
    for (int i = 0; i < 256; ++i) t->lit[i] = (code[(size_t)i] << 5) | len[(size_t)i];
    t->eob = (code[256] << 5) | len[256];
    for (uint32_t L = 3; L <= 258; ++L) {
        uint32_t sym, eb, ev;
        length_symbol(L, &sym, &eb, &ev);
        // length code, extra bits (LSB first), then the single distance code (1 bit, value 0)
        const uint32_t bits = code[sym] | (ev << len[sym]);
        const uint32_t nb = len[sym] + eb + 1u;
        t->match[L - 3] = (bits << 5) | nb;
    }
    // block header: BFINAL=1, BTYPE=10 (dynamic), HLIT=286, HDIST=4, HCLEN=19; the code-length
    // alphabet uses a flat 4-bit code for lengths 0..15 and no repeat symbols
    BitWriter w;
    w.put(1, 1);
    w.put(2, 2);
    w.put(BGZF_NLIT - 257, 5);
    w.put(4 - 1, 5);
    w.put(19 - 4, 4);
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (int i = 0; i < 19; ++i) w.put(order[i] >= 16 ? 0u : 4u, 3);
    for (uint32_t s = 0; s < BGZF_NLIT; ++s) w.put(bit_reverse(len[s], 4), 4);
    const uint8_t dist_len[4] = {0, 0, 0, 1}; // only distance symbol 3 (= distance 4) is used
    for (int s = 0; s < 4; ++s) w.put(bit_reverse(dist_len[s], 4), 4);
    t->hdr_nbits = w.nbits;
    memset(t->hdr, 0, sizeof t->hdr);
    for (size_t i = 0; i < w.words.size() && i < BGZF_HDR_WORDS; ++i) t->hdr[i] = w.words[i];
    // CRC-32 tables
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
        t->crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int k = 1; k < 4; ++k)
            t->crc_table[k][i] = (t->crc_table[k - 1][i] >> 8) ^ t->crc_table[0][t->crc_table[k - 1][i] & 0xFFu];
    for (uint32_t j = 0; j < 1024; ++j) t->crc_shift_chunk[j] = crc_x8n((uint64_t)BGZF_CHUNK * j);
    for (uint32_t j = 0; j < 256; ++j) t->crc_shift_byte[j] = crc_x8n(j);
}

} // namespace dnaf
