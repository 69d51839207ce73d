// gtz_host.h -- host side of the genotype-domain BGZF encoder (gtz_kernels.hip.h): Huffman tables for a
// deflate "dynamic" block with the full length / distance alphabets, the block header with run-length
// coded code lengths (RFC 1951, 3.2.7), and the CRC-32 constants that let a workgroup compute the CRC
// of VCF text it never reads.
//
// Why a second encoder: the text-domain encoder (bgzf_kernels.hip.h) reads the 80 KB/row text twice and
// only knows matches at distance 4.  A member that lies entirely in biallelic autosome rows is a pure
// function of the 2-bit genotype matrix and the rows' fixed columns, so it can be encoded from 1/16 of
// the bytes, with LZ77 matches found by direct-indexed tables over 5-cell keys (10 bits) -- the output
// is 30-35 % smaller at 20k samples.  Members that touch X / Y / MT or multi-allelic rows stay with the
// text-domain encoder.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "bgzf_host.h"

namespace dnaf {

constexpr uint32_t GTZ_NCLASS = 8;      // Huffman table sets, chosen per member by the minor-allele frequency of its first row
constexpr uint32_t GTZ_NSYM = 320;      // histogram layout per class: [0,286) literal/length symbols, [286,316) distance symbols
constexpr uint32_t GTZ_DIST0 = 286;
constexpr uint32_t GTZ_HDR_WORDS = 64;  // header bits of a class (<= 2048)
constexpr uint32_t GTZ_CHUNK_CELLS = 64; // cells (= 256 text bytes) tokenised by one thread
constexpr uint32_t GTZ_GAP_STRIDE = 264; // entries of the gap table (gap lengths 0..256)

// Entries are (bit pattern << 5) | bit count, patterns in LSB-first stream order.
struct GtzClassTables {
    uint32_t lit[256];   // literal byte -> code
    uint32_t len[256];   // match length L (index L - 3) -> length code + extra bits (<= 20 bits)
    uint32_t dist[32];   // distance symbol -> code (extra bits are appended by the kernel)
    uint32_t eob;
    uint32_t hdr_nbits;
    uint32_t hdr[GTZ_HDR_WORDS];
    // gap[L]: the L bytes between two events of a chunk (gtz_tokens.hip.h), every one repeating the byte four back:
    // L >= 3 one match (length code + extra bits + the code of distance 4), L = 0 nothing; L = 1 is the separator's
    // literal, '/' behind an allele a (this table) or the tab behind an allele b (the kernel takes lit['\t']).  At most 27 bits.
    uint32_t gap[GTZ_GAP_STRIDE];
    uint32_t len4[GTZ_CHUNK_CELLS + 8]; // LZ77 match of f cells (4 f bytes) -> length code + extra bits
    uint32_t unit_ok;    // 1: a gap code and the literal '0' / '1' behind it fit 32 bits together (what the kernel emits as one unit)
};

struct GtzTables {
    GtzClassTables cls[GTZ_NCLASS];
    uint32_t crc_table[256];                 // byte-wise CRC-32 table (for the rows' fixed columns)
    uint32_t crc_base[GTZ_CHUNK_CELLS + 1];  // CRC-32 of n cells "0/0\t"
    uint32_t crc_unit_a[GTZ_CHUNK_CELLS];    // change of a chunk's CRC when the first allele of the cell e cells before its end is '1'
    uint32_t crc_unit_b[GTZ_CHUNK_CELLS];    // ... the second allele
    uint32_t crc_shift_chunk[1024];          // x^(8*64*j) mod P
    uint32_t crc_shift_byte[256];            // x^(8*r) mod P
    // the two unit tables folded by pairs of cells: crc_nib[P][v] = change of a chunk's CRC when the four alleles of the
    // cells 2 P + 1 and 2 P before its end are v (bit 0 / 1: alleles a / b of the earlier cell, bit 2 / 3: of the later one)
    uint32_t crc_nib[(GTZ_CHUNK_CELLS / 2) * 16];
};

inline uint32_t crc32_bytes(const uint32_t* table, const uint8_t* p, size_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return ~c;
}

// code-length sequence -> (symbol, extra value) pairs of the code-length alphabet with repeats 16 / 17 / 18
inline void rle_code_lengths(const std::vector<uint8_t>& lens, std::vector<std::pair<uint8_t, uint8_t>>& out)
{
    size_t i = 0;
    while (i < lens.size()) {
        const uint8_t v = lens[i];
        size_t j = i;
        while (j < lens.size() && lens[j] == v) ++j;
        size_t run = j - i;
        if (v == 0) {
            while (run >= 11) { const size_t r = run > 138 ? 138 : run; out.push_back({18, (uint8_t)(r - 11)}); run -= r; }
            if (run >= 3) { out.push_back({17, (uint8_t)(run - 3)}); run = 0; }
            while (run--) out.push_back({0, 0});
        } else {
            out.push_back({v, 0});
            --run;
            while (run >= 3) { const size_t r = run > 6 ? 6 : run; out.push_back({16, (uint8_t)(r - 3)}); run -= r; }
            while (run--) out.push_back({v, 0});
        }
        i = j;
    }
}

// One class: litlen_hist[286], dist_hist[30] -> codes and the block header.  Every symbol gets a code.
inline void gtz_build_class(const uint64_t* litlen_hist, const uint64_t* dist_hist, GtzClassTables* t)
{
    std::vector<uint64_t> lf(BGZF_NLIT), df(30);
    for (uint32_t i = 0; i < BGZF_NLIT; ++i) lf[i] = litlen_hist[i] * 16 + 1; // smoothing well below one real count
    for (uint32_t i = 0; i < 30; ++i) df[i] = dist_hist[i] * 16 + 1;
    std::vector<uint8_t> ll, dl;
    huffman_lengths(lf, 15, ll);
    huffman_lengths(df, 15, dl);
    std::vector<uint32_t> lc, dc;
    canonical_codes(ll, lc);
    canonical_codes(dl, dc);
    for (int i = 0; i < 256; ++i) t->lit[i] = (lc[(size_t)i] << 5) | ll[(size_t)i];
    t->eob = (lc[256] << 5) | ll[256];
    for (uint32_t L = 3; L <= 258; ++L) {
        uint32_t sym, eb, ev;
        length_symbol(L, &sym, &eb, &ev);
        t->len[L - 3] = ((lc[sym] | (ev << ll[sym])) << 5) | (ll[sym] + eb);
    }
    memset(t->dist, 0, sizeof t->dist);
    for (uint32_t s = 0; s < 30; ++s) t->dist[s] = (dc[s] << 5) | dl[s];
    memset(t->gap, 0, sizeof t->gap);
    memset(t->len4, 0, sizeof t->len4);
    uint32_t widest = 0;
    for (uint32_t L = 3; L <= 256; ++L) {
        const uint32_t le = t->len[L - 3], ln = le & 31u, dn = dl[3];
        const uint32_t n = ln + dn;
        widest = n > widest ? n : widest;
        const uint32_t e = n <= 27 ? (((le >> 5) | (dc[3] << ln)) << 5) | n : 0u;
        t->gap[L] = e;
    }
    t->gap[1] = t->lit[(unsigned char)'/'];
    for (uint32_t f = 1; f <= GTZ_CHUNK_CELLS; ++f) t->len4[f] = t->len[4 * f - 3];
    const uint32_t lit_w = std::max(std::max(ll[(unsigned char)'0'], ll[(unsigned char)'1']), std::max(ll[(unsigned char)'/'], ll[(unsigned char)'\t']));
    t->unit_ok = widest <= 27 && widest + lit_w <= 32 ? 1u : 0u;

    // header: BFINAL=1, BTYPE=10, HLIT=286, HDIST=30, code lengths run-length coded with their own Huffman code
    std::vector<uint8_t> all(ll);
    all.insert(all.end(), dl.begin(), dl.end());
    std::vector<std::pair<uint8_t, uint8_t>> rle;
    rle_code_lengths(all, rle);
    std::vector<uint64_t> cf(19, 0);
    for (auto& e : rle) cf[e.first]++;
    std::vector<uint8_t> cl;
    huffman_lengths(cf, 7, cl);
    std::vector<uint32_t> cc;
    canonical_codes(cl, cc);
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cl[order[hclen - 1]] == 0) --hclen;
    BitWriter w;
    w.put(1, 1);
    w.put(2, 2);
    w.put(BGZF_NLIT - 257, 5);
    w.put(30 - 1, 5);
    w.put((uint32_t)hclen - 4, 4);
    for (int i = 0; i < hclen; ++i) w.put(cl[order[i]], 3);
    for (auto& e : rle) {
        w.put(cc[e.first], cl[e.first]);
        if (e.first == 16) w.put(e.second, 2);
        else if (e.first == 17) w.put(e.second, 3);
        else if (e.first == 18) w.put(e.second, 7);
    }
    t->hdr_nbits = w.nbits;
    memset(t->hdr, 0, sizeof t->hdr);
    for (size_t i = 0; i < w.words.size() && i < GTZ_HDR_WORDS; ++i) t->hdr[i] = w.words[i];
    if (w.nbits > GTZ_HDR_WORDS * 32) t->hdr_nbits = 0; // cannot happen: 316 lengths of <= 7+7 bits < 2048 x 2; checked by the caller
}

// hist: [GTZ_NCLASS][GTZ_NSYM] token counts of the sampled members.  A class with few samples borrows
// the pooled histogram, so its table is sensible even when the sample held none of its members.
inline bool gtz_build_tables(const uint32_t* hist, GtzTables* t)
{
    uint64_t pooled[GTZ_NSYM] = {0}, pooled_total = 0;
    for (uint32_t c = 0; c < GTZ_NCLASS; ++c)
        for (uint32_t s = 0; s < GTZ_NSYM; ++s) { pooled[s] += hist[c * GTZ_NSYM + s]; pooled_total += hist[c * GTZ_NSYM + s]; }
    for (uint32_t c = 0; c < GTZ_NCLASS; ++c) {
        uint64_t h[GTZ_NSYM], total = 0;
        for (uint32_t s = 0; s < GTZ_NSYM; ++s) { h[s] = hist[c * GTZ_NSYM + s]; total += h[s]; }
        if (total < 20000 && pooled_total) // weight the pooled counts down to about 20,000 tokens
            for (uint32_t s = 0; s < GTZ_NSYM; ++s) h[s] += pooled[s] * 20000 / (pooled_total + 1);
        gtz_build_class(h, h + GTZ_DIST0, &t->cls[c]);
        if (t->cls[c].hdr_nbits == 0) return false;
    }
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t v = i;
        for (int k = 0; k < 8; ++k) v = (v & 1u) ? (v >> 1) ^ CRC_POLY : v >> 1;
        t->crc_table[i] = v;
    }
    uint8_t cells[4 * GTZ_CHUNK_CELLS];
    for (uint32_t j = 0; j < GTZ_CHUNK_CELLS; ++j) { cells[4 * j] = '0'; cells[4 * j + 1] = '/'; cells[4 * j + 2] = '0'; cells[4 * j + 3] = '\t'; }
    for (uint32_t n = 0; n <= GTZ_CHUNK_CELLS; ++n) t->crc_base[n] = crc32_bytes(t->crc_table, cells, 4 * n);
    const uint32_t full = t->crc_base[GTZ_CHUNK_CELLS];
    for (uint32_t e = 0; e < GTZ_CHUNK_CELLS; ++e) {
        const uint32_t j = GTZ_CHUNK_CELLS - 1 - e;
        cells[4 * j] = '1';
        t->crc_unit_a[e] = crc32_bytes(t->crc_table, cells, sizeof cells) ^ full;
        cells[4 * j] = '0';
        cells[4 * j + 2] = '1';
        t->crc_unit_b[e] = crc32_bytes(t->crc_table, cells, sizeof cells) ^ full;
        cells[4 * j + 2] = '0';
    }
    for (uint32_t P = 0; P < GTZ_CHUNK_CELLS / 2; ++P)
        for (uint32_t v = 0; v < 16; ++v)
            t->crc_nib[P * 16 + v] = ((v & 1u) ? t->crc_unit_a[2 * P + 1] : 0u) ^ ((v & 2u) ? t->crc_unit_b[2 * P + 1] : 0u) ^
                                     ((v & 4u) ? t->crc_unit_a[2 * P] : 0u) ^ ((v & 8u) ? t->crc_unit_b[2 * P] : 0u);
    for (uint32_t j = 0; j < 1024; ++j) t->crc_shift_chunk[j] = crc_x8n((uint64_t)64 * j);
    for (uint32_t j = 0; j < 256; ++j) t->crc_shift_byte[j] = crc_x8n(j);
    return true;
}

} // namespace dnaf
