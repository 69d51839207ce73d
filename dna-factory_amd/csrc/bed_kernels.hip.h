// bed_kernels.hip.h -- PLINK 1 binary genotype rows (.bed, SNP-major) from the packed 2-bit matrix (gfx950).
//
// the reference hands population.vcf.gz to PLINK (plink_analysis.py) which first converts it to
// .bed / .bim / .fam; the packed matrix already is that information.  With A1 = ALT and A2 = REF in the .bim, a
// sample's two bits are (low, high): 11 homozygous A2 (0/0), 01... in PLINK's numbering value 2 = heterozygous,
// 0 = homozygous A1 (1/1), 1 = missing.  Haploid cells (males on X and Y, everyone on MT) are stored as
// homozygous, females on Y as missing.  Pure bit arithmetic on one 32-bit word (16 samples) per thread.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rowgen_kernels.hip.h"

namespace dnaf {

// 16 bits -> the even bit positions of 32 bits
__device__ __forceinline__ uint32_t bed_spread16(uint32_t v)
{
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

__global__ __launch_bounds__(256) void bed_kernel(const uint32_t* packed, const RowRec* recs, const uint8_t* male_nib, uint32_t S, uint32_t W,
                                                  uint32_t Wp, uint32_t G, uint64_t n_rows, uint8_t* bed)
{
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= n_rows * W) return;
    const uint64_t row = idx / W;
    const uint32_t w = (uint32_t)(idx - row * W);
    const uint32_t code = packed[row * Wp + w]; // rows are Wp words apart
    const uint32_t cls = recs[row].cls;
    uint32_t males = 0;
#pragma unroll
    for (uint32_t q = 0; q < 4; ++q)
        if (4u * w + q < G) males |= (uint32_t)(male_nib[4u * w + q] & 0xFu) << (4u * q);
    const uint32_t left = S - 16u * w; // samples from this word on (>= 1)
    const uint32_t valid = left >= 16u ? 0x55555555u : ((1u << (2u * left)) - 1u) & 0x55555555u;
    const uint32_t male = bed_spread16(males) & valid;
    const uint32_t a = code & 0x55555555u, b = (code >> 1) & 0x55555555u;
    const uint32_t hap = cls == 0u ? 0u : cls == 3u ? valid : male; // autosome: nobody; MT: everyone; X, Y: the males (common/snp.py:102-109)
    const uint32_t miss = cls == 2u ? valid & ~male : 0u; // females on Y (pop_factory.py:481-484)
    const uint32_t lo = (hap & ~a) | (~hap & ~(a | b));
    const uint32_t hi = (hap & ~a) | (~hap & ~(a & b));
    const uint32_t out = (miss | (~miss & lo & valid)) | ((~miss & hi & valid) << 1);
    const uint64_t row_bytes = ((uint64_t)S + 3u) / 4u;
    uint8_t* dst = bed + row * row_bytes + 4u * w;
    const uint32_t nbytes = (uint32_t)min((uint64_t)4, row_bytes - 4u * w);
    if (nbytes == 4u) __builtin_memcpy(dst, &out, 4);
    else
        for (uint32_t k = 0; k < nbytes; ++k) dst[k] = (uint8_t)(out >> (8u * k));
}

// packed rows Wp words apart -> dense rows of W words (dnaf_read_packed)
__global__ __launch_bounds__(256) void packed_compact_kernel(const uint32_t* src, uint32_t* dst, uint64_t n_rows, uint32_t W, uint32_t Wp)
{
    const uint64_t total = n_rows * W;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256u) {
        const uint64_t r = i / W;
        dst[i] = src[r * Wp + (i - r * W)];
    }
}

} // namespace dnaf
