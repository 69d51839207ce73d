// lds_rates.hip -- LDS cycles per wave-instruction of the accesses the text realignment could use: ds_read_b128 at byte
// offsets 0 / 1 / 4 / 8 from 16-byte alignment, ds_read2_b32 + ds_read2_b32 + ds_read_b32 (5 dwords), ds_write_b128 aligned
// and unaligned.  All waves of a CU hammer the LDS; time per instruction per CU = kernel time / (instructions per CU).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_rates lds_rates.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t __attribute__((address_space(3))) * lds_u32;

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t iters, uint32_t mis)
{
    __shared__ __attribute__((aligned(128))) uint8_t buf[4][4096 + 256];
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < sizeof buf / 4; i += 256) ((uint32_t*)buf)[i] = i;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(buf[w]) + 16u * lane + mis; // LDS byte address
    u32x4 acc = {0, 0, 0, 0};
    uint32_t a5 = 0;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t addr = base + ((i + u) & 1u) * 1024u;
            if (MODE == 0) { u32x4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr)); acc += v; }
            if (MODE == 1) { // 5 dwords: two ds_read2_b32 and a ds_read_b32
                uint32_t d0, d1, d2, d3, d4; uint64_t p0, p1;
                asm volatile("ds_read2_b32 %0, %3 offset1:1\n\tds_read2_b32 %1, %3 offset0:2 offset1:3\n\tds_read_b32 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(p0), "=v"(p1), "=v"(d4) : "v"(addr));
                d0 = (uint32_t)p0; d1 = (uint32_t)(p0 >> 32); d2 = (uint32_t)p1; d3 = (uint32_t)(p1 >> 32);
                acc.x += d0 ^ d1; acc.y += d2 ^ d3; a5 += d4;
            }
            if (MODE == 2) { asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(acc) : "memory"); acc.x += 1; }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w ^ a5) == 0x12345u) out[threadIdx.x] = acc.x;
}

int main()
{
    uint32_t* out; CK(hipMalloc((void**)&out, 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const uint32_t iters = 2000, nb = 256 * 7; // 7 workgroups of 4 waves per CU
    const char* names[3] = {"ds_read_b128", "ds_read2_b32 x2 + ds_read_b32 (5 dwords)", "ds_write_b128"};
    for (int mode = 0; mode < 3; ++mode)
        for (uint32_t mis : {0u, 1u, 2u, 4u, 8u, 12u, 13u}) {
            if (mode == 1 && (mis & 3u)) continue;
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CK(hipEventRecord(a, 0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, out, iters, mis);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, out, iters, mis);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, out, iters, mis);
                CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
            }
            const double inst_per_cu = 7.0 * 4 * iters * 8; // wave-instructions (groups of instructions for mode 1) per CU
            printf("%-44s misaligned by %2u B: %7.3f ms  %6.1f ns per wave-access per CU (= %5.1f cycles at 2.1 GHz)\n", names[mode], mis, best,
                   best * 1e6 / inst_per_cu, best * 1e6 / inst_per_cu * 2.1);
        }
    return 0;
}
