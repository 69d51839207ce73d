// microbench.hip -- gfx950 probes that size the row generator's two ceilings:
// Philox4x32-10 issue rate (integer VALU) and streaming-store rate vs byte misalignment.
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench microbench.hip ; run: ./microbench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int VARIANT>
__device__ __forceinline__ uint4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        if (VARIANT == 0) {
            hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0;
            hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
        } else {
            uint64_t p0, p1;
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p0) : "v"(c0), "s"(0xD2511F53u) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p1) : "v"(c2), "s"(0xCD9E8D57u) : "vcc");
            hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
        }
        if (VARIANT == 2) {
            c0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96); c1 = lo1;
            c2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96); c3 = lo0;
        } else { c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0; }
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_philox(uint32_t* out, uint32_t iters, uint32_t k0, uint32_t k1)
{
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        uint4 r = philox<VARIANT>(tid, i, 7, 9, k0, k1);
        acc ^= r.x ^ r.y ^ r.z ^ r.w;
    }
    if (acc == 0x12345678u) out[tid] = acc; // keep live, practically never stores
}

// each lane stores 16 B per iteration; a wave covers 1 KiB contiguous; base misaligned by m bytes
__global__ __launch_bounds__(256) void k_store16(uint8_t* dst, uint64_t bytes_per_wave, uint32_t m)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint8_t* p = dst + wave * bytes_per_wave + m + lane * 16;
    uint32_t v[4] = {0x092F3030u, 0x092F3031u, 0x092F3130u, 0x092F3131u};
    for (uint64_t o = 0; o + 1024 <= bytes_per_wave - 16; o += 1024) {
        v[0] += lane;
        __builtin_memcpy(p + o, v, 16);
    }
}

// text-stream shape of the row kernel: "rows" of row_bytes (any value, so every row has its own byte alignment),
// each cut into segments of seg_steps KiB, one wave per segment; the first pfx bytes of a row are skipped
__global__ __launch_bounds__(256) void k_store_rows(uint8_t* dst, uint64_t row_bytes, uint64_t n_rows, uint32_t nseg, uint32_t seg_steps, uint32_t pfx)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t item = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint64_t row = item / nseg;
    const uint32_t seg = (uint32_t)(item % nseg);
    if (row >= n_rows) return;
    const uint32_t steps = (uint32_t)((row_bytes - pfx) / 1024);
    const uint32_t s0 = seg * seg_steps, s1 = s0 + seg_steps < steps ? s0 + seg_steps : steps;
    uint8_t* p = dst + row * row_bytes + pfx + lane * 16 + (uint64_t)s0 * 1024;
    uint32_t v[4] = {0x092F3030u, 0x092F3031u, 0x092F3130u, 0x092F3131u};
    for (uint32_t s = s0; s < s1; ++s, p += 1024) {
        v[0] += lane;
        __builtin_memcpy(p, v, 16);
    }
}

typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
// same as k_store16 with non-temporal stores (global_store_dwordx4 ... nt)
__global__ __launch_bounds__(256) void k_store16_nt(uint8_t* dst, uint64_t bytes_per_wave, uint32_t m)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint8_t* p = dst + wave * bytes_per_wave + m + lane * 16;
    u32x4_a1 v = {0x092F3030u, 0x092F3031u, 0x092F3130u, 0x092F3131u};
    for (uint64_t o = 0; o + 1024 <= bytes_per_wave - 16; o += 1024) {
        v.x += lane;
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4_a1*>(p + o));
    }
}

__global__ __launch_bounds__(256) void k_store4(uint8_t* dst, uint64_t bytes_per_wave, uint32_t m)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint8_t* p = dst + wave * bytes_per_wave + m + lane * 4;
    uint32_t v = 0x092F3030u;
    for (uint64_t o = 0; o + 256 <= bytes_per_wave - 16; o += 256) {
        v += lane;
        __builtin_memcpy(p + o, &v, 4);
    }
}

__global__ __launch_bounds__(256) void k_store1(uint8_t* dst, uint64_t bytes_per_wave)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint8_t* p = dst + wave * bytes_per_wave + lane;
    uint8_t v = 3;
    for (uint64_t o = 0; o + 64 <= bytes_per_wave; o += 64) { v += lane; p[o] = v; }
}

// 64 contiguous bytes per wave-instruction, written by 16 lanes as dwords / by 4 lanes as dwordx4
__global__ __launch_bounds__(256) void k_store_sparse(uint8_t* dst, uint64_t bytes_per_wave, int lanes)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint8_t* p = dst + wave * bytes_per_wave;
    uint32_t v = 3;
    for (uint64_t o = 0; o + 64 <= bytes_per_wave; o += 64) {
        v += lane;
        if (lanes == 16) { if ((lane & 3) == 0) *reinterpret_cast<uint32_t*>(p + o + lane) = v; }
        else if (lanes == 4) { if ((lane & 15) == 0) *reinterpret_cast<uint4*>(p + o + lane) = make_uint4(v, v, v, v); }
        else p[o + lane] = (uint8_t)v;
    }
}

int main()
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t* d_out; CK(hipMalloc(&d_out, 1 << 26));
    float ms;
    // ---- Philox rate ----
    for (int variant = 0; variant < 3; ++variant) {
        for (int blocks_per_cu : {4, 8}) {
            const uint32_t blocks = 256 * blocks_per_cu, iters = 2000;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (variant == 0) hipLaunchKernelGGL(k_philox<0>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1u, 2u);
                else if (variant == 1) hipLaunchKernelGGL(k_philox<1>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1u, 2u);
                else hipLaunchKernelGGL(k_philox<2>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1u, 2u);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double calls = (double)blocks * 256 * iters;
            printf("philox variant=%s blocks/CU=%d: %.3f ms  %.3e calls/s  %.3e draws32/s\n", variant == 0 ? "mulhi+mullo" : variant == 1 ? "mad_u64_u32" : "mad_u64_u32+bitop3",
                   blocks_per_cu, ms, calls / (ms * 1e-3), 4 * calls / (ms * 1e-3));
        }
    }
    // ---- store rate vs misalignment ----
    const uint64_t total = 16ull << 30; // 16 GiB written per launch
    uint8_t* d_buf; CK(hipMalloc(&d_buf, total + 4096));
    for (uint64_t row_bytes : {81920ull, 80037ull, 80000ull}) {
        const uint64_t n_rows = total / row_bytes;
        for (uint32_t seg_steps : {32u, 80u}) {
            for (uint32_t pfx : {0u, 37u}) {
                const uint32_t steps = (uint32_t)((row_bytes - pfx) / 1024), nseg = (steps + seg_steps - 1) / seg_steps;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_store_rows, dim3((uint32_t)((n_rows * nseg + 3) / 4)), dim3(256), 0, 0, d_buf, row_bytes, n_rows, nseg, seg_steps, pfx);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
                }
                printf("rows of %llu B, %u KiB per wave, first %u B skipped: %.3f ms  %.1f GB/s\n", (unsigned long long)row_bytes, seg_steps, pfx, ms,
                       (double)n_rows * steps * 1024 / (ms * 1e-3) / 1e9);
            }
        }
    }
    for (uint64_t bpw : {32768ull, 1ull << 20}) {
        const uint64_t waves = total / bpw;
        const uint32_t blocks = (uint32_t)(waves / 4);
        for (uint32_t m : {0u, 4u, 8u, 1u, 2u, 3u, 5u, 13u}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_store16, dim3(blocks), dim3(256), 0, 0, d_buf, bpw, m);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double bytes = (double)waves * ((bpw - 16) / 1024) * 1024;
            printf("store16 bytes/wave=%llu misalign=%2u: %.3f ms  %.1f GB/s\n", (unsigned long long)bpw, m, ms, bytes / (ms * 1e-3) / 1e9);
        }
    }
    {
        const uint64_t bpw = 32768, waves = total / bpw; const uint32_t blocks = (uint32_t)(waves / 4);
        for (uint32_t m : {0u, 4u, 5u}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_store16_nt, dim3(blocks), dim3(256), 0, 0, d_buf, bpw, m);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double bytes = (double)waves * ((bpw - 16) / 1024) * 1024;
            printf("store16nt bytes/wave=%llu misalign=%2u: %.3f ms  %.1f GB/s\n", (unsigned long long)bpw, m, ms, bytes / (ms * 1e-3) / 1e9);
        }
    }
    {
        const uint64_t bpw = 32768, waves = total / bpw; const uint32_t blocks = (uint32_t)(waves / 4);
        for (uint32_t m : {0u, 1u, 2u}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_store4, dim3(blocks), dim3(256), 0, 0, d_buf, bpw, m);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double bytes = (double)waves * ((bpw - 16) / 256) * 256;
            printf("store4  bytes/wave=%llu misalign=%2u: %.3f ms  %.1f GB/s\n", (unsigned long long)bpw, m, ms, bytes / (ms * 1e-3) / 1e9);
        }
        const uint64_t bpw1 = 8192, waves1 = (1ull << 30) / bpw1;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_store1, dim3((uint32_t)(waves1 / 4)), dim3(256), 0, 0, d_buf, bpw1);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("store1  bytes/wave=%llu: %.3f ms  %.1f GB/s\n", (unsigned long long)bpw1, ms, (double)(1ull << 30) / (ms * 1e-3) / 1e9);
    }
    for (int lanes : {64, 16, 4}) {
        const uint64_t bpw1 = 8192, waves1 = (1ull << 30) / bpw1;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_store_sparse, dim3((uint32_t)(waves1 / 4)), dim3(256), 0, 0, d_buf, bpw1, lanes);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("store 64B/instr by %2d lanes: %.3f ms  %.1f GB/s\n", lanes, ms, (double)(1ull << 30) / (ms * 1e-3) / 1e9);
    }
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
