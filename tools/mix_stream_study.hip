// mix_stream_study.hip -- the store streams of the short-row kernel (rowmix_kernel) without its arithmetic.
//
// A wave owns a run of R rows of G groups (row = pfx fixed bytes + 16 G cell bytes, rows back to back, so every row has
// its own byte phase) and walks it in trips of 64 groups like the product kernel: per trip one 16-byte store per lane
// (the cells), one byte store per lane (the packed matrix), and the fixed columns of the rows that begin in the trip
// (byte stores).  Between trips a chain of `work` dependent v_mad_u64_u32 pairs stands in for the Philox group.
// What is measured: how fast this store pattern can go by itself (work = 0) and under how much arithmetic it hides.
//   variants (bit mask): 1 no packed stream, 2 no fixed columns, 4 cells only in whole trips back to back (two trips per
//   store block), 8 fixed columns by the lanes of ONE store per trip (lane = row x byte), 16 packed bytes staged in
//   registers and stored every 4th trip as dwords, 32 no cell stores (35: the arithmetic alone)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mix_stream_study mix_stream_study.hip
// usage: mix_stream_study [S] [rows] [work...]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef u32x4_t __attribute__((aligned(1))) u32x4_u_t;

template <int VAR>
__global__ __launch_bounds__(256) void k_mix(uint8_t* text, uint8_t* packed, uint32_t G, uint32_t pfx, uint32_t R, uint64_t n_rows, uint32_t work)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t run = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint64_t j0 = run * R;
    if (j0 >= n_rows) return;
    const uint32_t nrun = (uint32_t)min((uint64_t)R, n_rows - j0);
    const uint32_t row_bytes = pfx + 16u * G;
    uint8_t* run_text = text + j0 * row_bytes;
    uint8_t* run_packed = packed + j0 * G;
    const uint32_t n = nrun * G;
    uint32_t next_row = 0, next_q = 0;
    uint64_t acc = lane;
    uint32_t stash = 0;
    if (VAR & 4) { // the cells of two trips leave back to back (2 KiB blocks); packed bytes and fixed columns as in variant 0
        for (uint32_t qb = 0; qb < n; qb += 128u) {
            uint32_t dd[2];
            for (uint32_t t = 0; t < 2u; ++t) {
                for (uint32_t w = 0; w < work; ++w) {
                    acc = (uint64_t)(uint32_t)acc * 0xD2511F53u + (acc >> 32);
                    acc = (uint64_t)(uint32_t)acc * 0xCD9E8D57u + (acc >> 32);
                }
                dd[t] = (uint32_t)acc;
            }
            while (next_row < nrun && next_q < qb + 128u) {
                if (lane < pfx) run_text[next_row * row_bytes + lane] = (uint8_t)dd[0];
                ++next_row;
                next_q += G;
            }
#pragma unroll
            for (uint32_t t = 0; t < 2u; ++t) {
                const uint32_t q = qb + 64u * t + lane;
                const uint32_t row = q / G, g = q - row * G;
                if (q < n) {
                    const u32x4_t wv = {dd[t], dd[t] + 1u, dd[t] + 2u, dd[t] + 3u};
                    *reinterpret_cast<u32x4_u_t*>(run_text + row * row_bytes + pfx + 16u * g) = wv;
                }
            }
#pragma unroll
            for (uint32_t t = 0; t < 2u; ++t) {
                const uint32_t q = qb + 64u * t + lane;
                if (q < n && !(VAR & 1)) run_packed[q] = (uint8_t)dd[t];
            }
        }
        if (acc == 0x123456789ull) text[0] = 1;
        return;
    }
    for (uint32_t qb = 0; qb < n; qb += 64u) {
        const uint32_t q = qb + lane;
        const uint32_t row = q / G, g = q - row * G;
        for (uint32_t w = 0; w < work; ++w) { // dependent 64-bit multiply-adds: the Philox stand-in (two per step)
            acc = (uint64_t)(uint32_t)acc * 0xD2511F53u + (acc >> 32);
            acc = (uint64_t)(uint32_t)acc * 0xCD9E8D57u + (acc >> 32);
        }
        const uint32_t d = (uint32_t)acc;
        if (!(VAR & 2)) {
            if (VAR & 8) { // all rows that begin in this trip by one store: lane = (row of the trip, byte)
                const uint32_t first = next_row;
                while (next_row < nrun && next_q < qb + 64u) { ++next_row; next_q += G; }
                const uint32_t nb = next_row - first; // rows beginning here (<= 64 / G + 1)
                const uint32_t per = 64u / max(nb, 1u); // lanes per row
                const uint32_t rr = lane / per, b = lane - rr * per;
                if (nb && rr < nb)
                    for (uint32_t i = b; i < pfx; i += per) run_text[(first + rr) * row_bytes + i] = (uint8_t)d;
            } else {
                while (next_row < nrun && next_q < qb + 64u) {
                    if (lane < pfx) run_text[next_row * row_bytes + lane] = (uint8_t)d;
                    ++next_row;
                    next_q += G;
                }
            }
        }
        if (q < n) {
            const u32x4_t wv = {d, d + 1u, d + 2u, d + 3u};
            if (!(VAR & 32)) *reinterpret_cast<u32x4_u_t*>(run_text + row * row_bytes + pfx + 16u * g) = wv;
            if (!(VAR & 1)) {
                if (VAR & 64) { // as 1 KiB line-aligned blocks every 16th trip (what an LDS ring would emit; the bytes are placeholders)
                    if (((qb >> 6) & 15u) == 15u) {
                        uint8_t* blk = (uint8_t*)(((uintptr_t)(run_packed + (qb - 960u))) & ~(uintptr_t)127u);
                        *reinterpret_cast<u32x4_t*>(blk + 16u * lane) = u32x4_t{d, d, d, d};
                    }
                } else if (VAR & 16) {
                    stash = (stash >> 8) | (d << 24);
                    if (((qb >> 6) & 3u) == 3u) // trips 4k .. 4k+3: lane l holds bytes q = 256 k + 64 t + l, stored as one dword at 256 k + 4 l (a permutation of the bytes: same traffic)
                        *reinterpret_cast<uint32_t __attribute__((aligned(1)))*>(run_packed + (qb - 192u) + 4u * lane) = stash;
                } else
                    run_packed[q] = (uint8_t)d;
            }
        }
    }
    if (acc == 0x123456789ull) text[0] = 1;
}

template <int VAR>
static double run(uint8_t* text, uint8_t* packed, uint32_t G, uint32_t pfx, uint32_t R, uint64_t n_rows, uint32_t work, int reps)
{
    const uint64_t runs = (n_rows + R - 1) / R;
    const dim3 grid((uint32_t)((runs + 3) / 4)), block(256);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int i = 0; i < reps + 2; ++i) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_mix<VAR>, grid, block, 0, 0, text, packed, G, pfx, R, n_rows, work);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float t;
        CHECK(hipEventElapsedTime(&t, a, b));
        if (i >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char** argv)
{
    const uint32_t S = argc > 1 ? (uint32_t)atoi(argv[1]) : 200u;
    const uint64_t n_rows = argc > 2 ? (uint64_t)atoll(argv[2]) : 5000000ull;
    std::vector<uint32_t> works;
    for (int i = 3; i < argc; ++i) works.push_back((uint32_t)atoi(argv[i]));
    if (works.empty()) works = {0, 8, 16, 24};
    const uint32_t G = (S + 3) / 4, pfx = 45;
    const uint32_t R = std::min<uint32_t>(64, std::max<uint32_t>(1, (3200 + G - 1) / G));
    const uint64_t tbytes = n_rows * (uint64_t)(pfx + 16u * G), pbytes = n_rows * (uint64_t)G;
    uint8_t *text, *packed;
    CHECK(hipMalloc(&text, tbytes + 4096));
    CHECK(hipMalloc(&packed, pbytes + 4096));
    printf("S=%u G=%u rows=%llu R=%u: text %.3f GB + packed %.3f GB per launch\n", S, G, (unsigned long long)n_rows, R, tbytes / 1e9, pbytes / 1e9);
    for (uint32_t work : works) {
        const double t0 = run<0>(text, packed, G, pfx, R, n_rows, work, 5), t1 = run<1>(text, packed, G, pfx, R, n_rows, work, 5);
        const double t2 = run<2>(text, packed, G, pfx, R, n_rows, work, 5), t3 = run<3>(text, packed, G, pfx, R, n_rows, work, 5);
        const double t8 = run<8>(text, packed, G, pfx, R, n_rows, work, 5), t16 = run<16>(text, packed, G, pfx, R, n_rows, work, 5);
        const double t24 = run<24>(text, packed, G, pfx, R, n_rows, work, 5), t35 = run<35>(text, packed, G, pfx, R, n_rows, work, 5);
        const double t64 = run<64>(text, packed, G, pfx, R, n_rows, work, 5);
        const double t4 = run<4>(text, packed, G, pfx, R, n_rows, work, 5), t5 = run<5>(text, packed, G, pfx, R, n_rows, work, 5);
        printf("work %2u: all %.3f ms (%.0f GB/s) | no packed %.3f | no fixed columns %.3f | cells only %.3f | columns by one store %.3f | packed as dwords %.3f | both %.3f | no stores %.3f | 2 KiB blocks %.3f, no packed %.3f | packed as aligned KiB %.3f\n",
               work, t0, (tbytes + pbytes) / t0 / 1e6, t1, t2, t3, t8, t16, t24, t35, t4, t5, t64);
    }
    return 0;
}
