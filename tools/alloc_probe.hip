// alloc_probe.hip -- does the physical placement of a large buffer change the streaming-store rate?
// For each trial: allocate (hipMalloc, or the virtual-memory API with a chosen granularity), run a pure
// streaming-store kernel shaped like the row kernel's text stream (each wave walks its own 80 KB row in 1 KiB
// steps, rows byte-aligned only), report the rate, free.   hipcc --offload-arch=gfx950 -O3 -o alloc_probe alloc_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void rows_store(uint8_t* dst, uint64_t row_bytes, uint64_t n_rows)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t row = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    uint8_t* p = dst + row * row_bytes + 37 + lane * 16;
    uint32_t v[4] = {0x09302F30u, 0x09302F31u, 0x09312F30u, 0x09312F31u};
    for (uint64_t o = 0; o + 1024 <= row_bytes - 64; o += 1024) {
        v[0] += lane;
        __builtin_memcpy(p + o, v, 16);
    }
}

static float run(uint8_t* d, uint64_t row_bytes, uint64_t n_rows)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int i = 0; i < 6; ++i) {
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(rows_store, dim3((uint32_t)((n_rows + 3) / 4)), dim3(256), 0, 0, d, row_bytes, n_rows);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (i >= 2 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv)
{
    const uint64_t row_bytes = 80037, n_rows = 200000, bytes = row_bytes * n_rows + 4096;
    const int trials = argc > 1 ? atoi(argv[1]) : 6;
    for (int t = 0; t < trials; ++t) {
        uint8_t* d;
        CHECK(hipMalloc((void**)&d, bytes));
        const float ms = run(d, row_bytes, n_rows);
        printf("hipMalloc   trial %d: %.3f ms  %.0f GB/s\n", t, ms, (double)(row_bytes - 64) / 1024 * 1024 * n_rows / ms / 1e6);
        CHECK(hipFree(d));
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CHECK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CHECK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("VMM granularity: minimum %zu, recommended %zu\n", gmin, grec);
    for (int t = 0; t < trials; ++t) {
        const size_t gran = grec ? grec : gmin;
        const size_t size = (bytes + gran - 1) / gran * gran;
        hipMemGenericAllocationHandle_t h;
        void* va;
        CHECK(hipMemAddressReserve(&va, size, gran, nullptr, 0));
        CHECK(hipMemCreate(&h, size, &prop, 0));
        CHECK(hipMemMap(va, size, 0, h, 0));
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CHECK(hipMemSetAccess(va, size, &acc, 1));
        const float ms = run((uint8_t*)va, row_bytes, n_rows);
        printf("hipMemCreate trial %d: %.3f ms  %.0f GB/s\n", t, ms, (double)(row_bytes - 64) / 1024 * 1024 * n_rows / ms / 1e6);
        CHECK(hipMemUnmap(va, size));
        CHECK(hipMemRelease(h));
        CHECK(hipMemAddressFree(va, size));
    }
    return 0;
}
