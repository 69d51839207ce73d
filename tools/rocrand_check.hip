// rocrand_check.hip -- pins the sentence "draw n of RNG row r is rocRAND's rocrand_state_philox4x32_10 sequence for
// (seed, subsequence = r, offset = n)" (include/dnaf.h) against rocRAND itself.
// For every "seed row n" triple on stdin prints rocrand() after rocrand_init(seed, row, n, &state): first computed on
// the host (the engine is __host__ __device__), then -- when a GPU is present -- on the device.  tests/test_rocrand_kat.py
// compares both columns with the oracle's Philox draw.
// Build: hipcc --offload-arch=gfx950 -O2 -o rocrand_check rocrand_check.hip
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <cstdio>
#include <vector>

struct Q { unsigned long long seed, row, n; };

__global__ void k(const Q* q, unsigned int* out, int m)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(q[i].seed, q[i].row, q[i].n, &st);
    out[i] = rocrand(&st);
}

int main()
{
    std::vector<Q> q;
    Q t;
    while (scanf("%llu %llu %llu", &t.seed, &t.row, &t.n) == 3) q.push_back(t);
    const int m = (int)q.size();
    std::vector<unsigned int> host(m), dev(m, 0);
    for (int i = 0; i < m; ++i) {
        rocrand_device::philox4x32_10_engine e(q[i].seed, q[i].row, q[i].n);
        host[i] = e();
    }
    int n_dev = 0;
    bool on_gpu = hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0 && m > 0;
    if (on_gpu) {
        Q* dq; unsigned int* dout;
        on_gpu = hipMalloc((void**)&dq, m * sizeof(Q)) == hipSuccess && hipMalloc((void**)&dout, m * 4) == hipSuccess &&
                 hipMemcpy(dq, q.data(), m * sizeof(Q), hipMemcpyHostToDevice) == hipSuccess;
        if (on_gpu) {
            hipLaunchKernelGGL(k, dim3((m + 63) / 64), dim3(64), 0, 0, dq, dout, m);
            on_gpu = hipMemcpy(dev.data(), dout, m * 4, hipMemcpyDeviceToHost) == hipSuccess;
        }
    }
    for (int i = 0; i < m; ++i) {
        if (on_gpu) printf("%u %u\n", host[i], dev[i]);
        else printf("%u -\n", host[i]);
    }
    return 0;
}
