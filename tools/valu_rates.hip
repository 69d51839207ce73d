// valu_rates.hip -- cycles per wave-instruction of the integer VALU instructions the row kernel is made of, at the
// kernel's occupancy (8 waves per SIMD).  Each probe runs NCHAIN independent dependency chains of one instruction
// so that latency never limits; the time comes from s_memtime (shader cycles) per wave and from HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

#define REP8(x) x x x x x x x x

// one probe = 8 chains x 8 repeats = 64 instructions per loop trip
#define PROBE(NAME, ASM)                                                                                  \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint64_t* cyc, uint32_t iters, uint32_t s0, uint32_t s1)  \
    {                                                                                                     \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;                  \
        const uint64_t t0 = __builtin_amdgcn_s_memtime();                                                 \
        for (uint32_t i = 0; i < iters; ++i) {                                                            \
            REP8(ASM)                                                                                     \
        }                                                                                                 \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                 \
        uint32_t acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) >> 32); \
        if (acc == 0x12345678u) out[threadIdx.x] = acc;                                                   \
        if ((threadIdx.x & 63u) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                 \
    }

#define A8(fmt) asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
      "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "s"(s0), "s"(s1) : "vcc");

// operands: %0..%7 = 32-bit chains, %8..%15 = 64-bit chains, %16, %17 = SGPRs
#define F_XOR(n) "v_xor_b32 %" #n ", %16, %" #n "\n\t"
#define F_ADD(n) "v_add_u32 %" #n ", %16, %" #n "\n\t"
#define F_BITOP3V(n) "v_bitop3_b32 %" #n ", %" #n ", %" #n ", %17 bitop3:0x96\n\t"
#define F_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %16\n\t"
#define F_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %16\n\t"
#define F_MUL24(n) "v_mul_u32_u24 %" #n ", %16, %" #n "\n\t"
#define F_MULHI24(n) "v_mul_hi_u32_u24 %" #n ", %16, %" #n "\n\t"
#define F_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %16, %" #n "\n\t"
#define F_CMP(n) "v_cmp_lt_u32 vcc, %16, %" #n "\n\t"
#define F_ADDC(n) "v_addc_co_u32 %" #n ", vcc, %" #n ", %" #n ", vcc\n\t"
#define F_CMPADDC(n) "v_cmp_lt_u32 vcc, %16, %" #n "\n\tv_addc_co_u32 %" #n ", vcc, %" #n ", %" #n ", vcc\n\t"
#define F_MOV(n) "v_mov_b32 %" #n ", %" #n "\n\t"
#define F_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %" #n ", %16\n\t"
#define F_PERM(n) "v_perm_b32 %" #n ", %" #n ", %" #n ", %16\n\t"
#define F_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 1, %" #n "\n\t"
#define F_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %" #n ", vcc\n\t"
#define F_SUBCO(n) "v_sub_co_u32 %" #n ", vcc, %16, %" #n "\n\t"

PROBE(k_xor, A8(F_XOR))
PROBE(k_add, A8(F_ADD))
PROBE(k_bitop3v, A8(F_BITOP3V))
PROBE(k_mullo, A8(F_MULLO))
PROBE(k_mulhi, A8(F_MULHI))
PROBE(k_mul24, A8(F_MUL24))
PROBE(k_mulhi24, A8(F_MULHI24))
PROBE(k_mad24, A8(F_MAD24))
PROBE(k_cmp, A8(F_CMP))
PROBE(k_addc, A8(F_ADDC))
PROBE(k_cmpaddc, A8(F_CMPADDC))
PROBE(k_mov, A8(F_MOV))
PROBE(k_alignbit, A8(F_ALIGNBIT))
PROBE(k_perm, A8(F_PERM))
PROBE(k_lshlor, A8(F_LSHLOR))
PROBE(k_cndmask, A8(F_CNDMASK))
PROBE(k_subco, A8(F_SUBCO))

// v_mad_u64_u32 with its result feeding the next one's multiplicand (8 chains): dst 64-bit, src0 = low half of own chain
#define M8 asm volatile( \
    "v_mad_u64_u32 %0, vcc, %8, %16, 0\n\tv_mad_u64_u32 %1, vcc, %9, %16, 0\n\tv_mad_u64_u32 %2, vcc, %10, %16, 0\n\tv_mad_u64_u32 %3, vcc, %11, %16, 0\n\t" \
    "v_mad_u64_u32 %4, vcc, %12, %16, 0\n\tv_mad_u64_u32 %5, vcc, %13, %16, 0\n\tv_mad_u64_u32 %6, vcc, %14, %16, 0\n\tv_mad_u64_u32 %7, vcc, %15, %16, 0\n\t" \
    : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) \
    : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(s0), "s"(s1) : "vcc");
// the same with an SGPR-pair destination for the carry instead of vcc
#define M8S asm volatile( \
    "v_mad_u64_u32 %0, s[20:21], %8, %16, 0\n\tv_mad_u64_u32 %1, s[20:21], %9, %16, 0\n\tv_mad_u64_u32 %2, s[20:21], %10, %16, 0\n\tv_mad_u64_u32 %3, s[20:21], %11, %16, 0\n\t" \
    "v_mad_u64_u32 %4, s[20:21], %12, %16, 0\n\tv_mad_u64_u32 %5, s[20:21], %13, %16, 0\n\tv_mad_u64_u32 %6, s[20:21], %14, %16, 0\n\tv_mad_u64_u32 %7, s[20:21], %15, %16, 0\n\t" \
    : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) \
    : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(s0), "s"(s1) : "s20", "s21");
// multiply-add form (64-bit addend = own previous result)
#define M8A asm volatile( \
    "v_mad_u64_u32 %0, vcc, %8, %16, %0\n\tv_mad_u64_u32 %1, vcc, %9, %16, %1\n\tv_mad_u64_u32 %2, vcc, %10, %16, %2\n\tv_mad_u64_u32 %3, vcc, %11, %16, %3\n\t" \
    "v_mad_u64_u32 %4, vcc, %12, %16, %4\n\tv_mad_u64_u32 %5, vcc, %13, %16, %5\n\tv_mad_u64_u32 %6, vcc, %14, %16, %6\n\tv_mad_u64_u32 %7, vcc, %15, %16, %7\n\t" \
    : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) \
    : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "s"(s0), "s"(s1) : "vcc");
// alternating v_mad_u64_u32 and v_bitop3_b32 (the Philox mix), 8 + 8 per block
#define MB8 asm volatile( \
    "v_mad_u64_u32 %0, vcc, %8, %16, 0\n\tv_bitop3_b32 %8, %8, %8, %17 bitop3:0x96\n\tv_mad_u64_u32 %1, vcc, %9, %16, 0\n\tv_bitop3_b32 %9, %9, %9, %17 bitop3:0x96\n\t" \
    "v_mad_u64_u32 %2, vcc, %10, %16, 0\n\tv_bitop3_b32 %10, %10, %10, %17 bitop3:0x96\n\tv_mad_u64_u32 %3, vcc, %11, %16, 0\n\tv_bitop3_b32 %11, %11, %11, %17 bitop3:0x96\n\t" \
    "v_mad_u64_u32 %4, vcc, %12, %16, 0\n\tv_bitop3_b32 %12, %12, %12, %17 bitop3:0x96\n\tv_mad_u64_u32 %5, vcc, %13, %16, 0\n\tv_bitop3_b32 %13, %13, %13, %17 bitop3:0x96\n\t" \
    "v_mad_u64_u32 %6, vcc, %14, %16, 0\n\tv_bitop3_b32 %14, %14, %14, %17 bitop3:0x96\n\tv_mad_u64_u32 %7, vcc, %15, %16, 0\n\tv_bitop3_b32 %15, %15, %15, %17 bitop3:0x96\n\t" \
    : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7), \
      "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0), "s"(s1) : "vcc");
PROBE(k_mad64, M8)
PROBE(k_mad64s, M8S)
PROBE(k_mad64a, M8A)
PROBE(k_mad64_bitop3, MB8)

typedef void (*kern_t)(uint32_t*, uint64_t*, uint32_t, uint32_t, uint32_t);

int main()
{
    struct { const char* name; kern_t k; int per_trip; } probes[] = {
        {"v_xor_b32", k_xor, 64}, {"v_add_u32", k_add, 64}, {"v_mov_b32", k_mov, 64},
        {"v_bitop3_b32 (v,v,s)", k_bitop3v, 64},
        {"v_mul_lo_u32", k_mullo, 64}, {"v_mul_hi_u32", k_mulhi, 64}, {"v_mul_u32_u24", k_mul24, 64},
        {"v_mul_hi_u32_u24", k_mulhi24, 64}, {"v_mad_u32_u24", k_mad24, 64},
        {"v_mad_u64_u32 (+0, vcc)", k_mad64, 64}, {"v_mad_u64_u32 (+0, sgpr carry)", k_mad64s, 64}, {"v_mad_u64_u32 (+v64)", k_mad64a, 64},
        {"v_mad_u64_u32 + v_bitop3_b32 pairs", k_mad64_bitop3, 128},
        {"v_cmp_lt_u32 vcc", k_cmp, 64}, {"v_addc_co_u32", k_addc, 64}, {"v_cmp + v_addc pairs", k_cmpaddc, 128},
        {"v_sub_co_u32", k_subco, 64}, {"v_cndmask_b32", k_cndmask, 64},
        {"v_alignbit_b32", k_alignbit, 64}, {"v_perm_b32", k_perm, 64}, {"v_lshl_or_b32", k_lshlor, 64},
    };
    const uint32_t blocks = 256 * 8, iters = 4000; // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    uint32_t* out; uint64_t* cyc;
    CK(hipMalloc((void**)&out, 1024));
    CK(hipMalloc((void**)&cyc, blocks * 4 * sizeof(uint64_t)));
    std::vector<uint64_t> h(blocks * 4);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int waves = 8; waves >= 1; waves /= 2) {
        const uint32_t nb = 256 * waves; // waves per SIMD = workgroups per CU (one wave of each on every SIMD)
        printf("--- %d wave(s) per SIMD (%u workgroups of 256 threads)\n", waves, nb);
        for (auto& p : probes) {
            float best = 1e30f; double cyc_mean = 0;
            for (int r = 0; r < 3; ++r) {
                CK(hipEventRecord(a, 0));
                hipLaunchKernelGGL(p.k, dim3(nb), dim3(256), 0, 0, out, cyc, iters, 0x9E3779B9u, 0xBB67AE85u);
                CK(hipEventRecord(b, 0));
                CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) {
                    best = ms;
                    CK(hipMemcpy(h.data(), cyc, nb * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
                    cyc_mean = 0;
                    for (uint32_t i = 0; i < nb * 4; ++i) cyc_mean += (double)h[i];
                    cyc_mean /= nb * 4;
                }
            }
            const double n_inst = (double)p.per_trip * iters;         // per wave
            // per SIMD: `waves` waves share it; cycles per wave-instruction seen by the SIMD = wave cycles / (insts x waves)
            printf("%-36s %8.3f ms  wave-cycles/inst %6.2f  SIMD-cycles/inst %5.2f  (clock %.2f GHz by s_memtime/wall)\n", p.name, best,
                   cyc_mean / n_inst, cyc_mean / n_inst / waves, cyc_mean / (best * 1e6));
        }
    }
    return 0;
}
