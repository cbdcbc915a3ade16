// valu_rates.hip - per-instruction issue cost of the integer VALU ops the Myers
// recurrence is made of, measured on gfx950: N independent chains per lane, every CU
// filled with 8 waves per SIMD, wall time -> cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

#define KERNEL(name, body)                                                              \
    __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed)           \
    {                                                                                   \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3; \
        uint32_t b0 = a0 ^ 0x1234567, b1 = a1 ^ 0x89abcde, b2 = a2 + 77, b3 = a3 + 99;  \
        uint64_t q0 = ((uint64_t)a0 << 32) | b0, q1 = ((uint64_t)a1 << 32) | b1,        \
                 q2 = ((uint64_t)a2 << 32) | b2, q3 = ((uint64_t)a3 << 32) | b3;        \
        for (int i = 0; i < ITERS; ++i) {                                               \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { body }                 \
        }                                                                               \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3) >> 32); \
    }

// 4 independent instructions per body
KERNEL(k_and, asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %5\n v_and_b32 %2, %2, %6\n v_and_b32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_or3, asm volatile("v_or3_b32 %0, %0, %4, %5\n v_or3_b32 %1, %1, %5, %6\n v_or3_b32 %2, %2, %6, %7\n v_or3_b32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %0, %4, %5\n v_bfi_b32 %1, %1, %5, %6\n v_bfi_b32 %2, %2, %6, %7\n v_bfi_b32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_alignbit, asm volatile("v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %6, 31\n v_alignbit_b32 %3, %3, %7, 31" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_lshl_or, asm volatile("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %5\n v_lshl_or_b32 %2, %2, 1, %6\n v_lshl_or_b32 %3, %3, 1, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, 3, 29\n v_bfe_u32 %1, %1, 3, 29\n v_bfe_u32 %2, %2, 3, 29\n v_bfe_u32 %3, %3, 3, 29" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_lshl64, asm volatile("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %2, 1, %2\n v_lshlrev_b64 %3, 1, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));)
KERNEL(k_lshladd64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));)
KERNEL(k_addco, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_add_co_u32 %2, vcc, %2, %6\n v_addc_co_u32 %3, vcc, %3, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %5, vcc\n v_cndmask_b32 %2, %2, %6, vcc\n v_cndmask_b32 %3, %3, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %5\n v_xor_b32 %2, %2, %6\n v_xor_b32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %5, %6\n v_add3_u32 %2, %2, %6, %7\n v_add3_u32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_cmp, asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %5\n v_cmp_lt_u32 vcc, %2, %6\n v_cmp_lt_u32 vcc, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
// one dependent chain: latency
KERNEL(k_and_dep, asm volatile("v_and_b32 %0, %0, %1\n v_or_b32 %0, %0, %2\n v_and_b32 %0, %0, %3\n v_or_b32 %0, %0, %1" : "+v"(a0) : "v"(b0), "v"(b1), "v"(b2));)

typedef void (*kfn)(uint32_t*, uint32_t);

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t* out;
    struct { const char* name; kfn f; } ks[] = {
        {"v_and_b32", k_and}, {"v_xor_b32", k_xor}, {"v_or3_b32", k_or3}, {"v_bfi_b32", k_bfi}, {"v_alignbit_b32", k_alignbit},
        {"v_lshl_or_b32", k_lshl_or}, {"v_bfe_u32", k_bfe}, {"v_add3_u32", k_add3}, {"v_cndmask_b32", k_cndmask},
        {"v_cmp_lt_u32", k_cmp}, {"v_add_co+v_addc_co", k_addco}, {"v_lshlrev_b64", k_lshl64}, {"v_lshl_add_u64", k_lshladd64},
        {"dependent and/or chain", k_and_dep},
    };
    for (int wps = 1; wps <= 8; wps *= 2) {          // waves per SIMD
        const int blocks = cus * wps;                // 256 threads = 4 waves = one per SIMD
        CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
        printf("-- %d wave(s) per SIMD\n", wps);
        for (auto& k : ks) {
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            k.f<<<blocks, 256>>>(out, 1);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(a));
            k.f<<<blocks, 256>>>(out, 2);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            const double insts_per_simd = (double)ITERS * UNROLL * 4 * wps;   // wave-instructions per SIMD
            printf("%-26s %8.3f ms  %6.2f ns per wave-instr per SIMD  (%.2f cycles at 2.4 GHz)\n", k.name, ms,
                   ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
        }
        CHECK(hipFree(out));
    }
    return 0;
}
