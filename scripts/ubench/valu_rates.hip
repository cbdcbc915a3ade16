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
KERNEL(k_bitop3, asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0xde\n v_bitop3_b32 %1, %1, %5, %6 bitop3:0xde\n v_bitop3_b32 %2, %2, %6, %7 bitop3:0xde\n v_bitop3_b32 %3, %3, %7, %4 bitop3:0xde" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_not, asm volatile("v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
KERNEL(k_lshl32, asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 31, %1\n v_ashrrev_i32 %2, 3, %2\n v_lshlrev_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));)
KERNEL(k_minsub, asm volatile("v_min_u32 %0, %0, %4\n v_sub_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_max_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_addco_only, asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %5\n v_add_co_u32 %2, vcc, %2, %6\n v_add_co_u32 %3, vcc, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
KERNEL(k_addc_chain, asm volatile("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_addc_co_u32 %2, vcc, %2, %6, vcc\n v_addc_co_u32 %3, vcc, %3, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
KERNEL(k_and_or, asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %5, %6\n v_and_or_b32 %2, %2, %6, %7\n v_and_or_b32 %3, %3, %7, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_lshl_add, asm volatile("v_lshl_add_u32 %0, %0, 17, %4\n v_lshl_add_u32 %1, %1, 17, %5\n v_lshl_add_u32 %2, %2, 17, %6\n v_lshl_add_u32 %3, %3, 17, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_and_sgpr, asm volatile("v_and_b32 %0, s4, %0\n v_or_b32 %1, s5, %1\n v_add_u32 %2, s6, %2\n v_xor_b32 %3, 0x10000, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) :: "s4", "s5", "s6");)
KERNEL(k_cndmask_sgpr, asm volatile("v_cndmask_b32 %0, %0, %4, s[4:5]\n v_cndmask_b32 %1, %1, %5, s[4:5]\n v_cndmask_b32 %2, %2, %6, s[4:5]\n v_cndmask_b32 %3, %3, %7, s[4:5]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s4", "s5");)
KERNEL(k_cmp_cnd, asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_u32 vcc, %2, %6\n v_cndmask_b32 %3, %3, %7, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "vcc");)
KERNEL(k_lshl_c, asm volatile("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_lshr_c, asm volatile("v_lshrrev_b32 %0, 31, %0\n v_lshrrev_b32 %1, 31, %1\n v_lshrrev_b32 %2, 31, %2\n v_lshrrev_b32 %3, 31, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_ashr_c, asm volatile("v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_lshl_v, asm volatile("v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %5, %1\n v_lshlrev_b32 %2, %6, %2\n v_lshlrev_b32 %3, %7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_min, asm volatile("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %5\n v_min_u32 %2, %2, %6\n v_min_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_max, asm volatile("v_max_u32 %0, %0, %4\n v_max_u32 %1, %1, %5\n v_max_u32 %2, %2, %6\n v_max_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_sub, asm volatile("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %5\n v_sub_u32 %2, %2, %6\n v_sub_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_add, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %5\n v_add_u32 %2, %2, %6\n v_add_u32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_or, asm volatile("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %5\n v_or_b32 %2, %2, %6\n v_or_b32 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_and_lit, asm volatile("v_and_b32 %0, 0x10000, %0\n v_and_b32 %1, 0x10000, %1\n v_and_b32 %2, 0x10000, %2\n v_and_b32 %3, 0x10000, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_and_inl, asm volatile("v_and_b32 %0, 7, %0\n v_and_b32 %1, 7, %1\n v_and_b32 %2, 7, %2\n v_and_b32 %3, 7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_and_s, asm volatile("v_and_b32 %0, s4, %0\n v_and_b32 %1, s4, %1\n v_and_b32 %2, s4, %2\n v_and_b32 %3, s4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s4");)
KERNEL(k_add_s, asm volatile("v_add_u32 %0, s4, %0\n v_add_u32 %1, s4, %1\n v_add_u32 %2, s4, %2\n v_add_u32 %3, s4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s4");)
KERNEL(k_bitop3_2, asm volatile("v_bitop3_b32 %0, %0, %4, %4 bitop3:0xf1\n v_bitop3_b32 %1, %1, %5, %5 bitop3:0xf1\n v_bitop3_b32 %2, %2, %6, %6 bitop3:0xf1\n v_bitop3_b32 %3, %3, %7, %7 bitop3:0xf1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_mul24, asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %5\n v_mul_u32_u24 %2, %2, %6\n v_mul_u32_u24 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %4, %4\n v_perm_b32 %1, %1, %5, %5\n v_perm_b32 %2, %2, %6, %6\n v_perm_b32 %3, %3, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
KERNEL(k_pk_add, asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %5\n v_pk_add_u16 %2, %2, %6\n v_pk_add_u16 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
// mix as in the recurrence: 2-cycle and 4-cycle ops interleaved
KERNEL(k_mix, asm volatile("v_and_b32 %0, %0, %4\n v_bitop3_b32 %1, %1, %5, %6 bitop3:0xde\n v_or_b32 %2, %2, %6\n v_alignbit_b32 %3, %3, %7, 31" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1), "v"(b2), "v"(b3));)
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
        {"v_bitop3_b32", k_bitop3}, {"v_not_b32", k_not}, {"32-bit shifts", k_lshl32}, {"min/sub/add/max", k_minsub},
        {"v_add_co only", k_addco_only}, {"v_addc_co chain", k_addc_chain}, {"v_and_or_b32", k_and_or}, {"v_lshl_add_u32", k_lshl_add},
        {"VOP2 with sgpr/literal", k_and_sgpr}, {"v_cndmask sgpr mask", k_cndmask_sgpr}, {"v_cmp+v_cndmask", k_cmp_cnd},
        {"mix and/bitop3/or/alignbit", k_mix},
        {"lshl_c", k_lshl_c}, {"lshr_c", k_lshr_c}, {"ashr_c", k_ashr_c}, {"lshl_v", k_lshl_v}, {"min", k_min}, {"max", k_max}, {"sub", k_sub}, {"add", k_add}, {"or", k_or}, {"mov", k_mov}, {"and_lit", k_and_lit}, {"and_inl", k_and_inl}, {"and_s", k_and_s}, {"add_s", k_add_s}, {"bitop3_2", k_bitop3_2}, {"mul24", k_mul24}, {"perm", k_perm}, {"pk_add", k_pk_add},
        {"dependent and/or chain", k_and_dep},
    };
    for (int wps = 4; wps <= 4; wps *= 2) {          // waves per SIMD
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
