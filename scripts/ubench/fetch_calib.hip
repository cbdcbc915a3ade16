// fetch_calib.hip - what FETCH_SIZE reports for the access patterns of this library.
// MI355X_MICROARCH.md: FETCH_SIZE is half the bytes of a wide streaming read and "uncalibrated" elsewhere.
// Each kernel touches a known set of bytes exactly once in a 2 GiB buffer (far above the 256 MiB Infinity Cache);
// run under `rocprofv3 --pmc FETCH_SIZE` and compare with the counts printed here.
//   stream16     : 16 B per lane, consecutive lanes consecutive (the calibrated case)
//   gather16     : 16 B per lane, one lane per 256-byte slot (every lane its own line, 16 of 128 bytes used)
//   gather80     : five 16-byte loads per lane = 80 contiguous bytes at an 80-byte stride (pm32 rows before round 2)
//   gather80a    : the same 80 bytes at a 128-byte stride (pm32 rows now)
//   gather96u    : 96 contiguous bytes at an arbitrary 4-byte-aligned offset inside a 256-byte slot (a text window)
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

__global__ void stream16(const u32x4* p, uint32_t* out, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    u32x4 v = p[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = 1;
}
__global__ void gather16(const uint8_t* p, uint32_t* out, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    u32x4 v = *(const u32x4*)(p + i * 256);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = 1;
}
template <int STRIDE>
__global__ void gather80(const uint8_t* p, uint32_t* out, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const u32x4* q = (const u32x4*)(p + i * STRIDE);
    uint32_t a = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { u32x4 v = q[k]; a ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345678u) out[0] = 1;
}
__global__ void gather96u(const uint8_t* p, uint32_t* out, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const uint32_t off = (uint32_t)((i * 2654435761u) >> 7) % 40u * 4u;   // 0 .. 156, 4-byte aligned
    const u32x4_a4* q = (const u32x4_a4*)(p + i * 256 + off);
    uint32_t a = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) { u32x4 v = q[k]; a ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345678u) out[0] = 1;
}

int main()
{
    const size_t bytes = 2ull << 30;
    uint8_t* d; uint32_t* o;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    const int B = 256;
    size_t n;
    n = bytes / 16;  stream16<<<n / B, B>>>((const u32x4*)d, o, n);
    printf("stream16   bytes used %zu\n", n * 16);
    n = bytes / 256; gather16<<<n / B, B>>>(d, o, n);
    printf("gather16   bytes used %zu, 64-byte sectors %zu, 128-byte lines %zu\n", n * 16, n * 64, n * 128);
    n = bytes / 80 / B * B; gather80<80><<<n / B, B>>>(d, o, n);
    printf("gather80   bytes used %zu (dense: every byte of the range)\n", n * 80);
    n = bytes / 128; gather80<128><<<n / B, B>>>(d, o, n);
    printf("gather80a  bytes used %zu, 128-byte lines %zu\n", n * 80, n * 128);
    n = bytes / 256; gather96u<<<n / B, B>>>(d, o, n);
    printf("gather96u  bytes used %zu, 128-byte lines at most %zu\n", n * 96, n * 256);
    hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
