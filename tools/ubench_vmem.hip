// Vector-memory issue rate on gfx950: cycles one CU's texture-address / L1 path spends per wave-instruction for the access shapes of
// the tile kernel's gathers (L1-resident data, so this is the pipeline's rate, not cache misses).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_vmem.hip -o build/ubench/ubench_vmem && build/ubench/ubench_vmem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITER = 256;
constexpr int UNR = 8;

template <typename T> struct Acc { static __device__ int f(T v) { return (int)v; } };
template <> struct Acc<int2> { static __device__ int f(int2 v) { return v.x ^ v.y; } };
template <> struct Acc<int4> { static __device__ int f(int4 v) { return v.x ^ v.y ^ v.z ^ v.w; } };
struct int3a { int x, y, z; };
template <> struct Acc<int3a> { static __device__ int f(int3a v) { return v.x ^ v.y ^ v.z; } };

// offset of lane l in unrolled slot j of iteration it:  (((l * mul) >> rsh) << lsh) + (j * 512 + it * 64), wrapped to `mask`
template <typename T, int PERM = 0>
__global__ __launch_bounds__(256) void k_load(const char *__restrict__ base, uint32_t mul, uint32_t rsh, uint32_t lsh, uint32_t mask, int *out)
{
    uint32_t lane = threadIdx.x;
    if (PERM == 1) lane = 255u - lane;
    if (PERM == 2) lane = (lane & ~15u) | ((lane & 1u) << 3) | ((lane & 2u) << 1) | ((lane & 4u) >> 1) | ((lane & 8u) >> 3);
    uint32_t lo = ((lane * mul) >> rsh) << lsh;
    if (PERM == 3)      // split layout: odd entries in the upper half, entries = 2 mod 4 in the second quarter, multiples of 4 in the first
        lo = ((lane & 1u) ? 8192u + (lane >> 1) * mul : (lane & 2u) ? 4096u + (lane >> 2) * mul : (lane >> 2) * mul);
    int acc = 0;
    for (int it = 0; it < ITER; ++it) {
        T v[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const uint32_t off = (lo + (uint32_t)(j * 1024 + it * 64)) & mask & ~(uint32_t)(sizeof(T) > 4 ? 3 : sizeof(T) - 1);
            v[j] = *reinterpret_cast<const T *>(base + off);
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) acc += Acc<T>::f(v[j]);
    }
    if (acc == 0x7fffffff) out[threadIdx.x] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void k_store(char *__restrict__ base, uint32_t mul, uint32_t mask, int seed)
{
    const uint32_t lane = threadIdx.x;
    char *mine = base + (size_t)blockIdx.x * (mask + 1u);
    T v;
    __builtin_memset(&v, seed, sizeof(T));
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const uint32_t off = (lane * mul + (uint32_t)(j * 1024 + it * 64)) & mask & ~(uint32_t)(sizeof(T) - 1);
            *reinterpret_cast<T *>(mine + off) = v;
        }
    }
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    char *buf;
    CK(hipMalloc(&buf, 512u << 20));
    CK(hipMemset(buf, 1, 512u << 20));
    int *out;
    CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int blocks = cus * 8;
    auto report = [&](const char *name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 3;
        const double wave_instr_per_cu = (double)blocks * 4 * ITER * UNR / cus;
        printf("%-58s %8.3f ms  %7.2f cycles per wave-instruction per CU\n", name, ms, ms * 1e-3 * 2.4e9 / wave_instr_per_cu);
    };
    const uint32_t mask = (16u << 10) - 1u;      // 16 KiB window: L1-resident
    printf("device: %s  CUs=%d\nloads from a 16 KiB window (L1 hits), %d blocks x 256 threads:\n", prop.name, cus, blocks);
#define LD(T, mul, rsh, lsh, text) report(text, [&]() { hipLaunchKernelGGL(k_load<T>, dim3(blocks), dim3(256), 0, 0, (const char *)buf, (uint32_t)(mul), (uint32_t)(rsh), (uint32_t)(lsh), mask, out); })
    LD(int, 4, 0, 0, "dword, consecutive lanes (256 B per wave)");
    LD(int2, 8, 0, 0, "dwordx2, consecutive lanes (512 B per wave)");
    LD(int4, 16, 0, 0, "dwordx4, consecutive lanes (1 KiB per wave)");
    LD(uint16_t, 2, 0, 0, "ushort, lane stride 2 B (entries, K = 1)");
    LD(uint16_t, 6, 0, 0, "ushort, lane stride 6 B (K = 3)");
    LD(uint16_t, 10, 0, 0, "ushort, lane stride 10 B (K = 5)");
    LD(uint8_t, 1, 0, 0, "ubyte, lane stride 1 B");
    LD(int, 2, 2, 2, "dword holding a 2-byte entry, K = 1 (two lanes per dword)");
    LD(int, 6, 2, 2, "dword holding a 2-byte entry, K = 3");
    LD(int, 10, 2, 2, "dword holding a 2-byte entry, K = 5");
    LD(int, 4, 0, 0, "dword, lane stride 4 B (4-byte entries, K = 1)");
    LD(int, 12, 0, 0, "dword, lane stride 12 B (K = 3)");
    LD(int, 20, 0, 0, "dword, lane stride 20 B (K = 5)");
    LD(int2, 24, 0, 0, "dwordx2, lane stride 24 B (8-byte entries, K = 3)");
    LD(int2, 40, 0, 0, "dwordx2, lane stride 40 B (K = 5)");
    LD(int4, 1, 7, 4, "dwordx4 record of a 128-entry cell, K = 1 (one record per wave)");
    LD(int4, 5, 7, 4, "dwordx4 record of a 128-entry cell, K = 5 (3-4 records)");
    LD(int3a, 5, 7, 4, "dwordx3 record, K = 5");
    LD(int2, 5, 7, 3, "dwordx2 record of a 128-entry cell, K = 5");
    LD(int2, 5, 6, 3, "dwordx2 record of a 64-entry cell, K = 5");
    LD(int, 5, 7, 2, "dword record of a 128-entry cell, K = 5");
#define LDP(T, P, mul, rsh, lsh, text) report(text, [&]() { hipLaunchKernelGGL((k_load<T, P>), dim3(blocks), dim3(256), 0, 0, (const char *)buf, (uint32_t)(mul), (uint32_t)(rsh), (uint32_t)(lsh), mask, out); })
    LD(uint16_t, 4, 0, 0, "ushort, lane stride 4 B (64 B per 16 lanes)");
    LD(uint16_t, 8, 0, 0, "ushort, lane stride 8 B (128 B per 16 lanes)");
    LD(uint16_t, 16, 0, 0, "ushort, lane stride 16 B");
    LD(int, 8, 0, 0, "dword, lane stride 8 B (128 B per 16 lanes)");
    LD(int, 16, 0, 0, "dword, lane stride 16 B");
    LD(uint8_t, 3, 0, 0, "ubyte, lane stride 3 B");
    LD(uint8_t, 5, 0, 0, "ubyte, lane stride 5 B");
    LDP(uint16_t, 1, 2, 0, 0, "ushort, stride 2 B, lanes reversed");
    LDP(uint16_t, 2, 2, 0, 0, "ushort, stride 2 B, lanes permuted inside each 16");
    LDP(uint16_t, 3, 2, 0, 0, "ushort, split layout (three class regions), K = 1");
    LDP(uint16_t, 3, 6, 0, 0, "ushort, split layout (three class regions), K = 3");
    LDP(int, 2, 4, 0, 0, "dword, stride 4 B, lanes permuted inside each 16");
    LDP(int, 2, 12, 0, 0, "dword, stride 12 B, lanes permuted inside each 16");
    printf("stores into a private 16 KiB window per block:\n");
#define ST(T, mul, text) report(text, [&]() { hipLaunchKernelGGL(k_store<T>, dim3(blocks), dim3(256), 0, 0, buf, (uint32_t)(mul), mask, 3); })
    ST(int, 4, "dword store, consecutive lanes");
    ST(int4, 16, "dwordx4 store, consecutive lanes");
    ST(uint16_t, 2, "ushort store, consecutive lanes");
    return 0;
}
