// Store patterns of the tile kernel on gfx950: what the memory system sustains for 256 MiB written as the tile kernel writes
// it (per wave 24 separate 256-byte chunks: 8 images x 3 runs) against plain streaming.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_store.hip -o build/ub/ubench_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr uint32_t LQ = 24, E = 1u << LQ, H = E >> 1;            // 2^26 coefficients: ring of H lanes x 8 images

// MODE 0: tile pattern, one tile per workgroup in block order; 1: XCD-aware tile order; 2: each thread 24 consecutive-run dwords but
// images contiguous per workgroup (same chunk sizes, nearer addresses); 3: streaming dwordx4
template <int MODE>
__global__ __launch_bounds__(960) void k_store(int *__restrict__ out, int v)
{
    const uint32_t part = threadIdx.x / 192u, lane = threadIdx.x % 192u;
    uint32_t tile = blockIdx.x;
    if (MODE == 1) { const uint32_t per = gridDim.x >> 3, main = per << 3; if (blockIdx.x < main) tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3); }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        // sibling runs a third / a fifth of the ring apart, as the tile plan's 15 offsets
        const uint32_t start = (tile * 192u + (uint32_t)b * 2796203u + part * 1677722u) & (H - 1u);
        const uint32_t r = (start + lane) & (H - 1u);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[(size_t)r + (size_t)h * H + (size_t)j * E] = v + b + h + j;
    }
}

__global__ __launch_bounds__(256) void k_stream(int4 *out, size_t nvec, int v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const int4 d = make_int4(v, v + 1, v + 2, v + 3);
    for (; i < nvec; i += stride) out[i] = d;
}

// one dword per lane per iteration, consecutive lanes, a grid-stride loop (256-byte chunks per wave-instruction, streaming order)
__global__ __launch_bounds__(256) void k_stream1(int *out, size_t n, int v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = v;
}

int main()
{
    int *buf;
    const size_t n = (size_t)1 << 26;
    CK(hipMalloc(&buf, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char *name, auto launch) {
        for (int r = 0; r < 50; ++r) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 50; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
        printf("%-64s %8.4f ms  %7.1f GB/s\n", name, ms, n * 4 / (ms * 1e-3) / 1e9);
    };
    const unsigned tiles = (unsigned)(H / 2880u) + 1u;           // 15 runs x 192 lanes per tile (the last tile overlaps)
    report("tile pattern, block order (2913 x 960 threads)", [&]() { hipLaunchKernelGGL(k_store<0>, dim3(tiles), dim3(960), 0, 0, buf, 1); });
    report("tile pattern, XCD-aware tile order", [&]() { hipLaunchKernelGGL(k_store<1>, dim3(tiles), dim3(960), 0, 0, buf, 1); });
    report("streaming dwordx4, 65536 x 256 threads", [&]() { hipLaunchKernelGGL(k_stream, dim3(65536), dim3(256), 0, 0, (int4 *)buf, n / 4, 1); });
    report("streaming dword per lane, 65536 x 256 threads, grid-stride", [&]() { hipLaunchKernelGGL(k_stream1, dim3(65536), dim3(256), 0, 0, buf, n, 1); });
    report("streaming dword per lane, 8192 x 256 threads, grid-stride", [&]() { hipLaunchKernelGGL(k_stream1, dim3(8192), dim3(256), 0, 0, buf, n, 1); });
    return 0;
}
