// ubench_short.hip -- what a short launch costs on gfx950 whatever it computes: period of back-to-back kernels inside a HIP
// graph for (a) an empty kernel, (b) a kernel that only stores 4 MiB (the output of a 2^20-point window), (c) the same after a
// dependent VALU chain of a given length.  The floor under the C2-class windows.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_short tools/ubench_short.hip && build/ubench_short
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_empty(int *out) { if (out == nullptr) __builtin_trap(); }

__global__ __launch_bounds__(256) void k_store(int *out, int n8)
{
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(size_t)j * n8 + r] = (int)(r + j);
}

__global__ __launch_bounds__(256) void k_chain_store(int *out, int n8, int depth, int seed)
{
    const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
    int a = (int)r ^ seed, b = seed;
#pragma unroll 1
    for (int i = 0; i < depth; ++i) {                // 8 dependent VOP2 per trip
        a += b; b ^= a; a -= b >> 3; b += a; a ^= b; b -= a >> 5; a += b; b ^= a;
        asm volatile("" : "+v"(a), "+v"(b));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(size_t)j * n8 + r] = a + j * b;
}

template <typename F>
static float graph_period_us(hipStream_t st, int per_graph, int replays, F launch)
{
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < per_graph; ++i) launch();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return ms * 1e3f / (float)(per_graph * replays);
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int n8 = 1 << 17;                          // ring lanes of a 2^20-point window
    int *out;
    CK(hipMalloc(&out, (size_t)8 * n8 * sizeof(int)));
    const dim3 grid(n8 / 256), blk(256);
    printf("grid %u x 256 threads, 20 launches per graph, 200 replays\n", grid.x);
    printf("empty kernel                : %6.2f us per launch\n", graph_period_us(st, 20, 200, [&] { hipLaunchKernelGGL(k_empty, grid, blk, 0, st, out); }));
    printf("4 MiB of stores only        : %6.2f us\n", graph_period_us(st, 20, 200, [&] { hipLaunchKernelGGL(k_store, grid, blk, 0, st, out, n8); }));
    for (int depth : {0, 64, 128, 256, 512, 1024})
        printf("chain of %5d VOP2 + stores: %6.2f us\n", depth * 8, graph_period_us(st, 20, 200, [&] { hipLaunchKernelGGL(k_chain_store, grid, blk, 0, st, out, n8, depth, 12345); }));
    const dim3 grid1(n8 / 64), blk1(64);
    printf("one wave per workgroup (grid %u x 64):\n", grid1.x);
    for (int depth : {0, 256, 1024})
        printf("chain of %5d VOP2 + stores: %6.2f us\n", depth * 8, graph_period_us(st, 20, 200, [&] { hipLaunchKernelGGL(k_chain_store, grid1, blk1, 0, st, out, n8, depth, 12345); }));
    CK(hipFree(out));
    return 0;
}
