// ubench_anyorder.hip -- does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let a kernel start while its predecessor ON THE SAME
// STREAM is still running (no barrier bit on its AQL packet) on gfx950, and how much of a predecessor's last partial round of
// workgroups a successor can fill?  Background: the table strategy's two passes are B (build, one round of 2 048 workgroups,
// vector-issue bound, serial start per workgroup) and C (combine, 2 913 workgroups in 5.7 rounds of 512); B of call i+1 depends on
// nothing C of call i produces once the table scratch is double-buffered.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_anyorder tools/ubench_anyorder.hip && build/ubench_anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every wave spins until `ticks` of the 100 MHz wall clock have passed, then marks its slot
__global__ void k_spin(unsigned long long ticks, unsigned *counter)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) counter[blockIdx.x] = 1u;
}

static void launch(void (*k)(unsigned long long, unsigned *), dim3 g, dim3 b, hipStream_t st, unsigned long long ticks, unsigned *ctr, int flags)
{
    void *args[] = {&ticks, &ctr};
    CK(hipExtLaunchKernel(reinterpret_cast<const void *>(k), g, b, args, 0, st, nullptr, nullptr, flags));
}

static float pair_us(hipStream_t st, unsigned *ctr, dim3 ga, dim3 ba, unsigned long long ta, dim3 gb, dim3 bb, unsigned long long tb, int flag_b, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) { launch(k_spin, ga, ba, st, ta, ctr, 0); launch(k_spin, gb, bb, st, tb, ctr, flag_b); }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) { launch(k_spin, ga, ba, st, ta, ctr, 0); launch(k_spin, gb, bb, st, tb, ctr, flag_b); }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / (float)reps;
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned *ctr;
    CK(hipMalloc(&ctr, 4 * 65536));
    CK(hipMemset(ctr, 0, 4 * 65536));
    // (1) two half-chip kernels of 50 us each: 100 us serial, ~50 us if the second may start beside the first
    {
        const dim3 g(256), b(256);
        const float s = pair_us(st, ctr, g, b, 5000, g, b, 5000, 0, 50);
        const float o = pair_us(st, ctr, g, b, 5000, g, b, 5000, hipExtAnyOrderLaunch, 50);
        printf("two 256-workgroup kernels of 50 us : in order %7.1f us per pair, second launched any-order %7.1f us\n", s, o);
    }
    // (2) the shapes of the two passes: C = 2913 x 960 threads (2 per CU: 5.69 rounds of 10 us), B = 2048 x 256 threads (8 per CU: one round of 40 us)
    {
        const dim3 gc(2913), bc(960), gb(2048), bb(256);
        const float s = pair_us(st, ctr, gc, bc, 1000, gb, bb, 4000, 0, 50);
        const float o = pair_us(st, ctr, gc, bc, 1000, gb, bb, 4000, hipExtAnyOrderLaunch, 50);
        printf("C-shaped (2913 x 960, 10 us) then B-shaped (2048 x 256, 40 us): in order %7.1f us per pair, B any-order %7.1f us\n", s, o);
    }
    // (3) the same with the any-order kernel first in the pair (B of the next call behind C of this one, steady state)
    {
        const dim3 gc(2913), bc(960), gb(2048), bb(256);
        const float s = pair_us(st, ctr, gb, bb, 4000, gc, bc, 1000, 0, 50);
        printf("B-shaped then C-shaped, both in order: %7.1f us per pair\n", s);
    }
    CK(hipFree(ctr));
    return 0;
}
