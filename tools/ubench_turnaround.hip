// ubench_turnaround.hip -- what it costs gfx950 to retire one workgroup and start the next on the same CU ("turnaround"), for the
// workgroup shapes of the table strategy's combine pass: 960 threads, 64 VGPRs, 41 KB of LDS, two per CU.  Every wave spins for a
// fixed time T; a launch of R full rounds (R x 512 workgroups) then takes R x (T + turnaround) + launch.  If the turnaround is a
// sizeable fraction of the real kernel's ~11 us per workgroup, persistent workgroups are worth their loop; if not, they are not.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_turnaround tools/ubench_turnaround.hip && build/ubench_turnaround
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// BUSY = 0: poll the 100 MHz wall clock (s_memrealtime) with s_sleep in between; BUSY = 1: a dependent chain of `ticks` x 8 VOP2
// instructions, no memory traffic at all (thousands of waves polling the clock through the scalar cache turned out to slow the
// start of new workgroups themselves: the first version of this probe read 4.4 us of "turnaround" per round that the ALU form does not show)
#ifndef BUSY
#define BUSY 1
#endif
__device__ __forceinline__ void spin(unsigned long long ticks)
{
#if BUSY
    int a = (int)threadIdx.x, b = (int)ticks;
#pragma unroll 1
    for (unsigned long long i = 0; i < ticks; ++i) {
        a += b; b ^= a; a -= b >> 3; b += a; a ^= b; b -= a >> 5; a += b; b ^= a;
        asm volatile("" : "+v"(a), "+v"(b));
    }
    asm volatile("" :: "v"(a), "v"(b));
#else
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
#endif
}

template <int THREADS, int LDS_BYTES>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_spin(unsigned long long ticks, unsigned *counter)
{
    __shared__ int lds[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    if (LDS_BYTES) lds[threadIdx.x] = (int)threadIdx.x;
    asm volatile("v_mov_b32 v63, 0" ::: "v63");            // 64 VGPRs allocated per wave, like the tile kernel
    spin(ticks);
    if (threadIdx.x == 0) counter[blockIdx.x] = LDS_BYTES ? (unsigned)lds[0] + 1u : 1u;   // (one slot per workgroup: a shared atomic counter serialises at ~11 ns per workgroup and was the whole "dispatch cost" of the first version)
}

// persistent form: gridDim.x workgroups stride over `tiles` units of the same spin
template <int THREADS, int LDS_BYTES>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_spin_persistent(unsigned long long ticks, unsigned *counter, unsigned tiles)
{
    __shared__ int lds[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    if (LDS_BYTES) lds[threadIdx.x] = (int)threadIdx.x;
    asm volatile("v_mov_b32 v63, 0" ::: "v63");
    for (unsigned t = blockIdx.x; t < tiles; t += gridDim.x) {
        spin(ticks);
    }
    if (threadIdx.x == 0) counter[blockIdx.x] = LDS_BYTES ? (unsigned)lds[0] + 1u : 1u;   // (one slot per workgroup: a shared atomic counter serialises at ~11 ns per workgroup and was the whole "dispatch cost" of the first version)
}

template <typename F>
static float time_us(hipStream_t st, int reps, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch();
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
    const unsigned long long T = BUSY ? 300 : 1000;   // BUSY: 300 x 8 dependent VOP2 per wave (8 waves per SIMD: ~2400 x 8 x ~2.7 cycles ~ 20 us per round); else 10 us of the 100 MHz clock
    printf("%s; us per launch\n", BUSY ? "every wave runs a chain of 2400 dependent VOP2" : "every wave polls the wall clock for 10 us");
    printf("%-44s", "workgroups per launch (512 = one round):");
    for (int r : {1, 2, 3, 4, 6}) printf(" %6d", 512 * r);
    printf("   2913\n");
    auto row = [&](const char *name, auto kern, int threads, int per_cu) {
        printf("%-44s", name);
        for (int r : {1, 2, 3, 4, 6}) {
            const int wgs = 256 * per_cu * r;
            printf(" %6.1f", time_us(st, 30, [&] { hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), 0, st, T, ctr); }));
        }
        printf(" %6.1f\n", time_us(st, 30, [&] { hipLaunchKernelGGL(kern, dim3(2913 * (960 / threads)), dim3(threads), 0, st, T, ctr); }));
    };
    row("960 threads, 41 KB LDS, 2 per CU", k_spin<960, 41040>, 960, 2);
    row("960 threads, no LDS, 2 per CU", k_spin<960, 0>, 960, 2);
    row("320 threads, 13.7 KB LDS, 6 per CU", k_spin<320, 13680>, 320, 6);
    row("64 threads, no LDS, 32 per CU", k_spin<64, 0>, 64, 32);
    printf("persistent: 512 workgroups of 960 threads stride over the tiles\n%-44s", "tiles:");
    for (int r : {1, 2, 3, 4, 6}) printf(" %6d", 512 * r);
    printf("   2913\n%-44s", "960 threads, 41 KB LDS");
    for (int r : {1, 2, 3, 4, 6})
        printf(" %6.1f", time_us(st, 30, [&] { hipLaunchKernelGGL((k_spin_persistent<960, 41040>), dim3(512), dim3(960), 0, st, T, ctr, 512u * r); }));
    printf(" %6.1f\n", time_us(st, 30, [&] { hipLaunchKernelGGL((k_spin_persistent<960, 41040>), dim3(512), dim3(960), 0, st, T, ctr, 2913u); }));
    CK(hipFree(ctr));
    return 0;
}
