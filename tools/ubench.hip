// ubench.hip -- gfx950 micro-benchmarks that size the design (not part of the product):
//   * issue rate of the integer VALU instructions the CORDIC step can be built from
//   * streaming-store bandwidth (the declared roofline of the path) for plain / nontemporal stores
//   * gather bandwidth from a 128 MiB (c,s) table at the stride-k patterns of the combine pass
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITER = 2048;

#define OP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ __launch_bounds__(256) void k_valu(int *out, int seed)
{
    int a0 = threadIdx.x + seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    int b = seed * 3 + 1, c = seed + 5;
    long long w0 = a0, w1 = a1, w2 = a2, w3 = a3, w4 = a4, w5 = a5, w6 = a6, w7 = a7;
    for (int i = 0; i < ITER; ++i) {
        if constexpr (OP == 0) {  // v_add_u32
#define S(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 1) {  // v_ashrrev_i32
#define S(n) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a##n));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 2) {  // v_alignbit_b32
#define S(n) asm volatile("v_alignbit_b32 %0, %1, %0, 7" : "+v"(a##n) : "v"(b));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 3) {  // v_xad_u32
#define S(n) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 4) {  // v_add3_u32
#define S(n) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 5) {  // v_mul_lo_u32
#define S(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 6) {  // v_mul_hi_i32
#define S(n) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a##n) : "v"(b));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 7) {  // v_mad_i64_i32
#define S(n) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w##n) : "v"(b), "v"(c) : "vcc");
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 8) {  // v_lshl_add_u64
#define S(n) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w##n) : "v"(w7));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 9) {  // v_ashrrev_i64
#define S(n) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(w##n));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 10) {  // v_cndmask_b32 (vcc)
#define S(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##n) : "v"(b));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 11) {  // v_add_co_u32 + v_addc_co_u32 pair (64-bit add)
#define S(n) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a##n), "+v"(b) : "v"(c), "v"(a7) : "vcc");
            OP8(S)
#undef S
        } else if constexpr (OP == 12) {  // v_and_b32
#define S(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##n) : "v"(b));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 13) {  // v_sub_u32 with SGPR operand
#define S(n) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a##n) : "s"(seed));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 14) {  // v_cmp_lt_i32 -> sgpr pair + v_cndmask from sgpr pair
#define S(n) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##n) : "v"(b) : "vcc");
            OP8(S)
#undef S
        } else if constexpr (OP == 15) {  // v_lshl_add_u32
#define S(n) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a##n) : "v"(b));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 16) {  // v_mad_u32_u24
#define S(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 17) {  // v_pk_add? not for i32; v_bfe_i32
#define S(n) asm volatile("v_bfe_i32 %0, %0, 3, 20" : "+v"(a##n));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 18) {  // v_mul_i32_i24
#define S(n) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a##n) : "v"(b));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 20) {  // v_add_u32 with an SGPR operand
#define S(n) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a##n) : "s"(seed));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 21) {  // v_ashrrev_i32 by an SGPR amount
#define S(n) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(a##n) : "s"(seed));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 22) {  // v_and_b32 with an SGPR mask
#define S(n) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a##n) : "s"(seed));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 23) {  // v_mad_i64_i32 with an SGPR multiplicand
#define S(n) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w##n) : "s"(seed), "v"(c) : "vcc");
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 24) {  // v_cndmask_b32 with an SGPR-pair mask (VOP3 encoding)
            const unsigned long long m64 = 0x5555555555555555ull * (unsigned long long)seed;
#define S(n) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "s"(m64));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 25) {  // v_readlane_b32 -> SGPR -> dependent v_add
#define S(n) asm volatile("v_readlane_b32 s20, %0, 5\n\tv_add_u32 %0, s20, %0" : "+v"(a##n) : : "s20");
            OP8(S)
#undef S
        } else if constexpr (OP == 26) {  // v_bfe_u32
#define S(n) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(a##n));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 27) {  // v_bitop3_b32
#define S(n) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c" : "+v"(a##n) : "v"(b), "v"(c));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 28) {  // v_mad_i32_i24 with an SGPR operand
#define S(n) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a##n) : "s"(seed), "v"(c));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 29) {  // v_alignbit_b32 with an SGPR shift amount
#define S(n) asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(a##n) : "v"(b), "s"(seed));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 30) {  // v_cndmask_b32 (vcc) after one v_cmp: plain VOP2 selects
            asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(a0), "v"(b) : "vcc");
#define S(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##n) : "v"(b) : "vcc");
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 31) {  // v_sub_u32 (VGPR operands)
#define S(n) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a##n) : "v"(b));
            OP8(S) OP8(S)
#undef S
        } else if constexpr (OP == 19) {  // ds_read_b32 broadcast + v_add
            __shared__ int lds[64];
            if (i == 0) lds[threadIdx.x & 63] = seed;
#define S(n) a##n += lds[(i + n) & 63];
            OP8(S) OP8(S)
#undef S
        }
    }
    int r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b ^ (int)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7);
    if (r == 0x7fffffff) out[0] = r;
}

struct ValuCase { const char *name; void (*fn)(int *, int); int ops_per_iter; };

template <int MODE>
__global__ __launch_bounds__(256) void k_fill(int4 *out, size_t nvec, int v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    int4 d = make_int4(v, v + 1, v + 2, v + 3);
    for (; i < nvec; i += stride) {
        if constexpr (MODE == 0) out[i] = d;
        else { typedef int v4i __attribute__((ext_vector_type(4))); v4i dd = {d.x, d.y, d.z, d.w}; __builtin_nontemporal_store(dd, reinterpret_cast<v4i *>(&out[i])); }
    }
}

// one dword per lane (the natural "one lane per coefficient" store)
__global__ __launch_bounds__(256) void k_fill1(int *out, size_t n, int v)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = v + (int)i;
}

// gather (c,s) pairs at index (k*n) & mask, one lane per n, accumulate and store one dword
template <int K>
__global__ __launch_bounds__(256) void k_gather(const int2 *__restrict__ tab, unsigned mask, int *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int acc = 0;
#pragma unroll
    for (int k = 1; k <= K; ++k) {
        int2 v = tab[((unsigned)i * (unsigned)k) & mask];
        acc += v.x ^ v.y;
    }
    out[i] = acc;
}

// single harmonic stride test
__global__ __launch_bounds__(256) void k_gather1(const int2 *__restrict__ tab, unsigned mask, unsigned k, int *out, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int2 v = tab[((unsigned)i * k) & mask];
    out[i] = v.x ^ v.y;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  L2=%d  \n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.l2CacheSize);
    int *dout;
    CK(hipMalloc(&dout, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    ValuCase cases[] = {
        {"v_add_u32", k_valu<0>, 16}, {"v_ashrrev_i32", k_valu<1>, 16}, {"v_alignbit_b32", k_valu<2>, 16},
        {"v_xad_u32", k_valu<3>, 16}, {"v_add3_u32", k_valu<4>, 16}, {"v_mul_lo_u32", k_valu<5>, 16},
        {"v_mul_hi_i32", k_valu<6>, 16}, {"v_mad_i64_i32", k_valu<7>, 16}, {"v_lshl_add_u64", k_valu<8>, 16},
        {"v_ashrrev_i64", k_valu<9>, 16}, {"v_cndmask_b32", k_valu<10>, 16}, {"add_co+addc_co pair", k_valu<11>, 16},
        {"v_and_b32", k_valu<12>, 16}, {"v_sub_u32 sgpr", k_valu<13>, 16}, {"v_cmp+v_cndmask pair", k_valu<14>, 16},
        {"v_lshl_add_u32", k_valu<15>, 16}, {"v_mad_u32_u24", k_valu<16>, 16}, {"v_bfe_i32", k_valu<17>, 16},
        {"v_mul_i32_i24", k_valu<18>, 16}, {"lds bcast + add", k_valu<19>, 16},
        {"v_add_u32 sgpr", k_valu<20>, 16}, {"v_ashrrev_i32 sgpr", k_valu<21>, 16}, {"v_and_b32 sgpr", k_valu<22>, 16},
        {"v_mad_i64_i32 sgpr", k_valu<23>, 16}, {"v_cndmask_b32 sgpr-pair", k_valu<24>, 16}, {"v_readlane + v_add sgpr", k_valu<25>, 16},
        {"v_bfe_u32", k_valu<26>, 16}, {"v_bitop3_b32", k_valu<27>, 16}, {"v_mad_i32_i24 sgpr", k_valu<28>, 16},
        {"v_alignbit_b32 sgpr", k_valu<29>, 16}, {"v_cndmask_b32 vcc (1 cmp)", k_valu<30>, 16}, {"v_sub_u32 vgpr", k_valu<31>, 16},
    };
    const int blocks = prop.multiProcessorCount * 8;
    printf("\n%-24s %10s %14s %12s\n", "instruction", "ms", "Tops/s(lane)", "cyc/wave-instr/SIMD@2.4GHz");
    for (auto &c : cases) {
        hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, dout, 1);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(256), 0, 0, dout, 1);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 3;
        double ops = (double)blocks * 256 * ITER * c.ops_per_iter;
        double tops = ops / (ms * 1e-3) / 1e12;
        // wave-instructions per SIMD per second = ops/64/(CUs*4); cycles each = 2.4e9 / that
        double wi = ops / 64.0 / (prop.multiProcessorCount * 4.0) / (ms * 1e-3);
        printf("%-24s %10.3f %14.2f %12.2f\n", c.name, ms, tops, 2.4e9 / wi);
    }

    // ---- streaming stores ----
    const size_t bytes = 256ull << 20;
    int4 *buf;
    CK(hipMalloc(&buf, bytes));
    printf("\nfill 256 MiB:\n");
    for (int mode = 0; mode < 3; ++mode) {
        for (int gb : {2048, 8192, 65536, 1 << 20}) {
            if (mode == 2 && gb != (1 << 20)) continue;
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k_fill<0>, dim3(gb > (int)(bytes / 16 / 256) ? (int)(bytes / 16 / 256) : gb), dim3(256), 0, 0, buf, bytes / 16, 3);
                else if (mode == 1) hipLaunchKernelGGL(k_fill<1>, dim3(gb > (int)(bytes / 16 / 256) ? (int)(bytes / 16 / 256) : gb), dim3(256), 0, 0, buf, bytes / 16, 3);
                else hipLaunchKernelGGL(k_fill1, dim3((unsigned)(bytes / 4 / 256)), dim3(256), 0, 0, (int *)buf, bytes / 4, 3);
            };
            launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 10; ++r) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 10;
            printf("  %-14s grid=%8d  %8.3f ms  %8.1f GB/s\n", mode == 0 ? "dwordx4 plain" : mode == 1 ? "dwordx4 nt" : "dword/lane", gb, ms, bytes / (ms * 1e-3) / 1e9);
        }
    }
    {   // hipMemsetAsync reference
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; ++r) CK(hipMemsetAsync(buf, 1, bytes, 0));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  hipMemsetAsync               %8.3f ms  %8.1f GB/s\n", ms / 10, bytes / (ms / 10 * 1e-3) / 1e9);
    }

    // ---- gather from a 128 MiB table ----
    const size_t entries = 1ull << 24;
    int2 *tab;
    CK(hipMalloc(&tab, entries * 8));
    CK(hipMemset(tab, 1, entries * 8));
    const size_t n = 1ull << 26;
    printf("\ngather 8-B pairs from 128 MiB table, %zu lanes, out = 4 B/lane:\n", n);
    for (unsigned k = 1; k <= 6; ++k) {
        hipLaunchKernelGGL(k_gather1, dim3((unsigned)(n / 256)), dim3(256), 0, 0, tab, (unsigned)(entries - 1), k, (int *)buf, n);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_gather1, dim3((unsigned)(n / 256)), dim3(256), 0, 0, tab, (unsigned)(entries - 1), k, (int *)buf, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        printf("  stride k=%u: %8.3f ms  %7.2f Glanes/s  useful %7.1f GB/s\n", k, ms, n / (ms * 1e-3) / 1e9, n * 12.0 / (ms * 1e-3) / 1e9);
    }
    {
        hipLaunchKernelGGL(k_gather<6>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, tab, (unsigned)(entries - 1), (int *)buf, n);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_gather<6>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, tab, (unsigned)(entries - 1), (int *)buf, n);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        printf("  all 6 harmonics per lane: %8.3f ms  %7.2f Glanes/s\n", ms, n / (ms * 1e-3) / 1e9);
    }
    // quarter-size lane count with 4 outputs per lane (quadrant fold): 6 gathers + 4 stores per lane
    return 0;
}
