// ubench_issue.hip -- gfx950 issue-rate micro-benchmarks behind the round-3 table-build kernel (not part of the product):
//   * scalar-ALU throughput next to the vector ALU (does a CU's scalar unit keep up with 4 SIMDs?)
//   * the EXEC-masked rotation block of k_table_build_mirror in its candidate forms
//   * taken / not-taken scalar branches, LDS byte gathers
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_issue.hip -o tools/ubench_issue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITER = 2048;
#define R4(S) S S S S
#define R8(S) R4(S) R4(S)
#define R16(S) R8(S) R8(S)

template <int OP>
__global__ __launch_bounds__(256) void k_issue(int *out, int seed)
{
    __shared__ signed char lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (signed char)(i * seed);
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + seed, y = x ^ 0x55555555u, a = 0, b = 0, zacc = ~0u;
    int z = (int)(threadIdx.x * 977 + seed) - 30000;
    unsigned s0 = seed, s1 = seed + 1;
    unsigned long long sv = 0, zm = 0;
    for (int i = 0; i < ITER; ++i) {
        if constexpr (OP == 0) {          // 16 x s_add_u32 (two independent scalars)
            asm volatile(R8("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 3\n\t") : "+s"(s0), "+s"(s1) : : "scc");
        } else if constexpr (OP == 1) {   // 16 x v_add_u32
            asm volatile(R8("v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %2\n\t") : "+v"(x), "+v"(y) : "v"(a));
        } else if constexpr (OP == 2) {   // 16 x (v_add_u32 ; s_add_u32) interleaved
            asm volatile(R8("v_add_u32 %0, %0, %4\n\ts_add_u32 %2, %2, 1\n\tv_add_u32 %1, %1, %4\n\ts_add_u32 %3, %3, 3\n\t")
                         : "+v"(x), "+v"(y), "+s"(s0), "+s"(s1) : "v"(a) : "scc");
        } else if constexpr (OP == 3) {   // 16 v_add + 4 s_add
            asm volatile(R4("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\ts_add_u32 %2, %2, 1\n\t")
                         : "+v"(x), "+v"(y), "+s"(s0), "+s"(s1) : "v"(a) : "scc");
        } else if constexpr (OP == 4) {   // rotation block as committed: 10 VALU + 4 SALU, exec saved / restored
#define ROT(K) "v_lshrrev_b32 %[a], " #K ", %[y]\n\tv_lshrrev_b32 %[b], " #K ", %[x]\n\tv_cmp_eq_u32 vcc, 0, %[z]\n\ts_or_b64 %[zm], %[zm], vcc\n\t" \
               "s_mov_b64 %[sv], exec\n\tv_cmpx_gt_i32 vcc, 0, %[z]\n\tv_add_u32 %[x], %[x], %[a]\n\tv_sub_u32 %[y], %[y], %[b]\n\tv_add_u32 %[z], %[z], %[l]\n\t" \
               "s_andn2_b64 exec, %[sv], exec\n\tv_sub_u32 %[x], %[x], %[a]\n\tv_add_u32 %[y], %[y], %[b]\n\tv_sub_u32 %[z], %[z], %[l]\n\ts_mov_b64 exec, %[sv]\n\t"
            asm volatile(ROT(17) ROT(18) ROT(19) ROT(20) ROT(21) ROT(22) ROT(23) ROT(24)
                         : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [zm] "+s"(zm), [a] "=&v"(a), [b] "=&v"(b), [sv] "=&s"(sv) : [l] "s"(s0) : "vcc");
#undef ROT
        } else if constexpr (OP == 5) {   // rotation block: v_min zero tracking, s_not / s_mov -1: 10 VALU + 2 SALU
#define ROT(K) "v_lshrrev_b32 %[a], " #K ", %[y]\n\tv_lshrrev_b32 %[b], " #K ", %[x]\n\tv_min_u32 %[za], %[za], %[z]\n\t" \
               "v_cmpx_gt_i32 vcc, 0, %[z]\n\tv_add_u32 %[x], %[x], %[a]\n\tv_sub_u32 %[y], %[y], %[b]\n\tv_add_u32 %[z], %[z], %[l]\n\t" \
               "s_not_b64 exec, exec\n\tv_sub_u32 %[x], %[x], %[a]\n\tv_add_u32 %[y], %[y], %[b]\n\tv_sub_u32 %[z], %[z], %[l]\n\ts_mov_b64 exec, -1\n\t"
            asm volatile(ROT(17) ROT(18) ROT(19) ROT(20) ROT(21) ROT(22) ROT(23) ROT(24)
                         : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [za] "+v"(zacc), [a] "=&v"(a), [b] "=&v"(b) : [l] "s"(s0) : "vcc", "scc");
#undef ROT
        } else if constexpr (OP == 6) {   // same with the ROM word in a VGPR
            unsigned lv = s0;
#define ROT(K) "v_lshrrev_b32 %[a], " #K ", %[y]\n\tv_lshrrev_b32 %[b], " #K ", %[x]\n\tv_min_u32 %[za], %[za], %[z]\n\t" \
               "v_cmpx_gt_i32 vcc, 0, %[z]\n\tv_add_u32 %[x], %[x], %[a]\n\tv_sub_u32 %[y], %[y], %[b]\n\tv_add_u32 %[z], %[z], %[l]\n\t" \
               "s_not_b64 exec, exec\n\tv_sub_u32 %[x], %[x], %[a]\n\tv_add_u32 %[y], %[y], %[b]\n\tv_sub_u32 %[z], %[z], %[l]\n\ts_mov_b64 exec, -1\n\t"
            asm volatile(ROT(17) ROT(18) ROT(19) ROT(20) ROT(21) ROT(22) ROT(23) ROT(24)
                         : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [za] "+v"(zacc), [a] "=&v"(a), [b] "=&v"(b) : [l] "v"(lv) : "vcc", "scc");
#undef ROT
        } else if constexpr (OP == 7) {   // select form: no EXEC games -- sign mask, xor, sub: 12 VALU
#define ROT(K) "v_lshrrev_b32 %[a], " #K ", %[y]\n\tv_lshrrev_b32 %[b], " #K ", %[x]\n\tv_min_u32 %[za], %[za], %[z]\n\tv_ashrrev_i32 %[m], 31, %[z]\n\t" \
               "v_xor_b32 %[a], %[a], %[m]\n\tv_xor_b32 %[b], %[b], %[m]\n\tv_xor_b32 %[t], %[l], %[m]\n\t" \
               "v_sub_u32 %[x], %[x], %[a]\n\tv_add_u32 %[y], %[y], %[b]\n\tv_sub_u32 %[z], %[z], %[t]\n\t" \
               "v_add_u32 %[x], %[x], %[m]\n\tv_sub_u32 %[y], %[y], %[m]\n\tv_add_u32 %[z], %[z], %[m]\n\t"
            unsigned lv = s0, m, t;
            asm volatile(ROT(17) ROT(18) ROT(19) ROT(20) ROT(21) ROT(22) ROT(23) ROT(24)
                         : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [za] "+v"(zacc), [a] "=&v"(a), [b] "=&v"(b), [m] "=&v"(m), [t] "=&v"(t) : [l] "v"(lv));
#undef ROT
        } else if constexpr (OP == 8) {   // the old 64-bit rotation for reference (8 per iteration)
            long long X = ((long long)x << 1) | 1, Y = ((long long)y << 1) | 1;
#pragma unroll
            for (int k = 17; k < 25; ++k) {
                const int m = z >> 31, sg = m | 1, nsg = -sg;
                int ys = (int)(Y >> k), xs = (int)(X >> k);
                asm volatile("" : "+v"(ys), "+v"(xs));
                X += (long long)nsg * ys; Y += (long long)sg * xs;
                z += __mul24(nsg, (int)s0);
            }
            x = (unsigned)X; y = (unsigned)Y;
        } else if constexpr (OP == 9) {   // 8 not-taken scalar compare + branch pairs
            asm volatile(R8("s_cmp_eq_u32 %0, 0x12345\n\ts_cbranch_scc1 9f\n\t") "9:\n\t" : : "s"(s0) : "scc");
        } else if constexpr (OP == 10) {  // 8 taken scalar compare + branch pairs (each jumps over one instruction)
#define TB(n) "s_cmp_lg_u32 %1, 0x12345\n\ts_cbranch_scc1 " #n "f\n\tv_add_u32 %0, %0, 1\n" #n ":\n\t"
            asm volatile(TB(1) TB(2) TB(3) TB(4) TB(5) TB(6) TB(7) TB(8) : "+v"(x) : "s"(s0) : "scc");
#undef TB
        } else if constexpr (OP == 11) {  // 8 LDS byte gathers at lane-dependent addresses + a dependent add each
            unsigned addr = (x >> 7) & 4095u;
#pragma unroll
            for (int j = 0; j < 8; ++j) { x += (unsigned)(int)lds[addr]; addr = (addr * 5u + 77u) & 4095u; }
        } else if constexpr (OP == 12) {  // 16 x v_cmpx + restore
            asm volatile(R8("v_cmpx_gt_i32 vcc, 0, %0\n\ts_mov_b64 exec, -1\n\tv_cmpx_gt_i32 vcc, 1, %0\n\ts_mov_b64 exec, -1\n\t") : : "v"(z) : "vcc");
        } else if constexpr (OP == 13) {  // 16 x v_mul_hi_i32 + v_sub (the predictor's candidate form)
            asm volatile(R8("v_mul_hi_i32 %[a], %[x], %[y]\n\tv_sub_u32 %[z], %[z], %[a]\n\t") : [z] "+v"(z), [a] "=&v"(a) : [x] "v"(x), [y] "v"(y));
        }
    }
    unsigned r = x ^ y ^ a ^ b ^ (unsigned)z ^ s0 ^ s1 ^ (unsigned)sv ^ (unsigned)zm ^ zacc;
    if (r == 0x7fffffffu) out[0] = (int)r;
}

struct Case { const char *name; void (*fn)(int *, int); int units; const char *unit; };

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    int *dout;
    CK(hipMalloc(&dout, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    Case cases[] = {
        {"16 s_add_u32", k_issue<0>, 16, "instr"}, {"16 v_add_u32", k_issue<1>, 16, "instr"},
        {"16 v_add + 16 s_add interleaved", k_issue<2>, 16, "pair"}, {"16 v_add + 4 s_add", k_issue<3>, 4, "group of 4v+1s"},
        {"rotation: 10 VALU + 4 SALU (round-3 first form)", k_issue<4>, 8, "rotation"},
        {"rotation: v_min, s_not, s_mov -1 (10 VALU + 2 SALU)", k_issue<5>, 8, "rotation"},
        {"rotation: same, ROM word in a VGPR", k_issue<6>, 8, "rotation"},
        {"rotation: sign-mask xor/sub form (13 VALU, no EXEC)", k_issue<7>, 8, "rotation"},
        {"rotation: 64-bit mad form (round 2)", k_issue<8>, 8, "rotation"},
        {"s_cmp + s_cbranch not taken", k_issue<9>, 8, "pair"}, {"s_cmp + s_cbranch taken (+1 skipped VALU)", k_issue<10>, 8, "pair"},
        {"ds_read_i8 gather + dependent add", k_issue<11>, 8, "read"}, {"v_cmpx + s_mov exec", k_issue<12>, 16, "pair"},
        {"v_mul_hi_i32 + v_sub", k_issue<13>, 16, "pair"},
    };
    for (int wg_per_cu : {8, 4, 2}) {
        const int blocks = prop.multiProcessorCount * wg_per_cu;
        printf("\n%d waves per SIMD\n%-56s %10s %26s\n", wg_per_cu, "case", "ms", "cycles per unit per SIMD @2.4GHz");
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
            const double units_per_simd = (double)wg_per_cu * ITER * c.units;       // one wave per SIMD per workgroup
            printf("%-56s %10.3f %14.2f per %s\n", c.name, ms, ms * 1e-3 * 2.4e9 / units_per_simd, c.unit);
        }
    }
    return 0;
}
