// bhw_fused.hip -- fused strategy: whole-period work in one launch, no table
//
// Part of the hand-written HIP kernels for gfx950 (MI355X, CDNA4) behind include/bhw.h.  Hot path of the reference: phase
// accumulator -> CORDIC rotation chain (or Taylor LUT) -> weighted N-term cosine sum -> int32 coefficient (SURVEY section 8a
// rows a1-a11).  Integer semantics follow SURVEY App. A; reference lines are cited at each step.
#include "bhw_device.h"

namespace {

// Output stage of the fused kernels.  `through`: stores at agent scope (sc1), written through the L2 as they are produced.  Windows of
// up to 2^24 coefficients fit the L2s, and with write-back stores the whole window stayed dirty there until the end-of-kernel
// release wrote it back -- inside the gap before the next dependent launch: BH-4 2^20 / 24-bit 6.05 -> 5.64 us per window in a
// 20-call graph, BH-4 2^22 15.3 -> 13.1 us, BH-3 2^22 / 20-bit 10.2 -> 8.1 us (profiles/r04_short_windows_store_scope.txt).  Longer
// ones (explicit BHW_ALGO_FUSED up to phi_width 30, ownership parts of 2^26-point windows) keep write-back stores: at agent scope the
// tile kernel's 268 MB took 91 instead of 67 us (profiles/r04_ab_store_scope.txt) -- the same rule as k_table_combine_tile / k_tile9.
template <bool THROUGH>
__device__ __forceinline__ void emit_f(const BhwWinCfg &win, int32_t *__restrict__ out, uint64_t idx, int32_t w)
{
    if (win.apply_x) w = (int32_t)(((int64_t)__builtin_nontemporal_load(&win.apply_x[idx]) * (int64_t)w) >> win.apply_shift);
    if constexpr (THROUGH) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(out + idx), "v"(w) : "memory");
    else out[idx] = w;
}


// ---------------------------------------------------------------------------------------
// Fused fold kernel: whole-period work in ONE launch, no table.
//
// Lane r of the ring [0, N/8) owns the eight coefficients n = r + h*N/8 + j*N/4 (the quadrant + half-period fold of the tile
// kernel) and runs the first-quadrant CORDIC chains they need itself: two per odd harmonic (entries K*r and K*r + E/2), one
// per even harmonic -- 9 chains for 8 coefficients of a 7-term window instead of 48 in the direct kernel.  The 64 lanes of a
// wave are consecutive r, so for every chain their leaves are equally spaced in angle and share a rotation prefix exactly as
// the 64-leaf groups of k_table_build_shared do: phase 1 runs the (waves x chains) shared prefixes, one lane each, and parks
// them in LDS; phase 2 is one lane per r.  The rotation count is a run-time loop bound, so one instance serves every width.
//
// Used for (a) short whole windows (2^9 .. ~2^20 coefficients), where the table strategy is two dependent launches around a
// table round trip, and (b) interleaved ownership parts of a long window (bhw_generate_part_device): a device that owns 1/G of
// the ring needs 9/8G chains per coefficient of the whole window, below the table's 1/4 once G >= 5.
// The lanes of a launch are a list of runs of consecutive r (one run for a whole window; the 15 sibling runs of the tile plan,
// split where they wrap, for an ownership part).
// ---------------------------------------------------------------------------------------
// (kFoldRunsMax = 32, kFoldBlock = 256: bhw_plan.h)

struct BhwFoldPlan {
    uint32_t lut[34];                        // rescaled ROM as 32-bit words (quarter circle <= 2^32); [32], [33] = 0: the loop reads one ahead
    int64_t  x0;
    uint32_t n_iter, z_shr, z_shl, out_shr;
    uint32_t n_runs, phi_width, dat_width, ones_neg;
    uint32_t fast_mul, k24;                  // 1: every harmonic weight below 2^(W-3): one-instruction products (tile_harmonic FAST);
                                             // k24: first rotation the sign-product form of the narrow kernel is valid at (rot_mad24)
    uint32_t run0_r0, run0_end;              // = r0[0], r_end[0], next to the other scalars: a one-run launch reads one block of arguments
    uint32_t block, frames;                  // threads per workgroup of this launch (k_fold_direct: 64 or 256); identical frames to write (>= 1)
    uint32_t r0[kFoldRunsMax];               // first ring index of each run
    uint32_t r_end[kFoldRunsMax];            // one past its last
    uint32_t wg_first[kFoldRunsMax + 1];     // first workgroup of each run; [n_runs] = grid size
};

__host__ __device__ constexpr int fold_chains(int n_terms)      // first-quadrant chains per ring lane
{
    return n_terms == 2 ? 2 : n_terms == 3 ? 3 : n_terms == 4 ? 5 : n_terms == 5 ? 6 : 9;
}


// lutv: the rescaled ROM spread over the lanes of the wave (lane k holds lut[k]); v_readlane_b32 with the scalar rotation
// counter fetches a word in a few cycles.  (As a scalar load from the kernel arguments every rotation waited ~100+ cycles for
// its ROM word: short windows have too few waves to hide that, 9 us for a 2^16-point window.)
__device__ __forceinline__ void chain_from(int64_t &x, int64_t &y, int32_t &z, int k0, int n_iter, uint32_t lutv)
{
    int k = k0;                                                   // 1 <= k0 <= 20, n_iter <= 32
#pragma unroll 1
    for (; k < n_iter && k < kMad24From; ++k)                     // the first rotations: ROM words of 24 bits and more
        rot_step_dyn(x, y, z, k, (uint32_t)__builtin_amdgcn_readlane((int)lutv, k), false);
#pragma unroll 1
    for (; k < n_iter; ++k)
        rot_step_dyn(x, y, z, k, (uint32_t)__builtin_amdgcn_readlane((int)lutv, k), true);
}

// LOCKSTEP selects how phase 2 walks a lane's chains:
//   false: one chain after the other, each from its own split rotation (fewest rotations: the form for launches that fill the
//          chip, which are bound by vector issue);
//   true : all chains of the lane in one loop from the earliest split rotation of the wave -- a few rotations are repeated, but
//          the NCH independent rotations per iteration hide the ~15-cycle dependent-issue latency that a single chain exposes
//          when a launch has only a wave or two per SIMD (a 2^16-point window: 13 -> 6 us).
// Phase 1 parks the shared state after every prefix rotation (20 levels x tasks x 20 bytes of LDS), so either form picks its
// start level.
constexpr int kFoldLevels = (kPrefixMax < 32 ? kPrefixMax : 32) + 1;

//   FORM 2 (narrow): lockstep on 32-bit state, for configurations whose x, y, z fit 32-bit signed words (dat_width + out_shr <= 30).
//          Every wave runs the shared prefixes of its own chains in its first NCH lanes -- all of them up to the first rotation at
//          which ANY of them splits, so the hand-over is one wave-uniform level -- and broadcasts them with v_readlane: no LDS, no
//          barrier, no wave waiting for another one's serial phase.  The per-leaf rotations take the EXEC-masked form of the table
//          build (rot_narrow32 below: 9 plain VOP2 instructions and two scalar ones instead of v_mad_i64_i32 / v_alignbit_b32).
// One rotation of a 32-bit signed state; k and lutk are scalars.  Every lane of the wave is active on entry and on exit (phase 2
// runs whole waves: lanes beyond r_end compute and do not store).
__device__ __forceinline__ void rot_narrow32(int32_t &x, int32_t &y, int32_t &z, int k, uint32_t lutk)
{
    int32_t a, b;
    asm volatile("v_ashrrev_i32 %[a], %[k], %[y]\n\t"
                 "v_ashrrev_i32 %[b], %[k], %[x]\n\t"
                 "v_cmpx_gt_i32 vcc, 0, %[z]\n\t"
                 "v_add_u32 %[x], %[x], %[a]\n\t"
                 "v_sub_u32 %[y], %[y], %[b]\n\t"
                 "v_add_u32 %[z], %[l], %[z]\n\t"
                 "s_not_b64 exec, exec\n\t"
                 "v_sub_u32 %[x], %[x], %[a]\n\t"
                 "v_add_u32 %[y], %[y], %[b]\n\t"
                 "v_subrev_u32 %[z], %[l], %[z]\n\t"
                 "s_mov_b64 exec, -1"
                 : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [a] "=&v"(a), [b] "=&v"(b)
                 : [l] "s"(lutk), [k] "s"(k)
                 : "vcc", "scc");
}

// The same rotation as sign products: x -= sg * (y >> k), y += sg * (x >> k), z -= sg * lut[k] with sg = +1 / -1 in three
// v_mad_i32_i24 (24-bit factors, 32-bit addend) and no EXEC traffic.  Valid from rotation plan.k24 on, where the shifted coordinates
// and TWICE the ROM word fit 24 bits (the launcher derives it from the widths; 2 for a 2^20-point 24-bit window).
// The angle is carried doubled and odd, zz = 2 z + 1 (never 0; z >= 0 <=> zz >= 1), so that the sign is ONE instruction,
// sg = med3(zz, -1, 1): 7 vector instructions per rotation instead of 8 (shift-sign, or 1, negate before).  lut2 = 2 * lut[k].
__device__ __forceinline__ void rot_mad24(int32_t &x, int32_t &y, int32_t &zz, int k, uint32_t lut2)
{
    int32_t a, b, sg, ng;
    asm volatile("v_med3_i32 %[sg], %[z], -1, 1\n\t"
                 "v_ashrrev_i32 %[a], %[k], %[y]\n\t"
                 "v_ashrrev_i32 %[b], %[k], %[x]\n\t"
                 "v_sub_u32 %[ng], 0, %[sg]\n\t"
                 "v_mad_i32_i24 %[y], %[sg], %[b], %[y]\n\t"
                 "v_mad_i32_i24 %[x], %[ng], %[a], %[x]\n\t"
                 "v_mad_i32_i24 %[z], %[ng], %[l], %[z]"
                 : [x] "+v"(x), [y] "+v"(y), [z] "+v"(zz), [a] "=&v"(a), [b] "=&v"(b), [sg] "=&v"(sg), [ng] "=&v"(ng)
                 : [l] "s"(lut2), [k] "s"(k));
}

// Two / three chains in one statement, instruction by instruction side by side: a wave issues in order, and with one or two waves
// per SIMD (a 2^20-point window is 2 048 waves on 1 024 SIMDs) the dependent steps of a rotation (sign, negate, product)
// stall it unless other chains' instructions sit in between.
__device__ __forceinline__ void rot_mad24_x2(int32_t &x0, int32_t &y0, int32_t &z0, int32_t &x1, int32_t &y1, int32_t &z1, int k, uint32_t lut2)
{
    int32_t a0, b0, s0, n0, a1, b1, s1, n1;
    asm volatile("v_med3_i32 %[s0], %[z0], -1, 1\n\tv_med3_i32 %[s1], %[z1], -1, 1\n\t"
                 "v_ashrrev_i32 %[a0], %[k], %[y0]\n\tv_ashrrev_i32 %[a1], %[k], %[y1]\n\t"
                 "v_ashrrev_i32 %[b0], %[k], %[x0]\n\tv_ashrrev_i32 %[b1], %[k], %[x1]\n\t"
                 "v_sub_u32 %[n0], 0, %[s0]\n\tv_sub_u32 %[n1], 0, %[s1]\n\t"
                 "v_mad_i32_i24 %[y0], %[s0], %[b0], %[y0]\n\tv_mad_i32_i24 %[y1], %[s1], %[b1], %[y1]\n\t"
                 "v_mad_i32_i24 %[x0], %[n0], %[a0], %[x0]\n\tv_mad_i32_i24 %[x1], %[n1], %[a1], %[x1]\n\t"
                 "v_mad_i32_i24 %[z0], %[n0], %[l], %[z0]\n\tv_mad_i32_i24 %[z1], %[n1], %[l], %[z1]"
                 : [x0] "+v"(x0), [y0] "+v"(y0), [z0] "+v"(z0), [x1] "+v"(x1), [y1] "+v"(y1), [z1] "+v"(z1),
                   [a0] "=&v"(a0), [b0] "=&v"(b0), [s0] "=&v"(s0), [n0] "=&v"(n0), [a1] "=&v"(a1), [b1] "=&v"(b1), [s1] "=&v"(s1), [n1] "=&v"(n1)
                 : [l] "s"(lut2), [k] "s"(k));
}
__device__ __forceinline__ void rot_mad24_x3(int32_t &x0, int32_t &y0, int32_t &z0, int32_t &x1, int32_t &y1, int32_t &z1,
                                             int32_t &x2, int32_t &y2, int32_t &z2, int k, uint32_t lut2)
{
    int32_t a0, b0, s0, n0, a1, b1, s1, n1, a2, b2, s2, n2;
    asm volatile("v_med3_i32 %[s0], %[z0], -1, 1\n\tv_med3_i32 %[s1], %[z1], -1, 1\n\tv_med3_i32 %[s2], %[z2], -1, 1\n\t"
                 "v_ashrrev_i32 %[a0], %[k], %[y0]\n\tv_ashrrev_i32 %[a1], %[k], %[y1]\n\tv_ashrrev_i32 %[a2], %[k], %[y2]\n\t"
                 "v_ashrrev_i32 %[b0], %[k], %[x0]\n\tv_ashrrev_i32 %[b1], %[k], %[x1]\n\tv_ashrrev_i32 %[b2], %[k], %[x2]\n\t"
                 "v_sub_u32 %[n0], 0, %[s0]\n\tv_sub_u32 %[n1], 0, %[s1]\n\tv_sub_u32 %[n2], 0, %[s2]\n\t"
                 "v_mad_i32_i24 %[y0], %[s0], %[b0], %[y0]\n\tv_mad_i32_i24 %[y1], %[s1], %[b1], %[y1]\n\tv_mad_i32_i24 %[y2], %[s2], %[b2], %[y2]\n\t"
                 "v_mad_i32_i24 %[x0], %[n0], %[a0], %[x0]\n\tv_mad_i32_i24 %[x1], %[n1], %[a1], %[x1]\n\tv_mad_i32_i24 %[x2], %[n2], %[a2], %[x2]\n\t"
                 "v_mad_i32_i24 %[z0], %[n0], %[l], %[z0]\n\tv_mad_i32_i24 %[z1], %[n1], %[l], %[z1]\n\tv_mad_i32_i24 %[z2], %[n2], %[l], %[z2]"
                 : [x0] "+v"(x0), [y0] "+v"(y0), [z0] "+v"(z0), [x1] "+v"(x1), [y1] "+v"(y1), [z1] "+v"(z1), [x2] "+v"(x2), [y2] "+v"(y2), [z2] "+v"(z2),
                   [a0] "=&v"(a0), [b0] "=&v"(b0), [s0] "=&v"(s0), [n0] "=&v"(n0), [a1] "=&v"(a1), [b1] "=&v"(b1), [s1] "=&v"(s1), [n1] "=&v"(n1),
                   [a2] "=&v"(a2), [b2] "=&v"(b2), [s2] "=&v"(s2), [n2] "=&v"(n2)
                 : [l] "s"(lut2), [k] "s"(k));
}
// one rotation of all NCH chains of a lane (NCH = 2, 3, 5, 6, 9): groups of three, then what is left
template <int NCH>
__device__ __forceinline__ void rot_mad24_all(int32_t (&x)[NCH], int32_t (&y)[NCH], int32_t (&z)[NCH], int k, uint32_t lutk)
{
    constexpr int N3 = NCH / 3 * 3;
#pragma unroll
    for (int c = 0; c < N3; c += 3) rot_mad24_x3(x[c], y[c], z[c], x[c + 1], y[c + 1], z[c + 1], x[c + 2], y[c + 2], z[c + 2], k, lutk);
    if constexpr (NCH - N3 == 2) rot_mad24_x2(x[N3], y[N3], z[N3], x[N3 + 1], y[N3 + 1], z[N3 + 1], k, lutk);
    if constexpr (NCH - N3 == 1) rot_mad24(x[N3], y[N3], z[N3], k, lutk);
}

template <int NTERMS, int MODE, int FORM>
__global__ __launch_bounds__(kFoldBlock) void k_fold_direct(BhwWinCfg win, BhwFoldPlan plan, int32_t *__restrict__ out)
{
    constexpr bool LOCKSTEP = FORM == 1;
    using acc_t = typename std::conditional<MODE == 2, Sum32, int32_t>::type;
    constexpr int NCH = fold_chains(NTERMS);
    constexpr int kTasks = (kFoldBlock / 64) * NCH;
    __shared__ int64_t gx[kFoldLevels][kTasks], gy[kFoldLevels][kTasks];   // [level = rotations applied][wave * NCH + chain]
    __shared__ uint32_t gdz[kFoldLevels][kTasks];
    __shared__ int32_t gk[kTasks];
    BhwCordicCfg cfg;                                                     // tile_harmonic() reads ones_neg only
    cfg.ones_neg = plan.ones_neg;

    const uint32_t lq = plan.phi_width - 2;
    const uint32_t E = 1u << lq, emask = E - 1u, H = E >> 1;
    const uint32_t W = plan.dat_width;
    const int n_iter = (int)plan.n_iter;
    // the run of this workgroup.  One run (a whole window) reads its bounds at fixed kernel-argument offsets, in the same batch of
    // scalar loads as everything else; the search (at most kFoldRunsMax runs) costs two more dependent round trips of ~0.2 us each
    uint32_t run_r0 = plan.run0_r0, r_end = plan.run0_end, run_wg = 0u;
    uint32_t lutv_early = 0u;
    if constexpr (FORM == 2) {
        // lane k holds lut[k]: requested from the argument segment (plan follows win, 8-byte aligned) by hand, next to the scalar
        // loads, and waited for where the prefix first needs it -- the compiler would issue it after the scalar batch has returned
        constexpr uint32_t kPlanOffset = (uint32_t)((sizeof(BhwWinCfg) + 7u) & ~7u);
        const char *args = (const char *)__builtin_amdgcn_kernarg_segment_ptr() + kPlanOffset;   // (constant address space: C-style cast)
        const uint32_t off = (threadIdx.x & 31u) << 2;
        asm volatile("global_load_dword %0, %1, %2" : "=v"(lutv_early) : "v"(off), "s"(args));
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FORM == 2) {
        // short launches: every kernel argument the common path reads, requested before the first wait (left alone the compiler
        // fetches them in two or three dependent batches, a scalar-cache miss each)
        asm volatile("" :: "s"(plan.n_runs), "s"(run_r0), "s"(r_end), "s"(plan.x0), "s"(plan.n_iter), "s"(plan.z_shr), "s"(plan.z_shl),
                     "s"(plan.out_shr), "s"(plan.phi_width), "s"(plan.dat_width), "s"(plan.ones_neg), "s"(plan.fast_mul), "s"(plan.k24), "s"(plan.block),
                     "s"(win.aa[0]), "s"(win.aa[1]), "s"(win.aa[NTERMS > 2 ? 2 : 0]), "s"(win.aa[NTERMS > 3 ? 3 : 0]),
                     "s"(win.aa[NTERMS > 4 ? 4 : 0]), "s"(win.aa[NTERMS > 5 ? 5 : 0]), "s"(win.aa[NTERMS > 6 ? 6 : 0]), "s"(out));
    }
    if (plan.n_runs > 1u) {
        uint32_t run = 0;
        while (run + 1u < plan.n_runs && blockIdx.x >= plan.wg_first[run + 1u]) ++run;
        run_r0 = plan.r0[run];
        r_end = plan.r_end[run];
        run_wg = plan.wg_first[run];
    }
    const uint32_t wg_r0 = run_r0 + (blockIdx.x - run_wg) * plan.block;
    const uint32_t n_waves = plan.block >> 6;
    const uint32_t z_shr = plan.z_shr, z_shl = plan.z_shl, out_shr = plan.out_shr;

    // ---- phase 1: shared rotation prefix of every (wave, chain) ----
    // chain slot c -> harmonic K and half-period image: (1,0) (1,1) (2) (3,0) (3,1) (4) (5,0) (5,1) (6)
    if constexpr (FORM != 2) {
    if (threadIdx.x < n_waves * NCH) {
        const uint32_t wv = threadIdx.x / NCH, c = threadIdx.x % NCH;
        const uint32_t K = 2u * (c / 3u) + 1u + (c % 3u == 2u ? 1u : 0u);
        const uint32_t hodd = (c % 3u == 1u) ? 1u : 0u;
        const uint32_t rf = wg_r0 + (wv << 6);
        const uint32_t t0 = (K * rf + hodd * H) & emask;
        const uint32_t tl = t0 + 63u * K;                                 // last leaf, if the 64 leaves do not wrap past E
        const uint32_t z0f = (t0 >> z_shr) << z_shl;
        bool live = tl <= emask;                                          // wrapped groups are not contiguous in angle: no sharing
        const uint32_t span = live ? ((tl >> z_shr) << z_shl) - z0f : 0u;
        int64_t x = plan.x0, y = plan.x0;                                 // after rotation 0 (z0 >= 0 always adds)
        int32_t zf = (int32_t)(z0f - plan.lut[0]);
        int k = 1;
        gx[1][threadIdx.x] = x;
        gy[1][threadIdx.x] = y;
        gdz[1][threadIdx.x] = (uint32_t)zf - z0f;                         // z_level(leaf) = z0(leaf) + this, for every leaf of the group
#pragma unroll
        for (int kk = 1; kk < kFoldLevels - 1; ++kk) {
            if (live && kk < n_iter) {
                const int32_t zl = (int32_t)((uint32_t)zf + span);
                if ((zf < 0) != (zl < 0)) {
                    live = false;                                         // the group splits at rotation kk
                } else {
                    rot_step(x, y, zf, kk, plan.lut[kk]);
                    k = kk + 1;
                    gx[kk + 1][threadIdx.x] = x;
                    gy[kk + 1][threadIdx.x] = y;
                    gdz[kk + 1][threadIdx.x] = (uint32_t)zf - z0f;
                }
            }
        }
        gk[threadIdx.x] = k;
    }
    __syncthreads();
    }

    // ---- phase 2: one lane per r ----
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t r = wg_r0 + threadIdx.x;
    uint32_t lutv;                                                        // lane k (and k + 32) holds lut[k]
    if constexpr (FORM == 2) {
        lutv = lutv_early;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(lutv));
    } else lutv = plan.lut[threadIdx.x & 31u];
    const bool fast = plan.fast_mul != 0u;                                // (VHDL rule too: (q + 1) >> 1 on q = mul_hi, two-word sums)
    acc_t acc[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (MODE == 2) acc[h][j] = sum32_first(win.aa[0]);
            else acc[h][j] = win.aa[0];
        }
    // slot -> (K, half-period image) as compile-time functions of the slot
    auto slot_K = [](int slot) { return 2 * (slot / 3) + 1 + (slot % 3 == 2 ? 1 : 0); };
    auto slot_h = [](int slot) { return slot % 3 == 1 ? 1u : 0u; };
    int2 cs[NCH];
    if constexpr (LOCKSTEP) {
        int kc = 32;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int kq = __builtin_amdgcn_readfirstlane(gk[wave * NCH + c]);
            kc = kq < kc ? kq : kc;
        }
        int64_t x[NCH], y[NCH];
        int32_t z[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t i = wave * NCH + c;
            x[c] = gx[kc][i];
            y[c] = gy[kc][i];
            const uint32_t t = ((uint32_t)slot_K(c) * r + slot_h(c) * H) & emask;
            z[c] = (int32_t)(((t >> z_shr) << z_shl) + gdz[kc][i]);
        }
        int k = kc;
#pragma unroll 1
        for (; k < n_iter && k < kMad24From; ++k) {
            const uint32_t lutk = (uint32_t)__builtin_amdgcn_readlane((int)lutv, k);
#pragma unroll
            for (int c = 0; c < NCH; ++c) rot_step_dyn(x[c], y[c], z[c], k, lutk, false);
        }
#pragma unroll 1
        for (; k < n_iter; ++k) {
            const uint32_t lutk = (uint32_t)__builtin_amdgcn_readlane((int)lutv, k);
#pragma unroll
            for (int c = 0; c < NCH; ++c) rot_step_dyn(x[c], y[c], z[c], k, lutk, true);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) cs[c] = make_int2((int32_t)(x[c] >> out_shr), (int32_t)(y[c] >> out_shr));
    }
    if constexpr (FORM == 2) {
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t rf = wg_r0 + (wave << 6);                          // first ring lane of this wave
        // prefixes: lane c < NCH follows the group of chain c; the loop ends for all of them at the first split
        const uint32_t pc = lane < (uint32_t)NCH ? lane : 0u;
        const uint32_t pK = 2u * (pc / 3u) + 1u + (pc % 3u == 2u ? 1u : 0u), ph = (pc % 3u == 1u) ? 1u : 0u;
        const uint32_t t0 = (pK * rf + ph * H) & emask;
        const uint32_t tl = t0 + 63u * pK;
        const uint32_t z0f = (t0 >> z_shr) << z_shl;
        const bool whole = tl <= emask;                                   // wrapped groups are not contiguous in angle: no sharing
        const uint32_t span = (whole && lane < (uint32_t)NCH) ? ((tl >> z_shr) << z_shl) - z0f : 0u;   // idle lanes never split
        int32_t px = (int32_t)plan.x0, py = (int32_t)plan.x0;            // after rotation 0 (z0 >= 0 always adds)
        int32_t pz = (int32_t)(z0f - (uint32_t)__builtin_amdgcn_readlane((int)lutv, 0));
        int kc = 1;
        const int k_stop = __builtin_amdgcn_ballot_w64(!whole && lane < (uint32_t)NCH) != 0ull ? 1 : n_iter;   // a wrapped group: no prefix
        // Rotations 1 .. kc - 1 on the group states, until some group splits (sign of z differs at its two end leaves): one statement,
        // so that the serial depth of a level is what the arithmetic needs -- both outcomes of z side by side and one select, the
        // split test of the next level on the selected z, the branch on a compare issued four instructions earlier -- instead of the
        // compare / mask / branch ladder the compiler builds around a wave vote (190 -> ~90 cycles per level; every wave of a short
        // launch starts with ~10 of these).  Two instructions sit between a lane read or compare that writes scalar registers and
        // the vector instruction that reads them (gfx940+ hazard; nothing inserts no-ops inside a statement).
        {
            int32_t t, a, b, za, zb, xa, xb, ya, yb;
            uint32_t sl;
            uint64_t ng;
            asm volatile("s_cmp_ge_i32 %[kc], %[ks]\n\t"
                         "s_cbranch_scc1 1f\n"
                         "0:\n\t"
                         "v_add_u32 %[t], %[pz], %[sp]\n\t"
                         "v_readlane_b32 %[sl], %[lv], %[kc]\n\t"
                         "v_xor_b32 %[t], %[t], %[pz]\n\t"
                         "v_cmp_gt_i32_e64 %[ng], 0, %[pz]\n\t"
                         "v_cmp_gt_i32 vcc, 0, %[t]\n\t"
                         "v_ashrrev_i32 %[a], %[kc], %[py]\n\t"
                         "v_ashrrev_i32 %[b], %[kc], %[px]\n\t"
                         "v_add_u32 %[za], %[sl], %[pz]\n\t"
                         "v_subrev_u32 %[zb], %[sl], %[pz]\n\t"
                         "s_cbranch_vccnz 1f\n\t"
                         "v_cndmask_b32_e64 %[pz], %[zb], %[za], %[ng]\n\t"
                         "v_add_u32 %[xa], %[px], %[a]\n\t"
                         "v_sub_u32 %[xb], %[px], %[a]\n\t"
                         "v_sub_u32 %[ya], %[py], %[b]\n\t"
                         "v_add_u32 %[yb], %[py], %[b]\n\t"
                         "s_add_u32 %[kc], %[kc], 1\n\t"
                         "v_cndmask_b32_e64 %[px], %[xb], %[xa], %[ng]\n\t"
                         "v_cndmask_b32_e64 %[py], %[yb], %[ya], %[ng]\n\t"
                         "s_cmp_lt_i32 %[kc], %[ks]\n\t"
                         "s_cbranch_scc1 0b\n"
                         "1:"
                         : [px] "+v"(px), [py] "+v"(py), [pz] "+v"(pz), [kc] "+s"(kc), [t] "=&v"(t), [a] "=&v"(a), [b] "=&v"(b),
                           [za] "=&v"(za), [zb] "=&v"(zb), [xa] "=&v"(xa), [xb] "=&v"(xb), [ya] "=&v"(ya), [yb] "=&v"(yb),
                           [sl] "=&s"(sl), [ng] "=&s"(ng)
                         : [sp] "v"(span), [lv] "v"(lutv), [ks] "s"(k_stop)
                         : "vcc", "scc");
        }
        const uint32_t pdz = (uint32_t)pz - z0f;
        int32_t x[NCH], y[NCH], z[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            x[c] = __builtin_amdgcn_readlane(px, c);
            y[c] = __builtin_amdgcn_readlane(py, c);
            const uint32_t t = ((uint32_t)slot_K(c) * r + slot_h(c) * H) & emask;
            z[c] = (int32_t)(((t >> z_shr) << z_shl) + (uint32_t)__builtin_amdgcn_readlane((int)pdz, c));
        }
        int k = kc;
        const int k24 = (int)plan.k24 < n_iter ? (int)plan.k24 : n_iter;
#pragma unroll 1
        for (; k < k24; ++k) {                                            // (only waves whose prefix ended before rotation k24)
            const uint32_t lutk = (uint32_t)__builtin_amdgcn_readlane((int)lutv, k);
#pragma unroll
            for (int c = 0; c < NCH; ++c) rot_narrow32(x[c], y[c], z[c], k, lutk);
        }
        // from here on the angle is carried as 2 z + 1 (rot_mad24); it is not read after the last rotation
        const uint32_t lutv2 = lutv << 1;
#pragma unroll
        for (int c = 0; c < NCH; ++c) z[c] = (int32_t)(((uint32_t)z[c] << 1) | 1u);
#pragma unroll 1
        for (; k < n_iter; ++k) {
            const uint32_t lut2 = (uint32_t)__builtin_amdgcn_readlane((int)lutv2, k);
            rot_mad24_all<NCH>(x, y, z, k, lut2);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) cs[c] = make_int2(x[c] >> out_shr, y[c] >> out_shr);
    }
    auto chain = [&](const uint32_t slot, const uint32_t K, const uint32_t hodd) -> int2 {
        if constexpr (FORM != 0) return cs[slot];
        const uint32_t i = wave * NCH + slot;                             // scalar: the parked state is read as a broadcast
        const int k0 = __builtin_amdgcn_readfirstlane(gk[i]);
        int64_t x = gx[k0][i], y = gy[k0][i];
        const uint32_t t = (K * r + hodd * H) & emask;
        int32_t z = (int32_t)(((t >> z_shr) << z_shl) + gdz[k0][i]);
        chain_from(x, y, z, k0, n_iter, lutv);
        return make_int2((int32_t)(x >> out_shr), (int32_t)(y >> out_shr));
    };
#define BHW_FD_TERM(K, HH, CS, ACC, OFF)                                                                 \
    if (fast) {                                                           /* scalar branch */        \
        tile_harmonic<K, MODE, ring_qbase(K, HH), ring_qbits(K, HH), true>(cfg, (int32_t)((uint32_t)win.aa[K] << (34u - W)), W, CS, \
                                                                                ((uint32_t)K * (r + (uint32_t)HH * H)) >> lq, sv); \
        tile_accumulate<K, OFF, true>(sv, ACC);                                                 \
    } else {                                                                                         \
        tile_harmonic<K, MODE, ring_qbase(K, HH), ring_qbits(K, HH)>(cfg, win.aa[K], W, CS, ((uint32_t)K * (r + (uint32_t)HH * H)) >> lq, sv); \
        tile_accumulate<K, OFF>(sv, ACC);                                                            \
    }
#define BHW_FD_HARMONIC(K)                                                                           \
    if constexpr (NTERMS > K) {                                                                      \
        constexpr uint32_t slot = ((K - 1) / 2) * 3 + ((K & 1) ? 0 : 2);                             \
        int32_t sv[4];                                                                               \
        const int2 cs0 = chain(slot, K, 0u);                                                         \
        if constexpr ((K & 1) != 0) {                                                                \
            BHW_FD_TERM(K, 0, cs0, acc[0], 0)                                                        \
            const int2 cs1 = chain(slot + 1u, K, 1u);                                                \
            BHW_FD_TERM(K, 1, cs1, acc[1], 0)                                                        \
        } else {                                                                                     \
            if (fast) {                                                                              \
                tile_harmonic<K, MODE, ring_qbase(K, 0), ring_qbits(K, 0), true>(cfg, (int32_t)((uint32_t)win.aa[K] << (34u - W)), W, cs0, ((uint32_t)K * r) >> lq, sv); \
                tile_accumulate<K, 0, true>(sv, acc[0]);                                        \
                tile_accumulate<K, K / 2, true>(sv, acc[1]);                                    \
            } else {                                                                                 \
                tile_harmonic<K, MODE, ring_qbase(K, 0), ring_qbits(K, 0)>(cfg, win.aa[K], W, cs0, ((uint32_t)K * r) >> lq, sv); \
                tile_accumulate<K, 0>(sv, acc[0]);                                                   \
                tile_accumulate<K, K / 2>(sv, acc[1]);                                               \
            }                                                                                        \
        }                                                                                            \
    }
    BHW_FD_HARMONIC(1) BHW_FD_HARMONIC(2) BHW_FD_HARMONIC(3) BHW_FD_HARMONIC(4) BHW_FD_HARMONIC(5) BHW_FD_HARMONIC(6)
#undef BHW_FD_TERM
#undef BHW_FD_HARMONIC
    if (r >= r_end) return;
    int32_t vout[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (MODE == 2) vout[h][j] = w32_final<BHW_COMBINE_VHDL>(acc[h][j], W, NTERMS);
            else vout[h][j] = (int32_t)((uint32_t)acc[h][j] << (32u - W)) >> (32u - W);   // (win_t)(...) wrap to W bits
        }
    if (plan.frames <= 1u) {                                              // scalar
        // (ONE wave-uniform branch around the eight stores, not one per store: the short windows count tenths of a microsecond)
        if (plan.phi_width <= 24u) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) emit_f<true>(win, out, (uint64_t)(r + (uint32_t)h * H) + (uint64_t)j * E, vout[h][j]);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) emit_f<false>(win, out, (uint64_t)(r + (uint32_t)h * H) + (uint64_t)j * E, vout[h][j]);
        }
        return;
    }
    // Batched identical frames (bhw_generate_batched_device: the coefficient stream is periodic, src/bh_win_7term.vhd:92-97): the
    // lane's eight coefficients go into every frame of this workgroup row's share -- the frames are divided over gridDim.y, each
    // row computing the period again (a few microseconds of latency) instead of a second kernel reading it back.  The whole batch
    // is far larger than the L2s: plain write-back stores.
    const uint32_t per = (plan.frames + gridDim.y - 1u) / gridDim.y;
    const uint32_t f0 = blockIdx.y * per, f1 = f0 + per < plan.frames ? f0 + per : plan.frames;
    for (uint32_t f = f0; f < f1; ++f) {
        int32_t *frame = out + ((uint64_t)f << plan.phi_width);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) frame[(uint64_t)(r + (uint32_t)h * H) + (uint64_t)j * E] = vout[h][j];
    }
}

// Fused fold kernel, short-launch form: the chains of 64 ring lanes SPLIT OVER THE FOUR WAVES of a workgroup (one per SIMD).
// A launch of a few thousand lanes is bound by the serial depth of one wave -- prefix, then 5 .. 9 chains -- not by issue
// slots; here wave w takes chains w, w + 4, w + 8 of the same 64 lanes (at most three), runs their prefixes in its first lanes,
// broadcasts them with v_readlane (no LDS, no barrier), walks them together from the earliest split level (each chain joining
// at its own), and hands the (c, s) pairs over through LDS; waves 0 and 1 then sum the h = 0 / h = 1 images.
// The z recurrence takes the short path z += sg * (-lut) (sign, or, mad: three dependent instructions instead of four).
template <bool MAD24>
__device__ __forceinline__ void rot_step_lat(int64_t &x, int64_t &y, int32_t &z, int k, int32_t nlutk)
{
    const int32_t sg = (z >> 31) | 1;
    int32_t ys = (int32_t)__builtin_amdgcn_alignbit((uint32_t)((uint64_t)y >> 32), (uint32_t)y, (uint32_t)k);
    int32_t xs = (int32_t)__builtin_amdgcn_alignbit((uint32_t)((uint64_t)x >> 32), (uint32_t)x, (uint32_t)k);
    asm volatile("" : "+v"(ys), "+v"(xs));
    if constexpr (MAD24) z += __mul24(sg, nlutk);                // |lut[k]| < 2^23 from rotation kMad24From on
    else z += sg * nlutk;                                        // 32-bit product: exact modulo 2^32 for every ROM word
    x -= (int64_t)sg * (int64_t)ys;
    y += (int64_t)sg * (int64_t)xs;
}

// SW = waves per 64 ring lanes.  Round 5: one chain per wave (SW = the chain count, 576 threads for a 7-term window) instead of four
// waves with up to three chains each: a launch this short has about one wave per SIMD either way, so what counts is the serial
// depth of a wave -- one chain's dependent rotations instead of three chains' worth of issue (profiles/r05_short_windows_split.txt).
template <int NTERMS, int MODE, int SW>
__global__ __launch_bounds__(64 * SW) void k_fold_split(BhwWinCfg win, BhwFoldPlan plan, int32_t *__restrict__ out)
{
    using acc_t = typename std::conditional<MODE == 2, Sum32, int32_t>::type;
    constexpr int NCH = fold_chains(NTERMS);
    static_assert(SW >= 2 && SW <= 16, "waves 0 and 1 sum the two half-period images");
    constexpr int MAXC = (NCH + SW - 1) / SW;                             // chains per wave
    __shared__ int2 cs_s[NCH][64];
    BhwCordicCfg cfg;                                                     // tile_harmonic() reads ones_neg only
    cfg.ones_neg = plan.ones_neg;
    const uint32_t lq = plan.phi_width - 2;
    const uint32_t E = 1u << lq, emask = E - 1u, H = E >> 1;
    const uint32_t W = plan.dat_width;
    const int n_iter = (int)plan.n_iter;
    uint32_t run_r0 = plan.run0_r0, r_end = plan.run0_end, run_wg = 0u;     // (see k_fold_direct)
    if (plan.n_runs > 1u) {
        uint32_t run = 0;
        while (run + 1u < plan.n_runs && blockIdx.x >= plan.wg_first[run + 1u]) ++run;
        run_r0 = plan.r0[run];
        r_end = plan.r_end[run];
        run_wg = plan.wg_first[run];
    }
    const uint32_t wg_r0 = run_r0 + (blockIdx.x - run_wg) * 64u;          // 64 ring lanes per workgroup
    const uint32_t z_shr = plan.z_shr, z_shl = plan.z_shl, out_shr = plan.out_shr;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t r = wg_r0 + lane;
    const int32_t nlutv = -(int32_t)plan.lut[lane & 31u];                 // lane k holds -lut[k]

    auto chain_K = [](uint32_t c) { return 2u * (c / 3u) + 1u + (c % 3u == 2u ? 1u : 0u); };
    auto chain_h = [](uint32_t c) { return (c % 3u == 1u) ? 1u : 0u; };

    // ---- phase 1: lane i < MAXC runs the shared prefix of this wave's chain i (chain index wave + SW i) ----
    int64_t px = plan.x0, py = plan.x0;
    uint32_t pdz = 0u;
    int pk = 1;
    {
        const uint32_t ci = lane < (uint32_t)MAXC ? lane : 0u;
        uint32_t c = wave + (uint32_t)SW * ci;
        if (c >= (uint32_t)NCH) c = wave < (uint32_t)NCH ? wave : 0u;     // idle lanes / waves repeat a valid chain
        const uint32_t K = chain_K(c), hodd = chain_h(c);
        const uint32_t t0 = (K * wg_r0 + hodd * H) & emask;
        const uint32_t tl = t0 + 63u * K;
        const uint32_t z0f = (t0 >> z_shr) << z_shl;
        bool live = tl <= emask;
        const uint32_t span = live ? ((tl >> z_shr) << z_shl) - z0f : 0u;
        int32_t zf = (int32_t)(z0f - plan.lut[0]);
        constexpr int kmax = kPrefixMax < 32 ? kPrefixMax : 32;
#pragma unroll
        for (int kk = 1; kk < kmax; ++kk) {
            if (live && kk < n_iter) {
                const int32_t zl = (int32_t)((uint32_t)zf + span);
                if ((zf < 0) != (zl < 0)) {
                    live = false;
                } else {
                    rot_step(px, py, zf, kk, plan.lut[kk]);
                    pk = kk + 1;
                }
            }
        }
        pdz = (uint32_t)zf - z0f;
    }
    // ---- phase 2: this wave's chains, every lane its own leaf ----
    int64_t x[MAXC], y[MAXC];
    int32_t z[MAXC];
    int k0[MAXC];
    int kc = 32;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const uint32_t c = wave + (uint32_t)SW * (uint32_t)i;
        const uint32_t cc = c < (uint32_t)NCH ? c : 0u;
        const uint32_t xl = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)px, i), xh = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)px >> 32), i);
        const uint32_t yl = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)py, i), yh = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)py >> 32), i);
        x[i] = (int64_t)(((uint64_t)xh << 32) | xl);
        y[i] = (int64_t)(((uint64_t)yh << 32) | yl);
        const uint32_t dz = (uint32_t)__builtin_amdgcn_readlane((int)pdz, i);
        k0[i] = c < (uint32_t)NCH ? __builtin_amdgcn_readlane(pk, i) : 32;     // chains this wave does not have never start
        const uint32_t t = (chain_K(cc) * r + chain_h(cc) * H) & emask;
        z[i] = (int32_t)(((t >> z_shr) << z_shl) + dz);
        kc = k0[i] < kc ? k0[i] : kc;
    }
    int k = kc;
#pragma unroll 1
    for (; k < n_iter && k < kMad24From; ++k) {
        const int32_t nlutk = __builtin_amdgcn_readlane(nlutv, k);
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (k >= k0[i]) rot_step_lat<false>(x[i], y[i], z[i], k, nlutk);   // scalar guard: k0 is wave-uniform
    }
#pragma unroll 1
    for (; k < n_iter; ++k) {
        const int32_t nlutk = __builtin_amdgcn_readlane(nlutv, k);
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
            if (k >= k0[i]) rot_step_lat<true>(x[i], y[i], z[i], k, nlutk);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const uint32_t c = wave + (uint32_t)SW * (uint32_t)i;
        if (c < (uint32_t)NCH) cs_s[c][lane] = make_int2((int32_t)(x[i] >> out_shr), (int32_t)(y[i] >> out_shr));
    }
    __syncthreads();
    if (wave >= 2u || r >= r_end) return;

    // ---- combine: wave h sums the four images n = r + h*N/8 + j*N/4 ----
    auto combine = [&](auto hc) {
        constexpr int HH = decltype(hc)::value;
        acc_t acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (MODE == 2) acc[j] = sum32_first(win.aa[0]);
            else acc[j] = win.aa[0];
        }
#define BHW_FS_HARMONIC(K)                                                                                            \
        if constexpr (NTERMS > K) {                                                                                   \
            constexpr uint32_t slot = ((K - 1) / 2) * 3 + ((K & 1) ? (uint32_t)HH : 2u);                              \
            int32_t sv[4];                                                                                            \
            if constexpr ((K & 1) != 0) {                                                                             \
                tile_harmonic<K, MODE, ring_qbase(K, HH), ring_qbits(K, HH)>(cfg, win.aa[K], W, cs_s[slot][lane],     \
                                                                            ((uint32_t)K * (r + (uint32_t)HH * H)) >> lq, sv); \
                tile_accumulate<K, 0>(sv, acc);                                                                       \
            } else {                                                                                                  \
                tile_harmonic<K, MODE, ring_qbase(K, 0), ring_qbits(K, 0)>(cfg, win.aa[K], W, cs_s[slot][lane], ((uint32_t)K * r) >> lq, sv); \
                tile_accumulate<K, (HH ? K / 2 : 0)>(sv, acc);                                                        \
            }                                                                                                         \
        }
        BHW_FS_HARMONIC(1) BHW_FS_HARMONIC(2) BHW_FS_HARMONIC(3) BHW_FS_HARMONIC(4) BHW_FS_HARMONIC(5) BHW_FS_HARMONIC(6)
#undef BHW_FS_HARMONIC
        int32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (MODE == 2) v[j] = w32_final<BHW_COMBINE_VHDL>(acc[j], W, NTERMS);
            else v[j] = (int32_t)((uint32_t)acc[j] << (32u - W)) >> (32u - W);
        }
        if (plan.phi_width <= 24u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) emit_f<true>(win, out, (uint64_t)(r + (uint32_t)HH * H) + (uint64_t)j * E, v[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) emit_f<false>(win, out, (uint64_t)(r + (uint32_t)HH * H) + (uint64_t)j * E, v[j]);
        }
    };
    if (wave == 0u) combine(std::integral_constant<int, 0>{});
    else combine(std::integral_constant<int, 1>{});
}

} // namespace

int bhwk_fold_direct(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const BhwFoldRun *runs, uint32_t n_runs, int32_t *d_out, uint32_t frames)
{
    if (!n_runs) return 0;
    if (n_runs > (uint32_t)kFoldRunsMax || !bhwk_fold_direct_applicable(c)) return (int)hipErrorInvalidValue;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    BhwFoldPlan plan;
    memset(&plan, 0, sizeof plan);
    for (uint32_t k = 0; k < 32; ++k) plan.lut[k] = (uint32_t)c.lut[k];
    plan.x0 = c.x0;
    plan.n_iter = c.n_iter;
    plan.z_shr = c.z_shr;
    plan.z_shl = c.z_shl;
    plan.out_shr = c.out_shr;
    plan.n_runs = n_runs;
    plan.phi_width = c.phi_width;
    plan.dat_width = c.dat_width;
    plan.ones_neg = c.ones_neg;
    plan.frames = frames ? frames : 1u;
    plan.fast_mul = c.dat_width >= 3 ? 1u : 0u;                          // either cosine-sum rule (the VHDL one keeps its two-word sums)
    for (uint32_t k = 1; k < w.n_terms && plan.fast_mul; ++k) {
        const int64_t lim = (int64_t)1 << (c.dat_width - 3);
        if ((int64_t)w.aa[k] >= lim || (int64_t)w.aa[k] <= -lim) plan.fast_mul = 0u;
    }
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_runs; ++i) total += runs[i].r_end - runs[i].r0;
    // form of the kernel for this many lanes (bhwp_fold_form, bhw_plan.cpp: the measurements behind the thresholds are quoted there)
    const int form = bhwp_fold_form(c, w, total);
    const bool split = form == BHWP_FOLD_SPLIT, narrow = form == BHWP_FOLD_NARROW, lockstep = form == BHWP_FOLD_LOCKSTEP;
    // short launches: one wave per workgroup spreads the few waves over more CUs
    // (one wave per workgroup for the longer launches of the barrier-free narrow form too: no gain, profiles/r04_short_windows_store_scope.txt)
    const uint32_t block = total <= 64u * 1024u ? 64u : (uint32_t)kFoldBlock;
    uint32_t wg = 0;
    for (uint32_t i = 0; i < n_runs; ++i) {
        plan.r0[i] = runs[i].r0;
        plan.r_end[i] = runs[i].r_end;
        plan.wg_first[i] = wg;
        wg += (runs[i].r_end - runs[i].r0 + block - 1u) / block;
    }
    plan.wg_first[n_runs] = wg;
    plan.block = block;
    plan.run0_r0 = plan.r0[0];
    plan.run0_end = plan.r_end[0];
    if (!wg) return 0;
    const int mode = (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0);
    // batched identical frames: the frames are divided over grid.y -- enough rows that the launch has ~4 waves per SIMD to keep
    // the store stream going (each row computes the period again: microseconds of latency, no traffic)
    uint32_t gy = 1u;
    if (plan.frames > 1u) {
        if (split || w.apply_x != nullptr) return (int)hipErrorInvalidValue;    // (the caller sends these through one period + replicate)
        const uint32_t waves_x = wg * (block >> 6);
        gy = (4096u + waves_x - 1u) / (waves_x ? waves_x : 1u);
        if (gy > plan.frames) gy = plan.frames;
        if (gy < 1u) gy = 1u;
    }
    const dim3 grid(wg, gy), blk(block);
    plan.k24 = bhwp_fold_k24(c);
    const uint32_t split_waves = (uint32_t)(w.n_terms == 2 ? 2 : w.n_terms == 3 ? 3 : w.n_terms == 4 ? 5 : w.n_terms == 5 ? 6 : 9);   // fold_chains(): one chain per wave
    dim3 grid_s(0), blk_s(64u * split_waves);
    if (split) {
        uint32_t wgs = 0;
        for (uint32_t i = 0; i < n_runs; ++i) {
            plan.wg_first[i] = wgs;
            wgs += (runs[i].r_end - runs[i].r0 + 63u) / 64u;
        }
        plan.wg_first[n_runs] = wgs;
        grid_s = dim3(wgs);
    }
#define BHW_FD_NT_M(NT, M)                                                                                  \
    do {                                                                                                    \
        if (split) BHW_LAUNCH((k_fold_split<NT, M, fold_chains(NT)>), grid_s, blk_s, 0, st, w, plan, d_out); \
        else if (narrow)   BHW_LAUNCH((k_fold_direct<NT, M, 2>), grid, blk, 0, st, w, plan, d_out);      \
        else if (lockstep) BHW_LAUNCH((k_fold_direct<NT, M, 1>), grid, blk, 0, st, w, plan, d_out);      \
        else               BHW_LAUNCH((k_fold_direct<NT, M, 0>), grid, blk, 0, st, w, plan, d_out);      \
    } while (0)
#define BHW_FD_NT(NT)                                                                                       \
    do {                                                                                                    \
        if (mode == 0)      BHW_FD_NT_M(NT, 0);                                                             \
        else if (mode == 1) BHW_FD_NT_M(NT, 1);                                                             \
        else                BHW_FD_NT_M(NT, 2);                                                             \
    } while (0)
    switch (w.n_terms) {
    case 2: BHW_FD_NT(2); break;
    case 3: BHW_FD_NT(3); break;
    case 4: BHW_FD_NT(4); break;
    case 5: BHW_FD_NT(5); break;
    case 7: BHW_FD_NT(7); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef BHW_FD_NT
#undef BHW_FD_NT_M
    return finish(hipSuccess);
}

