// bhw_direct.hip -- direct strategy (one lane per coefficient), cordic() sweeps, replicate
//
// Part of the hand-written HIP kernels for gfx950 (MI355X, CDNA4) behind include/bhw.h.  Hot path of the reference: phase
// accumulator -> CORDIC rotation chain (or Taylor LUT) -> weighted N-term cosine sum -> int32 coefficient (SURVEY section 8a
// rows a1-a11).  Integer semantics follow SURVEY App. A; reference lines are cited at each step.
#include "bhw_device.h"

namespace {

// ---------------------------------------------------------------------------------------
// Direct kernel: one lane per coefficient, K-1 CORDIC chains per lane.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_direct(BhwCordicCfg cfg, BhwWinCfg win, uint64_t n0, uint64_t count,
                                                    int32_t *__restrict__ out)
{
    __shared__ T lut_s[32];
    stage_lut<T>(cfg, lut_s);
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t mask = (cfg.phi_width >= 32) ? 0xFFFFFFFFu : ((1u << cfg.phi_width) - 1u);
    const uint32_t n = (uint32_t)(n0 + i) & mask;                   // phase counter wraps: bh_win_7term.vhd:92-97
    int64_t acc = win.aa[0];
    for (uint32_t k = 1; k < win.n_terms; ++k) {
        const uint32_t theta = (k * n) & mask;                     // ph_ink += k: bh_win_7term.vhd:187-194 | cordic(k*i): win_function.cpp:361-366
        int32_t c, s;
        cordic_full<T>(cfg, lut_s, theta, c, s);
        combine_term(acc, win.aa[k], c, k, cfg.dat_width, win.combine);
    }
    emit(win, out, i, combine_final(acc, cfg.dat_width, win.combine, win.n_terms));
}

// sin/cos sweep: cordic() alone.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_sincos(BhwCordicCfg cfg, uint64_t theta0, uint64_t count,
                                                    int32_t *__restrict__ d_sin, int32_t *__restrict__ d_cos)
{
    __shared__ T lut_s[32];
    stage_lut<T>(cfg, lut_s);
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t mask = (cfg.phi_width >= 32) ? 0xFFFFFFFFu : ((1u << cfg.phi_width) - 1u);
    int32_t c, s;
    cordic_full<T>(cfg, lut_s, (uint32_t)(theta0 + i) & mask, c, s);
    if (d_sin) d_sin[i] = s;
    if (d_cos) d_cos[i] = c;
}

// ---------------------------------------------------------------------------------------
// Direct kernel, fast form: one lane per coefficient, K-1 full CORDIC chains per lane in the same
// 8-instruction rotation step as the table build (no sharing between lanes), rescaled ROM staged in LDS
// and read with immediate offsets, one coalesced int32 store per lane.
// ---------------------------------------------------------------------------------------
template <int NITER>
__global__ __launch_bounds__(kBlock) void k_direct_fast(BhwCordicCfg cfg, BhwWinCfg win, uint64_t n0, uint64_t count,
                                                         int32_t *__restrict__ out)
{
    __shared__ uint32_t lut_s[32];
    if (threadIdx.x < 32) lut_s[threadIdx.x] = (uint32_t)cfg.lut[threadIdx.x];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t pw = cfg.phi_width;
    const uint32_t mask = (pw >= 32) ? 0xFFFFFFFFu : ((1u << pw) - 1u);
    const uint32_t tmask = (1u << (pw - 2)) - 1u;
    const uint32_t n = (uint32_t)(n0 + i) & mask;
    int64_t acc = win.aa[0];
    for (uint32_t k = 1; k < win.n_terms; ++k) {
        const uint32_t theta = (k * n) & mask;
        const uint32_t u = (theta & tmask) >> cfg.z_shr;
        int64_t x = cfg.x0, y = cfg.x0;                                          // rotation 0 always adds (z0 >= 0)
        int32_t z = (int32_t)((u << cfg.z_shl) - lut_s[0]);
#pragma unroll
        for (int r = 1; r < NITER; ++r) rot_step(x, y, z, r, lut_s[r]);
        int32_t c, s;
        quadrant_map(theta >> (pw - 2), (int32_t)(x >> cfg.out_shr), (int32_t)(y >> cfg.out_shr), cfg.ones_neg, c, s);
        combine_term(acc, win.aa[k], c, k, cfg.dat_width, win.combine);
    }
    emit(win, out, i, combine_final(acc, cfg.dat_width, win.combine, win.n_terms));
}

// sin/cos sweep, fast form (cordic() alone in the mad-form rotation).
template <int NITER>
__global__ __launch_bounds__(kBlock) void k_sincos_fast(BhwCordicCfg cfg, uint64_t theta0, uint64_t count,
                                                         int32_t *__restrict__ d_sin, int32_t *__restrict__ d_cos)
{
    __shared__ uint32_t lut_s[32];
    if (threadIdx.x < 32) lut_s[threadIdx.x] = (uint32_t)cfg.lut[threadIdx.x];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t pw = cfg.phi_width;
    const uint32_t mask = (pw >= 32) ? 0xFFFFFFFFu : ((1u << pw) - 1u);
    const uint32_t theta = (uint32_t)(theta0 + i) & mask;
    const uint32_t u = (theta & ((1u << (pw - 2)) - 1u)) >> cfg.z_shr;
    int64_t x = cfg.x0, y = cfg.x0;
    int32_t z = (int32_t)((u << cfg.z_shl) - lut_s[0]);
#pragma unroll
    for (int r = 1; r < NITER; ++r) rot_step(x, y, z, r, lut_s[r]);
    int32_t c, s;
    quadrant_map(theta >> (pw - 2), (int32_t)(x >> cfg.out_shr), (int32_t)(y >> cfg.out_shr), cfg.ones_neg, c, s);
    if (d_sin) d_sin[i] = s;
    if (d_cos) d_cos[i] = c;
}

// ---------------------------------------------------------------------------------------
// Replicate: frames copies of one period (the stream is periodic: bh_win_7term.vhd:92-97,176-197).
// Store-only after one 16-byte read per lane; grid.y strides over frames.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_replicate16(const int4 *__restrict__ frame, uint64_t frame_vec, uint32_t frames,
                                                         int4 *__restrict__ out)
{
    const uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v >= frame_vec) return;
    const int4 d = frame[v];
    for (uint32_t f = blockIdx.y; f < frames; f += gridDim.y) out[(uint64_t)f * frame_vec + v] = d;
}

__global__ __launch_bounds__(kBlock) void k_replicate4(const int32_t *__restrict__ frame, uint64_t frame_len, uint32_t frames,
                                                        int32_t *__restrict__ out)
{
    const uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (v >= frame_len) return;
    const int32_t d = frame[v];
    for (uint32_t f = blockIdx.y; f < frames; f += gridDim.y) out[(uint64_t)f * frame_len + v] = d;
}

} // namespace

int bhwk_direct(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, uint64_t n0, uint64_t count, int32_t *d_out)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    if (c.dat_width + c.out_shr <= 34 && c.n_iter >= 7) {   // |x| < 2^33, quarter circle <= 2^32: the mad-form rotation applies
        const dim3 grid(grid_for(count)), block(kBlock);
        switch (c.n_iter) {
#define BHW_CASE(N) case N: BHW_LAUNCH(k_direct_fast<N>, grid, block, 0, st, c, w, n0, count, d_out); break;
            BHW_CASE(7) BHW_CASE(8) BHW_CASE(9) BHW_CASE(10) BHW_CASE(11) BHW_CASE(12) BHW_CASE(13) BHW_CASE(14)
            BHW_CASE(15) BHW_CASE(16) BHW_CASE(17) BHW_CASE(18) BHW_CASE(19) BHW_CASE(20) BHW_CASE(21) BHW_CASE(22)
            BHW_CASE(23) BHW_CASE(24) BHW_CASE(25) BHW_CASE(26) BHW_CASE(27) BHW_CASE(28) BHW_CASE(29) BHW_CASE(30)
            BHW_CASE(31) BHW_CASE(32)
#undef BHW_CASE
        default: return (int)hipErrorInvalidValue;
        }
        return finish(hipSuccess);
    }
    if (c.wide) BHW_LAUNCH(k_direct<int64_t>, dim3(grid_for(count)), dim3(kBlock), 0, st, c, w, n0, count, d_out);
    else        BHW_LAUNCH(k_direct<int32_t>, dim3(grid_for(count)), dim3(kBlock), 0, st, c, w, n0, count, d_out);
    return finish(hipSuccess);
}

int bhwk_sincos(const BhwLaunch &l, const BhwCordicCfg &c, uint64_t theta0, uint64_t count, int32_t *d_sin, int32_t *d_cos)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    // One whole period from 2^16 phases on: the shared-prefix chains of the table build (one chain per first-quadrant angle, a
    // 64-leaf group's common rotations run once) with the four quadrant images written straight out -- a quarter of the chains of
    // the per-phase kernel and about half of their rotations.
    if (c.dat_width + c.out_shr <= 34 && c.n_iter >= 7 && c.z_shr == 0 && c.phi_width >= 16 &&
        count == (1ull << c.phi_width)) {
        return bhwk_sincos_sweep(l, c, theta0, d_sin, d_cos);
    }
    if (c.dat_width + c.out_shr <= 34 && c.n_iter >= 7) {
        const dim3 grid(grid_for(count)), block(kBlock);
        switch (c.n_iter) {
#define BHW_CASE(N) case N: BHW_LAUNCH(k_sincos_fast<N>, grid, block, 0, st, c, theta0, count, d_sin, d_cos); break;
            BHW_CASE(7) BHW_CASE(8) BHW_CASE(9) BHW_CASE(10) BHW_CASE(11) BHW_CASE(12) BHW_CASE(13) BHW_CASE(14)
            BHW_CASE(15) BHW_CASE(16) BHW_CASE(17) BHW_CASE(18) BHW_CASE(19) BHW_CASE(20) BHW_CASE(21) BHW_CASE(22)
            BHW_CASE(23) BHW_CASE(24) BHW_CASE(25) BHW_CASE(26) BHW_CASE(27) BHW_CASE(28) BHW_CASE(29) BHW_CASE(30)
            BHW_CASE(31) BHW_CASE(32)
#undef BHW_CASE
        default: return (int)hipErrorInvalidValue;
        }
        return finish(hipSuccess);
    }
    if (c.wide) BHW_LAUNCH(k_sincos<int64_t>, dim3(grid_for(count)), dim3(kBlock), 0, st, c, theta0, count, d_sin, d_cos);
    else        BHW_LAUNCH(k_sincos<int32_t>, dim3(grid_for(count)), dim3(kBlock), 0, st, c, theta0, count, d_sin, d_cos);
    return finish(hipSuccess);
}

int bhwk_replicate(const BhwLaunch &l, const int32_t *d_frame, uint64_t frame_len, uint32_t frames, int32_t *d_out)
{
    if (!frames || !frame_len) return 0;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    const bool vec = (frame_len % 4 == 0) && (((uintptr_t)d_frame | (uintptr_t)d_out) % 16 == 0);
    const uint64_t items = vec ? frame_len / 4 : frame_len;
    const unsigned gx = grid_for(items);
    // the fill rate on MI355X peaks with >= 64K workgroups in flight (profiles/r01_ubench_gfx950.txt: 5.8 TB/s at 2K
    // blocks, 6.95 TB/s at 64K), so spread the frames over grid.y until there are about that many
    unsigned gy = (unsigned)((65536u + gx - 1) / gx);
    if (gy > frames) gy = frames;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    if (vec) BHW_LAUNCH(k_replicate16, dim3(gx, gy), dim3(kBlock), 0, st, (const int4 *)d_frame, items, frames, (int4 *)d_out);
    else     BHW_LAUNCH(k_replicate4, dim3(gx, gy), dim3(kBlock), 0, st, d_frame, items, frames, d_out);
    return finish(hipSuccess);
}

