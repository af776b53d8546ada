// bhw_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4).
//
// Hot path of the reference: phase accumulator -> CORDIC rotation chain (or Taylor LUT)
// -> weighted N-term cosine sum -> int32 coefficient.
// wave = 64 lanes; one lane per output coefficient in the direct kernels; the rescaled
// arctangent ROM is staged in LDS once per workgroup; stores are coalesced int32.
//
// Integer semantics follow SURVEY App. A; reference lines are cited at each step.
#include <hip/hip_runtime.h>
#include "bhw_internal.h"

namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------
// CORDIC rotation chain, first quadrant.  T = int32_t when the state fits 32 bits, else int64_t.
//   hls/windows/win_function.cpp:110-125 | cpp/cordic_sincos.cpp:49-63 | src/cordic_dds.vhd:197-213
// The typed-store wraps of the HLS/VHDL models (W+2 / W+PRECISION bits) can never fire:
// |x|,|y| <= 2^W * 1.0002 and |z| <= 2^W stay inside the state width (checked exhaustively by the
// oracle's wrap counter in tests/test_oracle.py), so they are not re-applied here.
// ---------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void cordic_q1(const T *__restrict__ lut, T x0, T z, int n_iter, T &xo, T &yo)
{
    T x = x0, y = 0;
#pragma unroll 4
    for (int k = 0; k < n_iter; ++k) {
        const T xs = x >> k, ys = y >> k;
        const bool neg = z < 0;
        const T l = lut[k];
        x = neg ? x + ys : x - ys;
        y = neg ? y - xs : y + xs;
        z = neg ? z + l : z - l;
    }
    xo = x;
    yo = y;
}

// quadrant map: hls/windows/win_function.cpp:135-150 | cpp/cordic_sincos.cpp:70-86 | src/cordic_dds.vhd:232-246
__device__ __forceinline__ void quadrant_map(uint32_t q, int32_t c, int32_t s, uint32_t ones_neg, int32_t &oc, int32_t &os)
{
    const int32_t nc = ones_neg ? ~c : -c;
    const int32_t ns = ones_neg ? ~s : -s;
    oc = (q == 0) ? c : (q == 1) ? ns : (q == 2) ? nc : s;
    os = (q == 0) ? s : (q == 1) ? c : (q == 2) ? ns : nc;
}

template <typename T>
__device__ __forceinline__ void cordic_full(const BhwCordicCfg &cfg, const T *lut, uint32_t theta, int32_t &oc, int32_t &os)
{
    const uint32_t pw = cfg.phi_width;
    const uint32_t q = theta >> (pw - 2);                          // two MSBs of the phase
    const uint32_t t = theta & ((1u << (pw - 2)) - 1u);
    const T z0 = (T)((T)(t >> cfg.z_shr) << cfg.z_shl);            // win_function.cpp:91-96 | cordic_sincos.cpp:31-36 | cordic_dds.vhd:159-166
    T x, y;
    cordic_q1<T>(lut, (T)cfg.x0, z0, (int)cfg.n_iter, x, y);
    const int32_t c = (int32_t)(x >> cfg.out_shr);                 // win_function.cpp:128-129 | cordic_dds.vhd:218-219
    const int32_t s = (int32_t)(y >> cfg.out_shr);
    quadrant_map(q, c, s, cfg.ones_neg, oc, os);
}

__device__ __forceinline__ int64_t wrap_bits(int64_t v, uint32_t bits)
{
    const uint32_t sh = 64u - bits;
    return (int64_t)((uint64_t)v << sh) >> sh;
}

// Accumulate one harmonic.  HLS rule: hls/windows/win_function.cpp:368-375;
// VHDL rule: src/bh_win_7term.vhd:353-402 (slice, round) -- SURVEY App. A.4/A.6.
__device__ __forceinline__ void combine_term(int64_t &acc, int32_t a, int32_t cosv, uint32_t k, uint32_t W, uint32_t combine)
{
    const int64_t prod = (int64_t)a * (int64_t)cosv;
    int64_t m = prod >> (W - 2);
    if (combine == BHW_COMBINE_VHDL) {
        const int64_t r = wrap_bits(m, W + 1);
        m = wrap_bits((r >> 1) + (r & 1), W);
    }
    acc += (k & 1u) ? -m : m;
}

// Final stage.  HLS: (win_t)(a0 - m1 + ...) win_function.cpp:375; VHDL: bh_win_7term.vhd:427-438, hamming_win.vhd:220-231.
__device__ __forceinline__ int32_t combine_final(int64_t acc, uint32_t W, uint32_t combine, uint32_t n_terms)
{
    if (combine == BHW_COMBINE_VHDL) {
        if (n_terms == 2) {
            const int64_t S = wrap_bits(acc, W + 1);
            acc = (S >> 1) + (S & 1);
        } else {
            const int64_t S = wrap_bits(acc, W + 2);
            acc = (S >> 2) + ((S >> 1) & 1);
        }
    }
    return (int32_t)wrap_bits(acc, W);
}

template <typename T>
__device__ __forceinline__ void stage_lut(const BhwCordicCfg &cfg, T *lut_s)
{
    if (threadIdx.x < 32) lut_s[threadIdx.x] = (T)cfg.lut[threadIdx.x];
    __syncthreads();
}

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
    out[i] = combine_final(acc, cfg.dat_width, win.combine, win.n_terms);
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
// Table strategy, pass 1: first-quadrant (c, s) for every distinct CORDIC input
//   u in [0, 2^(PW-2-z_shr)),  z0 = u << z_shl.
// Every harmonic of every coefficient evaluates this same function (the quadrant field is
// applied after the rotation), so the whole window needs only 2^(PW-2-z_shr) chains.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_table_build(BhwCordicCfg cfg, uint32_t entries, int2 *__restrict__ table)
{
    __shared__ T lut_s[32];
    stage_lut<T>(cfg, lut_s);
    const uint32_t u = blockIdx.x * kBlock + threadIdx.x;
    if (u >= entries) return;
    T x, y;
    cordic_q1<T>(lut_s, (T)cfg.x0, (T)((T)u << cfg.z_shl), (int)cfg.n_iter, x, y);
    table[u] = make_int2((int32_t)(x >> cfg.out_shr), (int32_t)(y >> cfg.out_shr));
}

// Table strategy, pass 2 (general form): one lane per coefficient, K-1 gathers.
__global__ __launch_bounds__(kBlock) void k_table_combine(BhwCordicCfg cfg, BhwWinCfg win, const int2 *__restrict__ table,
                                                           uint64_t n0, uint64_t count, int32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t pw = cfg.phi_width;
    const uint32_t mask = (1u << pw) - 1u;
    const uint32_t tmask = (1u << (pw - 2)) - 1u;
    const uint32_t n = (uint32_t)(n0 + i) & mask;
    int64_t acc = win.aa[0];
    for (uint32_t k = 1; k < win.n_terms; ++k) {
        const uint32_t theta = (k * n) & mask;
        const int2 cs = table[(theta & tmask) >> cfg.z_shr];
        int32_t c, s;
        quadrant_map(theta >> (pw - 2), cs.x, cs.y, cfg.ones_neg, c, s);
        combine_term(acc, win.aa[k], c, k, cfg.dat_width, win.combine);
    }
    out[i] = combine_final(acc, cfg.dat_width, win.combine, win.n_terms);
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

// ---------------------------------------------------------------------------------------
// Taylor feeder: quarter-wave ROM + 1st-order correction (src/taylor_sincos.vhd:141-253,
// src/tay1_order.vhd:112-146,501-502,585-616).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void taylor_full(const BhwTaylorCfg &t, uint32_t cnt, int32_t &oc, int32_t &os)
{
    const uint32_t pw = t.phi_width, W = t.dat_width, L = t.lut_size;
    const uint32_t q = cnt >> (pw - 2);
    uint32_t addr, f = 0;
    if (t.mode == 0)      addr = (cnt & ((1u << (pw - 2)) - 1u)) << (L - pw + 2);     // taylor_sincos.vhd:157-161
    else if (t.mode == 1) addr = cnt & ((1u << L) - 1u);                             // :164-167
    else {                                                                           // :190-191
        addr = (cnt >> (pw - L - 2)) & ((1u << L) - 1u);
        f = cnt & ((1u << (pw - L - 2)) - 1u);
    }
    const int2 sc = reinterpret_cast<const int2 *>(t.rom)[addr];
    int64_t S = sc.x, C = sc.y, c = C, s = S;
    if (t.mode == 2) {
        const int64_t m = ((int64_t)t.pi_word * (int64_t)f) & 0xFFFFFF;              // tay1_order.vhd:133-146
        const uint32_t X = t.xshift;                                                 // :112
        if (W < 19) {                                                                // :192-208,501-502
            // (C << X) - m*S needs up to 18 + 39 + 1 bits: fits int64
            c = wrap_bits(((C << X) - m * S) >> X, W);
            s = wrap_bits(((S << X) + m * C) >> X, W);
        } else {                                                                     // :585-616
            const int64_t dc = wrap_bits((m * S) >> X, W);
            const int64_t ds = wrap_bits((m * C) >> X, W);
            c = wrap_bits(C - dc, W);
            s = wrap_bits(S + ds, W);
            const int64_t sat = ((int64_t)1 << (W - 1)) - 1;
            if (c < 0) c = sat;
            if (s < 0) s = sat;
        }
    }
    const int64_t nc = wrap_bits(-c, W), ns = wrap_bits(-s, W);                      // taylor_sincos.vhd:240-253
    oc = (int32_t)((q == 0) ? c : (q == 1) ? ns : (q == 2) ? nc : s);
    os = (int32_t)((q == 0) ? s : (q == 1) ? c : (q == 2) ? ns : nc);
}

__global__ __launch_bounds__(kBlock) void k_taylor_window(BhwTaylorCfg t, BhwWinCfg win, uint64_t n0, uint64_t count,
                                                           int32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t mask = (1u << t.phi_width) - 1u;
    const uint32_t n = (uint32_t)(n0 + i) & mask;
    int64_t acc = win.aa[0];
    for (uint32_t k = 1; k < win.n_terms; ++k) {
        int32_t c, s;
        if (k == 1) {
            taylor_full(t, n, c, s);
        } else {
            // 2nd harmonic = a second generator with PHASE_WIDTH-1 on its own +1 counter: bh_win_3term.vhd:221-226
            BhwTaylorCfg t2 = t;
            t2.phi_width = t.phi_width - 1;
            const int d = (int)t2.phi_width - (int)t2.lut_size;
            t2.mode = d < 2 ? 0u : d == 2 ? 1u : 2u;
            t2.pi_word = t.pad[0];
            taylor_full(t2, n & (mask >> 1), c, s);
        }
        combine_term(acc, win.aa[k], c, k, t.dat_width, win.combine);
    }
    out[i] = combine_final(acc, t.dat_width, win.combine, win.n_terms);
}

__global__ __launch_bounds__(kBlock) void k_taylor_sincos(BhwTaylorCfg t, uint64_t theta0, uint64_t count,
                                                           int32_t *__restrict__ d_sin, int32_t *__restrict__ d_cos)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    int32_t c, s;
    taylor_full(t, (uint32_t)(theta0 + i) & ((1u << t.phi_width) - 1u), c, s);
    if (d_sin) d_sin[i] = s;
    if (d_cos) d_cos[i] = c;
}

inline unsigned grid_for(uint64_t count) { return (unsigned)((count + kBlock - 1) / kBlock); }

inline int finish(hipError_t e)
{
    if (e == hipSuccess) e = hipGetLastError();
    return (int)e;
}

} // namespace

#define BHW_SET_DEVICE(l)                                   \
    do {                                                    \
        hipError_t e__ = hipSetDevice((l).device);          \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

int bhwk_direct(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, uint64_t n0, uint64_t count, int32_t *d_out)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    if (c.wide) hipLaunchKernelGGL(k_direct<int64_t>, dim3(grid_for(count)), dim3(kBlock), 0, st, c, w, n0, count, d_out);
    else        hipLaunchKernelGGL(k_direct<int32_t>, dim3(grid_for(count)), dim3(kBlock), 0, st, c, w, n0, count, d_out);
    return finish(hipSuccess);
}

int bhwk_sincos(const BhwLaunch &l, const BhwCordicCfg &c, uint64_t theta0, uint64_t count, int32_t *d_sin, int32_t *d_cos)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    if (c.wide) hipLaunchKernelGGL(k_sincos<int64_t>, dim3(grid_for(count)), dim3(kBlock), 0, st, c, theta0, count, d_sin, d_cos);
    else        hipLaunchKernelGGL(k_sincos<int32_t>, dim3(grid_for(count)), dim3(kBlock), 0, st, c, theta0, count, d_sin, d_cos);
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
    // enough workgroups to fill 256 CUs several times over, without one block per (chunk, frame)
    unsigned gy = (unsigned)((4096 + gx - 1) / gx);
    if (gy > frames) gy = frames;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    if (vec) hipLaunchKernelGGL(k_replicate16, dim3(gx, gy), dim3(kBlock), 0, st, (const int4 *)d_frame, items, frames, (int4 *)d_out);
    else     hipLaunchKernelGGL(k_replicate4, dim3(gx, gy), dim3(kBlock), 0, st, d_frame, items, frames, d_out);
    return finish(hipSuccess);
}

int bhwk_table_build(const BhwLaunch &l, const BhwCordicCfg &c, int32_t *d_table)
{
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    const uint32_t entries = 1u << (c.phi_width - 2 - c.z_shr);
    if (c.wide) hipLaunchKernelGGL(k_table_build<int64_t>, dim3(grid_for(entries)), dim3(kBlock), 0, st, c, entries, (int2 *)d_table);
    else        hipLaunchKernelGGL(k_table_build<int32_t>, dim3(grid_for(entries)), dim3(kBlock), 0, st, c, entries, (int2 *)d_table);
    return finish(hipSuccess);
}

int bhwk_table_combine(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_table,
                       uint64_t n0, uint64_t count, int32_t *d_out)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    hipLaunchKernelGGL(k_table_combine, dim3(grid_for(count)), dim3(kBlock), 0, st, c, w, (const int2 *)d_table, n0, count, d_out);
    return finish(hipSuccess);
}

int bhwk_taylor_window(const BhwLaunch &l, const BhwTaylorCfg &t, const BhwWinCfg &w, uint64_t n0, uint64_t count, int32_t *d_out)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    hipLaunchKernelGGL(k_taylor_window, dim3(grid_for(count)), dim3(kBlock), 0, (hipStream_t)l.stream, t, w, n0, count, d_out);
    return finish(hipSuccess);
}

int bhwk_taylor_sincos(const BhwLaunch &l, const BhwTaylorCfg &t, uint64_t theta0, uint64_t count, int32_t *d_sin, int32_t *d_cos)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    hipLaunchKernelGGL(k_taylor_sincos, dim3(grid_for(count)), dim3(kBlock), 0, (hipStream_t)l.stream, t, theta0, count, d_sin, d_cos);
    return finish(hipSuccess);
}
