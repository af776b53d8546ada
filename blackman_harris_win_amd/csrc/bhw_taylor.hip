// bhw_taylor.hip -- Taylor feeder (quarter-wave ROM + first-order correction)
//
// Part of the hand-written HIP kernels for gfx950 (MI355X, CDNA4) behind include/bhw.h.  Hot path of the reference: phase
// accumulator -> CORDIC rotation chain (or Taylor LUT) -> weighted N-term cosine sum -> int32 coefficient (SURVEY section 8a
// rows a1-a11).  Integer semantics follow SURVEY App. A; reference lines are cited at each step.
#include "bhw_device.h"

namespace {

// ---------------------------------------------------------------------------------------
// Taylor feeder: quarter-wave ROM + 1st-order correction (src/taylor_sincos.vhd:141-253,
// src/tay1_order.vhd:112-146,501-502,585-616).
// ---------------------------------------------------------------------------------------
// first-quadrant part: (sin', cos') of the ROM entry + 1st-order correction for the phase bits below the quadrant field
__device__ __forceinline__ void taylor_q1(const BhwTaylorCfg &t, const int2 *rom, uint32_t cnt, int64_t &s, int64_t &c)
{
    const uint32_t pw = t.phi_width, W = t.dat_width, L = t.lut_size;
    uint32_t addr, f = 0;
    if (t.mode == 0)      addr = (cnt & ((1u << (pw - 2)) - 1u)) << (L - pw + 2);     // taylor_sincos.vhd:157-161
    else if (t.mode == 1) addr = cnt & ((1u << L) - 1u);                             // :164-167
    else {                                                                           // :190-191
        addr = (cnt >> (pw - L - 2)) & ((1u << L) - 1u);
        f = cnt & ((1u << (pw - L - 2)) - 1u);
    }
    const int2 sc = rom[addr];
    const int64_t S = sc.x, C = sc.y;
    c = C;
    s = S;
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
}

__device__ __forceinline__ void taylor_full(const BhwTaylorCfg &t, uint32_t cnt, int32_t &oc, int32_t &os)
{
    const uint32_t q = cnt >> (t.phi_width - 2);
    int64_t s, c;
    taylor_q1(t, reinterpret_cast<const int2 *>(t.rom), cnt, s, c);
    const int64_t nc = wrap_bits(-c, t.dat_width), ns = wrap_bits(-s, t.dat_width);  // taylor_sincos.vhd:240-253
    oc = (int32_t)((q == 0) ? c : (q == 1) ? ns : (q == 2) ? nc : s);
    os = (int32_t)((q == 0) ? s : (q == 1) ? c : (q == 2) ? ns : nc);
}

// Generator of harmonic k = m * 2^v: taylor_sincos at PHASE_WIDTH - v (bh_win_3term.vhd:221-226 for k = 2; continued to
// k = 3..6 by BHW_SIN_TAYLOR_ALL, include/bhw.h).  pad[v-1] holds that generator's pi word.
__device__ __forceinline__ BhwTaylorCfg taylor_gen(const BhwTaylorCfg &t, uint32_t v)
{
    BhwTaylorCfg g = t;
    g.phi_width = t.phi_width - v;
    const int d = (int)g.phi_width - (int)g.lut_size;
    g.mode = d < 2 ? 0u : d == 2 ? 1u : 2u;
    g.pi_word = v == 0 ? t.pi_word : t.pad[v - 1];
    return g;
}

// ---- 32-bit forms for dat_width <= 16 (every product a*v and every sum fits int32; same results; wrap32 is defined above) ----
__device__ __forceinline__ void taylor_q1_narrow(const BhwTaylorCfg &t, const int2 *rom, uint32_t cnt, int32_t &s, int32_t &c)
{
    const uint32_t pw = t.phi_width, W = t.dat_width, L = t.lut_size;
    uint32_t addr, f = 0;
    if (t.mode == 0)      addr = (cnt & ((1u << (pw - 2)) - 1u)) << (L - pw + 2);
    else if (t.mode == 1) addr = cnt & ((1u << L) - 1u);
    else {
        addr = (cnt >> (pw - L - 2)) & ((1u << L) - 1u);
        f = cnt & ((1u << (pw - L - 2)) - 1u);
    }
    const int2 sc = rom[addr];
    s = sc.x;
    c = sc.y;
    if (t.mode == 2) {                                       // W < 19 path: tay1_order.vhd:192-208,501-502
        const int32_t m = (int32_t)((t.pi_word * f) & 0xFFFFFFu);
        const uint32_t X = t.xshift;
        // ((C << X) - m*S) >> X == C + ((-(m*S)) >> X) because C << X is a multiple of 2^X.  m < pi * 2^18 < 2^20, so for
        // X <= 32 (LUT_SIZE <= 13) the floor shifts are one v_mul_hi_i32 each: (m*v) >> X == mulhi(m << (32-X), v).
        int32_t dc, ds;
        if (X <= 32u && L >= 2u) {
            const int32_t ms = (int32_t)((uint32_t)m << (32u - X));
            dc = __mulhi(ms, -sc.x);
            ds = __mulhi(ms, sc.y);
        } else {
            dc = (int32_t)((-((int64_t)m * sc.x)) >> X);
            ds = (int32_t)(((int64_t)m * sc.y) >> X);
        }
        c = wrap32(sc.y + dc, W);
        s = wrap32(sc.x + ds, W);
    }
}

__device__ __forceinline__ int32_t narrow_term(int32_t a, int32_t v, uint32_t W, uint32_t combine)
{
    int32_t m = (a * v) >> (W - 2);                          // |a|,|v| < 2^15: exact in int32
    if (combine == BHW_COMBINE_VHDL) {
        const int32_t r = wrap32(m, W + 1);
        m = wrap32((r >> 1) + (r & 1), W);
    }
    return m;
}

__device__ __forceinline__ int32_t narrow_final(int32_t acc, uint32_t W, uint32_t combine, uint32_t n_terms)
{
    if (combine == BHW_COMBINE_VHDL) {
        if (n_terms == 2) {
            const int32_t S = wrap32(acc, W + 1);
            acc = (S >> 1) + (S & 1);
        } else {
            const int32_t S = wrap32(acc, W + 2);
            acc = (S >> 2) + ((S >> 1) & 1);
        }
    }
    return wrap32(acc, W);
}

// ---- 32-bit forms for dat_width >= 19 (the wide rounding variant, tay1_order.vhd:585-616) with 24 <= 19+L <= 32 ----
// m < 2^24 and the ROM entries are in [0, 2^(W-1)), so (m*v) >> X is one v_mul_hi_u32 of (m << (32-X)) and v, already
// inside W bits; C - dc cannot leave the W-bit range, S + ds can (and then saturates, as does a negative c).
__device__ __forceinline__ void taylor_q1_w32(const BhwTaylorCfg &t, const int2 *rom, uint32_t cnt, int32_t &s, int32_t &c)
{
    const uint32_t pw = t.phi_width, W = t.dat_width, L = t.lut_size;
    const uint32_t addr = (cnt >> (pw - L - 2)) & ((1u << L) - 1u);             // taylor_sincos.vhd:190-191 (mode 2 only)
    const uint32_t f = cnt & ((1u << (pw - L - 2)) - 1u);
    const int2 sc = rom[addr];
    const uint32_t m = (t.pi_word * f) & 0xFFFFFFu;                             // tay1_order.vhd:133-146
    const uint32_t ms = m << (32u - t.xshift);
    const int32_t dc = (int32_t)__umulhi(ms, (uint32_t)sc.x);
    const int32_t ds = (int32_t)__umulhi(ms, (uint32_t)sc.y);
    const int32_t sat = (int32_t)((1u << (W - 1)) - 1u);
    c = sc.y - dc;                                                              // :595-596, in range without a wrap
    s = wrap32((int32_t)((uint32_t)sc.x + (uint32_t)ds), W);
    if (c < 0) c = sat;                                                         // :602-616
    if (s < 0) s = sat;
}

// Whole-period Taylor window, quadrant fold: lane r in [0, N/4) owns n = r + j*N/4.  The first generator's quadrant
// is then simply j; the 3-term window's second generator (PHASE_WIDTH-1, bh_win_3term.vhd:221-226) sees phase
// n mod N/2 = r + (j & 1) * N/4, i.e. quadrant (r / (N/8)) + 2*(j & 1) of its own period.  One ROM read and one
// Taylor correction per generator serve four coefficients; the quarter-wave ROM is staged in LDS.
constexpr int kTaylorRomLds = 4096;     // entries (32 KiB); larger ROMs are read from global memory

// Each thread takes four consecutive r so that every image is written with one 16-byte store per lane (the dword-per-
// lane store rate on MI355X is ~4.5 TB/s, the 16-byte rate ~6.9 TB/s: profiles/r01_ubench_gfx950.txt).
// FAST: both generators take the 1st-order-correction path (PHASE_WIDTH - LUT_SIZE > 3) and the ROM fits LDS -- the usual case.
// ARITH: 0 generic 64-bit, 1 int32 for dat_width <= 16 (weights inside 16 bits), 2 int32 for dat_width >= 19 with 24 <= 19+L <= 32
template <int ARITH, uint32_t COMBINE, uint32_t NTERMS, bool FAST>
__global__ __launch_bounds__(kBlock) void k_taylor_window_fold(BhwTaylorCfg t, BhwWinCfg win, int32_t *__restrict__ out)
{
    __shared__ int2 rom_s[kTaylorRomLds];
    const uint32_t depth = 1u << t.lut_size;
    const bool in_lds = FAST || depth <= (uint32_t)kTaylorRomLds;
    if constexpr (FAST) { t.mode = 2u; }
    if (in_lds) {
        for (uint32_t i = threadIdx.x; i < depth; i += kBlock) rom_s[i] = reinterpret_cast<const int2 *>(t.rom)[i];
        __syncthreads();
    }
    const int2 *rom_g = reinterpret_cast<const int2 *>(t.rom);
    const uint32_t E = 1u << (t.phi_width - 2);                  // a multiple of 4 (PW >= 5 is required by the caller)
    const uint32_t W = t.dat_width;
    constexpr bool NARROW = (ARITH == 1);
    using val_t = typename std::conditional<ARITH == 2, Sum32, typename std::conditional<ARITH == 1, int32_t, int64_t>::type>::type;
    using trig_t = typename std::conditional<ARITH == 0, int64_t, int32_t>::type;
    auto add_term = [&](val_t &a, int32_t weight, int32_t v, uint32_t k) {
        if constexpr (ARITH == 2) {
            w32_term<COMBINE>(a, weight, v, k, W);
        } else if constexpr (NARROW) {
            const int32_t m = narrow_term(weight, v, W, COMBINE);
            a += (k & 1u) ? -m : m;
        } else {
            combine_term(a, weight, v, k, W, COMBINE);
        }
    };
    auto neg = [&](trig_t v) -> int32_t {
        if constexpr (ARITH != 0) return wrap32(-(int32_t)v, W);
        else return (int32_t)wrap_bits(-(int64_t)v, W);
    };
    auto zero = [&]() -> val_t { if constexpr (ARITH == 2) return Sum32{0, 0}; else return (val_t)0; };
    auto first = [&]() -> val_t {                                 // a_0
        if constexpr (ARITH == 2) {
            if constexpr (COMBINE == BHW_COMBINE_HLS) return Sum32{win.aa[0], 0};
            else return sum32_first(win.aa[0]);
        } else return (val_t)win.aa[0];
    };
    // generators by valuation v = 0, 1, 2 of the harmonic number (k = 1,3,5 | 2,6 | 4)
    BhwTaylorCfg tg[3] = {taylor_gen(t, 0), taylor_gen(t, NTERMS > 2 ? 1 : 0), taylor_gen(t, NTERMS > 4 ? 2 : 0)};
    if constexpr (FAST) { tg[0].mode = tg[1].mode = tg[2].mode = 2u; }
    // harmonic K of lane r: phase (m*r) mod 2^(PW-v) in generator v; image j sits K*j quadrants further on
    auto harmonic = [&](auto kc, uint32_t r, val_t (&acc)[4]) {
        constexpr uint32_t K = decltype(kc)::value;
        constexpr uint32_t V = (K & 1u) ? 0u : (K & 2u) ? 1u : 2u, M = K >> V;
        const BhwTaylorCfg &g = tg[V];
        const uint32_t cnt = (M * r) & ((4u * E >> V) - 1u);
        const uint32_t q0 = (K == 1u) ? 0u : cnt >> (g.phi_width - 2u);
        trig_t s, c;
        if constexpr (ARITH == 2)      { taylor_q1_w32(g, rom_s, cnt, s, c); }   // FAST only: ROM in LDS, correction path
        else if constexpr (NARROW)     { if (in_lds) taylor_q1_narrow(g, rom_s, cnt, s, c); else taylor_q1_narrow(g, rom_g, cnt, s, c); }
        else                           { if (in_lds) taylor_q1(g, rom_s, cnt, s, c);        else taylor_q1(g, rom_g, cnt, s, c); }
        const int32_t p0 = (int32_t)c, p1 = neg(s), p2 = neg(c), p3 = (int32_t)s;   // quadrant 0..3: taylor_sincos.vhd:240-253
        if constexpr (K == 1u) {
            add_term(acc[0], win.aa[1], p0, 1);
            add_term(acc[1], win.aa[1], p1, 1);
            add_term(acc[2], win.aa[1], p2, 1);
            add_term(acc[3], win.aa[1], p3, 1);
        } else {
            const bool b0 = q0 & 1u, b1 = q0 & 2u;
            const int32_t r0 = b0 ? p1 : p0, r1 = b0 ? p2 : p1, r2 = b0 ? p3 : p2, r3 = b0 ? p0 : p3;
            const int32_t sv[4] = {b1 ? r2 : r0, b1 ? r3 : r1, b1 ? r0 : r2, b1 ? r1 : r3};
            if constexpr ((K & 3u) == 0u) {                      // all four images in one quadrant
                val_t one = zero();
                add_term(one, win.aa[K], sv[0], K);
                acc[0] += one; acc[1] += one; acc[2] += one; acc[3] += one;
            } else if constexpr ((K & 1u) == 0u) {               // images alternate between two quadrants
                val_t even = zero(), odd = zero();
                add_term(even, win.aa[K], sv[0], K);
                add_term(odd, win.aa[K], sv[2], K);
                acc[0] += even; acc[1] += odd; acc[2] += even; acc[3] += odd;
            } else {
                add_term(acc[0], win.aa[K], sv[0], K);
                add_term(acc[1], win.aa[K], sv[K & 3u], K);
                add_term(acc[2], win.aa[K], sv[(2u * K) & 3u], K);
                add_term(acc[3], win.aa[K], sv[(3u * K) & 3u], K);
            }
        }
    };
    // grid-stride over 1024-coefficient-wide chunks of r: the ROM staging above is paid once per workgroup
    for (uint32_t r0 = (blockIdx.x * kBlock + threadIdx.x) * 4u; r0 < E; r0 += gridDim.x * kBlock * 4u) {
    int32_t res[4][4];                                           // [image j][i]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t r = r0 + (uint32_t)i;
        val_t acc[4] = {first(), first(), first(), first()};
        harmonic(std::integral_constant<uint32_t, 1>{}, r, acc);
        if constexpr (NTERMS > 2) harmonic(std::integral_constant<uint32_t, 2>{}, r, acc);
        if constexpr (NTERMS > 3) harmonic(std::integral_constant<uint32_t, 3>{}, r, acc);
        if constexpr (NTERMS > 4) harmonic(std::integral_constant<uint32_t, 4>{}, r, acc);
        if constexpr (NTERMS > 5) {
            harmonic(std::integral_constant<uint32_t, 5>{}, r, acc);
            harmonic(std::integral_constant<uint32_t, 6>{}, r, acc);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (ARITH == 2) res[j][i] = w32_final<COMBINE>(acc[j], W, NTERMS);
            else if constexpr (NARROW) res[j][i] = narrow_final(acc[j], W, COMBINE, NTERMS);
            else res[j][i] = combine_final(acc[j], W, COMBINE, NTERMS);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint64_t idx = (uint64_t)r0 + (uint64_t)j * E;
        if (win.apply_x || (((uintptr_t)out) & 15u)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) emit(win, out, idx + i, res[j][i]);
        } else {
            *reinterpret_cast<int4 *>(out + idx) = make_int4(res[j][0], res[j][1], res[j][2], res[j][3]);
        }
    }
    }
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
        // harmonic k = m * 2^v: generator of PHASE_WIDTH - v at phase (m*n) mod 2^(PW-v)  (k = 2: bh_win_3term.vhd:221-226)
        const uint32_t v = (uint32_t)__builtin_ctz(k);
        taylor_full(taylor_gen(t, v), ((k * n) & mask) >> v, c, s);
        combine_term(acc, win.aa[k], c, k, t.dat_width, win.combine);
    }
    emit(win, out, i, combine_final(acc, t.dat_width, win.combine, win.n_terms));
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

} // namespace

int bhwk_taylor_window(const BhwLaunch &l, const BhwTaylorCfg &t, const BhwWinCfg &w, uint64_t n0, uint64_t count, int32_t *d_out)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    BHW_LAUNCH(k_taylor_window, dim3(grid_for(count)), dim3(kBlock), 0, (hipStream_t)l.stream, t, w, n0, count, d_out);
    return finish(hipSuccess);
}

int bhwk_taylor_window_fold(const BhwLaunch &l, const BhwTaylorCfg &t, const BhwWinCfg &w, int32_t *d_out)
{
    BHW_SET_DEVICE(l);
    const uint32_t E = 1u << (t.phi_width - 2);
    // dat_width <= 16 with weights inside the W-bit range: every product and sum fits int32
    bool narrow = t.dat_width <= 16;
    for (uint32_t k = 0; k < w.n_terms; ++k) narrow = narrow && w.aa[k] < (1 << 15) && w.aa[k] >= -(1 << 15);
    unsigned blocks = grid_for(E / 4);
    // grid-stride above 4 096 workgroups.  (A cap of 8 192 reads 3 % better per isolated call -- 0.0465 -> 0.0450 ms -- and 3 % worse
    // back to back, which is what the kernel's own duration shows: 42.0 -> 43.3 us under rocprofv3.  4 096 stays.)
    if (blocks > 4096u) blocks = 4096u;
    const dim3 grid(blocks);
    hipStream_t st = (hipStream_t)l.stream;
#define BHW_TAYLOR_FOLD(ARITH, COMBINE, NT) BHW_LAUNCH((k_taylor_window_fold<ARITH, COMBINE, NT, true>), grid, dim3(kBlock), 0, st, t, w, d_out)
    const bool vhdl = w.combine == BHW_COMBINE_VHDL;
    // every generator in use (PHASE_WIDTH - v, v <= vmax) on the 1st-order-correction path, ROM in LDS
    const int vmax = w.n_terms > 4 ? 2 : w.n_terms > 2 ? 1 : 0;
    const bool fast = (1u << t.lut_size) <= (uint32_t)kTaylorRomLds && (int)t.phi_width - vmax - (int)t.lut_size > 2;
    if (!fast) {
        // a quarter-wave ROM beyond LDS (LUT_SIZE > 12) or a generator without the correction stage (PHASE_WIDTH - LUT_SIZE <= 3):
        // the general one-lane-per-coefficient kernel over the period (the fold kernel is instantiated in its usual form only)
        BHW_LAUNCH(k_taylor_window, dim3(grid_for(4ull * E)), dim3(kBlock), 0, st, t, w, (uint64_t)0, (uint64_t)(4ull * E), d_out);
        return finish(hipSuccess);
    }
    // 32-bit arithmetic: int16-sized operands (W <= 16), or the wide rounding variant with its shift inside one mul_hi
    const int arith = narrow ? 1 : (fast && t.dat_width >= 19 && t.xshift >= 24 && t.xshift <= 32) ? 2 : 0;
#define BHW_TAYLOR_FOLD_NT(NT)                                                                                      \
    do {                                                                                                            \
        if (arith == 1)      { if (vhdl) BHW_TAYLOR_FOLD(1, BHW_COMBINE_VHDL, NT); else BHW_TAYLOR_FOLD(1, BHW_COMBINE_HLS, NT); } \
        else if (arith == 2) { if (vhdl) BHW_TAYLOR_FOLD(2, BHW_COMBINE_VHDL, NT); else BHW_TAYLOR_FOLD(2, BHW_COMBINE_HLS, NT); } \
        else                 { if (vhdl) BHW_TAYLOR_FOLD(0, BHW_COMBINE_VHDL, NT); else BHW_TAYLOR_FOLD(0, BHW_COMBINE_HLS, NT); } \
    } while (0)
    switch (w.n_terms) {
    case 2: BHW_TAYLOR_FOLD_NT(2); break;
    case 3: BHW_TAYLOR_FOLD_NT(3); break;
    case 4: BHW_TAYLOR_FOLD_NT(4); break;
    case 5: BHW_TAYLOR_FOLD_NT(5); break;
    case 7: BHW_TAYLOR_FOLD_NT(7); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef BHW_TAYLOR_FOLD_NT
#undef BHW_TAYLOR_FOLD
    return finish(hipSuccess);
}

int bhwk_taylor_sincos(const BhwLaunch &l, const BhwTaylorCfg &t, uint64_t theta0, uint64_t count, int32_t *d_sin, int32_t *d_cos)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    BHW_LAUNCH(k_taylor_sincos, dim3(grid_for(count)), dim3(kBlock), 0, (hipStream_t)l.stream, t, theta0, count, d_sin, d_cos);
    return finish(hipSuccess);
}

