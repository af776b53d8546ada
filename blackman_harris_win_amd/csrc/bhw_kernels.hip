// bhw_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4).
//
// Hot path of the reference: phase accumulator -> CORDIC rotation chain (or Taylor LUT)
// -> weighted N-term cosine sum -> int32 coefficient.
// wave = 64 lanes; one lane per output coefficient in the direct kernels; the rescaled
// arctangent ROM is staged in LDS once per workgroup; stores are coalesced int32.
//
// Integer semantics follow SURVEY App. A; reference lines are cited at each step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <type_traits>
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

// Table layouts.  Natural: entry u at index u.  Split: the table is stored as three runs
//   [ u % 4 == 0 | u % 4 == 2 | u odd ]  so that the even harmonics (t = 2r, 4r, 6r only ever touch even /
// multiple-of-4 entries) read dense runs instead of every 2nd / 4th entry of a line.  (Four runs by u % 4 -- a plain
// rotate of the index -- cost 5 % more: a wave of consecutive odd-harmonic lanes then reads four 128-byte pieces
// instead of one 256-byte piece and two of 128, profiles/r01_ab_inproc.txt.)
// KCLASS states what the caller knows about u at compile time (from the harmonic number): 0 nothing, 2 u is even,
// 4 u is a multiple of 4.  Branch-free on purpose: as a ?: chain the compiler emits exec-mask branches per gather.
// SPLIT: -1 decided at run time by `split`, 0 / 1 known at compile time (no branch around the gather: the compiler can then
// batch the loads of a harmonic instead of waiting on each one).
template <int KCLASS = 0, int SPLIT = -1>
__device__ __forceinline__ uint32_t tab_index(uint32_t u, uint32_t log2_entries, uint32_t split)
{
    if (SPLIT == 0 || (SPLIT < 0 && !split)) return u;
    if constexpr (KCLASS == 4) return u >> 2;
    const uint32_t e = 1u << log2_entries;
    const uint32_t mid = (e >> 2) & (0u - ((u >> 1) & 1u));            // u % 4 == 2 -> second run
    if constexpr (KCLASS == 2) return (u >> 2) + mid;
    const uint32_t odd = u & 1u;
    return (u >> (2u - odd)) + (odd ? (e >> 1) : mid);
}

// Packed tables (z_shr == 0 only).  (c, s)(t) is smooth in t, so whole-period tile calls store less than 8 bytes per entry.
// Both formats are exact by construction and keep the layout (index) of the plain table; cfg.tab_dlog selects:
//   6      "delta16": inside an aligned block of 64 entries (c, s) moves by at most 63 * 2 pi * 2^(W-2-PW) (+ the CORDIC's
//          rounding noise of a few LSB), which fits int16 whenever W - PW <= 8 (bhwk_packed_ok).  One dword per entry = the
//          two 16-bit differences to the block's first entry; the first entries are int2 records at cfg.tab_coarse
//          (8 bytes per 64 entries).  Two adds to unpack.
//   7..9   "residual": between two exact records 2^d entries apart the curve deviates from the straight line through them by
//          the CORDIC's own rounding noise (a few LSB: <= 32 rotations of < 1 LSB each, in x and in the residual angle) plus
//          < 1 LSB of curvature (d is chosen for that, bhwk_resid_dlog).  Two bytes per entry = that deviation for c and s;
//          int4 records {c, s, dc, ds} at cfg.tab_coarse (16 bytes per 2^d entries).  Build and combine evaluate the same
//          integer predictor  rec.c + ((rec.dc * (t mod 2^d)) >> d),  so the reconstruction is exact as long as the deviation
//          fits int8 (tests/test_oracle.py::test_residual_format_margin measures <= 40 over every model and width).
//   23..25 "nibble" (16 + d): the residual format with the two deviations in 4-bit fields, one byte per entry.  The deviations
//          of the 32-bit HLS model stay within -5 .. 6 over the whole 2^24-entry table of a 2^26-point window (measured with the
//          oracle); like the other packed formats it is used only after the build kernel has checked every entry of the
//          configuration (a model whose noise is wider -- the cpp model reaches 10 -- falls back to the byte fields).
// The combine pass is bound by table + output traffic as much as by arithmetic, and these cut the table's share to 1/2 .. 1/8.
constexpr uint32_t kPackLog = 6;
constexpr uint32_t kNibbleFlag = 16;            // cfg.tab_dlog = kNibbleFlag + d
__host__ __device__ constexpr uint32_t fmt_cell_log(uint32_t tab_dlog) { return tab_dlog & (kNibbleFlag - 1u); }
__host__ __device__ constexpr int fmt_of(uint32_t tab_dlog) { return tab_dlog == 0 ? 0 : tab_dlog == kPackLog ? 1 : tab_dlog >= kNibbleFlag ? 3 : 2; }

__device__ __forceinline__ int2 tab_predict(const int4 rec, uint32_t f, uint32_t d)
{
    return make_int2(rec.x + (__mul24(rec.z, (int32_t)f) >> d), rec.y + (__mul24(rec.w, (int32_t)f) >> d));   // |dc|, |ds| < 2^17, f < 2^9
}

// v representable as a two's-complement field of `bits` bits
__device__ __forceinline__ bool fits_bits(int32_t v, uint32_t bits)
{
    return (uint32_t)(v + (1 << (bits - 1))) < (1u << bits);
}

// Table reads address as (scalar base) + (32-bit byte offset per lane): the tables are below 4 GiB, and with the offset held
// in 32 bits the compiler emits the saddr form  global_load v, v_off, s[base:base+1]  instead of a 64-bit add per address
// (the tile kernel issues 54 such loads per thread).
template <typename T>
__device__ __forceinline__ T ld_off(const void *__restrict__ base, uint32_t byte_off)
{
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

// Entry `u` of the table, stored at byte offset `boff`.  FMT: -1 format read from cfg.tab_dlog at run time, 0 plain int2 entries,
// 1 delta16, 2 residual.
// Residual format: cell size and in-cell mask as the caller holds them.  On gfx950 a VOP2 instruction with an SGPR operand
// issues in ~4.1 cycles against ~2.5 with VGPR / inline-constant operands (profiles/r02_ubench_gfx950.txt), so the tile
// kernel, which shifts and masks by these per gather, keeps them in VGPRs; everyone else passes the scalars.
#ifndef BHW_TILE_DBG
#define BHW_TILE_DBG 0      // timing experiments only (tools/ab_inproc.py with AB_NOCHECK): 4 no stores
#endif
struct ResidK {
    uint32_t d;       // log2 of the cell size
    uint32_t fmask;   // 2^d - 1
};

template <int FMT = -1>
__device__ __forceinline__ int2 tab_fetch(const BhwCordicCfg &cfg, const void *__restrict__ table, uint32_t u, uint32_t idx)
{
    if (FMT == 0 || (FMT < 0 && cfg.tab_dlog == 0)) return ld_off<int2>(table, idx << 3);
    if (FMT == 1 || (FMT < 0 && cfg.tab_dlog == kPackLog)) {
        const uint32_t e = ld_off<uint32_t>(table, idx << 2);
        const int2 base = ld_off<int2>(cfg.tab_coarse, (u >> kPackLog) << 3);
        return make_int2(base.x + (int32_t)(int16_t)(e & 0xFFFFu), base.y + ((int32_t)e >> 16));
    }
    const uint32_t d = fmt_cell_log(cfg.tab_dlog);
    const int2 p = tab_predict(ld_off<int4>(cfg.tab_coarse, (u >> d) << 4), u & ((1u << d) - 1u), d);
    if (FMT == 3 || (FMT < 0 && cfg.tab_dlog >= kNibbleFlag)) {
        const uint32_t e = ld_off<uint8_t>(table, idx);
        return make_int2(p.x + ((int32_t)(e << 28) >> 28), p.y + ((int32_t)(e << 24) >> 28));
    }
    const uint32_t e = ld_off<uint16_t>(table, idx << 1);
    return make_int2(p.x + (int32_t)(int8_t)(e & 0xFFu), p.y + (int32_t)(int8_t)(e >> 8));
}

// Residual / nibble entry of the tile kernel from the UNMASKED angle theta = K * (r + g * E/2) (u = theta mod E), its residual
// word e already loaded.  LDS: the record comes from the copy the wave staged in shared memory; `bias` is the byte address of
// the record of "cell 0" as that window sees it and absorbs the whole turns of theta (no run of such a tile wraps), so the cell
// index needs no mask: shift, shift-add, ds_read.  Otherwise the record is read from the table's record array.
template <int FMT, bool LDS>
__device__ __forceinline__ int2 resid_value(const BhwCordicCfg &cfg, uint32_t theta, uint32_t emask, const ResidK &rk, const char *lrec, uint32_t bias,
                                            uint32_t e)
{
    static_assert(FMT == 2 || FMT == 3, "residual / nibble entries");
    int4 rec;
    if constexpr (LDS) rec = *reinterpret_cast<const int4 *>(lrec + (((theta >> rk.d) << 4) + bias));
    else rec = ld_off<int4>(cfg.tab_coarse, ((theta & emask) >> rk.d) << 4);
    // (the predictor as two shifts + two v_mul_hi_i32 on doubled slopes -- one instruction fewer per gather -- measured slower:
    // 0.1358 -> 0.1367 ms, profiles/r02_ab_tile_memory_path.txt)
    const int2 p = tab_predict(rec, theta & rk.fmask, rk.d);
    if constexpr (FMT == 3) return make_int2(p.x + ((int32_t)(e << 28) >> 28), p.y + ((int32_t)(e << 24) >> 28));
    else return make_int2(p.x + (int32_t)(int8_t)(e & 0xFFu), p.y + (int32_t)(int8_t)(e >> 8));
}

template <int KCLASS = 0, int FMT = -1, int SPLIT = -1>
__device__ __forceinline__ int2 tab_load(const BhwCordicCfg &cfg, const void *__restrict__ table, uint32_t u, uint32_t log2_entries)
{
    return tab_fetch<FMT>(cfg, table, u, tab_index<KCLASS, SPLIT>(u, log2_entries, cfg.tab_split));
}

// Split layout, entries of one residue class (the odd harmonics of a lane: u = K*r has r's class for every odd K, and so has
// the half-period image u + E/2):
//   index(u) = (u >> s) | base,   s = 1 (u odd) or 2 (u even),  base = E/2 (u odd), E/4 (u % 4 == 2), 0 (u % 4 == 0)
// -- the regions are sized so that base never overlaps the shifted index.  split_class() folds the class once per run into one
// word the gathers of the run's odd harmonics share:
//   plain (8-byte) and delta16 (4-byte) entries:  cls = (base << LB) | (LB - s),  byte offset = ((u << cls) | cls) & ~(2^LB - 1)
//     (LB = log2 of the entry size; the hardware takes the shift amount from the low five bits of cls);
//   residual (2-byte) entries:  cls = (E if base != 0) | (s - 1),  byte offset = (((theta & (E-1)) | cls) >> cls) & ~1
//     -- base << s is E for both non-zero bases, the low bit the amount sets in u is shifted out, and the mask of theta rides in
//     the same v_and_or_b32: three instructions per gather from the unmasked angle (nibble tables are not split: resid_offset).
template <int FMT>
__device__ __forceinline__ uint32_t split_class(uint32_t r, uint32_t log2_entries)
{
    const uint32_t e = 1u << log2_entries;
    const uint32_t s = (r & 1u) ? 1u : 2u;
    if constexpr (FMT == 2 || FMT == 3) return ((r & 3u) ? e : 0u) | (s - 1u);      // (FMT 3: unused, natural layout)
    constexpr uint32_t LB = FMT == 1 ? 2u : 3u;
    const uint32_t base = (r & 1u) ? (e >> 1) : (((r >> 1) & 1u) ? (e >> 2) : 0u);
    return (base << LB) | (LB - s);                                  // left by LB - s (0 .. 2)
}

// plain / delta16 entry of class `cls`
template <int FMT>
__device__ __forceinline__ int2 tab_load_class(const BhwCordicCfg &cfg, const void *__restrict__ table, uint32_t u, uint32_t cls)
{
    static_assert(FMT == 0 || FMT == 1, "residual / nibble entries: resid_offset + resid_value");
    if constexpr (FMT == 1) {                                    // 4 bytes per entry
        const uint32_t boff = ((u << (cls & 31u)) | cls) & ~3u;
        const uint32_t e = ld_off<uint32_t>(table, boff);
        const int2 base = ld_off<int2>(cfg.tab_coarse, (u >> kPackLog) << 3);
        return make_int2(base.x + (int32_t)(int16_t)(e & 0xFFFFu), base.y + ((int32_t)e >> 16));
    } else {                                                     // 8 bytes per entry
        const uint32_t boff = ((u << (cls & 31u)) | cls) & ~7u;
        return ld_off<int2>(table, boff);
    }
}

// Byte offset of the residual word of entry u = K * rg mod E (rg = r + g * E/2), tile kernel.
// Nibble format (one byte per entry, natural layout): u itself.  With one-byte entries a wave's gather of harmonic K spans
// K * 64 bytes either way, the K = 1 gathers become unit-stride loads (4.7 instead of 16.4 cycles of the CU's address path,
// profiles/r02_ubench_vmem.txt) and the address is one instruction: -1.0 % on the whole call against the split layout.
// Residual format (two bytes per entry, split layout: natural is 2.9 % slower there).  Odd K: the class word (split_class).
// Even K: u = 2w or 4w with w = (K/2) rg or (K/4) rg, and the index is a bit field of w --
// (u >> 2) + (E/4 if u % 4 == 2) = w[lq-2:1] | w[0] << (lq-2), or u >> 2 = w mod E/4 -- three instructions / two.
template <int FMT, int K>
__device__ __forceinline__ uint32_t resid_offset(uint32_t rg, uint32_t theta, uint32_t cls, uint32_t lq, uint32_t emask)
{
    static_assert(FMT == 2 || FMT == 3, "residual / nibble entries");
    if constexpr (FMT == 3) {                                       // nibble tables keep the natural layout (table_layout)
        if constexpr (K <= 2) return theta;                        // K (r + g E/2) < E for r < E/2: nothing to wrap
        else return theta & emask;
    }
    else if constexpr ((K & 1) != 0) {
        return (((theta & emask) | cls) >> (cls & 31u)) & ~1u;
    } else if constexpr (K % 4 == 2) {
        const uint32_t w = (uint32_t)(K / 2) * rg;
        return (w & ((1u << (lq - 1u)) - 2u)) | ((w & 1u) << (lq - 1u));
    } else {
        const uint32_t w = (uint32_t)(K / 4) * rg;
        return (w << 1) & ((1u << (lq - 1u)) - 2u);
    }
}

// `head` = (c, s) of the first entry of u's 64-entry block (delta16; the caller holds it: lane 0 of the wave);
// `rec` = the residual format's record of u's cell (wave-uniform: a 64-entry block lies inside one cell).
template <int FMT = -1>
__device__ __forceinline__ void tab_store(void *__restrict__ table, uint32_t u, uint32_t log2_entries, uint32_t split, uint32_t dlog,
                                          void *coarse, int32_t c, int32_t s, int2 head, int4 rec, uint32_t *check_flag)
{
    const uint32_t idx = tab_index(u, log2_entries, split);
    if (FMT == 0 || (FMT < 0 && dlog == 0)) {
        reinterpret_cast<int2 *>(table)[idx] = make_int2(c, s);
    } else if (FMT == 1 || (FMT < 0 && dlog == kPackLog)) {
        const int32_t dc = c - head.x, ds = s - head.y;
        if (check_flag && !(fits_bits(dc, 16) && fits_bits(ds, 16))) atomicOr(check_flag, 1u);
        reinterpret_cast<uint32_t *>(table)[idx] = ((uint32_t)dc & 0xFFFFu) | ((uint32_t)ds << 16);
        if ((u & ((1u << kPackLog) - 1u)) == 0u) reinterpret_cast<int2 *>(coarse)[u >> kPackLog] = make_int2(c, s);
    } else {
        const uint32_t d = fmt_cell_log(dlog);
        const int2 p = tab_predict(rec, u & ((1u << d) - 1u), d);
        const int32_t dc = c - p.x, ds = s - p.y;
        if (FMT == 3 || (FMT < 0 && dlog >= kNibbleFlag)) {
            if (check_flag && !(fits_bits(dc, 4) && fits_bits(ds, 4))) atomicOr(check_flag, 1u);
            reinterpret_cast<uint8_t *>(table)[idx] = (uint8_t)(((uint32_t)dc & 0xFu) | (((uint32_t)ds & 0xFu) << 4));
        } else {
            if (check_flag && !(fits_bits(dc, 8) && fits_bits(ds, 8))) atomicOr(check_flag, 1u);
            reinterpret_cast<uint16_t *>(table)[idx] = (uint16_t)(((uint32_t)dc & 0xFFu) | (((uint32_t)ds & 0xFFu) << 8));
        }
    }
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

// Output stage.  Plain generation stores the coefficient; the fused apply (the window feeds a
// multiplier in front of an FFT) stores (x[i] * w[i]) >> shift -- exact 64-bit product like int_multNxN_dsp48.vhd:102,
// floor shift, low 32 bits -- so the coefficient vector never round-trips through HBM.
__device__ __forceinline__ void emit(const BhwWinCfg &win, int32_t *__restrict__ out, uint64_t idx, int32_t w)
{
    if (win.apply_x) w = (int32_t)(((int64_t)win.apply_x[idx] * (int64_t)w) >> win.apply_shift);
    out[idx] = w;
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
// Table strategy, pass 1: first-quadrant (c, s) for every distinct CORDIC input
//   u in [0, 2^(PW-2-z_shr)),  z0 = u << z_shl.
// Every harmonic of every coefficient evaluates this same function (the quadrant field is
// applied after the rotation), so the whole window needs only 2^(PW-2-z_shr) chains.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_table_build(BhwCordicCfg cfg, uint32_t entries, void *__restrict__ table)
{
    __shared__ T lut_s[32];
    stage_lut<T>(cfg, lut_s);
    const uint32_t u = blockIdx.x * kBlock + threadIdx.x;
    if (u >= entries) return;
    T x, y;
    cordic_q1<T>(lut_s, (T)cfg.x0, (T)((T)u << cfg.z_shl), (int)cfg.n_iter, x, y);
    const int32_t c = (int32_t)(x >> cfg.out_shr), sn = (int32_t)(y >> cfg.out_shr);
    // a wave holds one aligned 64-entry block (the packed format needs entries >= 64, see bhwk_packed_ok)
    const int2 head = make_int2(__builtin_amdgcn_readfirstlane(c), __builtin_amdgcn_readfirstlane(sn));
    const int4 rec = cfg.tab_dlog > kPackLog ? reinterpret_cast<const int4 *>(cfg.tab_coarse)[u >> fmt_cell_log(cfg.tab_dlog)] : make_int4(0, 0, 0, 0);
    tab_store(table, u, cfg.phi_width - 2 - cfg.z_shr, cfg.tab_split, cfg.tab_dlog, const_cast<void *>(cfg.tab_coarse), c, sn, head, rec,
              cfg.tab_check);
}

// ---------------------------------------------------------------------------------------
// Table strategy, pass 1, shared-prefix form.
//
// Leaves u (table entries) are contiguous in angle: z0(u) = u << z_shl.  At rotation k every leaf of
// a group takes the same decision as long as sign(z_k) agrees at the group's two end leaves (z_k is
// the same affine function of u for all of them), and then (x_k, y_k) is one value for the group.
//   phase 1: one lane per group of 64 leaves runs the chain from k = 1 until the first rotation at
//            which the group's end leaves disagree (or kPrefixMax), and parks (x, y, z_first, k) in LDS;
//   phase 2: each wave takes a group, broadcasts the parked state, and runs only the remaining
//            rotations with one lane per leaf.
// Rotation step in "mad" form (x, y 64-bit; z 32-bit; sg = +1 / -1 = the decision):
//   x += (-sg) * lo32(y >> k);  y += sg * lo32(x >> k)      -> v_ashrrev_i64 / v_mad_i64_i32
//   z += (-sg) * lut[k]                                       -> v_mad_i32_i24 once lut[k] < 2^23
// Valid when |x|,|y| < 2^33 and the quarter circle <= 2^32 (all models at W <= 32; VHDL: W+P <= 34):
// rotation 0 always adds (z0 >= 0), giving x1 = y1 = x0 and z1 = z0 - lut[0], which fits int32.
// ---------------------------------------------------------------------------------------
#ifndef BHW_HEADS_UNROLL
#define BHW_HEADS_UNROLL 1     // the cell-head chains unrolled on scalar ROM words: every workgroup starts with them (and the group prefixes), and a
                              // rolled loop on the LDS copy of the ROM makes that start 120 instructions longer: 0.1468 -> 0.1451 ms per window
#endif
#ifndef BHW_PREFIX_MAX
#define BHW_PREFIX_MAX 20
#endif
constexpr int kPrefixMax = BHW_PREFIX_MAX;   // deepest rotation a 64-leaf group is followed to in phase 1
constexpr int kGroupsPerWg = 64;    // phase 1: at most one group per lane of the first wave (plan.groups_per_wg <= 64)
constexpr int kBuildThreads = 256;  // phase 2: four waves, 16 groups each
constexpr int kHeadsMax = 40;       // residual format: cells of one workgroup (4096 entries >> 7 = 32) + 2

struct BhwBuildPlan {
    uint32_t lut[32];    // the rescaled ROM as 32-bit words (entries fit: quarter circle <= 2^32)
    uint32_t entries;    // 2^(PW-2-z_shr), a multiple of 64
    uint32_t n_iter;
    uint32_t z_shl;
    uint32_t out_shr;
    uint32_t log2_entries;
    uint32_t tab_split;
    uint32_t tab_dlog;        // packed table format (see tab_load)
    uint32_t pad0;
    const void *tab_coarse;
    uint32_t groups_per_wg;   // 4, 16 or 64: small tables use small workgroups so the grid still fills the chip
    uint32_t pad;
    int64_t  x0;
    uint32_t *check_flag;     // packed formats: set to 1 when a difference does not fit its field (NULL: configuration already verified)
};

// lut[k] < 2^23 for every k >= 9 whenever the fast path is legal (lut[k] <= atan(2^-k) 2^33 / pi), so the
// z update is one v_mad_i32_i24 there; earlier rotations use the three-op form.
constexpr int kMad24From = 9;

__device__ __forceinline__ void rot_step(int64_t &x, int64_t &y, int32_t &z, int k, uint32_t lutk)
{
    const int32_t m = z >> 31;                 // -1 when z < 0
    const int32_t sg = m | 1;                  // decision: -1 rotate back, +1 rotate forward
    const int32_t nsg = -sg;
    int32_t ys = (int32_t)(y >> k);
    int32_t xs = (int32_t)(x >> k);
    asm volatile("" : "+v"(ys), "+v"(xs));     // both shifts read the old state before either update lands
    x += (int64_t)nsg * (int64_t)ys;
    y += (int64_t)sg * (int64_t)xs;
    if (k >= kMad24From) z += __mul24(nsg, (int32_t)lutk);
    else                 z = (int32_t)((uint32_t)z - lutk + ((2u * lutk) & (uint32_t)m));
}

// FMT: table format as a template parameter (0 plain, 1 delta16, 2 residual): no format branches around the stores.
template <int NITER, int FMT>
__global__ __launch_bounds__(kBuildThreads) void k_table_build_shared(BhwBuildPlan plan, void *__restrict__ table)
{
    __shared__ int64_t gx[kGroupsPerWg];
    __shared__ int64_t gy[kGroupsPerWg];
    __shared__ int32_t gz[kGroupsPerWg];
    __shared__ int32_t gk[kGroupsPerWg];
    __shared__ uint32_t lut_s[32];
    if (threadIdx.x < 32) lut_s[threadIdx.x] = plan.lut[threadIdx.x];
    __syncthreads();

    constexpr int n_iter = NITER;
    const uint32_t s = plan.z_shl;
    const uint32_t gpw = plan.groups_per_wg;
    const uint32_t group0 = blockIdx.x * gpw;
    const uint32_t n_groups = plan.entries >> 6;

    // Residual format: the records {c, s, dc, ds} of the cells this workgroup touches.  Lanes of the second wave run the full
    // chain (the very rot_step of the leaves) at the cell starts -- heads cell_lo .. cell_lo + n_cell, plus head cell_lo - 1
    // for the table's last cell, whose end point is not an entry and which reuses the slope of the cell before it -- while
    // the first wave runs the group prefixes; cells that start inside this workgroup are also written out for the combine pass.
    __shared__ int32_t hc[kHeadsMax], hs[kHeadsMax];
    const uint32_t d = fmt_cell_log(plan.tab_dlog);
    constexpr bool resid = (FMT == 2 || FMT == 3);            // (FMT 4: no table, see the store below)
    const uint32_t cells_total = resid ? plan.entries >> d : 0u;
    uint32_t cell_lo = 0, n_cell = 0;
    if (resid) {
        const uint32_t u_end = ((group0 + gpw) << 6) < plan.entries ? ((group0 + gpw) << 6) : plan.entries;
        cell_lo = (group0 << 6) >> d;
        n_cell = ((u_end - 1u) >> d) - cell_lo + 1u;
    }
    if (resid && threadIdx.x >= 64u && threadIdx.x < 64u + n_cell + 2u) {
        const uint32_t t = threadIdx.x - 64u;
        const int64_t cell = (t <= n_cell) ? (int64_t)cell_lo + t : (int64_t)cell_lo - 1;
        if (cell >= 0 && cell < (int64_t)cells_total) {
            int64_t x = plan.x0, y = plan.x0;
            int32_t z = (int32_t)((((uint32_t)cell << d) << s) - lut_s[0]);
#if BHW_HEADS_UNROLL
#pragma unroll
            for (int r = 1; r < n_iter; ++r) rot_step(x, y, z, r, plan.lut[r]);
#else
#pragma unroll 1
            for (int r = 1; r < n_iter; ++r) rot_step(x, y, z, r, lut_s[r]);
#endif
            hc[t] = (int32_t)(x >> plan.out_shr);
            hs[t] = (int32_t)(y >> plan.out_shr);
        }
    }

    // ---- phase 1: shared prefix of each 64-leaf group ----
    if (threadIdx.x < gpw) {
        const uint32_t g = group0 + threadIdx.x;
        const uint32_t u_first = g << 6;
        int64_t x = plan.x0, y = plan.x0;                                        // after rotation 0
        int32_t zf = (int32_t)((u_first << s) - lut_s[0]);
        const uint32_t span = 63u << s;                                          // z_last - z_first
        int k = 1;
        bool live = g < n_groups;
        // unrolled: immediate shifts and scalar ROM words; this serial chain is the latency every workgroup starts with
        constexpr int kmax = n_iter < kPrefixMax ? n_iter : kPrefixMax;
#pragma unroll
        for (int kk = 1; kk < kmax; ++kk) {
            if (live) {
                const int32_t zl = (int32_t)((uint32_t)zf + span);
                if ((zf < 0) != (zl < 0)) {
                    live = false;                                                // the group splits at rotation kk
                } else {
                    rot_step(x, y, zf, kk, plan.lut[kk]);
                    k = kk + 1;
                }
            }
        }
        gx[threadIdx.x] = x;
        gy[threadIdx.x] = y;
        gz[threadIdx.x] = zf;
        gk[threadIdx.x] = k;
    }
    __syncthreads();

    auto record_of = [&](uint32_t cell) -> int4 {                  // cell in [cell_lo, cell_lo + n_cell)
        const uint32_t t = cell - cell_lo;
        if (cell + 1u < cells_total) return make_int4(hc[t], hs[t], hc[t + 1] - hc[t], hs[t + 1] - hs[t]);
        const uint32_t tp = t ? t - 1u : n_cell + 1u;                // last cell of the table: slope of the cell before it
        return make_int4(hc[t], hs[t], hc[t] - hc[tp], hs[t] - hs[tp]);
    };
    auto record = [&](uint32_t cell) -> int4 {                     // the same for a wave-uniform cell: scalar control flow
        return record_of(__builtin_amdgcn_readfirstlane(cell));
    };
    if (resid && threadIdx.x < n_cell) {
        const uint32_t cell = cell_lo + threadIdx.x;
        if ((cell << d) >= (group0 << 6))                            // starts inside this workgroup's entries: this one writes it
            reinterpret_cast<int4 *>(const_cast<void *>(plan.tab_coarse))[cell] = record_of(cell);
    }

    // ---- phase 2: one wave per group, one lane per leaf, remaining rotations only ----
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;   // g and what derives from it stay scalar
    // table index of leaf (g, lane) = idx_a + g * idx_m: natural layout 64 g + lane; split layout per residue class of the lane
    // (64 g + lane has the lane's residue mod 4), tab_index() folded into two per-lane constants
    uint32_t idx_a = lane, idx_m = 64u;
    if (plan.tab_split) {
        const uint32_t e = 1u << plan.log2_entries;
        if (lane & 1u)      { idx_a = (e >> 1) + (lane >> 1); idx_m = 32u; }
        else if (lane & 2u) { idx_a = (e >> 2) + (lane >> 2); idx_m = 16u; }
        else                { idx_a = lane >> 2;              idx_m = 16u; }
    }
    for (uint32_t gi = wave; gi < gpw; gi += kBuildThreads / 64) {
        const uint32_t g = __builtin_amdgcn_readfirstlane(group0 + gi);
        if (g >= n_groups) break;
        int64_t x = gx[gi], y = gy[gi];
        int32_t z = (int32_t)((uint32_t)gz[gi] + (lane << s));
        const int k0 = __builtin_amdgcn_readfirstlane(gk[gi]);
        // Fully unrolled: every shift is an immediate and every lut entry a scalar kernel argument; k0 is
        // wave-uniform, so each guard is one scalar compare-and-branch.
#pragma unroll
        for (int k = 1; k < NITER; ++k) {
            if (k >= kPrefixMax || k >= k0) rot_step(x, y, z, k, plan.lut[k]);   // k0 <= kPrefixMax: no guard (one basic block) beyond it
        }
        const int32_t c = (int32_t)(x >> plan.out_shr), sn = (int32_t)(y >> plan.out_shr);
        const int2 head = make_int2(__builtin_amdgcn_readfirstlane(c), __builtin_amdgcn_readfirstlane(sn));   // leaf 0 of the group
        const uint32_t idx = idx_a + g * idx_m;
        if constexpr (FMT == 4) {
            // sin / cos sweep over one whole period (bhwk_sincos): no table -- the leaf's first-quadrant pair goes out as its four
            // quadrant images, phase q*E + u at stream position (phase - theta0) mod N.  table = d_sin, tab_coarse = d_cos (either
            // may be NULL), pad0 = the model's negation rule, pad = theta0 mod N.
            const uint32_t u = (g << 6) + lane, nmask = 4u * plan.entries - 1u;
            int32_t *d_sin = reinterpret_cast<int32_t *>(table), *d_cos = reinterpret_cast<int32_t *>(const_cast<void *>(plan.tab_coarse));
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) {
                int32_t oc, os;
                quadrant_map(q, c, sn, plan.pad0, oc, os);
                const uint32_t i = (q * plan.entries + u - plan.pad) & nmask;
                if (d_sin) d_sin[i] = os;
                if (d_cos) d_cos[i] = oc;
            }
        } else if constexpr (FMT == 0) {
            reinterpret_cast<int2 *>(table)[idx] = make_int2(c, sn);
        } else if constexpr (FMT == 1) {
            const int32_t dc = c - head.x, ds = sn - head.y;
            if (plan.check_flag && !(fits_bits(dc, 16) && fits_bits(ds, 16))) atomicOr(plan.check_flag, 1u);   // scalar guard: verified configurations skip it
            reinterpret_cast<uint32_t *>(table)[idx] = ((uint32_t)dc & 0xFFFFu) | ((uint32_t)ds << 16);
            if (lane == 0u) reinterpret_cast<int2 *>(const_cast<void *>(plan.tab_coarse))[g] = head;   // block = group
        } else {
            const int4 rec = record((g << 6) >> d);                                                     // wave-uniform
            const int2 p = tab_predict(rec, ((g << 6) & ((1u << d) - 1u)) + lane, d);
            const int32_t dc = c - p.x, ds = sn - p.y;
            // (byte-wide stores make this variant 2 % slower than the two-byte one at equal instruction counts, 72.3 against 70.7 us;
            // staging a workgroup's entries in shared memory and writing them as 16-byte packets costs more than it saves, +1.5 us)
            if constexpr (FMT == 3) {
                if (plan.check_flag && !(fits_bits(dc, 4) && fits_bits(ds, 4))) atomicOr(plan.check_flag, 1u);
                reinterpret_cast<uint8_t *>(table)[idx] = (uint8_t)(((uint32_t)dc & 0xFu) | (((uint32_t)ds & 0xFu) << 4));
            } else {
                if (plan.check_flag && !(fits_bits(dc, 8) && fits_bits(ds, 8))) atomicOr(plan.check_flag, 1u);
                reinterpret_cast<uint16_t *>(table)[idx] = (uint16_t)(((uint32_t)dc & 0xFFu) | (((uint32_t)ds & 0xFFu) << 8));
            }
        }
    }
}

// Table strategy, pass 1, octant-mirror form (residual / nibble formats in the split layout; BHW_BUILD_MIRROR).
//
// After rotation 0 the state is the 45-degree vector (x0, x0) and 2 * lut[0] is exactly a quarter turn (checked by the launcher),
// so the chain of u' = E - u is the chain of u with x and y swapped, z negated and every decision flipped -- bit for bit, floors
// included -- as long as no z_k on the way is exactly 0 (z >= 0 rotates forward in both).  Zero events are common only in the last
// rotations (|z| is a few units there), so:
//   * only u in [0, E/2) run a chain of their own; u' in (E/2, E) are images of u in [1, E/2 - 1]; the middle entry E/2 is a
//     deferred chain of the last workgroup;
//   * at rotation KS = NITER - BHW_MIRROR_TAIL the image state is taken as (y, x, -z) and the last rotations run for both;
//   * a lane that met z_k == 0 before KS (one v_cmp per shared rotation, collected in a scalar mask; about 1 % of the lanes, but
//     every second wave has one) does not store its image: it appends u to a worklist in shared memory, and after the groups
//     the workgroup's first lanes run those images as chains of their own (a wave-wide replay in place costs more than the
//     symmetry saves);
//   * a group whose shared prefix itself met z == 0 (only its leaf 0 can) recomputes that one image from scratch;
//   * the records of the image cells come from chains of their own (third wave), never from the symmetry.
// Per pair of entries 9 shared + 2 x 6 own rotations instead of 2 x 15.
#ifndef BHW_MIRROR_TAIL
#define BHW_MIRROR_TAIL 6
#endif
#ifndef BHW_MIRROR_GPW
#define BHW_MIRROR_GPW 64
#endif
#ifndef BHW_MIRROR_EXACT
#define BHW_MIRROR_EXACT 1          // own chains for u in [0, E/2) only (a power-of-two grid, no straggler workgroup); entry E/2 goes through the worklist.  0: one more group [E/2, E/2 + 64)
#endif
#ifndef BHW_MIRROR_WORKMAX
#define BHW_MIRROR_WORKMAX 512      // worklist slots per workgroup (about 40 used); a build with 4 exercises the overflow path
#endif
template <int NITER, int FMT>
__global__ __launch_bounds__(kBuildThreads) void k_table_build_mirror(BhwBuildPlan plan, void *__restrict__ table)
{
    static_assert(FMT == 2 || FMT == 3, "residual / nibble entries");
    __shared__ int64_t gx[kGroupsPerWg];
    __shared__ int64_t gy[kGroupsPerWg];
    __shared__ int32_t gz[kGroupsPerWg];
    __shared__ int32_t gk[kGroupsPerWg];
    __shared__ uint32_t gflag[kGroupsPerWg];
    __shared__ uint32_t lut_s[32];
    constexpr uint32_t kWorkMax = BHW_MIRROR_WORKMAX;                // images to run as chains of their own (expected ~40 per workgroup)
    __shared__ uint32_t work_n;
    __shared__ uint32_t work_u[kWorkMax];
    if (threadIdx.x < 32) lut_s[threadIdx.x] = plan.lut[threadIdx.x];
    constexpr int n_iter = NITER;
    constexpr int KS = NITER - BHW_MIRROR_TAIL;                      // image state taken at this rotation
    const uint32_t s = plan.z_shl;
    constexpr uint32_t gpw = BHW_MIRROR_GPW;                         // own groups per workgroup
    const uint32_t group0 = blockIdx.x * gpw;
    const uint32_t E = plan.entries;
    const uint32_t n_groups = (E >> 7) + (BHW_MIRROR_EXACT ? 0u : 1u);   // groups that run chains of their own
    const uint32_t u_lo = group0 << 6;
    const uint32_t u_hi = ((group0 + gpw) << 6) < (n_groups << 6) ? ((group0 + gpw) << 6) : (n_groups << 6);   // own entries [u_lo, u_hi)
    const uint32_t m_last = (E >> 1) - (BHW_MIRROR_EXACT ? 1u : 64u);    // u in [1, m_last] also produce the image E - u
    if (threadIdx.x == 0) {
        work_n = 0u;
        if (BHW_MIRROR_EXACT && u_hi == (E >> 1)) {                  // the middle entry E/2 (its own image): one more deferred chain
            work_u[0] = E >> 1;
            work_n = 1u;
        }
    }
    __syncthreads();


    // records {c, s, dc, ds} of the cells this workgroup stores into: w = 0 its own range, w = 1 the image range.  Heads
    // lo .. lo + n, plus head lo - 1 for the table's last cell (see k_table_build_shared), by full chains of the second / third wave.
    __shared__ int32_t hc[2][kHeadsMax], hs[2][kHeadsMax];
    const uint32_t d = fmt_cell_log(plan.tab_dlog);
    const uint32_t cells_total = E >> d;
    uint32_t cell_lo[2] = {0u, 0u}, n_cell[2] = {0u, 0u}, r_lo[2] = {u_lo, 0u}, r_hi[2] = {u_hi, 0u};   // entry ranges [r_lo, r_hi)
    {
        const uint32_t a = u_lo > 1u ? u_lo : 1u, b = (u_hi - 1u) < m_last ? (u_hi - 1u) : m_last;        // sources a .. b
        if (a <= b) { r_lo[1] = E - b; r_hi[1] = E - a + 1u; }
        if (BHW_MIRROR_EXACT && a <= b && b == (E >> 1) - 1u) r_lo[1] = E >> 1;      // the middle entry belongs to this image range
    }
#pragma unroll
    for (int w = 0; w < 2; ++w)
        if (r_hi[w] > r_lo[w]) {
            cell_lo[w] = r_lo[w] >> d;
            n_cell[w] = ((r_hi[w] - 1u) >> d) - cell_lo[w] + 1u;
        }
    {
        const uint32_t w = (threadIdx.x >> 6) - 1u;                  // wave 1 -> own range, wave 2 -> image range
        const uint32_t t = threadIdx.x & 63u;
        if (w < 2u && n_cell[w] && t < n_cell[w] + 2u) {
            const int64_t cell = (t <= n_cell[w]) ? (int64_t)cell_lo[w] + t : (int64_t)cell_lo[w] - 1;
            if (cell >= 0 && cell < (int64_t)cells_total) {
                int64_t x = plan.x0, y = plan.x0;
                int32_t z = (int32_t)((((uint32_t)cell << d) << s) - lut_s[0]);
#if BHW_HEADS_UNROLL
#pragma unroll
                for (int r = 1; r < n_iter; ++r) rot_step(x, y, z, r, plan.lut[r]);
#else
#pragma unroll 1
                for (int r = 1; r < n_iter; ++r) rot_step(x, y, z, r, lut_s[r]);
#endif
                hc[w][t] = (int32_t)(x >> plan.out_shr);
                hs[w][t] = (int32_t)(y >> plan.out_shr);
            }
        }
    }

    // ---- phase 1: shared prefix of each 64-leaf group (never past KS: the image state is taken there) ----
    constexpr int kcap = KS < kPrefixMax ? KS : kPrefixMax;
    if (threadIdx.x < gpw) {
        const uint32_t g = group0 + threadIdx.x;
        const uint32_t u_first = g << 6;
        int64_t x = plan.x0, y = plan.x0;                                        // after rotation 0
        int32_t zf = (int32_t)((u_first << s) - lut_s[0]);
        const uint32_t span = 63u << s;                                          // z_last - z_first
        int k = 1;
        bool live = g < n_groups;
        uint32_t zero0 = 0u;                                                     // leaf 0 met z == 0 inside the shared prefix
#pragma unroll
        for (int kk = 1; kk < kcap; ++kk) {
            if (live) {
                const int32_t zl = (int32_t)((uint32_t)zf + span);
                if ((zf < 0) != (zl < 0)) {
                    live = false;                                                // the group splits at rotation kk
                } else {
                    zero0 |= (uint32_t)(zf == 0);
                    rot_step(x, y, zf, kk, plan.lut[kk]);
                    k = kk + 1;
                }
            }
        }
        gx[threadIdx.x] = x;
        gy[threadIdx.x] = y;
        gz[threadIdx.x] = zf;
        gk[threadIdx.x] = k;
        gflag[threadIdx.x] = zero0;
    }
    __syncthreads();

    // the records themselves, once per workgroup: the groups read them back with one ds_read_b128 each
    __shared__ int4 hrec[2][kHeadsMax];
    {
        const uint32_t w = threadIdx.x >> 6, t = threadIdx.x & 63u;
        if (w < 2u && t < n_cell[w]) {
            const uint32_t cell = cell_lo[w] + t;
            int4 r;
            if (cell + 1u < cells_total) r = make_int4(hc[w][t], hs[w][t], hc[w][t + 1] - hc[w][t], hs[w][t + 1] - hs[w][t]);
            else {
                const uint32_t tp = t ? t - 1u : n_cell[w] + 1u;     // last cell of the table: slope of the cell before it
                r = make_int4(hc[w][t], hs[w][t], hc[w][t] - hc[w][tp], hs[w][t] - hs[w][tp]);
            }
            hrec[w][t] = r;
            if ((cell << d) >= r_lo[w])                              // its first entry is stored by this workgroup
                reinterpret_cast<int4 *>(const_cast<void *>(plan.tab_coarse))[cell] = r;
        }
    }
    __syncthreads();
    auto record_of = [&](int w, uint32_t cell) -> int4 { return hrec[w][cell - cell_lo[w]]; };   // cell in [cell_lo[w], cell_lo[w] + n_cell[w])
    auto record = [&](int w, uint32_t cell) -> int4 {              // the same for a wave-uniform cell (a broadcast read)
        return record_of(w, __builtin_amdgcn_readfirstlane(cell));
    };

    // one entry: deviation from the record's straight line, checked and packed
    auto store_entry = [&](uint32_t idx, int32_t c, int32_t sn, const int4 rec, uint32_t pos) {
        const int2 p = tab_predict(rec, pos, d);
        const int32_t dc = c - p.x, ds = sn - p.y;
        if constexpr (FMT == 3) {
            if (plan.check_flag && !(fits_bits(dc, 4) && fits_bits(ds, 4))) atomicOr(plan.check_flag, 1u);
            reinterpret_cast<uint8_t *>(table)[idx] = (uint8_t)(((uint32_t)dc & 0xFu) | (((uint32_t)ds & 0xFu) << 4));
        } else {
            if (plan.check_flag && !(fits_bits(dc, 8) && fits_bits(ds, 8))) atomicOr(plan.check_flag, 1u);
            reinterpret_cast<uint16_t *>(table)[idx] = (uint16_t)(((uint32_t)dc & 0xFFu) | (((uint32_t)ds & 0xFFu) << 8));
        }
    };

    // ---- phase 2: one wave per group, one lane per leaf (and its image) ----
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;   // g and the cell arithmetic below stay scalar
    // index of leaf (g, lane) = idx_a + g * idx_m; index of its image E - u = idx_i - (that) -- split layout per residue class of the lane
    uint32_t idx_a, idx_m, idx_i;
    if (lane & 1u)      { idx_a = (E >> 1) + (lane >> 1); idx_m = 32u; idx_i = E + (E >> 1) - 1u; }
    else if (lane & 2u) { idx_a = (E >> 2) + (lane >> 2); idx_m = 16u; idx_i = (E >> 1) + (E >> 2) - 1u; }
    else                { idx_a = lane >> 2;              idx_m = 16u; idx_i = E >> 2; }
    if (!plan.tab_split) { idx_a = lane; idx_m = 64u; idx_i = E; }  // natural layout (nibble tables): index u, image E - u
    const uint32_t fmask = (1u << d) - 1u;
    for (uint32_t gi = wave; gi < gpw; gi += kBuildThreads / 64) {
        const uint32_t g = __builtin_amdgcn_readfirstlane(group0 + gi);   // (kept in a vector register otherwise, and the cell arithmetic with it)
        if (g >= n_groups) break;
        int64_t x = gx[gi], y = gy[gi];
        int32_t z = (int32_t)((uint32_t)gz[gi] + (lane << s));
        const int k0 = __builtin_amdgcn_readfirstlane(gk[gi]);
        const uint32_t gf = __builtin_amdgcn_readfirstlane(gflag[gi]);
        uint64_t zmask = 0ull;                                       // lanes whose z_k was exactly 0 at a rotation before KS
#pragma unroll
        for (int k = 1; k < KS; ++k) {
            if (k >= kcap || k >= k0) {
                zmask |= __builtin_amdgcn_ballot_w64(z == 0);
                rot_step(x, y, z, k, plan.lut[k]);
            }
        }
        int64_t x2 = y, y2 = x;                                      // image chain at rotation KS
        int32_t z2 = -z;
        bool deferred = false;                                       // this lane's image goes to the worklist
        if (zmask != 0ull) {                                         // scalar
            const uint32_t u_ = (g << 6) + lane;
            if (((zmask >> lane) & 1ull) != 0ull && u_ >= 1u && u_ <= m_last) {
                const uint32_t slot = atomicAdd(&work_n, 1u);
                if (slot < kWorkMax) { work_u[slot] = u_; deferred = true; }
                else {                                               // list full (never seen): the image chain from scratch, in place
                    int64_t xf = plan.x0, yf = plan.x0;
                    int32_t zf = (int32_t)(((E - u_) << s) - lut_s[0]);
#pragma unroll 1
                    for (int r = 1; r < KS; ++r) rot_step(xf, yf, zf, r, lut_s[r]);
                    x2 = xf; y2 = yf; z2 = zf;
                }
            }
        }
#pragma unroll
        for (int k = KS; k < NITER; ++k) {
            rot_step(x, y, z, k, plan.lut[k]);
            rot_step(x2, y2, z2, k, plan.lut[k]);
        }
        const uint32_t g6 = __builtin_amdgcn_readfirstlane(g << 6);   // back in a scalar register (merged with the zero-event path's copy it
                                                                      // lands in a vector one, and the wave-uniform cell arithmetic below with it)
        const uint32_t u = g6 + lane;
        const uint32_t idx = idx_a + g * idx_m;
        store_entry(idx, (int32_t)(x >> plan.out_shr), (int32_t)(y >> plan.out_shr), record(0, g6 >> d), (g6 & fmask) + lane);
        int32_t c2 = (int32_t)(x2 >> plan.out_shr), s2 = (int32_t)(y2 >> plan.out_shr);
        if (gf != 0u && lane == 0u && u >= 1u && u <= m_last) {       // rare (scalar test first): the shared prefix is not mirrored for leaf 0
            int64_t xf = plan.x0, yf = plan.x0;
            int32_t zf = (int32_t)(((E - u) << s) - lut_s[0]);
#pragma unroll 1
            for (int r = 1; r < n_iter; ++r) rot_step(xf, yf, zf, r, lut_s[r]);
            c2 = (int32_t)(xf >> plan.out_shr);
            s2 = (int32_t)(yf >> plan.out_shr);
        }
        if (u >= 1u && u <= m_last && !deferred) {
            // images E - 64g - 63 .. E - 64g, descending with the lane: one cell, or two when lane 0's image opens the next one
            const uint32_t um = E - u;
            const uint32_t top = E - g6, cell_a = (top - 63u) >> d, cell_b = top >> d;             // wave-uniform
            const uint32_t cell_c = cell_a >= cell_lo[1] ? cell_a : cell_lo[1];                      // (sources above m_last are masked off)
            int4 rec2 = record(1, cell_c);
            if (__builtin_amdgcn_readfirstlane(cell_b) != __builtin_amdgcn_readfirstlane(cell_c)) { // scalar branch, 1 group in 2^(d-6)
                const int4 rb = record(1, cell_b);
                if (lane == 0u) rec2 = rb;
            }
            store_entry(idx_i - idx, c2, s2, rec2, um & fmask);
        }
    }
    // ---- the deferred images: one lane each, the whole chain ----
    __syncthreads();
    const uint32_t n_work = work_n < kWorkMax ? work_n : kWorkMax;
    for (uint32_t i = threadIdx.x; i < n_work; i += kBuildThreads) {
        const uint32_t um = E - work_u[i];
        int64_t xf = plan.x0, yf = plan.x0;
        int32_t zf = (int32_t)((um << s) - lut_s[0]);
#pragma unroll 1
        for (int r = 1; r < n_iter; ++r) rot_step(xf, yf, zf, r, lut_s[r]);
        store_entry(tab_index(um, plan.log2_entries, plan.tab_split), (int32_t)(xf >> plan.out_shr), (int32_t)(yf >> plan.out_shr), record_of(1, um >> d), um & fmask);
    }
}

// Table strategy, pass 1, small tables: one lane per entry, the whole chain unrolled in the mad form, plain natural layout.
// Below ~2^20 entries the shared-prefix kernel is bound by the latency of its serial prefix phase (8.3 us for 2^18 entries);
// 64 independent chains per wave finish sooner.
template <int NITER>
__global__ __launch_bounds__(kBlock) void k_table_build_plain(BhwCordicCfg cfg, uint32_t entries, int2 *__restrict__ table)
{
    __shared__ uint32_t lut_s[32];
    if (threadIdx.x < 32) lut_s[threadIdx.x] = (uint32_t)cfg.lut[threadIdx.x];
    __syncthreads();
    const uint32_t u = blockIdx.x * kBlock + threadIdx.x;
    if (u >= entries) return;
    int64_t x = cfg.x0, y = cfg.x0;                                              // rotation 0 always adds (z0 >= 0)
    int32_t z = (int32_t)((u << cfg.z_shl) - lut_s[0]);
#pragma unroll
    for (int r = 1; r < NITER; ++r) rot_step(x, y, z, r, lut_s[r]);
    table[u] = make_int2((int32_t)(x >> cfg.out_shr), (int32_t)(y >> cfg.out_shr));
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

// Table strategy, pass 2 (general form): one lane per coefficient, K-1 gathers.
__global__ __launch_bounds__(kBlock) void k_table_combine(BhwCordicCfg cfg, BhwWinCfg win, const void *__restrict__ table,
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
        const int2 cs = tab_load(cfg, table, (theta & tmask) >> cfg.z_shr, pw - 2 - cfg.z_shr);
        int32_t c, s;
        quadrant_map(theta >> (pw - 2), cs.x, cs.y, cfg.ones_neg, c, s);
        combine_term(acc, win.aa[k], c, k, cfg.dat_width, win.combine);
    }
    emit(win, out, i, combine_final(acc, cfg.dat_width, win.combine, win.n_terms));
}

// Table strategy, pass 2, whole-period form ("quadrant fold").  Lane r in [0, N/4) owns the four
// coefficients n = r + j*N/4.  For harmonic k their phases k*n = k*r + j*k*N/4 differ only in the
// quadrant field, which every model applies AFTER the rotation (win_function.cpp:86-88,135-150 |
// cordic_sincos.cpp:25,70-86 | cordic_dds.vhd:170-172,232-246), so one (c, s) gather serves all four.
__global__ __launch_bounds__(kBlock) void k_table_combine_fold(BhwCordicCfg cfg, BhwWinCfg win, const void *__restrict__ table,
                                                                int32_t *__restrict__ out)
{
    const uint32_t pw = cfg.phi_width;
    const uint32_t quarter = 1u << (pw - 2);
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= quarter) return;
    const uint32_t mask = (pw >= 32) ? 0xFFFFFFFFu : ((1u << pw) - 1u);
    const uint32_t tmask = quarter - 1u;
    const uint32_t W = cfg.dat_width;
    int64_t acc0 = win.aa[0], acc1 = acc0, acc2 = acc0, acc3 = acc0;
    for (uint32_t k = 1; k < win.n_terms; ++k) {
        const uint32_t theta = (k * r) & mask;
        const uint32_t q = theta >> (pw - 2);
        const int2 cs = tab_load(cfg, table, (theta & tmask) >> cfg.z_shr, pw - 2 - cfg.z_shr);
        const int32_t nc = cfg.ones_neg ? ~cs.x : -cs.x;
        const int32_t ns = cfg.ones_neg ? ~cs.y : -cs.y;
        // cosine in quadrant 0..3: c, -s, -c, s
        int64_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        combine_term(p0, win.aa[k], cs.x, 0, W, win.combine);
        combine_term(p1, win.aa[k], ns, 0, W, win.combine);
        combine_term(p2, win.aa[k], nc, 0, W, win.combine);
        combine_term(p3, win.aa[k], cs.y, 0, W, win.combine);
        if (k & 1u) { p0 = -p0; p1 = -p1; p2 = -p2; p3 = -p3; }
        // image j sits in quadrant (q + j*k) & 3
        const uint32_t q0 = q, q1 = (q + k) & 3u, q2 = (q + 2u * k) & 3u, q3 = (q + 3u * k) & 3u;
        acc0 += q0 == 0 ? p0 : q0 == 1 ? p1 : q0 == 2 ? p2 : p3;
        acc1 += q1 == 0 ? p0 : q1 == 1 ? p1 : q1 == 2 ? p2 : p3;
        acc2 += q2 == 0 ? p0 : q2 == 1 ? p1 : q2 == 2 ? p2 : p3;
        acc3 += q3 == 0 ? p0 : q3 == 1 ? p1 : q3 == 2 ? p2 : p3;
    }
    emit(win, out, r, combine_final(acc0, W, win.combine, win.n_terms));
    emit(win, out, r + quarter, combine_final(acc1, W, win.combine, win.n_terms));
    emit(win, out, r + 2u * quarter, combine_final(acc2, W, win.combine, win.n_terms));
    emit(win, out, r + 3u * quarter, combine_final(acc3, W, win.combine, win.n_terms));
}

// ---------------------------------------------------------------------------------------
// Table strategy, pass 2, super-tile form (z_shr == 0).
//
// Lane r gathers entry t_k = k*r mod E (E = N/4) for harmonic k.  A run of consecutive r therefore
// reads every k-th entry of a span, and the other k-1 residues of that span are wanted by the runs
// r + i*inv(k) mod E.  A workgroup takes the 15 runs  r0 + i3*inv3 + i5*inv5 + [0, B)  together: for
// k = 3 the three i3-siblings interleave into one dense span, for k = 5 the five i5-siblings do, k = 6
// reads E[3r'] (dense for the same reason), k = 1, 2, 4 are dense in the split layout on their own.
// Every table line is then fetched for entries that are all used, instead of 1/k of them.
// The 15 run offsets are ~E/15 apart, so tiles m = 0..n_tiles-1 cover the ring once; the few entries
// covered twice at the seams are recomputed with identical results (idempotent stores).
// ---------------------------------------------------------------------------------------
// Tile shape (profiles/r01_ab_inproc.txt): 5 thread groups x 192 lanes.  Thread group p holds the three inv3-siblings of the
// 15-run tile with i5 = p (24 sums per thread, ~52 VGPRs), so two 960-thread workgroups fit a CU and the gathers of one
// overlap the arithmetic of the other: 0.1907 ms vs 0.1945 ms for 3 groups x 256 lanes (40 sums per thread, one workgroup per CU).
#ifndef BHW_TILE_THREADS
#define BHW_TILE_THREADS 960
#endif
#ifndef BHW_TILE_LANES
#define BHW_TILE_LANES 192
#endif
#ifndef BHW_TILE_WAVES
#define BHW_TILE_WAVES 8      // waves per SIMD the tile kernel is register-allocated for
#endif
constexpr int kTileThreads = BHW_TILE_THREADS;
constexpr int kTileLanes = BHW_TILE_LANES;     // tile width in lanes; the tile's 15 runs are split over kTileThreads / kTileLanes thread groups

struct BhwTilePlan {
    uint32_t offs[16];   // (i3*inv3 + i5*inv5) mod ring, index i3 + 3*i5; padded by repeating the last run
    uint32_t n_tiles;    // tiles that cover the ring once
    uint32_t tile0;      // first tile of this launch (interleaved ownership parts launch a sub-range of the tiles)
    uint32_t img_mask;   // MASKED instances: bit 2j + h set = image (h, j), i.e. stream indices [(2j + h) N/8, +N/8), is wanted
    uint32_t n0mod;      // MASKED instances: stream index (mod N) that `out` points at; image m lands at ((m N/8 - n0mod) mod N)
};

__device__ __forceinline__ int32_t wrap32(int32_t v, uint32_t bits)
{
    const uint32_t sh = 32u - bits;
    return (int32_t)((uint32_t)v << sh) >> sh;
}


// W-bit sums of the cosine-sum rules in 32-bit registers.  HLS rule: everything modulo 2^32, wrapped to W bits at the end.
// VHDL rule: the sum needs W+2 bits, so it is carried as S = 4*hi + lo (hi modulo 2^32, lo a small exact integer):
//   b_k = wrap_W((P >> (W-1)) + ((P >> (W-2)) & 1))   == the slice-and-round of bh_win_7term.vhd:353-402 on the 2W-bit product P
//   S>>2 = hi + (lo>>2),  (S>>1)&1 = (lo>>1)&1,  S>>1 = 2*hi + (lo>>1),  S&1 = lo&1
struct Sum32 {
    int32_t hi, lo;
    __device__ __forceinline__ Sum32 &operator+=(const Sum32 &o) { hi += o.hi; lo += o.lo; return *this; }
};

__device__ __forceinline__ int32_t acc_value(int32_t v) { return v; }
__device__ __forceinline__ int32_t acc_value(const Sum32 &v) { return v.hi; }

template <uint32_t COMBINE>
__device__ __forceinline__ void w32_term(Sum32 &acc, int32_t a, int32_t v, uint32_t k, uint32_t W)
{
    const int64_t P = (int64_t)a * (int64_t)v;
    if constexpr (COMBINE == BHW_COMBINE_HLS) {
        const int32_t m = (int32_t)(P >> (W - 2));
        acc.hi += (k & 1u) ? -m : m;
    } else {
        const int32_t b = wrap32((int32_t)(P >> (W - 1)) + (int32_t)(((uint32_t)P >> (W - 2)) & 1u), W);
        if (k & 1u) { acc.hi -= b >> 2; acc.lo -= b & 3; }
        else        { acc.hi += b >> 2; acc.lo += b & 3; }
    }
}

template <uint32_t COMBINE>
__device__ __forceinline__ int32_t w32_final(const Sum32 &acc, uint32_t W, uint32_t n_terms)
{
    if constexpr (COMBINE == BHW_COMBINE_HLS) return wrap32(acc.hi, W);
    else if (n_terms == 2) return wrap32(2 * acc.hi + (acc.lo >> 1) + (acc.lo & 1), W);          // hamming_win.vhd:214-228
    else return wrap32(acc.hi + (acc.lo >> 2) + ((acc.lo >> 1) & 1), W);                         // bh_win_7term.vhd:409-435
}

// MODE 0: HLS cosine-sum, two's-complement quadrant map, sums kept modulo 2^32 (exact: the result is
//         wrapped to W <= 32 bits anyway, win_function.cpp:375);  MODE 1: same with the one's-complement map of
//         the cpp model;  MODE 2: VHDL cosine-sum (either quadrant map): per-product slice-and-round b_k in 32 bits, the
//         W+2-bit sum carried as 4*hi + lo (Sum32 above).
// sv[i] = the harmonic's term for an image whose quadrant is q + i (MODE 0/1: already signed (-1)^K; MODE 2: b_k, sign applied
// when it is accumulated)
// QBASE / QBITS: what the caller knows about q at compile time.  A ring lane r < N/8 turns harmonic K through fewer than K/2 + 1
// quadrants, so q - QBASE takes 1 (QBITS 0), 2 (QBITS 1) or more (QBITS 2: plain two-bit rotation, QBASE 0) values: harmonics 1 and
// 2 need no run-time rotation at all, harmonics 3 and 4 one select per slot instead of two (ring_quadrants() below).
// FAST (HLS rule only): every |a_k| < 2^(W-3), so a_k << (34 - W) fits int32 and  (a_k * v) >> (W-2)  is the high half of the
// 32 x 32 product of that pre-shifted weight -- one v_mul_hi_i32 instead of v_mad_i64_i32 + v_ashrrev_i64, the low half telling
// whether the shifted-out bits were zero.  The caller passes the pre-shifted weight as `a`.
#ifndef BHW_TILE_ROT
#define BHW_TILE_ROT 1      // how the quadrant rotation is written: 0 plain selects (the compiler turns them into three compare + select pairs per slot), 1 quadrant bits opaque to it (-1.7 %)
#endif
template <int K, int MODE, int QBASE = 0, int QBITS = 2, bool FAST = false>
__device__ __forceinline__ void tile_harmonic(const BhwCordicCfg &cfg, const int32_t a, const uint32_t W, const int2 cs, const uint32_t q,
                                              int32_t (&sv)[4])
{
    int32_t p0, p1, p2, p3;                                // cosine term in quadrant 0..3: c, -s, -c, s
    if constexpr (FAST && MODE != 2) {
        // four one-instruction products with the quadrant's own operand (-v or ~v); the harmonic's sign (-1)^K is NOT applied
        // here: all four candidates carry it alike, so tile_accumulate<K, OFF, true> subtracts instead of adding for odd K
        p0 = __mulhi(a, cs.x);
        p3 = __mulhi(a, cs.y);
        if constexpr (MODE == 1) {
            p1 = __mulhi(a, ~cs.y);
            p2 = __mulhi(a, ~cs.x);
        } else {
            // a * (-v) is the same 64-bit product as (-a) * v: the negation moves to the weight (a scalar), two vector
            // instructions fewer per gather (the callers keep a > -2^31; a table value is never -2^31)
            const int32_t na = -a;
            p1 = __mulhi(na, cs.y);
            p2 = __mulhi(na, cs.x);
        }
    } else if constexpr (MODE == 2) {
        const int32_t nc = cfg.ones_neg ? ~cs.x : -cs.x;
        const int32_t ns = cfg.ones_neg ? ~cs.y : -cs.y;
        auto slice_round = [&](int32_t v) -> int32_t {     // bh_win_7term.vhd:353-402 on the 2W-bit product (see Sum32)
            const int64_t P = (int64_t)a * (int64_t)v;
            return wrap32((int32_t)(P >> (W - 1)) + (int32_t)(((uint32_t)P >> (W - 2)) & 1u), W);
        };
        p0 = slice_round(cs.x);
        p1 = slice_round(ns);
        p2 = slice_round(nc);
        p3 = slice_round(cs.y);
    } else {
        const uint32_t sh = W - 2;                         // mlt_k = (a_k * c_k) >> (NWIDTH-2), win_function.cpp:368-373
        if constexpr (MODE == 1) {
            const int32_t nc = ~cs.x, ns = ~cs.y;
            const int32_t m0 = (int32_t)(((int64_t)a * cs.x) >> sh), m1 = (int32_t)(((int64_t)a * ns) >> sh);
            const int32_t m2 = (int32_t)(((int64_t)a * nc) >> sh), m3 = (int32_t)(((int64_t)a * cs.y) >> sh);
            p0 = (K & 1) ? -m0 : m0;                       // a0 - m1 + m2 - m3 + ...
            p1 = (K & 1) ? -m1 : m1;
            p2 = (K & 1) ? -m2 : m2;
            p3 = (K & 1) ? -m3 : m3;
        } else {
            // two's-complement map: the products with -c and -s come from the same 64-bit product,
            //   floor(-P / 2^sh) = -(floor(P / 2^sh) + (P mod 2^sh != 0)),  all modulo 2^32 (sh <= 30)
            const int64_t Pc = (int64_t)a * cs.x, Ps = (int64_t)a * cs.y;
            const uint32_t low = (1u << sh) - 1u;
            const int32_t mc = (int32_t)(Pc >> sh), ms = (int32_t)(Ps >> sh);
            const int32_t uc = mc + ((((uint32_t)Pc) & low) != 0u), us = ms + ((((uint32_t)Ps) & low) != 0u);   // = -m(-c), -m(-s)
            p0 = (K & 1) ? -mc : mc;
            p1 = (K & 1) ? us : -us;
            p2 = (K & 1) ? uc : -uc;
            p3 = (K & 1) ? -ms : ms;
        }
    }
    // rotate the four candidates by q so that image j (quadrant q + j*K) reads a fixed slot
#if BHW_TILE_ROT == 1
    // quadrant bits as opaque 0 / 1 values: one compare per bit, plain selects (left to itself the compiler turns the selects into
    // an indexed read of the four candidates and that into a chain of three compare + select pairs per slot)
    if constexpr (QBITS == 2) {
        static_assert(QBASE == 0, "two-bit rotation takes q as it is");
        uint32_t q0 = q & 1u, q1 = q & 2u;
        asm("" : "+v"(q0), "+v"(q1));
        const bool b0 = q0 != 0u, b1 = q1 != 0u;
        const int32_t r0 = b0 ? p1 : p0, r1 = b0 ? p2 : p1, r2 = b0 ? p3 : p2, r3 = b0 ? p0 : p3;
        sv[0] = b1 ? r2 : r0;
        sv[1] = b1 ? r3 : r1;
        sv[2] = b1 ? r0 : r2;
        sv[3] = b1 ? r1 : r3;
    } else if constexpr (QBITS == 0) {
        const int32_t p[4] = {p0, p1, p2, p3};
#pragma unroll
        for (int i = 0; i < 4; ++i) sv[i] = p[(i + QBASE) & 3];
    } else {
        uint32_t q0 = (q ^ (uint32_t)QBASE) & 1u;
        asm("" : "+v"(q0));
        const bool b0 = q0 != 0u;
        auto pick = [&](int i) -> int32_t { return (i & 3) == 0 ? p0 : (i & 3) == 1 ? p1 : (i & 3) == 2 ? p2 : p3; };
        sv[0] = b0 ? pick(QBASE + 1) : pick(QBASE);
        sv[1] = b0 ? pick(QBASE + 2) : pick(QBASE + 1);
        sv[2] = b0 ? pick(QBASE + 3) : pick(QBASE + 2);
        sv[3] = b0 ? pick(QBASE + 4) : pick(QBASE + 3);
    }
#else
    if constexpr (QBITS == 2) {
        static_assert(QBASE == 0, "two-bit rotation takes q as it is");
        const bool b0 = q & 1u, b1 = q & 2u;
        const int32_t r0 = b0 ? p1 : p0, r1 = b0 ? p2 : p1, r2 = b0 ? p3 : p2, r3 = b0 ? p0 : p3;
        sv[0] = b1 ? r2 : r0;
        sv[1] = b1 ? r3 : r1;
        sv[2] = b1 ? r0 : r2;
        sv[3] = b1 ? r1 : r3;
    } else {
        const int32_t p[4] = {p0, p1, p2, p3};
        if constexpr (QBITS == 0) {                                   // q == QBASE for every lane
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[i] = p[(i + QBASE) & 3];
        } else {                                                      // q is QBASE or QBASE + 1
            const bool b0 = ((q ^ (uint32_t)QBASE) & 1u) != 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) sv[i] = b0 ? p[(i + QBASE + 1) & 3] : p[(i + QBASE) & 3];
        }
    }
#endif
}

// Quadrants harmonic K can be in for a ring lane r in [0, N/8) (image h: r + h * N/8): theta / (N/4) lies in [K*h/2, K*h/2 + K/2).
// first = the lowest quadrant, count = how many (1, 2 or more).  Only the odd harmonics have an h = 1 gather.
__host__ __device__ constexpr int ring_quadrant_first(int K, int h) { return (K * h) >> 1; }
__host__ __device__ constexpr int ring_quadrant_count(int K, int h) { return ((K * h + K - 1) >> 1) - ((K * h) >> 1) + 1; }
__host__ __device__ constexpr int ring_qbits(int K, int h) { return ring_quadrant_count(K, h) == 1 ? 0 : ring_quadrant_count(K, h) == 2 ? 1 : 2; }
__host__ __device__ constexpr int ring_qbase(int K, int h) { return ring_qbits(K, h) == 2 ? 0 : (ring_quadrant_first(K, h) & 3); }

// Wave-uniform quadrant (tile kernel, tiles in which no run crosses a multiple of a quarter turn): the rotation of the four
// candidates becomes a scalar branch to the accumulate code of that quadrant instead of 4 - 8 vector selects per gather:
//   acc[j] -/+= sv[(j*K + OFF + q) & 3],  q a scalar.
// One inline-assembly statement per gather holds the compare, the branches and the adds of every case: written as C++ control
// flow the compiler sinks the adds below the join and leaves a register move per slot in the cases (the selects again).
// QBITS as in tile_harmonic: 1 = q is QBASE or QBASE + 1 (two cases), 2 = any quadrant (four cases).
template <int K, int OFF, int QBASE, int QBITS>
__device__ __forceinline__ void tile_accumulate_uniform(uint32_t q, const int32_t (&sv)[4], int32_t (&acc)[4])
{
    auto S = [&](int j, int Q) -> int32_t { return sv[(j * K + OFF + Q) & 3]; };
    if constexpr (QBITS == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (K & 1) ? acc[j] - S(j, QBASE) : acc[j] + S(j, QBASE);
    } else if constexpr (QBITS == 1) {
#define BHW_UNI2(OP)                                                                                                   \
        asm("s_cmp_eq_u32 %12, %13\n\ts_cbranch_scc0 1f\n\t"                                                          \
            OP " %0, %0, %4\n\t" OP " %1, %1, %5\n\t" OP " %2, %2, %6\n\t" OP " %3, %3, %7\n\ts_branch 2f\n1:\n\t"      \
            OP " %0, %0, %8\n\t" OP " %1, %1, %9\n\t" OP " %2, %2, %10\n\t" OP " %3, %3, %11\n2:"                        \
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                                   \
            : "v"(S(0, QBASE)), "v"(S(1, QBASE)), "v"(S(2, QBASE)), "v"(S(3, QBASE)),                                  \
              "v"(S(0, QBASE + 1)), "v"(S(1, QBASE + 1)), "v"(S(2, QBASE + 1)), "v"(S(3, QBASE + 1)), "s"(q), "n"(QBASE) : "scc")
        if constexpr (K & 1) BHW_UNI2("v_sub_u32"); else BHW_UNI2("v_add_u32");
#undef BHW_UNI2
    } else {
#define BHW_UNI4(OP)                                                                                                   \
        asm("s_cmp_lt_u32 %20, 2\n\ts_cbranch_scc0 2f\n\ts_cmp_eq_u32 %20, 0\n\ts_cbranch_scc0 1f\n\t"                   \
            OP " %0, %0, %4\n\t" OP " %1, %1, %5\n\t" OP " %2, %2, %6\n\t" OP " %3, %3, %7\n\ts_branch 4f\n1:\n\t"      \
            OP " %0, %0, %8\n\t" OP " %1, %1, %9\n\t" OP " %2, %2, %10\n\t" OP " %3, %3, %11\n\ts_branch 4f\n2:\n\t"   \
            "s_cmp_eq_u32 %20, 2\n\ts_cbranch_scc0 3f\n\t"                                                             \
            OP " %0, %0, %12\n\t" OP " %1, %1, %13\n\t" OP " %2, %2, %14\n\t" OP " %3, %3, %15\n\ts_branch 4f\n3:\n\t" \
            OP " %0, %0, %16\n\t" OP " %1, %1, %17\n\t" OP " %2, %2, %18\n\t" OP " %3, %3, %19\n4:"                      \
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                                   \
            : "v"(S(0, 0)), "v"(S(1, 0)), "v"(S(2, 0)), "v"(S(3, 0)), "v"(S(0, 1)), "v"(S(1, 1)), "v"(S(2, 1)), "v"(S(3, 1)), \
              "v"(S(0, 2)), "v"(S(1, 2)), "v"(S(2, 2)), "v"(S(3, 2)), "v"(S(0, 3)), "v"(S(1, 3)), "v"(S(2, 3)), "v"(S(3, 3)), \
              "s"(q) : "scc")
        if constexpr (K & 1) BHW_UNI4("v_sub_u32"); else BHW_UNI4("v_add_u32");
#undef BHW_UNI4
    }
}
// VHDL rule: the W+2-bit sums are carried as 4*hi + lo (Sum32), a term b adds b >> 2 to hi and b & 3 to lo -- the two halves of
// the four candidates once, then the same scalar-branched accumulate for each half.
template <int K, int OFF, int QBASE, int QBITS>
__device__ __forceinline__ void tile_accumulate_uniform(uint32_t q, const int32_t (&sv)[4], Sum32 (&acc)[4])
{
    int32_t svh[4], svl[4], h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { svh[i] = sv[i] >> 2; svl[i] = sv[i] & 3; h[i] = acc[i].hi; l[i] = acc[i].lo; }
    tile_accumulate_uniform<K, OFF, QBASE, QBITS>(q, svh, h);
    tile_accumulate_uniform<K, OFF, QBASE, QBITS>(q, svl, l);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i].hi = h[i]; acc[i].lo = l[i]; }
}

// image j of a lane sits K*j quadrants after image 0; OFF = extra quadrants of this half-period image (even K: K/2)
// UNSIGNED: the candidates come without the harmonic's sign (tile_harmonic FAST): odd harmonics are subtracted
template <int K, int OFF, bool UNSIGNED = false>
__device__ __forceinline__ void tile_accumulate(const int32_t (&sv)[4], int32_t (&acc)[4])
{
    if constexpr (UNSIGNED && (K & 1)) {
        acc[0] -= sv[OFF & 3];
        acc[1] -= sv[(K + OFF) & 3];
        acc[2] -= sv[(2 * K + OFF) & 3];
        acc[3] -= sv[(3 * K + OFF) & 3];
    } else {
        acc[0] += sv[OFF & 3];
        acc[1] += sv[(K + OFF) & 3];
        acc[2] += sv[(2 * K + OFF) & 3];
        acc[3] += sv[(3 * K + OFF) & 3];
    }
}
template <int K, int OFF, bool UNSIGNED = false>
__device__ __forceinline__ void tile_accumulate(const int32_t (&sv)[4], Sum32 (&acc)[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int32_t b = sv[(j * K + OFF) & 3];
        if (K & 1) { acc[j].hi -= b >> 2; acc[j].lo -= b & 3; }
        else       { acc[j].hi += b >> 2; acc[j].lo += b & 3; }
    }
}

// Quadrant fold for whole periods below the tile threshold, plain natural table: lane r in [0, N/4) owns n = r + j*N/4.  The
// harmonic loop is unrolled (NTERMS) so the K-1 gathers issue together, and the arithmetic is that of the tile kernel
// (32-bit forms of both cosine-sum rules).
template <int NTERMS, int MODE>
__global__ __launch_bounds__(kBlock) void k_table_combine_fold_t(BhwCordicCfg cfg, BhwWinCfg win, const void *__restrict__ table,
                                                                  int32_t *__restrict__ out)
{
    using acc_t = typename std::conditional<MODE == 2, Sum32, int32_t>::type;
    const uint32_t lq = cfg.phi_width - 2;
    const uint32_t E = 1u << lq, emask = E - 1u;
    const uint32_t r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= E) return;
    const uint32_t W = cfg.dat_width;
    int2 cs[NTERMS];
#pragma unroll
    for (int k = 1; k < NTERMS; ++k)
        cs[k] = reinterpret_cast<const int2 *>(table)[(((uint32_t)k * r) & emask) >> cfg.z_shr];
    acc_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (MODE == 2) acc[j] = Sum32{win.aa[0] >> 2, win.aa[0] & 3};
        else acc[j] = win.aa[0];
    }
    int32_t sv[4];
#define BHW_FOLD_HARMONIC(K)                                                                       \
    if constexpr (NTERMS > K) {                                                                    \
        tile_harmonic<K, MODE>(cfg, win.aa[K], W, cs[K], ((uint32_t)K * r) >> lq, sv);             \
        tile_accumulate<K, 0>(sv, acc);                                                            \
    }
    BHW_FOLD_HARMONIC(1) BHW_FOLD_HARMONIC(2) BHW_FOLD_HARMONIC(3) BHW_FOLD_HARMONIC(4) BHW_FOLD_HARMONIC(5) BHW_FOLD_HARMONIC(6)
#undef BHW_FOLD_HARMONIC
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int32_t v;
        if constexpr (MODE == 2) v = w32_final<BHW_COMBINE_VHDL>(acc[j], W, NTERMS);
        else v = (int32_t)((uint32_t)acc[j] << (32u - W)) >> (32u - W);            // (win_t)(...) wrap to W bits
        emit(win, out, (uint64_t)r + (uint64_t)j * E, v);
    }
}

// Residual-format records through LDS (BHW_TILE_LDSREC).  On gfx950 a vector load whose lanes do not read consecutive elements
// costs the CU's address path ~16 cycles per wave-instruction whatever its width, a unit-stride one ~4.7
// (profiles/r02_ubench_vmem.txt), and the tile kernel issues 54 such loads per thread: the address path, not the vector ALU, is
// what bounds it.  The records (16 bytes per cell of 2^d >= 128 entries) of the cells a wave's three runs touch are few -- for
// harmonic K a run of 192 lanes spans K * 191 + 1 entries -- so every wave copies them into shared memory once (three load
// instructions) and its 27 gathers read them with ds_read_b128 instead.  Set (K, g) of run b holds rec_slots(K) consecutive cells
// from the cell of the run's first lane; tiles in which a run wraps around the ring or a harmonic's entries wrap around the table
// (about 1.6 % of the 2 913 x 5 thread groups of a 2^26-point window: every wave decides for its own three runs) take the global loads.
#ifndef BHW_TILE_LDSREC
#define BHW_TILE_LDSREC 1
#endif
#ifndef BHW_TILE_NTSTORE
#define BHW_TILE_NTSTORE 0     // nontemporal coefficient stores: +7 % time (0.1497 -> 0.1600 ms), measured again in round 2
#endif
#ifndef BHW_TILE_UNIQ
#define BHW_TILE_UNIQ 1       // scalar quadrants in the tiles that take the LDS path (quad_switch)
#endif
constexpr int kRecSets = 9;                                      // (K, g): (1,0) (1,1) (2) (3,0) (3,1) (4) (5,0) (5,1) (6)
__host__ __device__ constexpr int rec_set_K(int si) { return si < 2 ? 1 : si == 2 ? 2 : si < 5 ? 3 : si == 5 ? 4 : si < 8 ? 5 : 6; }
__host__ __device__ constexpr int rec_set_g(int si) { return (si == 1 || si == 4 || si == 7) ? 1 : 0; }
__host__ __device__ constexpr int rec_set_index(int K, int g) { return K == 1 ? g : K == 2 ? 2 : K == 3 ? 3 + g : K == 4 ? 5 : K == 5 ? 6 + g : 8; }
__host__ __device__ constexpr int rec_slots(int K) { return ((K * (kTileLanes - 1) + 127) >> 7) + 1; }   // cells of >= 128 entries
__host__ __device__ constexpr int rec_set_base(int si) { int s = 0; for (int i = 0; i < si; ++i) s += rec_slots(rec_set_K(i)); return s; }
constexpr int kRecPerRun = rec_set_base(kRecSets);               // 57 for 192-lane runs
static_assert(kRecPerRun <= 64, "slot -> (K, g, j) table");
struct RecMeta { uint8_t v[64]; };
constexpr RecMeta make_rec_meta()
{
    RecMeta m{};
    int s = 0;
    for (int si = 0; si < kRecSets; ++si)
        for (int j = 0; j < rec_slots(rec_set_K(si)); ++j) m.v[s++] = (uint8_t)(rec_set_K(si) | (rec_set_g(si) << 3) | (j << 4));
    return m;
}
__device__ const RecMeta kRecMeta = make_rec_meta();

// Two orders of the tile kernel's work, both measured (profiles/r02_ab_tile_memory_path.txt):
//   harmonic-major (default for the HLS cosine-sum): every harmonic over the thread's three runs, 24 stores at the end;
//   run-major (BHW_TILE_RUNMAJOR; default for the VHDL cosine-sum): one run at a time -- its six harmonics, then its eight
//     stores -- with the nine residual words of the next run requested before the current one is worked on (BHW_TILE_PREFETCH,
//     one register each).  8 sums live instead of 24: the VHDL-rule instance drops from 93 to 66 registers (5 -> 7 waves per
//     SIMD) and runs 3.7 % faster; the HLS-rule instance (64 registers either way) ties with harmonic-major (0.1479 against
//     0.1470 ms) and, without the prefetch, loses 10 % (fewer independent gathers in flight per harmonic).
// Requesting all 27 residual words of a thread up front in harmonic-major order is slower (0.1562 ms: registers), as a
// 320-thread / 64-lane tile shape with five waves per SIMD it ties (0.1467 ms): the kernel's remaining stall is not the latency
// of its own loads.  (Two gathers per register through global_load_ubyte_d16 / _d16_hi is not available: with SRAM ECC a d16
// load clears the other half.)
#ifndef BHW_TILE_RUNMAJOR
#define BHW_TILE_RUNMAJOR (NB >= 15 && (MODE == 2 || !FAST))      // also the 64-bit-product form (caller-scaled weights): no registers to spare otherwise
#endif
#ifndef BHW_TILE_PREFETCH
#define BHW_TILE_PREFETCH (NB >= 15 && (MODE == 2 || !FAST))
#endif
__host__ __device__ constexpr int gather_order(int K, int b, int g)     // consumption order of the tile kernel's gathers, NR = 3
{
    return K == 1 ? b * 2 + g : K == 2 ? 6 + b : K == 3 ? 9 + b * 2 + g : K == 4 ? 15 + b : K == 5 ? 18 + b * 2 + g : 24 + b;
}

// Lane r in [0, E/2) owns the eight coefficients n = r + h*E/2 + j*E (h = 0,1; j = 0..3).  For even k the
// two h-images share one gather (k*E/2 is a whole number of quadrants); for odd k the second image reads
// entry t + E/2, another dense span of the same tile.
// MASKED: the launch produces only some of the eight images (a contiguous index range of the window that is a whole number of
// eighths -- one device's contiguous shard of a window split over 2, 4 or 8): a gather is skipped when no wanted image reads it
// (odd harmonics: the h = 0 / h = 1 gathers), sums of an unwanted half are not formed, unwanted images are not stored.
template <int NB, int MODE, int FMT, bool FAST = false, bool MASKED = false>
__global__ __launch_bounds__(kTileThreads) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? 4 : BHW_TILE_WAVES))) void k_table_combine_tile(BhwCordicCfg cfg, BhwWinCfg win, BhwTilePlan tp,
                                                                      const void *__restrict__ table, int32_t *__restrict__ out)
{
    using acc_t = typename std::conditional<MODE == 2, Sum32, int32_t>::type;
    const uint32_t lq = cfg.phi_width - 2;
    const uint32_t E = 1u << lq, emask = E - 1u, H = E >> 1, hmask = H - 1u;
    const uint32_t W = cfg.dat_width;
    // thread group `part` of the workgroup takes runs [part*NR, part*NR + NR) of the tile (registers: NR*8 sums)
    constexpr int kParts = (NB >= 15) ? kTileThreads / kTileLanes : 1;
    constexpr int NR = (NB + kParts - 1) / kParts;
    constexpr int kLanes = kTileThreads / kParts;
    // Lane -> r inside a run.  A run is kLanes consecutive r starting at (tile base + offs[b]), an arbitrary address, so a
    // plain "lane i takes start + i" makes every wave's 256-byte output chunk straddle three cache lines (two partial).
    // Rotating the lanes by the start's offset inside a 64-element block gives every wave an aligned block instead; only
    // wave 0 is split (head of the first block + tail of the last).  Same set of r, same gathers, full-line stores:
    // -1.25 % on the whole call (profiles/r01_ab_inproc.txt).
    // Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share an L2).  BHW_TILE_XCD = 1 renumbers the tiles so
    // that each XCD sweeps a contiguous eighth of the ring (neighbouring tiles share table lines at their run boundaries and the
    // 16-byte records of the residual format).
#ifndef BHW_TILE_XCD
#define BHW_TILE_XCD 1
#endif
    uint32_t tile_of_block = blockIdx.x;
    if (BHW_TILE_XCD) {
        const uint32_t per = gridDim.x >> 3, main = per << 3;       // tiles per XCD in the evenly divisible part
        if (blockIdx.x < main) tile_of_block = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    }
    const uint32_t part = __builtin_amdgcn_readfirstlane(threadIdx.x / kLanes);   // wave-uniform: kLanes is a multiple of 64
    const uint32_t lane_in_part = threadIdx.x % kLanes;
    constexpr bool kLdsRec = BHW_TILE_LDSREC && (FMT == 2 || FMT == 3) && NB >= 15 && kLanes == kTileLanes;
    uint32_t rec_meta = 0;                                           // slot -> (K, g, j) of the record staging below, fetched first
    if constexpr (kLdsRec) rec_meta = kRecMeta.v[threadIdx.x & 63u];
    uint32_t rr[NR], starts[NR];
#pragma unroll
    for (int b = 0; b < NR; ++b) {
        const uint32_t start = ((tile_of_block + tp.tile0) * kLanes + tp.offs[part * NR + b]) & hmask;   // scalar; offs padded with copies of the last run
        starts[b] = start;
        rr[b] = (start + (lane_in_part + kLanes - (start & 63u)) % kLanes) & hmask;
    }
    // records of the cells this wave's runs touch, staged in shared memory (see BHW_TILE_LDSREC above)
    constexpr int kWavesWg = kTileThreads / 64;
    __shared__ int4 rec_s[kLdsRec ? kWavesWg * NR * kRecPerRun : 1];
    uint32_t rbias[kRecSets][NR];                                    // scalar: byte offset of "cell 0" of set si, run b in rec_s
    uint32_t qpack[NR];                                              // scalar: quadrant of set si, run b in bits 2 si, 2 si + 1
    bool wraps = false;
    uint32_t cls[NR];                                                // residue class of r in the split layout (odd harmonics)
#pragma unroll
    for (int b = 0; b < NR; ++b) cls[b] = split_class<FMT>(rr[b], lq);
    constexpr bool kPrefetch = BHW_TILE_PREFETCH && kLdsRec && NR == 3;
    uint32_t land[kPrefetch ? 27 : 1];                               // residual words, one per gather (gather_order)
    // MASKED: which half-period images (h = 0: even image numbers, h = 1: odd) this launch wants at all
    const bool want0 = !MASKED || (tp.img_mask & 0x55u) != 0u, want1 = !MASKED || (tp.img_mask & 0xAAu) != 0u;
    auto issue_runs = [&](auto run_tag) {
        constexpr int B0 = decltype(run_tag)::value < 0 ? 0 : decltype(run_tag)::value, B1 = decltype(run_tag)::value < 0 ? NR : B0 + 1;
#define BHW_TILE_ISSUE(K)                                                                                \
        if (win.n_terms > K) {                                                                           \
            constexpr int NG = (K & 1) ? 2 : 1;                                                          \
            _Pragma("unroll") for (int b = B0; b < B1; ++b) {                                            \
                _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                         \
                    if (MASKED && NG == 2 && !(g ? want1 : want0)) continue;                             \
                    if constexpr (FMT == 2 || FMT == 3) {                                               \
                        const uint32_t rg = rr[b] + (uint32_t)g * H;                                     \
                        const uint32_t boff = resid_offset<FMT, K>(rg, (uint32_t)K * rg, cls[b], lq, emask); \
                        land[gather_order(K, b, g)] = FMT == 3 ? (uint32_t)ld_off<uint8_t>(table, boff) : (uint32_t)ld_off<uint16_t>(table, boff); \
                    }                                                                                    \
                }                                                                                        \
            }                                                                                            \
        }
        BHW_TILE_ISSUE(1) BHW_TILE_ISSUE(2) BHW_TILE_ISSUE(3) BHW_TILE_ISSUE(4) BHW_TILE_ISSUE(5) BHW_TILE_ISSUE(6)
#undef BHW_TILE_ISSUE
        __builtin_amdgcn_sched_barrier(0);                           // the loads stay here (left alone the scheduler sinks them to their uses)
    };
    if constexpr (kLdsRec) {
        const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const uint32_t d = fmt_cell_log(cfg.tab_dlog);
#pragma unroll
        for (int b = 0; b < NR; ++b) {
            wraps |= starts[b] + (uint32_t)kLanes > H;
            qpack[b] = 0u;
#pragma unroll
            for (int si = 0; si < kRecSets; ++si) {
                const uint32_t K = rec_set_K(si);
                const uint32_t th0 = K * (starts[b] + (uint32_t)rec_set_g(si) * H);
                const uint32_t u0 = th0 & emask;
                qpack[b] |= ((th0 >> lq) & 3u) << (2 * si);
                wraps |= u0 + K * (uint32_t)(kLanes - 1) > emask;
                rbias[si][b] = (((wave * NR + b) * kRecPerRun + (uint32_t)rec_set_base(si)) << 4) - ((th0 >> d) << 4);   // cell of the unmasked angle (resid_value)
            }
        }
        if constexpr (kPrefetch) {
            if (!wraps) {
                if constexpr (BHW_TILE_RUNMAJOR) issue_runs(std::integral_constant<int, 0>{});   // run 0 now, run b + 1 while run b is worked on
                else issue_runs(std::integral_constant<int, -1>{});
            }
        }
        if (!wraps) {
            // lane s < kRecPerRun copies slot s of each of the wave's NR runs: the NR loads are in flight together
            const uint32_t s = threadIdx.x & 63u;
            if (s < (uint32_t)kRecPerRun) {
                const uint32_t meta = rec_meta;
                int4 rec[NR];
#pragma unroll
                for (int b = 0; b < NR; ++b) {
                    const uint32_t u0 = ((meta & 7u) * (starts[b] + ((meta >> 3) & 1u) * H)) & emask;
                    const uint32_t cell = ((u0 >> d) + (meta >> 4)) & ((E >> d) - 1u);
                    rec[b] = ld_off<int4>(cfg.tab_coarse, cell << 4);
                }
#pragma unroll
                for (int b = 0; b < NR; ++b) rec_s[(wave * NR + b) * kRecPerRun + s] = rec[b];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                         // the wave's own LDS writes are ordered before its reads
        }
    }
#ifndef BHW_TILE_VGPR_CONSTS
#define BHW_TILE_VGPR_CONSTS 0      // measured: 1 is 1 % slower (profiles/r02_ab_tile_kernel_steps.txt) -- register pressure outweighs the cheaper operands
#endif
#ifdef BHW_TILE_FIXD      // timing experiment: the cell size as a compile-time constant (immediate shifts and mask)
    ResidK rk{BHW_TILE_FIXD, (1u << BHW_TILE_FIXD) - 1u};
#else
    ResidK rk{fmt_cell_log(cfg.tab_dlog), (1u << fmt_cell_log(cfg.tab_dlog)) - 1u};
#endif
    uint32_t emask_v = emask, lq_v = lq;                            // per-gather shift / mask operands: VGPR copies (see ResidK)
#if BHW_TILE_VGPR_CONSTS
    asm volatile("" : "+v"(rk.d), "+v"(rk.fmask), "+v"(emask_v), "+v"(lq_v));
#endif
    // Run-major order (BHW_TILE_RUNMAJOR): one run at a time -- its six harmonics, then its eight stores -- instead of every
    // harmonic over the three runs and 24 stores at the end: 8 sums live instead of 24, and the stores of a wave are spread
    // over its life.  `run_tag` selects the runs a pass covers: -1 all (harmonic-major), else that one.
    acc_t acc[NR][2][4];
    auto init_acc = [&](auto run_tag) {
        constexpr int B0 = decltype(run_tag)::value < 0 ? 0 : decltype(run_tag)::value, B1 = decltype(run_tag)::value < 0 ? NR : B0 + 1;
#pragma unroll
        for (int b = B0; b < B1; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (MODE == 2) acc[b][h][j] = Sum32{win.aa[0] >> 2, win.aa[0] & 3};
                    else acc[b][h][j] = win.aa[0];
                }
    };

    auto harmonics = [&](auto lds_tag, auto run_tag) {
    constexpr bool LDS = decltype(lds_tag)::value;
    constexpr int B0 = decltype(run_tag)::value < 0 ? 0 : decltype(run_tag)::value, B1 = decltype(run_tag)::value < 0 ? NR : B0 + 1;
    const char *lrec = reinterpret_cast<const char *>(rec_s);
#define BHW_TILE_HARMONIC(K)                                                                             \
    if (win.n_terms > K) {                                                                               \
        constexpr int NG = (K & 1) ? 2 : 1;                                                              \
        constexpr int KC = (K % 4 == 0) ? 4 : (K % 2 == 0) ? 2 : 0;   /* split layout only exists at z_shr == 0 */ \
        int2 cs[NR][NG];                                                                                 \
        _Pragma("unroll") for (int b = B0; b < B1; ++b) {                                                 \
            _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                             \
                if (MASKED && NG == 2 && !(g ? want1 : want0)) continue;                                 \
                const uint32_t rg = rr[b] + (uint32_t)g * H;                                             \
                const uint32_t theta = (uint32_t)K * rg;                                                 \
                if constexpr (NB > 1 && (FMT == 2 || FMT == 3)) {                                        \
                    uint32_t bias = rbias[rec_set_index(K, g)][b];                                       \
                    if constexpr (LDS) asm("" : "+s"(bias));    /* one scalar: the record address is shift, shift-add */ \
                    uint32_t e;                                                                          \
                    if constexpr (kPrefetch && LDS) e = land[gather_order(K, b, g)];                     \
                    else {                                                                               \
                        const uint32_t boff = resid_offset<FMT, K>(rg, theta, cls[b], lq, emask_v);     \
                        e = FMT == 3 ? (uint32_t)ld_off<uint8_t>(table, boff) : (uint32_t)ld_off<uint16_t>(table, boff); \
                    }                                                                                    \
                    cs[b][g] = resid_value<FMT, LDS>(cfg, theta, emask_v, rk, lrec, bias, e);            \
                } else if constexpr (NB > 1 && (K & 1)) cs[b][g] = tab_load_class<FMT>(cfg, table, theta & emask_v, cls[b]); \
                else if constexpr (NB > 1) {                                                             \
                    const uint32_t u = theta & emask_v;                                                  \
                    cs[b][g] = tab_fetch<FMT>(cfg, table, u, tab_index<KC, 1>(u, lq, 1u));              \
                } else cs[b][g] = tab_load<KC, FMT, -1>(cfg, table, (theta & emask) >> cfg.z_shr, lq - cfg.z_shr); \
            }                                                                                            \
        }                                                                                                \
        _Pragma("unroll") for (int b = B0; b < B1; ++b) {                                                 \
            int32_t sv[4];                                                                               \
            /* only quadrant bits 0,1 of theta >> lq are used */                                         \
            const int32_t aK = FAST ? (int32_t)((uint32_t)win.aa[K] << (34u - W)) : win.aa[K];          \
            if constexpr (LDS && BHW_TILE_UNIQ && (FAST || MODE == 2)) {                                 \
                /* no run of this tile crosses a quarter turn: the quadrants are scalars (qpack) */      \
                const uint32_t q0 = (qpack[b] >> (2 * rec_set_index(K, 0))) & 3u;                        \
                if (NG == 1 || want0) {                                                                  \
                    tile_harmonic<K, MODE, 0, 0, FAST>(cfg, aK, W, cs[b][0], 0u, sv);                    \
                    if (want0) tile_accumulate_uniform<K, 0, ring_qbase(K, 0), ring_qbits(K, 0)>(q0, sv, acc[b][0]); \
                }                                                                                        \
                if constexpr (NG == 1) {                                                                 \
                    if (want1) tile_accumulate_uniform<K, K / 2, ring_qbase(K, 0), ring_qbits(K, 0)>(q0, sv, acc[b][1]); \
                } else if (want1) {                                                                      \
                    tile_harmonic<K, MODE, 0, 0, FAST>(cfg, aK, W, cs[b][1], 0u, sv);                    \
                    const uint32_t q1 = (qpack[b] >> (2 * rec_set_index(K, 1))) & 3u;                    \
                    tile_accumulate_uniform<K, 0, ring_qbase(K, 1), ring_qbits(K, 1)>(q1, sv, acc[b][1]); \
                }                                                                                        \
                continue;                                                                                \
            }                                                                                            \
            if (NG == 1 || want0) {                                                                      \
                tile_harmonic<K, MODE, ring_qbase(K, 0), ring_qbits(K, 0), FAST>(cfg, aK, W, cs[b][0], ((uint32_t)K * rr[b]) >> lq_v, sv); \
                if (want0) tile_accumulate<K, 0, FAST>(sv, acc[b][0]);                                         \
            }                                                                                            \
            if constexpr (NG == 2) {                                                                     \
                if (want1) {                                                                             \
                    tile_harmonic<K, MODE, ring_qbase(K, 1), ring_qbits(K, 1), FAST>(cfg, aK, W, cs[b][NG - 1], ((uint32_t)K * (rr[b] + H)) >> lq_v, sv); \
                    tile_accumulate<K, 0, FAST>(sv, acc[b][1]);                                                \
                }                                                                                        \
            } else {                                                                                     \
                /* even K: the second half-period image reads the same entry K/2 quadrants further on */ \
                if (want1) tile_accumulate<K, K / 2, FAST>(sv, acc[b][1]);                                     \
            }                                                                                            \
        }                                                                                                \
    }
    BHW_TILE_HARMONIC(1)
    BHW_TILE_HARMONIC(2)
    BHW_TILE_HARMONIC(3)
    BHW_TILE_HARMONIC(4)
    BHW_TILE_HARMONIC(5)
    BHW_TILE_HARMONIC(6)
#undef BHW_TILE_HARMONIC
    };
    auto run_harmonics = [&](auto run_tag) {
        if constexpr (kLdsRec) {
            if (wraps) harmonics(std::false_type{}, run_tag);        // wave-uniform
            else harmonics(std::true_type{}, run_tag);
        } else harmonics(std::false_type{}, run_tag);
    };

    auto final_value = [&](int b, int h, int j) -> int32_t {
        if constexpr (MODE == 2) return w32_final<BHW_COMBINE_VHDL>(acc[b][h][j], W, win.n_terms);
        else return (int32_t)((uint32_t)acc[b][h][j] << (32u - W)) >> (32u - W);   // (win_t)(...) wrap to W bits
    };
    auto store_runs = [&](auto run_tag) {
    constexpr int B0 = decltype(run_tag)::value < 0 ? 0 : decltype(run_tag)::value, B1 = decltype(run_tag)::value < 0 ? NR : B0 + 1;
    if (win.apply_x == nullptr) {                                    // wave-uniform
        // image (h, j) starts at out + h*H + j*E, a scalar address the lane adds its 32-bit byte offset r * 4 to (saddr stores;
        // the empty asm keeps the compiler from folding the image offset back into a 64-bit vector add per store)
        auto store_all = [&](auto full_width) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MASKED && !((tp.img_mask >> (2 * j + h)) & 1u)) continue;
                    uint64_t img_off = (uint64_t)h * H + (uint64_t)j * E;
                    if (MASKED) img_off = (img_off - tp.n0mod) & (4ull * E - 1ull);      // position of the image in the caller's range
                    asm volatile("" : "+s"(img_off));
                    int32_t *img = out + img_off;
#pragma unroll
                    for (int b = B0; b < B1; ++b) {
                        int32_t v;
                        if constexpr (MODE != 2 && decltype(full_width)::value) v = acc_value(acc[b][h][j]);   // W == 32: nothing to wrap
                        else v = final_value(b, h, j);
#if BHW_TILE_DBG & 4
                        if (v == 0x12345)
#endif
#if BHW_TILE_NTSTORE
                        __builtin_nontemporal_store(v, reinterpret_cast<int32_t *>(reinterpret_cast<char *>(img) + (rr[b] << 2)));
#else
                        *reinterpret_cast<int32_t *>(reinterpret_cast<char *>(img) + (rr[b] << 2)) = v;
#endif
                    }
                }
        };
        if (W == 32u) store_all(std::true_type{});
        else store_all(std::false_type{});
    } else {
        // Fused apply (emit()): one run at a time, its eight x samples fetched together before they are used
#pragma unroll
        for (int b = B0; b < B1; ++b) {
            int32_t xv[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[h][j] = win.apply_x[(uint64_t)(rr[b] + (uint32_t)h * H + (uint32_t)j * E)];
            // all eight in registers before the first store (otherwise each load is sunk next to its use: load, wait, store, eight times)
            asm volatile("" : "+v"(xv[0][0]), "+v"(xv[0][1]), "+v"(xv[0][2]), "+v"(xv[0][3]),
                              "+v"(xv[1][0]), "+v"(xv[1][1]), "+v"(xv[1][2]), "+v"(xv[1][3]));
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    out[(uint64_t)(rr[b] + (uint32_t)h * H + (uint32_t)j * E)] =
                        (int32_t)(((int64_t)xv[h][j] * (int64_t)final_value(b, h, j)) >> win.apply_shift);
        }
    }
    };
    if constexpr (BHW_TILE_RUNMAJOR && NR == 3) {
        if constexpr (kPrefetch) { if (!wraps) issue_runs(std::integral_constant<int, 1>{}); }
        init_acc(std::integral_constant<int, 0>{}); run_harmonics(std::integral_constant<int, 0>{}); store_runs(std::integral_constant<int, 0>{});
        if constexpr (kPrefetch) { if (!wraps) issue_runs(std::integral_constant<int, 2>{}); }
        init_acc(std::integral_constant<int, 1>{}); run_harmonics(std::integral_constant<int, 1>{}); store_runs(std::integral_constant<int, 1>{});
        init_acc(std::integral_constant<int, 2>{}); run_harmonics(std::integral_constant<int, 2>{}); store_runs(std::integral_constant<int, 2>{});
    } else {
        init_acc(std::integral_constant<int, -1>{}); run_harmonics(std::integral_constant<int, -1>{}); store_runs(std::integral_constant<int, -1>{});
    }
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
constexpr int kFoldRunsMax = 32;
constexpr int kFoldBlock = 256;

struct BhwFoldPlan {
    uint32_t lut[34];                        // rescaled ROM as 32-bit words (quarter circle <= 2^32); [32], [33] = 0: the loop reads one ahead
    int64_t  x0;
    uint32_t n_iter, z_shr, z_shl, out_shr;
    uint32_t n_runs, phi_width, dat_width, ones_neg;
    uint32_t fast_mul, pad0;                 // 1: every harmonic weight below 2^(W-3): one-instruction products (tile_harmonic FAST)
    uint32_t r0[kFoldRunsMax];               // first ring index of each run
    uint32_t r_end[kFoldRunsMax];            // one past its last
    uint32_t wg_first[kFoldRunsMax + 1];     // first workgroup of each run; [n_runs] = grid size
};

__host__ __device__ constexpr int fold_chains(int n_terms)      // first-quadrant chains per ring lane
{
    return n_terms == 2 ? 2 : n_terms == 3 ? 3 : n_terms == 4 ? 5 : n_terms == 5 ? 6 : 9;
}

// Remaining rotations of one chain, k0 <= k < n_iter, as a rolled loop on a scalar counter: k0 and n_iter are wave-uniform, the
// shift amount of v_alignbit_b32 and the ROM word are scalar operands, so a rotation is the 8 vector instructions of the
// unrolled rot_step, one v_readlane_b32 and ~3 scalar ones.  (Unrolled with a
// scalar guard per rotation the kernel carried ~6 scalar instructions for every one of the 31 possible rotations of every
// chain, executed or not -- more scalar than vector work, and the scalar unit is shared by the CU's four SIMDs.)
__device__ __forceinline__ void rot_step_dyn(int64_t &x, int64_t &y, int32_t &z, int k, uint32_t lutk, bool mad24)
{
    const int32_t m = z >> 31;
    const int32_t sg = m | 1;
    const int32_t nsg = -sg;
    int32_t ys = (int32_t)__builtin_amdgcn_alignbit((uint32_t)((uint64_t)y >> 32), (uint32_t)y, (uint32_t)k);   // lo32(y >> k), 1 <= k <= 31
    int32_t xs = (int32_t)__builtin_amdgcn_alignbit((uint32_t)((uint64_t)x >> 32), (uint32_t)x, (uint32_t)k);
    asm volatile("" : "+v"(ys), "+v"(xs));
    x += (int64_t)nsg * (int64_t)ys;
    y += (int64_t)sg * (int64_t)xs;
    if (mad24) z += __mul24(nsg, (int32_t)lutk);
    else       z = (int32_t)((uint32_t)z - lutk + ((2u * lutk) & (uint32_t)m));
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

template <int NTERMS, int MODE, bool LOCKSTEP>
__global__ __launch_bounds__(kFoldBlock) void k_fold_direct(BhwWinCfg win, BhwFoldPlan plan, int32_t *__restrict__ out)
{
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
    uint32_t run = 0;                                                     // scalar search: at most kFoldRunsMax runs
    while (run + 1u < plan.n_runs && blockIdx.x >= plan.wg_first[run + 1u]) ++run;
    const uint32_t wg_r0 = plan.r0[run] + (blockIdx.x - plan.wg_first[run]) * blockDim.x;
    const uint32_t r_end = plan.r_end[run];
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t z_shr = plan.z_shr, z_shl = plan.z_shl, out_shr = plan.out_shr;

    // ---- phase 1: shared rotation prefix of every (wave, chain) ----
    // chain slot c -> harmonic K and half-period image: (1,0) (1,1) (2) (3,0) (3,1) (4) (5,0) (5,1) (6)
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

    // ---- phase 2: one lane per r ----
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t r = wg_r0 + threadIdx.x;
    const uint32_t lutv = plan.lut[threadIdx.x & 31u];                    // lane k (and k + 32) holds lut[k]
    const bool fast = MODE != 2 && plan.fast_mul != 0u;
    acc_t acc[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (MODE == 2) acc[h][j] = Sum32{win.aa[0] >> 2, win.aa[0] & 3};
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
    auto chain = [&](const uint32_t slot, const uint32_t K, const uint32_t hodd) -> int2 {
        if constexpr (LOCKSTEP) return cs[slot];
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
        tile_harmonic<K, MODE, ring_qbase(K, HH), ring_qbits(K, HH), MODE != 2>(cfg, (int32_t)((uint32_t)win.aa[K] << (34u - W)), W, CS, \
                                                                                ((uint32_t)K * (r + (uint32_t)HH * H)) >> lq, sv); \
        tile_accumulate<K, OFF, MODE != 2>(sv, ACC);                                                 \
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
                tile_harmonic<K, MODE, ring_qbase(K, 0), ring_qbits(K, 0), MODE != 2>(cfg, (int32_t)((uint32_t)win.aa[K] << (34u - W)), W, cs0, ((uint32_t)K * r) >> lq, sv); \
                tile_accumulate<K, 0, MODE != 2>(sv, acc[0]);                                        \
                tile_accumulate<K, K / 2, MODE != 2>(sv, acc[1]);                                    \
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
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int32_t v;
            if constexpr (MODE == 2) v = w32_final<BHW_COMBINE_VHDL>(acc[h][j], W, NTERMS);
            else v = (int32_t)((uint32_t)acc[h][j] << (32u - W)) >> (32u - W);         // (win_t)(...) wrap to W bits
            emit(win, out, (uint64_t)(r + (uint32_t)h * H) + (uint64_t)j * E, v);
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

template <int NTERMS, int MODE>
__global__ __launch_bounds__(256) void k_fold_split(BhwWinCfg win, BhwFoldPlan plan, int32_t *__restrict__ out)
{
    using acc_t = typename std::conditional<MODE == 2, Sum32, int32_t>::type;
    constexpr int NCH = fold_chains(NTERMS);
    constexpr int MAXC = (NCH + 3) / 4;                                   // chains per wave
    __shared__ int2 cs_s[NCH][64];
    BhwCordicCfg cfg;                                                     // tile_harmonic() reads ones_neg only
    cfg.ones_neg = plan.ones_neg;
    const uint32_t lq = plan.phi_width - 2;
    const uint32_t E = 1u << lq, emask = E - 1u, H = E >> 1;
    const uint32_t W = plan.dat_width;
    const int n_iter = (int)plan.n_iter;
    uint32_t run = 0;
    while (run + 1u < plan.n_runs && blockIdx.x >= plan.wg_first[run + 1u]) ++run;
    const uint32_t wg_r0 = plan.r0[run] + (blockIdx.x - plan.wg_first[run]) * 64u;      // 64 ring lanes per workgroup
    const uint32_t r_end = plan.r_end[run];
    const uint32_t z_shr = plan.z_shr, z_shl = plan.z_shl, out_shr = plan.out_shr;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t r = wg_r0 + lane;
    const int32_t nlutv = -(int32_t)plan.lut[lane & 31u];                 // lane k holds -lut[k]

    auto chain_K = [](uint32_t c) { return 2u * (c / 3u) + 1u + (c % 3u == 2u ? 1u : 0u); };
    auto chain_h = [](uint32_t c) { return (c % 3u == 1u) ? 1u : 0u; };

    // ---- phase 1: lane i < MAXC runs the shared prefix of this wave's chain i (chain index wave + 4 i) ----
    int64_t px = plan.x0, py = plan.x0;
    uint32_t pdz = 0u;
    int pk = 1;
    {
        const uint32_t ci = lane < (uint32_t)MAXC ? lane : 0u;
        uint32_t c = wave + 4u * ci;
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
        const uint32_t c = wave + 4u * (uint32_t)i;
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
        const uint32_t c = wave + 4u * (uint32_t)i;
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
            if constexpr (MODE == 2) acc[j] = Sum32{win.aa[0] >> 2, win.aa[0] & 3};
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
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int32_t v;
            if constexpr (MODE == 2) v = w32_final<BHW_COMBINE_VHDL>(acc[j], W, NTERMS);
            else v = (int32_t)((uint32_t)acc[j] << (32u - W)) >> (32u - W);
            emit(win, out, (uint64_t)(r + (uint32_t)HH * H) + (uint64_t)j * E, v);
        }
    };
    if (wave == 0u) combine(std::integral_constant<int, 0>{});
    else combine(std::integral_constant<int, 1>{});
}

// ---------------------------------------------------------------------------------------
// Run-length kernel: whole periods of configurations that drop phase bits (z_shr > 0).
//
// Models A and C discard the low PW - W phase bits before the rotation (cpp/cordic_sincos.cpp:31-36, src/cordic_dds.vhd:159-162),
// so only 2^(W-2) first-quadrant results exist however long the window is (2^14 pairs = 128 KiB at 16 bits: L2-resident) and
// harmonic K of consecutive coefficients reads the SAME table entry for 2^z_shr / K of them in a row.  The coefficient stream
// is therefore piecewise constant between "breakpoints" where some harmonic steps to its next entry, and a lane that owns a
// run of consecutive ring positions does full work (gather, products, quadrant rotation) only twice per harmonic -- at the
// two ends of its run -- and O(1) work per breakpoint in between:
//   thread   = kRlRun = 16 consecutive ring lanes r (times the eight quadrant / half-period images);
//   setup    : per chain (harmonic K, image h) the rotated term candidates at r_first and r_last and the position i_K of the
//              one breakpoint in between (at most one while (NTERMS-1) * 16 <= 2^z_shr); delta_K = terms(last) - terms(first);
//   sweep    : i = 0 .. 15: acc += delta_K where i == i_K (a wave-wide vote skips harmonics nobody steps at this i);
//   output   : 16 values per image go through a swizzled LDS tile so that every store instruction writes 1 KiB of
//              consecutive addresses (16 bytes per lane).
// About 10 VALU instructions per coefficient instead of ~35, so these configurations run at the store rate.
// All sums are plain int32: HLS rule modulo 2^32 as in the tile kernel; VHDL rule needs W + 2 <= 30 bits (z_shr > 0 means
// W < PW <= 30 anyway; the launcher checks).
// ---------------------------------------------------------------------------------------
#ifndef BHW_RL_RUN
#define BHW_RL_RUN 16
#endif
constexpr int kRlRun = BHW_RL_RUN;               // 8 or 16 (the swizzles below assume a multiple of 4 granule rows)
#ifndef BHW_RL_BLOCK
#define BHW_RL_BLOCK 64
#endif
#ifndef BHW_RL_DIRECT_STORE
#define BHW_RL_DIRECT_STORE 0      // 1: every lane stores its own 16-byte granules (64-byte stride across lanes), no LDS tile
#endif
constexpr int kRlBlock = BHW_RL_BLOCK;

// LDS tile of one (wave, image): 1024 values as 256 granules of 16 bytes, granule index XOR-swizzled inside rows of eight
// so that both the producer pattern (granule 4*lane + c) and the consumer pattern (granule 64*s + lane) are conflict-free.
__device__ __forceinline__ uint32_t rl_swizzle(uint32_t g) { return (g & ~7u) | ((g ^ (g >> 3)) & 7u); }
// the same for 8-byte granules (rows of sixteen): dat_width <= 16 keeps the tile as int16, half the LDS, twice the waves per CU
__device__ __forceinline__ uint32_t rl_swizzle16(uint32_t g) { return (g & ~15u) | ((g ^ (g >> 4)) & 15u); }

template <int NTERMS, int MODE, bool NARROW>
__global__ __launch_bounds__(kRlBlock) void k_runlength_window(BhwCordicCfg cfg, BhwWinCfg win, const int2 *__restrict__ table,
                                                              int32_t *__restrict__ out)
{
#if !BHW_RL_DIRECT_STORE
    using gran_t = typename std::conditional<NARROW, uint2, int4>::type;  // four coefficients: 4 x int16 or 4 x int32
    __shared__ gran_t tile[kRlBlock / 64][4][16 * kRlRun];                        // [wave][image j][granule]: 8 / 16 KiB per wave
#endif
    const uint32_t lq = cfg.phi_width - 2;
    const uint32_t E = 1u << lq, emask = E - 1u, H = E >> 1;
    const uint32_t W = cfg.dat_width, zs = cfg.z_shr, zmask = (1u << zs) - 1u;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t r_wave = (blockIdx.x * kRlBlock + (wave << 6)) * kRlRun;     // first ring lane of this wave's 1024
    const uint32_t r0 = r_wave + lane * kRlRun;
    constexpr uint32_t R1 = kRlRun - 1;

    auto finish = [&](int32_t acc) -> int32_t {
        if constexpr (MODE == 2) {
            if constexpr (NTERMS == 2) { const int32_t S = wrap32(acc, W + 1); return wrap32((S >> 1) + (S & 1), W); }     // hamming_win.vhd:214-228
            else { const int32_t S = wrap32(acc, W + 2); return wrap32((S >> 2) + ((S >> 1) & 1), W); }                     // bh_win_7term.vhd:409-435
        } else return wrap32(acc, W);                                                                                      // win_function.cpp:375
    };

#pragma unroll 1
    for (uint32_t h = 0; h < 2; ++h) {
        int32_t acc[4] = {win.aa[0], win.aa[0], win.aa[0], win.aa[0]};
        int32_t dlt[NTERMS][4];
        uint32_t brk[NTERMS];
#define BHW_RL_SETUP(K)                                                                                                  \
        if constexpr (NTERMS > K) {                                                                                      \
            const uint32_t hodd = (K & 1) ? h : 0u;                                                                      \
            const uint32_t tha = (uint32_t)K * (r0 + hodd * H), thb = tha + (uint32_t)K * R1;   /* phases of the run's two ends */ \
            const uint32_t ta = tha & emask, tb = thb & emask;                                                           \
            const int2 csa = table[ta >> zs], csb = table[tb >> zs];                                                     \
            int32_t sva[4], svb[4];                                                                                      \
            tile_harmonic<K, MODE>(cfg, win.aa[K], W, csa, tha >> lq, sva);                                              \
            tile_harmonic<K, MODE>(cfg, win.aa[K], W, csb, thb >> lq, svb);                                              \
            /* first i at which the entry index (t >> z_shr, quadrant included) differs from the one at i = 0 */         \
            brk[K] = ((zmask + 1u) - (ta & zmask) + (uint32_t)K - 1u) / (uint32_t)K;                                     \
            constexpr int OFF = (K & 1) ? 0 : K / 2;                       /* even K: image h = 1 sits K/2 quadrants on */  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                              \
                const int ia = (j * K) & 3, ib = (j * K + OFF) & 3;                                                      \
                int32_t va = h ? sva[ib] : sva[ia], vb = h ? svb[ib] : svb[ia];                                          \
                if constexpr (MODE == 2 && (K & 1)) { va = -va; vb = -vb; }    /* VHDL rule: b_k enters with (-1)^k */    \
                acc[j] += va;                                                                                            \
                dlt[K][j] = vb - va;                                                                                     \
            }                                                                                                            \
        }
        BHW_RL_SETUP(1) BHW_RL_SETUP(2) BHW_RL_SETUP(3) BHW_RL_SETUP(4) BHW_RL_SETUP(5) BHW_RL_SETUP(6)
#undef BHW_RL_SETUP
        // sweep the run; every four positions one 16-byte granule per image goes to the LDS tile
#pragma unroll
        for (int c4 = 0; c4 < kRlRun / 4; ++c4) {
            int32_t v[4][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t i = (uint32_t)(c4 * 4 + e);
                if (i) {
#define BHW_RL_STEP(K)                                                                                                   \
                    if constexpr (NTERMS > K) {                                                                          \
                        if (__builtin_amdgcn_ballot_w64(brk[K] == i)) {            /* wave-uniform: usually nobody */   \
                            const bool mine = brk[K] == i;                                                              \
                            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j] += mine ? dlt[K][j] : 0;               \
                        }                                                                                               \
                    }
                    BHW_RL_STEP(1) BHW_RL_STEP(2) BHW_RL_STEP(3) BHW_RL_STEP(4) BHW_RL_STEP(5) BHW_RL_STEP(6)
#undef BHW_RL_STEP
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j][e] = finish(acc[j]);
            }
#if BHW_RL_DIRECT_STORE
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<int4 *>(out + (uint64_t)(r0 + 4u * (uint32_t)c4 + h * H) + (uint64_t)j * E) = make_int4(v[j][0], v[j][1], v[j][2], v[j][3]);
        }
#else
            if constexpr (NARROW) {
                const uint32_t g = rl_swizzle16((uint32_t)(kRlRun / 4) * lane + (uint32_t)c4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    tile[wave][j][g] = make_uint2(((uint32_t)v[j][0] & 0xFFFFu) | ((uint32_t)v[j][1] << 16),
                                                  ((uint32_t)v[j][2] & 0xFFFFu) | ((uint32_t)v[j][3] << 16));
            } else {
                const uint32_t g = rl_swizzle((uint32_t)(kRlRun / 4) * lane + (uint32_t)c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[wave][j][g] = make_int4(v[j][0], v[j][1], v[j][2], v[j][3]);
            }
        }
        __syncthreads();
        // store instruction s of image j: lane l writes ring lanes r_wave + 256 s + 4 l .. + 3 (1 KiB per wave instruction)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int sgrp = 0; sgrp < kRlRun / 4; ++sgrp) {
                int4 d;
                if constexpr (NARROW) {
                    const uint2 q = tile[wave][j][rl_swizzle16(64u * (uint32_t)sgrp + lane)];
                    d = make_int4((int32_t)(int16_t)(q.x & 0xFFFFu), (int32_t)q.x >> 16, (int32_t)(int16_t)(q.y & 0xFFFFu), (int32_t)q.y >> 16);
                } else {
                    d = tile[wave][j][rl_swizzle(64u * (uint32_t)sgrp + lane)];
                }
                const uint64_t idx = (uint64_t)(r_wave + 256u * (uint32_t)sgrp + 4u * lane + h * H) + (uint64_t)j * E;
                *reinterpret_cast<int4 *>(out + idx) = d;
            }
        __syncthreads();
#endif
    }
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
            else return Sum32{win.aa[0] >> 2, win.aa[0] & 3};
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

inline unsigned grid_for(uint64_t count) { return (unsigned)((count + kBlock - 1) / kBlock); }

// Launches go through hipLaunchKernel, which returns the launch status itself: the thread's hipGetLastError() state is
// neither read nor cleared here, so an error left behind by another library is not swallowed and not blamed on this call.
// The API layer (bhw_api.cpp) has already made l.device the current device.
thread_local hipError_t t_launch_err = hipSuccess;

template <typename T> struct same_type { using type = T; };

template <typename... KArgs>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, hipStream_t st, typename same_type<KArgs>::type... args)
{
    void *ptrs[] = {(void *)&args...};
    const hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(kernel), grid, block, ptrs, 0, st);
    if (e != hipSuccess && t_launch_err == hipSuccess) t_launch_err = e;
}
#define BHW_LAUNCH(kernel, grid, block, shmem, st, ...) launch(kernel, grid, block, st, __VA_ARGS__)

inline int finish(hipError_t e)
{
    if (e == hipSuccess) e = t_launch_err;
    t_launch_err = hipSuccess;
    return (int)e;
}

} // namespace

#define BHW_SET_DEVICE(l) ((void)(l))

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
#ifndef BHW_SINCOS_FOLD_MIN_PW
#define BHW_SINCOS_FOLD_MIN_PW 16
#endif
    if (c.dat_width + c.out_shr <= 34 && c.n_iter >= 7 && c.z_shr == 0 && c.phi_width >= BHW_SINCOS_FOLD_MIN_PW &&
        count == (1ull << c.phi_width)) {
        BhwBuildPlan plan;
        for (uint32_t k = 0; k < 32; ++k) plan.lut[k] = (uint32_t)c.lut[k];
        plan.entries = 1u << (c.phi_width - 2);
        plan.n_iter = c.n_iter;
        plan.z_shl = c.z_shl;
        plan.out_shr = c.out_shr;
        plan.log2_entries = c.phi_width - 2;
        plan.tab_split = 0;
        plan.tab_dlog = 0;
        plan.pad0 = c.ones_neg;
        plan.tab_coarse = d_cos;
        plan.x0 = c.x0;
        plan.check_flag = nullptr;
        const unsigned groups = plan.entries >> 6;
        plan.groups_per_wg = groups >= 64u * 1024u ? 64u : groups >= 16u * 1024u ? 16u : 4u;
        plan.pad = (uint32_t)(theta0 & ((1ull << c.phi_width) - 1ull));
        const dim3 grid((groups + plan.groups_per_wg - 1) / plan.groups_per_wg), block(kBuildThreads);
        switch (c.n_iter) {
#define BHW_CASE(N) case N: BHW_LAUNCH((k_table_build_shared<N, 4>), grid, block, 0, st, plan, (void *)d_sin); break;
            BHW_CASE(7) BHW_CASE(8) BHW_CASE(9) BHW_CASE(10) BHW_CASE(11) BHW_CASE(12) BHW_CASE(13) BHW_CASE(14)
            BHW_CASE(15) BHW_CASE(16) BHW_CASE(17) BHW_CASE(18) BHW_CASE(19) BHW_CASE(20) BHW_CASE(21) BHW_CASE(22)
            BHW_CASE(23) BHW_CASE(24) BHW_CASE(25) BHW_CASE(26) BHW_CASE(27) BHW_CASE(28) BHW_CASE(29) BHW_CASE(30)
            BHW_CASE(31) BHW_CASE(32)
#undef BHW_CASE
        default: return (int)hipErrorInvalidValue;
        }
        return finish(hipSuccess);
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
#ifndef BHW_REPL_BLOCKS
#define BHW_REPL_BLOCKS 65536
#endif
    unsigned gy = (unsigned)((BHW_REPL_BLOCKS + gx - 1) / gx);
    if (gy > frames) gy = frames;
    if (gy < 1) gy = 1;
    if (gy > 65535) gy = 65535;
    if (vec) BHW_LAUNCH(k_replicate16, dim3(gx, gy), dim3(kBlock), 0, st, (const int4 *)d_frame, items, frames, (int4 *)d_out);
    else     BHW_LAUNCH(k_replicate4, dim3(gx, gy), dim3(kBlock), 0, st, d_frame, items, frames, d_out);
    return finish(hipSuccess);
}

// Packed (delta16) table format applies when the (c, s) drift across a 64-entry block fits int16 with margin:
// 63 * 2 pi * 2^(W-2-PW) + noise < 2^15  <=>  W - PW <= 8  (25.4 k at W - PW = 8).  Amplitude is 2^(W-2) for every model.
bool bhwk_packed_ok(const BhwCordicCfg &c)
{
    if (c.z_shr != 0 || c.phi_width < 8) return false;
    return (int)c.dat_width - (int)c.phi_width <= 8;
}

// Residual format: largest d <= 9 for which the straight line between records 2^d entries apart stays within half an LSB of
// the true curve: (2 pi 2^d / 2^PW)^2 / 8 * 2^(W-2) <= 0.5.  0 = not applicable (d = 6 is left to delta16).
uint32_t bhwk_resid_dlog(const BhwCordicCfg &c)
{
    if (c.z_shr != 0 || c.phi_width < 20 || c.dat_width + c.out_shr > 34 || c.n_iter < 7) return 0;
    const int amp_bits = (int)c.dat_width - 2;                       // |c|, |s| <= 2^(W-2) (+1)
    const int twice_d = 2 * (int)c.phi_width - amp_bits - 4;         // 4.93 * 2^(2d - 2PW + W - 2) <= 0.5
    int d = twice_d / 2;
    if (d > 9) d = 9;
    if (d <= (int)kPackLog) return 0;                                // a 64-leaf build group must sit inside one cell
    if ((int)c.phi_width - 2 - d < 2) return 0;
    return (uint32_t)d;
}

// The layout goes with the format: nibble tables are always in the natural order (resid_offset), whatever the caller asked for.
static BhwCordicCfg table_layout(const BhwCordicCfg &c)
{
    BhwCordicCfg n = c;
    if (fmt_of(c.tab_dlog) == 3) n.tab_split = 0u;
    return n;
}

// octant mirror (k_table_build_mirror): residual / nibble entries, tables of 2^20 entries and more, and the
// exact quarter turn 2 * lut[0] == E << z_shl the symmetry rests on (true for every model at z_shr == 0; checked, not assumed)
#ifndef BHW_BUILD_MIRROR
#define BHW_BUILD_MIRROR 1
#endif
static bool build_mirror_applies(const BhwCordicCfg &c, uint32_t entries)
{
    const int fmt = fmt_of(c.tab_dlog);
    return BHW_BUILD_MIRROR && (fmt == 2 || fmt == 3) && (c.tab_split || fmt == 3) && c.z_shr == 0 && entries >= (1u << 20) && c.n_iter >= 21 &&
           c.dat_width + c.out_shr <= 34 && 2ull * (uint64_t)(uint32_t)c.lut[0] == ((uint64_t)entries << c.z_shl);
}

int bhwk_table_build(const BhwLaunch &l, const BhwCordicCfg &c_in, int32_t *d_table)
{
    const BhwCordicCfg c = table_layout(c_in);
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    const uint32_t entries = 1u << (c.phi_width - 2 - c.z_shr);
    // shared-prefix kernel: needs whole 64-leaf groups, |x| < 2^33 and a quarter circle <= 2^32
    const bool fits = (c.dat_width + c.out_shr <= 34);
    if (fits && c.n_iter >= 7 && entries < (1u << 20) && c.tab_dlog == 0 && !c.tab_split) {
        const dim3 grid(grid_for(entries)), block(kBlock);
        switch (c.n_iter) {
#define BHW_CASE(N) case N: BHW_LAUNCH(k_table_build_plain<N>, grid, block, 0, st, c, entries, (int2 *)d_table); break;
            BHW_CASE(7) BHW_CASE(8) BHW_CASE(9) BHW_CASE(10) BHW_CASE(11) BHW_CASE(12) BHW_CASE(13) BHW_CASE(14)
            BHW_CASE(15) BHW_CASE(16) BHW_CASE(17) BHW_CASE(18) BHW_CASE(19) BHW_CASE(20) BHW_CASE(21) BHW_CASE(22)
            BHW_CASE(23) BHW_CASE(24) BHW_CASE(25) BHW_CASE(26) BHW_CASE(27) BHW_CASE(28) BHW_CASE(29) BHW_CASE(30)
            BHW_CASE(31) BHW_CASE(32)
#undef BHW_CASE
        default: return (int)hipErrorInvalidValue;
        }
        return finish(hipSuccess);
    }
    if (entries >= 64 && fits && c.n_iter >= 2) {
        BhwBuildPlan plan;
        for (uint32_t k = 0; k < 32; ++k) plan.lut[k] = (uint32_t)c.lut[k];
        plan.entries = entries;
        plan.n_iter = c.n_iter;
        plan.z_shl = c.z_shl;
        plan.out_shr = c.out_shr;
        plan.log2_entries = c.phi_width - 2 - c.z_shr;
        plan.tab_split = c.tab_split;
        plan.tab_dlog = c.tab_dlog;
        plan.pad0 = 0;
        plan.tab_coarse = c.tab_coarse;
        plan.x0 = c.x0;
        plan.check_flag = c.tab_check;
        const unsigned groups = entries >> 6;
        plan.groups_per_wg = groups >= 64u * 1024u ? 64u : groups >= 16u * 1024u ? 16u : 4u;
        plan.pad = 0;
        const dim3 grid((groups + plan.groups_per_wg - 1) / plan.groups_per_wg), block(kBuildThreads);
        // plain tables for every rotation count; the packed formats (whole-period tile calls at z_shr == 0, i.e. PW >= 22 and
        // therefore at least 21 rotations: the VHDL model at PW == W runs W - 1 of them) from 21 rotations on
        const int fmt = fmt_of(c.tab_dlog);
        if (c.n_iter < 21 && fmt != 0) return (int)hipErrorInvalidValue;
        if (build_mirror_applies(c, entries)) {
            const unsigned own_groups = (entries >> 7) + (BHW_MIRROR_EXACT ? 0u : 1u);
            const dim3 mgrid((own_groups + BHW_MIRROR_GPW - 1) / BHW_MIRROR_GPW);
            plan.groups_per_wg = BHW_MIRROR_GPW;
            switch (c.n_iter) {
#define BHW_CASE_M(N) case N: if (fmt == 2) BHW_LAUNCH((k_table_build_mirror<N, 2>), mgrid, block, 0, st, plan, (void *)d_table); \
                              else          BHW_LAUNCH((k_table_build_mirror<N, 3>), mgrid, block, 0, st, plan, (void *)d_table); break;
                BHW_CASE_M(21) BHW_CASE_M(22) BHW_CASE_M(23) BHW_CASE_M(24) BHW_CASE_M(25) BHW_CASE_M(26) BHW_CASE_M(27) BHW_CASE_M(28)
                BHW_CASE_M(29) BHW_CASE_M(30) BHW_CASE_M(31) BHW_CASE_M(32)
#undef BHW_CASE_M
            default: return (int)hipErrorInvalidValue;
            }
            return finish(hipSuccess);
        }
#define BHW_LAUNCH_BUILD(N, F) BHW_LAUNCH((k_table_build_shared<N, F>), grid, block, 0, st, plan, (void *)d_table)
#define BHW_CASE(N) case N: BHW_LAUNCH_BUILD(N, 0); break;
#define BHW_CASE_T(N) case N: if (fmt == 0) BHW_LAUNCH_BUILD(N, 0); else if (fmt == 1) BHW_LAUNCH_BUILD(N, 1); else if (fmt == 2) BHW_LAUNCH_BUILD(N, 2); else BHW_LAUNCH_BUILD(N, 3); break;
        switch (c.n_iter) {
            BHW_CASE(7) BHW_CASE(8) BHW_CASE(9) BHW_CASE(10) BHW_CASE(11) BHW_CASE(12) BHW_CASE(13) BHW_CASE(14)
            BHW_CASE(15) BHW_CASE(16) BHW_CASE(17) BHW_CASE(18) BHW_CASE(19) BHW_CASE(20)
            BHW_CASE_T(21) BHW_CASE_T(22) BHW_CASE_T(23) BHW_CASE_T(24) BHW_CASE_T(25) BHW_CASE_T(26) BHW_CASE_T(27) BHW_CASE_T(28)
            BHW_CASE_T(29) BHW_CASE_T(30) BHW_CASE_T(31) BHW_CASE_T(32)
        default: return (int)hipErrorInvalidValue;
        }
#undef BHW_CASE
#undef BHW_CASE_T
#undef BHW_LAUNCH_BUILD
        return finish(hipSuccess);
    }
    if (c.tab_dlog > kPackLog) return (int)hipErrorInvalidValue;        // residual records come from the shared-prefix kernel only
    if (c.wide) BHW_LAUNCH(k_table_build<int64_t>, dim3(grid_for(entries)), dim3(kBlock), 0, st, c, entries, (void *)d_table);
    else        BHW_LAUNCH(k_table_build<int32_t>, dim3(grid_for(entries)), dim3(kBlock), 0, st, c, entries, (void *)d_table);
    return finish(hipSuccess);
}

int bhwk_table_combine(const BhwLaunch &l, const BhwCordicCfg &c_in, const BhwWinCfg &w, const int32_t *d_table,
                       uint64_t n0, uint64_t count, int32_t *d_out)
{
    const BhwCordicCfg c = table_layout(c_in);
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    BHW_LAUNCH(k_table_combine, dim3(grid_for(count)), dim3(kBlock), 0, st, c, w, (const void *)d_table, n0, count, d_out);
    return finish(hipSuccess);
}

int bhwk_table_combine_fold(const BhwLaunch &l, const BhwCordicCfg &c_in, const BhwWinCfg &w, const int32_t *d_table, int32_t *d_out)
{
    const BhwCordicCfg c = table_layout(c_in);
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    const uint32_t quarter = 1u << (c.phi_width - 2);
    if (c.tab_dlog == 0 && !c.tab_split) {                          // the usual case: plain natural table
        const int mode = (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0);
        const dim3 grid(grid_for(quarter)), block(kBlock);
#define BHW_FOLD_NT(NT)                                                                                                   \
        do {                                                                                                              \
            if (mode == 0)      BHW_LAUNCH((k_table_combine_fold_t<NT, 0>), grid, block, 0, st, c, w, (const void *)d_table, d_out); \
            else if (mode == 1) BHW_LAUNCH((k_table_combine_fold_t<NT, 1>), grid, block, 0, st, c, w, (const void *)d_table, d_out); \
            else                BHW_LAUNCH((k_table_combine_fold_t<NT, 2>), grid, block, 0, st, c, w, (const void *)d_table, d_out); \
        } while (0)
        switch (w.n_terms) {
        case 2: BHW_FOLD_NT(2); return finish(hipSuccess);
        case 3: BHW_FOLD_NT(3); return finish(hipSuccess);
        case 4: BHW_FOLD_NT(4); return finish(hipSuccess);
        case 5: BHW_FOLD_NT(5); return finish(hipSuccess);
        case 7: BHW_FOLD_NT(7); return finish(hipSuccess);
        default: break;
        }
#undef BHW_FOLD_NT
    }
    BHW_LAUNCH(k_table_combine_fold, dim3(grid_for(quarter)), dim3(kBlock), 0, st, c, w, (const void *)d_table, d_out);
    return finish(hipSuccess);
}


static uint32_t inv_mod_pow2(uint32_t a, uint32_t log2m)
{
    uint32_t x = a;                      // Newton iteration: x <- x (2 - a x), doubles the correct bits
    for (int i = 0; i < 6; ++i) x *= 2u - a * x;
    return log2m >= 32 ? x : (x & ((1u << log2m) - 1u));
}

bool bhwk_tile_applicable(const BhwCordicCfg &c, const BhwWinCfg &w)
{
    // With dropped phase bits (z_shr > 0) consecutive lanes share table entries, so the gathers are dense on their
    // own: such tables take the one-run form of the kernel over the natural layout.
    (void)w;
    // below 2^22 coefficients a grid of 960-thread tiles leaves CUs idle; the one-lane-per-four fold kernel has many more,
    // smaller workgroups and wins there (2^20: 15.0 vs 18.7 us, 2^21: 20.5 vs 21.2, 2^22: 36.0 vs 25.8; BH-7)
#ifndef BHW_TILE_MIN_PW
#define BHW_TILE_MIN_PW 22
#endif
    return c.phi_width >= BHW_TILE_MIN_PW && c.phi_width <= 30;
}

// The tile plan of a configuration: run offsets on the ring [0, N/8), lanes per run and tile, tiles that cover the ring.
static void make_tile_plan(const BhwCordicCfg &c, const BhwWinCfg &w, BhwTilePlan &tp, int &nb, uint32_t &lanes)
{
    const uint32_t lq = c.phi_width - 2, E = 1u << (lq - 1);   // the lane ring is [0, N/8): each lane owns r and r + N/8
    const uint32_t inv3 = inv_mod_pow2(3, lq - 1), inv5 = inv_mod_pow2(5, lq - 1);
    const int nb3 = (c.z_shr == 0 && w.n_terms > 3) ? 3 : 1, nb5 = (c.z_shr == 0 && w.n_terms > 5) ? 5 : 1;
    nb = nb3 * nb5;
    uint32_t sorted[15];
    for (int i5 = 0; i5 < nb5; ++i5)
        for (int i3 = 0; i3 < nb3; ++i3) {
            const uint32_t o = (uint32_t)(((uint64_t)i3 * inv3 + (uint64_t)i5 * inv5) & (E - 1u));
            // 3 thread groups: group p holds the five inv5-siblings of i3 = p (k = 5 dense per thread);
            // 5 thread groups (BHW_TILE_THREADS = 5 * BHW_TILE_LANES): group p holds the three inv3-siblings of i5 = p
            if (kTileThreads / kTileLanes == 5 && nb == 15) tp.offs[i3 + nb3 * i5] = o;
            else tp.offs[i5 + nb5 * i3] = o;
            sorted[i3 + nb3 * i5] = o;
        }
    for (int i = nb; i < 16; ++i) tp.offs[i] = tp.offs[nb - 1];
    // tiles needed so that every run class sweeps past the start of the next one around the ring
    for (int i = 1; i < nb; ++i)
        for (int j = i; j > 0 && sorted[j - 1] > sorted[j]; --j) { uint32_t t = sorted[j]; sorted[j] = sorted[j - 1]; sorted[j - 1] = t; }
    uint64_t maxgap = 0;
    for (int i = 0; i < nb; ++i) {
        const uint64_t nxt = (i + 1 < nb) ? sorted[i + 1] : (uint64_t)sorted[0] + E;
        if (nxt - sorted[i] > maxgap) maxgap = nxt - sorted[i];
    }
    lanes = (nb >= 15) ? (uint32_t)kTileLanes : (uint32_t)kTileThreads;
    tp.n_tiles = (uint32_t)((maxgap + lanes - 1) / lanes);
    tp.tile0 = 0;
}

// Tiles [tile0, tile0 + tile_count) of the plan (tile_count 0: all of them).
int bhwk_table_combine_tile_range(const BhwLaunch &l, const BhwCordicCfg &c_in, const BhwWinCfg &w, const int32_t *d_table, int32_t *d_out,
                                  uint32_t tile0, uint32_t tile_count, uint32_t img_mask, uint32_t n0mod)
{
    const BhwCordicCfg c = table_layout(c_in);
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    BhwTilePlan tp;
    int nb;
    uint32_t lanes;
    make_tile_plan(c, w, tp, nb, lanes);
    if (tile_count == 0) { tile0 = 0; tile_count = tp.n_tiles; }
    if (tile0 + tile_count > tp.n_tiles) return (int)hipErrorInvalidValue;
    tp.tile0 = tile0;
    tp.img_mask = img_mask & 0xFFu;
    tp.n0mod = n0mod;
    const bool masked = tp.img_mask != 0xFFu;                       // some of the eight images only (bhwk_tile_images_applicable)
    if (masked && (nb != 15 || w.apply_x != nullptr || tp.img_mask == 0u)) return (int)hipErrorInvalidValue;
    const int mode = (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0);
    const dim3 grid(tile_count), block(kTileThreads);
    // one-instruction products (tile_harmonic FAST): HLS rule, 15-run tiles, every harmonic weight below 2^(W-3) in magnitude
    // (the built-in weights are: a_k <= 0.49 * 2^(W-1 or W-2)); caller-scaled weights beyond that take the 64-bit products
#ifndef BHW_TILE_FASTMUL
#define BHW_TILE_FASTMUL 1
#endif
    bool fast = BHW_TILE_FASTMUL && mode != 2 && nb == 15 && c.dat_width >= 3;
    for (uint32_t k = 1; k < w.n_terms && fast; ++k) {
        const int64_t lim = (int64_t)1 << (c.dat_width - 3);
        fast = (int64_t)w.aa[k] < lim && (int64_t)w.aa[k] > -lim;       // (> : the kernel also multiplies by the negated pre-shifted weight)
    }
#define BHW_LAUNCH_TILE_MFK(NB, M, F, K)                                                                                 \
    do {                                                                                                                 \
        if (c.tab_dlog == 0)             BHW_LAUNCH((k_table_combine_tile<NB, M, 0, F, K>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out); \
        else if (c.tab_dlog == kPackLog) BHW_LAUNCH((k_table_combine_tile<NB, M, 1, F, K>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out); \
        else if (c.tab_dlog < kNibbleFlag) BHW_LAUNCH((k_table_combine_tile<NB, M, 2, F, K>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out); \
        else                             BHW_LAUNCH((k_table_combine_tile<NB, M, 3, F, K>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out); \
    } while (0)
#define BHW_LAUNCH_TILE_MF(NB, M, F)                                                                                     \
    do {                                                                                                                 \
        if (NB == 15 && masked) BHW_LAUNCH_TILE_MFK(NB, M, F, (NB == 15));                                               \
        else                    BHW_LAUNCH_TILE_MFK(NB, M, F, false);                                                    \
    } while (0)
#define BHW_LAUNCH_TILE_M(NB, M)                                                                                         \
    do {                                                                                                                 \
        if (NB == 15 && M != 2 && fast) BHW_LAUNCH_TILE_MF(NB, M, (NB == 15 && M != 2));                                 \
        else                            BHW_LAUNCH_TILE_MF(NB, M, false);                                                \
    } while (0)
#define BHW_LAUNCH_TILE(NB)                                                                                              \
    do {                                                                                                                 \
        if (mode == 0)      BHW_LAUNCH_TILE_M(NB, 0);                                                                    \
        else if (mode == 1) BHW_LAUNCH_TILE_M(NB, 1);                                                                    \
        else                BHW_LAUNCH_TILE_M(NB, 2);                                                                    \
    } while (0)
    if (nb == 15) BHW_LAUNCH_TILE(15);
    else if (nb == 3) BHW_LAUNCH_TILE(3);
    else BHW_LAUNCH_TILE(1);
#undef BHW_LAUNCH_TILE_M
#undef BHW_LAUNCH_TILE_MF
#undef BHW_LAUNCH_TILE_MFK
#undef BHW_LAUNCH_TILE
    return finish(hipSuccess);
}

int bhwk_table_combine_tile(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_table, int32_t *d_out)
{
    return bhwk_table_combine_tile_range(l, c, w, d_table, d_out, 0, 0, 0xFFu, 0u);
}

// A contiguous index range that is a whole number of eighths of the window (and less than all of it) can be produced by the
// tile kernel as a subset of its eight images: `*img_mask` = the images, `*n0mod` = n0 mod N (see BhwTilePlan).
bool bhwk_tile_images_applicable(const BhwCordicCfg &c, const BhwWinCfg &w, uint64_t n0, uint64_t count, uint32_t *img_mask, uint32_t *n0mod)
{
    if (!bhwk_tile_applicable(c, w) || c.z_shr != 0 || w.apply_x != nullptr || w.n_terms <= 5) return false;   // 15-run tiles only
    const uint64_t N = 1ull << c.phi_width, eighth = N >> 3;
    if (count == 0 || count >= N || (count % eighth) != 0 || (n0 % eighth) != 0) return false;
    const uint32_t m0 = (uint32_t)((n0 % N) / eighth), n_img = (uint32_t)(count / eighth);
    uint32_t mask = 0;
    for (uint32_t i = 0; i < n_img; ++i) mask |= 1u << ((m0 + i) & 7u);
    *img_mask = mask;
    *n0mod = (uint32_t)(n0 % N);
    return true;
}

// Kernel names of the table strategy's two passes for a resolved configuration (bhw_describe_plan: profilers, bench labels).
// Mirrors the dispatch in bhwk_table_build / bhwk_table_combine_tile_range / bhwk_table_combine_fold.
void bhwk_describe_table(const BhwCordicCfg &c_in, const BhwWinCfg &w, bool tiled, char *build, char *combine, size_t len)
{
    const BhwCordicCfg c = table_layout(c_in);
    const uint32_t entries = 1u << (c.phi_width - 2 - c.z_shr);
    const bool fits = (c.dat_width + c.out_shr <= 34);
    const int fmt = fmt_of(c.tab_dlog);
    if (fits && c.n_iter >= 7 && entries < (1u << 20) && c.tab_dlog == 0 && !c.tab_split) snprintf(build, len, "k_table_build_plain<%u>", c.n_iter);
    else if (entries >= 64 && fits && c.n_iter >= 2)
        snprintf(build, len, build_mirror_applies(c, entries) ? "k_table_build_mirror<%u,%d>" : "k_table_build_shared<%u,%d>", c.n_iter, fmt);
    else snprintf(build, len, "k_table_build<%s>", c.wide ? "int64_t" : "int32_t");
    const int mode = (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0);
    if (tiled) {
        const int nb3 = (c.z_shr == 0 && w.n_terms > 3) ? 3 : 1, nb5 = (c.z_shr == 0 && w.n_terms > 5) ? 5 : 1;
        snprintf(combine, len, "k_table_combine_tile<%d,%d,%d>", nb3 * nb5, mode, fmt);
    } else if (c.tab_dlog == 0 && !c.tab_split) snprintf(combine, len, "k_table_combine_fold_t<%u,%d>", w.n_terms, mode);
    else snprintf(combine, len, "k_table_combine_fold");
}

// Interleaved ownership (bhw_generate_part_device): the ring lanes of part `part` of `n_parts`, as runs of consecutive r.
// Where the tile kernel applies the parts are contiguous ranges of its tiles, i.e. the plan's sibling runs (so a part can be
// produced by the tile kernel over the full table or by the fused kernel, with the same ownership); elsewhere they are
// contiguous ranges of the ring in 64-lane units.  Runs that wrap the ring are split; neighbouring parts overlap by the few
// lanes the tile plan covers twice at its seams (identical values).
int bhwk_part_runs(const BhwCordicCfg &c, const BhwWinCfg &w, uint32_t part, uint32_t n_parts, BhwFoldRun *runs, uint32_t *tile0, uint32_t *tile_count)
{
    const uint32_t H = 1u << (c.phi_width - 3);
    *tile0 = *tile_count = 0;
    if (n_parts < 1) n_parts = 1;
    if (!bhwk_tile_applicable(c, w)) {
        const uint32_t units = (H + 63u) >> 6;
        const uint32_t a = (uint32_t)((uint64_t)units * part / n_parts) << 6, b = (uint32_t)((uint64_t)units * (part + 1u) / n_parts) << 6;
        runs[0] = BhwFoldRun{a < H ? a : H, b < H ? b : H};
        return runs[0].r_end > runs[0].r0 ? 1 : 0;
    }
    BhwTilePlan tp;
    int nb;
    uint32_t lanes;
    make_tile_plan(c, w, tp, nb, lanes);
    const uint32_t t0 = (uint32_t)((uint64_t)tp.n_tiles * part / n_parts), t1 = (uint32_t)((uint64_t)tp.n_tiles * (part + 1u) / n_parts);
    *tile0 = t0;
    *tile_count = t1 - t0;
    if (t1 == t0) return 0;
    const uint64_t len = (uint64_t)(t1 - t0) * lanes;
    int n = 0;
    for (int b = 0; b < nb; ++b) {
        if (len >= H) { runs[0] = BhwFoldRun{0u, H}; return 1; }
        const uint32_t start = (uint32_t)(((uint64_t)t0 * lanes + tp.offs[b]) & (H - 1u));
        if (start + len <= H) runs[n++] = BhwFoldRun{start, (uint32_t)(start + len)};
        else {
            runs[n++] = BhwFoldRun{start, H};
            runs[n++] = BhwFoldRun{0u, (uint32_t)(start + len - H)};
        }
    }
    return n;
}

bool bhwk_fold_direct_applicable(const BhwCordicCfg &c)
{
    // rot_step's forms: |x| < 2^33 and a quarter circle <= 2^32; ring of at least one wave
    return c.dat_width + c.out_shr <= 34 && c.phi_width >= 9 && c.phi_width <= 30 && c.n_iter >= 2;
}

int bhwk_fold_direct(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const BhwFoldRun *runs, uint32_t n_runs, int32_t *d_out)
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
    plan.fast_mul = (w.combine == BHW_COMBINE_HLS && c.dat_width >= 3) ? 1u : 0u;
    for (uint32_t k = 1; k < w.n_terms && plan.fast_mul; ++k) {
        const int64_t lim = (int64_t)1 << (c.dat_width - 3);
        if ((int64_t)w.aa[k] >= lim || (int64_t)w.aa[k] <= -lim) plan.fast_mul = 0u;
    }
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_runs; ++i) total += runs[i].r_end - runs[i].r0;
    // short launches: one wave per workgroup spreads the few waves over more CUs
    const uint32_t block = total <= 64u * 1024u ? 64u : (uint32_t)kFoldBlock;
    uint32_t wg = 0;
    for (uint32_t i = 0; i < n_runs; ++i) {
        plan.r0[i] = runs[i].r0;
        plan.r_end[i] = runs[i].r_end;
        plan.wg_first[i] = wg;
        wg += (runs[i].r_end - runs[i].r0 + block - 1u) / block;
    }
    plan.wg_first[n_runs] = wg;
    if (!wg) return 0;
    const int mode = (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0);
    const dim3 grid(wg), blk(block);
    // fewer than ~4 waves per SIMD in the whole launch: latency-bound, walk the chains in lockstep
#ifndef BHW_FD_LOCKSTEP_MAX
#define BHW_FD_LOCKSTEP_MAX (1u << 18)
#endif
    const bool lockstep = total <= BHW_FD_LOCKSTEP_MAX;
    // short launches, form of the kernel: 2 = chains split over four waves per 64 lanes (k_fold_split), 1 = lockstep, 0 = sequential
#ifndef BHW_FD_SMALL_MODE
#define BHW_FD_SMALL_MODE 2
#endif
    // measured per call (profiles/r02_ab_fused_lockstep.txt): split 7.8 / lockstep 9.0 us at 2^13 lanes (BH-7 2^16), 6.9 / 7.7 at
    // 2^15 (BH-5 2^18), 9.7 / 9.7 at 2^16, 9.0 / 8.2 at 2^17 (BH-4 2^20): split up to 2^15 lanes, lockstep up to 2^18
#ifndef BHW_FD_SPLIT_MAX
#define BHW_FD_SPLIT_MAX (1u << 15)
#endif
    const bool split = BHW_FD_SMALL_MODE == 2 && total <= BHW_FD_SPLIT_MAX;
    dim3 grid_s(0), blk_s(256);
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
        if (split) BHW_LAUNCH((k_fold_split<NT, M>), grid_s, blk_s, 0, st, w, plan, d_out);               \
        else if (lockstep && BHW_FD_SMALL_MODE >= 1) BHW_LAUNCH((k_fold_direct<NT, M, true>), grid, blk, 0, st, w, plan, d_out); \
        else          BHW_LAUNCH((k_fold_direct<NT, M, false>), grid, blk, 0, st, w, plan, d_out);          \
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

// Run-length kernel: z_shr > 0, at most one entry step per harmonic inside a 16-lane run, ring a multiple of the workgroup's
// 2048 lanes, plain natural table, 16-byte aligned output, no fused apply; VHDL rule in int32 needs W + 2 <= 30.
bool bhwk_runlength_applicable(const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_out)
{
    if (c.z_shr == 0 || c.tab_dlog != 0 || c.tab_split != 0 || w.apply_x != nullptr) return false;
    if (c.phi_width < 15 || c.phi_width > 30) return false;                       // ring (2^(PW-3)) >= 2048 lanes
    if (((w.n_terms - 1u) * (uint32_t)kRlRun) > (1u << c.z_shr)) return false;
    if (c.phi_width - 2u - c.z_shr < 2u) return false;                            // H a multiple of 2^z_shr
    if (w.combine != BHW_COMBINE_HLS && c.dat_width > 28) return false;
    return (((uintptr_t)d_out) & 15u) == 0;
}

int bhwk_runlength_window(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_table, int32_t *d_out)
{
    if (!bhwk_runlength_applicable(c, w, d_out)) return (int)hipErrorInvalidValue;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    const uint32_t H = 1u << (c.phi_width - 3);
    const dim3 grid(H / (kRlBlock * kRlRun)), block(kRlBlock);
    const int mode = (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0);
#ifndef BHW_RL_NARROW_MAX
#define BHW_RL_NARROW_MAX 16
#endif
    const bool narrow = c.dat_width <= BHW_RL_NARROW_MAX;                       // coefficients fit int16: half-size LDS tile
#define BHW_RL_NT_M(NT, M)                                                                                                     \
    do {                                                                                                                       \
        if (narrow) BHW_LAUNCH((k_runlength_window<NT, M, true>), grid, block, 0, st, c, w, (const int2 *)d_table, d_out);     \
        else        BHW_LAUNCH((k_runlength_window<NT, M, false>), grid, block, 0, st, c, w, (const int2 *)d_table, d_out);    \
    } while (0)
#define BHW_RL_NT(NT)                                                                                                          \
    do {                                                                                                                       \
        if (mode == 0)      BHW_RL_NT_M(NT, 0);                                                                                \
        else if (mode == 1) BHW_RL_NT_M(NT, 1);                                                                                \
        else                BHW_RL_NT_M(NT, 2);                                                                                \
    } while (0)
    switch (w.n_terms) {
    case 2: BHW_RL_NT(2); break;
    case 3: BHW_RL_NT(3); break;
    case 4: BHW_RL_NT(4); break;
    case 5: BHW_RL_NT(5); break;
    case 7: BHW_RL_NT(7); break;
    default: return (int)hipErrorInvalidValue;
    }
#undef BHW_RL_NT
#undef BHW_RL_NT_M
    return finish(hipSuccess);
}

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
#ifndef BHW_TAYLOR_BLOCKS
#define BHW_TAYLOR_BLOCKS 4096u
#endif
    if (blocks > BHW_TAYLOR_BLOCKS) blocks = BHW_TAYLOR_BLOCKS;
    const dim3 grid(blocks);
    hipStream_t st = (hipStream_t)l.stream;
#define BHW_TAYLOR_FOLD(ARITH, COMBINE, NT)                                                                         \
    do {                                                                                                            \
        if (fast) BHW_LAUNCH((k_taylor_window_fold<ARITH, COMBINE, NT, true>), grid, dim3(kBlock), 0, st, t, w, d_out);  \
        else      BHW_LAUNCH((k_taylor_window_fold<(ARITH == 2 ? 0 : ARITH), COMBINE, NT, false>), grid, dim3(kBlock), 0, st, t, w, d_out); \
    } while (0)
    const bool vhdl = w.combine == BHW_COMBINE_VHDL;
    // every generator in use (PHASE_WIDTH - v, v <= vmax) on the 1st-order-correction path, ROM in LDS
    const int vmax = w.n_terms > 4 ? 2 : w.n_terms > 2 ? 1 : 0;
    const bool fast = (1u << t.lut_size) <= (uint32_t)kTaylorRomLds && (int)t.phi_width - vmax - (int)t.lut_size > 2;
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

// ---------------------------------------------------------------------------------------
// Variant generators.  One lane per sample, 64-bit state wrapped to the entity's vector widths.
// ---------------------------------------------------------------------------------------
// cordic_dds48 (src/cordic_dds48.vhd:160-258) and cordic_dds_scaled (src/cordic_dds_scaled.vhd:176-283)
// The SIZE- and DWPH-bit stores of the entities never wrap (|x|, |y| <= 2^(SIZE-2) * 1.0000..., |z| < 2^(DWPH-2) + atan
// terms; tests/test_oracle.py::test_variant_generator_wraps_never_fire), so the kernel keeps plain 64-bit state.
// Rotation ii in "mad" form once the shifted operand fits 32 bits (SIZE <= 48: ii >= 16), as in rot_step; the first
// rotations use 64-bit select-and-add.  dds48 :233-251: z >= 0 -> x += y>>ii, y -= x>>ii, z -= rom; else the opposite.
template <int II>
__device__ __forceinline__ void prerot_step(int64_t &x, int64_t &y, int64_t &z, int64_t rom, bool last)
{
    const int32_t m = (int32_t)(z >> 63);               // -1 when z < 0
    if constexpr (II >= 16) {
        const int32_t sg = m | 1;                       // -1 when z < 0, +1 otherwise
        const int32_t nsg = -sg;
        int32_t ys = (int32_t)(y >> II), xs = (int32_t)(x >> II);
        asm volatile("" : "+v"(ys), "+v"(xs));          // both shifts read the old state
        x += (int64_t)sg * (int64_t)ys;
        y += (int64_t)nsg * (int64_t)xs;
        if (!last) z += (int64_t)nsg * (int64_t)(int32_t)rom;       // rom < 2^(45-II) here
    } else {
        const int64_t ys = y >> II, xs = x >> II;
        const bool neg = m != 0;
        x += neg ? -ys : ys;
        y += neg ? xs : -xs;
        if (!last) z += neg ? rom : -rom;
    }
}

template <int NITER>
__global__ __launch_bounds__(kBlock) void k_sincos_prerot(BhwPrerotCfg c, uint64_t theta0, uint64_t count,
                                                           int32_t *__restrict__ d_sin, int32_t *__restrict__ d_cos)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t PW = c.phi_width;
    const uint64_t theta = (theta0 + i) & ((1ull << PW) - 1ull);
    const uint32_t q = (uint32_t)(theta >> (PW - 2)) & 3u;              // dds48 :167
    const uint64_t low = theta & ((1ull << (PW - 2)) - 1ull);
    uint64_t t = theta;                                                 // init_t :169-186
    int64_t x = c.gain, y = 0;                                          // init_x / init_y :191-216
    if (q == 1u)      { t = low;                      x = 0; y = -c.gain; }
    else if (q == 2u) { t = (3ull << (PW - 2)) | low; x = 0; y = c.gain; }
    int64_t z = wrap_bits((int64_t)(t << (c.dwph - PW)), c.dwph);       // init_z :163-164 (scaled :180-186): sign-extend the phase
#define BHW_PREROT(II) if constexpr (NITER > II) prerot_step<II>(x, y, z, c.lut[II], II + 1 == NITER);
    BHW_PREROT(0) BHW_PREROT(1) BHW_PREROT(2) BHW_PREROT(3) BHW_PREROT(4) BHW_PREROT(5) BHW_PREROT(6) BHW_PREROT(7)
    BHW_PREROT(8) BHW_PREROT(9) BHW_PREROT(10) BHW_PREROT(11) BHW_PREROT(12) BHW_PREROT(13) BHW_PREROT(14) BHW_PREROT(15)
    BHW_PREROT(16) BHW_PREROT(17) BHW_PREROT(18) BHW_PREROT(19) BHW_PREROT(20) BHW_PREROT(21) BHW_PREROT(22) BHW_PREROT(23)
    BHW_PREROT(24) BHW_PREROT(25) BHW_PREROT(26) BHW_PREROT(27) BHW_PREROT(28) BHW_PREROT(29) BHW_PREROT(30) BHW_PREROT(31)
#undef BHW_PREROT
    if (d_sin) d_sin[i] = (int32_t)(y >> (c.size - NITER));             // :257-258  top DATA_WIDTH bits
    if (d_cos) d_cos[i] = (int32_t)(x >> (c.size - NITER));
}

// The same generators over one whole period (count == 2^PHASE_WIDTH, any start phase): inside a quadrant consecutive phases are
// consecutive angles on one start vector, so a group of 64 phases shares its rotations until the first one whose threshold
// falls inside the group -- exactly the structure of k_table_build_shared, with the quadrant's start vector per group.  Phase 1:
// one lane per group runs the shared prefix (these are the expensive rotations: 64-bit select-and-add below stage 16) and parks
// it in LDS; phase 2: one wave per group, one lane per phase, only the remaining stages.
template <int NITER>
__global__ __launch_bounds__(kBuildThreads) void k_prerot_sweep(BhwPrerotCfg c, uint32_t theta0, int32_t *__restrict__ d_sin, int32_t *__restrict__ d_cos)
{
    __shared__ int64_t gx[kGroupsPerWg], gy[kGroupsPerWg], gz[kGroupsPerWg];
    __shared__ int32_t gk[kGroupsPerWg];
    const uint32_t PW = c.phi_width, zs = c.dwph - PW;
    const uint32_t group0 = blockIdx.x * kGroupsPerWg, n_groups = 1u << (PW - 6);
    constexpr int kmax = NITER < kPrefixMax ? NITER : kPrefixMax;
    if (threadIdx.x < (uint32_t)kGroupsPerWg && group0 + threadIdx.x < n_groups) {
        const uint64_t theta = (uint64_t)(group0 + threadIdx.x) << 6;       // first phase of the group
        const uint32_t q = (uint32_t)(theta >> (PW - 2)) & 3u;              // dds48 :167
        const uint64_t low = theta & ((1ull << (PW - 2)) - 1ull);
        uint64_t t = theta;                                                 // init_t :169-186
        int64_t x = c.gain, y = 0;                                          // init_x / init_y :191-216
        if (q == 1u)      { t = low;                      x = 0; y = -c.gain; }
        else if (q == 2u) { t = (3ull << (PW - 2)) | low; x = 0; y = c.gain; }
        int64_t zf = wrap_bits((int64_t)(t << zs), c.dwph);                 // init_z: sign-extended phase
        const int64_t span = (int64_t)63 << zs;
        int k = 0;
        bool live = true;
#define BHW_PRE(II)                                                                        \
        if constexpr (kmax > II) {                                                          \
            if (live) {                                                                     \
                if ((zf < 0) != (zf + span < 0)) live = false;                              \
                else { prerot_step<II>(x, y, zf, c.lut[II], II + 1 == NITER); k = II + 1; } \
            }                                                                               \
        }
        BHW_PRE(0) BHW_PRE(1) BHW_PRE(2) BHW_PRE(3) BHW_PRE(4) BHW_PRE(5) BHW_PRE(6) BHW_PRE(7) BHW_PRE(8) BHW_PRE(9)
        BHW_PRE(10) BHW_PRE(11) BHW_PRE(12) BHW_PRE(13) BHW_PRE(14) BHW_PRE(15) BHW_PRE(16) BHW_PRE(17) BHW_PRE(18) BHW_PRE(19)
        BHW_PRE(20) BHW_PRE(21) BHW_PRE(22) BHW_PRE(23)
#undef BHW_PRE
        gx[threadIdx.x] = x; gy[threadIdx.x] = y; gz[threadIdx.x] = zf; gk[threadIdx.x] = k;
    }
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t nmask = (1u << PW) - 1u;                                  // PW <= 32 here: a sweep of 2^PW phases in int32 indices
    for (uint32_t gi = wave; gi < (uint32_t)kGroupsPerWg; gi += kBuildThreads / 64) {
        const uint32_t g = group0 + gi;
        if (g >= n_groups) break;
        int64_t x = gx[gi], y = gy[gi];
        int64_t z = gz[gi] + ((int64_t)lane << zs);
        const int k0 = __builtin_amdgcn_readfirstlane(gk[gi]);
#define BHW_POST(II) if constexpr (NITER > II) { if (II >= kmax || II >= k0) prerot_step<II>(x, y, z, c.lut[II], II + 1 == NITER); }
        BHW_POST(0) BHW_POST(1) BHW_POST(2) BHW_POST(3) BHW_POST(4) BHW_POST(5) BHW_POST(6) BHW_POST(7)
        BHW_POST(8) BHW_POST(9) BHW_POST(10) BHW_POST(11) BHW_POST(12) BHW_POST(13) BHW_POST(14) BHW_POST(15)
        BHW_POST(16) BHW_POST(17) BHW_POST(18) BHW_POST(19) BHW_POST(20) BHW_POST(21) BHW_POST(22) BHW_POST(23)
        BHW_POST(24) BHW_POST(25) BHW_POST(26) BHW_POST(27) BHW_POST(28) BHW_POST(29) BHW_POST(30) BHW_POST(31)
#undef BHW_POST
        const uint32_t i = ((g << 6) + lane - theta0) & nmask;
        if (d_sin) d_sin[i] = (int32_t)(y >> (c.size - NITER));             // :257-258  top DATA_WIDTH bits
        if (d_cos) d_cos[i] = (int32_t)(x >> (c.size - NITER));
    }
}

// cordic_atan2 (src/cordic_atan2.vhd:126-213).  The B = ANGLE_WIDTH + PRECISION bit registers do wrap (PRECISION 1 with
// full-scale inputs), so the state is kept shifted left by (word size - B): overflow of the word then *is* the B-bit wrap, and
// the only extra work is clearing the low bits that an arithmetic right shift drags in.  U = uint32_t when B <= 32 (half the
// instructions of the 64-bit form: every shift, add and select is one 32-bit operation), uint64_t otherwise.
template <typename U>
__global__ __launch_bounds__(kBlock) void k_atan2(BhwAtan2Cfg c, uint64_t count, const int32_t *__restrict__ d_x,
                                                   const int32_t *__restrict__ d_y, int32_t *__restrict__ d_phi)
{
    using S = typename std::make_signed<U>::type;
    constexpr uint32_t WB = 8u * sizeof(U);
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t IW = c.input_width, AW = c.angle_width, B = AW + c.precision, sh = WB - B;
    const uint64_t im = (1ull << IW) - 1ull;                             // IW <= 32
    const uint64_t ux = (uint64_t)(int64_t)d_x[i] & im, uy = (uint64_t)(int64_t)d_y[i] & im;
    const uint32_t sx = (uint32_t)(ux >> (IW - 1)) & 1u, sy = (uint32_t)(uy >> (IW - 1)) & 1u;
    const uint64_t lowm = (1ull << (AW - 1)) - 1ull;
    const U keep = (U)~(((U)1 << sh) - (U)1);
    U X = (U)((sx ? ~ux : ux) & lowm) << sh;                             // :142-147
    U Y = (U)((sy ? ~uy : uy) & lowm) << sh;
    U Z = 0;                                                            // :152
    for (uint32_t ii = 0; ii + 1 < AW; ++ii) {                          // :172-190
        const U xs = (U)((S)X >> ii) & keep, ys = (U)((S)Y >> ii) & keep;
        const U rom = (U)c.lut[ii] << sh;
        const bool pos = (S)Y >= 0;
        X = pos ? X + ys : X - ys;
        Y = pos ? Y - xs : Y + xs;
        Z = pos ? Z - rom : Z + rom;
    }
    // :194  sigZ(ANGLE_WIDTH-1)(B-1 downto PRECISION): the top ANGLE_WIDTH bits of the B-bit word
    const int64_t phi = (int64_t)((S)Z >> (WB - AW));
    const int64_t pi_word = (int64_t)1 << (AW - 2);                     // PHI_PI :112
    const uint32_t quad = (sx << 1) | sy;                               // :126-128
    const int64_t out = quad == 0u ? phi : quad == 1u ? phi + pi_word : quad == 2u ? -phi : phi - pi_word;   // :207-213
    d_phi[i] = (int32_t)wrap_bits(out, AW);
}

int bhwk_sincos_prerot(const BhwLaunch &l, const BhwPrerotCfg &c, uint64_t theta0, uint64_t count, int32_t *d_sin, int32_t *d_cos)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    if (c.phi_width >= 16 && c.phi_width <= 30 && count == (1ull << c.phi_width)) {      // one whole period: shared rotation prefixes
        const uint32_t groups = 1u << (c.phi_width - 6);
        const dim3 grid((groups + kGroupsPerWg - 1) / kGroupsPerWg), block(kBuildThreads);
        const uint32_t th0 = (uint32_t)(theta0 & ((1ull << c.phi_width) - 1ull));
        switch (c.dat_width) {
#define BHW_CASE(N) case N: BHW_LAUNCH(k_prerot_sweep<N>, grid, block, 0, st, c, th0, d_sin, d_cos); break;
            BHW_CASE(8) BHW_CASE(9) BHW_CASE(10) BHW_CASE(11) BHW_CASE(12) BHW_CASE(13) BHW_CASE(14) BHW_CASE(15) BHW_CASE(16)
            BHW_CASE(17) BHW_CASE(18) BHW_CASE(19) BHW_CASE(20) BHW_CASE(21) BHW_CASE(22) BHW_CASE(23) BHW_CASE(24)
            BHW_CASE(25) BHW_CASE(26) BHW_CASE(27) BHW_CASE(28) BHW_CASE(29) BHW_CASE(30) BHW_CASE(31) BHW_CASE(32)
#undef BHW_CASE
        default: return (int)hipErrorInvalidValue;
        }
        return finish(hipSuccess);
    }
    const dim3 grid(grid_for(count)), block(kBlock);
    switch (c.dat_width) {                                              // DATA_WIDTH stages, unrolled
#define BHW_CASE(N) case N: BHW_LAUNCH(k_sincos_prerot<N>, grid, block, 0, st, c, theta0, count, d_sin, d_cos); break;
        BHW_CASE(8) BHW_CASE(9) BHW_CASE(10) BHW_CASE(11) BHW_CASE(12) BHW_CASE(13) BHW_CASE(14) BHW_CASE(15) BHW_CASE(16)
        BHW_CASE(17) BHW_CASE(18) BHW_CASE(19) BHW_CASE(20) BHW_CASE(21) BHW_CASE(22) BHW_CASE(23) BHW_CASE(24)
        BHW_CASE(25) BHW_CASE(26) BHW_CASE(27) BHW_CASE(28) BHW_CASE(29) BHW_CASE(30) BHW_CASE(31) BHW_CASE(32)
#undef BHW_CASE
    default: return (int)hipErrorInvalidValue;
    }
    return finish(hipSuccess);
}

int bhwk_atan2(const BhwLaunch &l, const BhwAtan2Cfg &c, uint64_t count, const int32_t *d_x, const int32_t *d_y, int32_t *d_phi)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    if (c.angle_width + c.precision <= 32u) BHW_LAUNCH(k_atan2<uint32_t>, dim3(grid_for(count)), dim3(kBlock), 0, (hipStream_t)l.stream, c, count, d_x, d_y, d_phi);
    else                                    BHW_LAUNCH(k_atan2<uint64_t>, dim3(grid_for(count)), dim3(kBlock), 0, (hipStream_t)l.stream, c, count, d_x, d_y, d_phi);
    return finish(hipSuccess);
}

int bhwk_taylor_sincos(const BhwLaunch &l, const BhwTaylorCfg &t, uint64_t theta0, uint64_t count, int32_t *d_sin, int32_t *d_cos)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    BHW_LAUNCH(k_taylor_sincos, dim3(grid_for(count)), dim3(kBlock), 0, (hipStream_t)l.stream, t, theta0, count, d_sin, d_cos);
    return finish(hipSuccess);
}
