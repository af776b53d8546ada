// bhw_device.h -- device helpers, table formats, cosine-sum arithmetic and launch helpers shared by the kernel translation units
//
// Part of the hand-written HIP kernels for gfx950 (MI355X, CDNA4) behind include/bhw.h.  Hot path of the reference: phase
// accumulator -> CORDIC rotation chain (or Taylor LUT) -> weighted N-term cosine sum -> int32 coefficient (SURVEY section 8a
// rows a1-a11).  Integer semantics follow SURVEY App. A; reference lines are cited at each step.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include "bhw_plan.h"

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
//   23..25 "nibble" (16 + d): the residual format with the two deviations in 4-bit fields, one byte per entry:
//          byte = (dev_c + 8) | (dev_s + 8) << 4, both fields unsigned; the records of these formats are {c - 8, s - 8, 2 dc, 2 ds}
//          (tab_predict_nib), so an entry is  record line + field  with no sign extension.  The deviations
//          of the 32-bit HLS model stay within -5 .. 6 over the whole 2^24-entry table of a 2^26-point window (measured with the
//          oracle); like the other packed formats it is used only after the build kernel has checked every entry of the
//          configuration (a model whose noise is wider -- the cpp model reaches 10 -- falls back to the byte fields).
//   55..57 "nibble + escapes" (48 + d): the nibble format with the low field 0 (deviation -8) reserved as a marker: that entry's exact pair is in
//          the hash table (kEscSlots slots, esc_lookup) of the build workgroup that stored it, at cfg.tab_esc.  For the models whose
//          noise is a little wider than the fields -- cpp, VHDL at 32 bits: 547 / 932 of the 2^24 entries of a 2^26-point
//          window, at most 36 / 74 in one workgroup (tests/test_oracle.py::test_nibble_escape_capacity; the records of this format
//          carry another + 1, which centres the deviations and halves the count) -- so that they too read one byte per entry.  The tile kernel tests the minimum of a harmonic's low fields (one v_min per gather, one branch per
//          harmonic) and resolves the rare marked lane on the scalar unit (esc_fix_wave); a workgroup whose table would fill
//          beyond kEscFill sets the check word and the configuration falls back to the byte fields.
// The combine pass is bound by table + output traffic as much as by arithmetic, and these cut the table's share to 1/2 .. 1/8.
// (kPackLog, kNibbleFlag, fmt_cell_log(), fmt_of(), table_layout(): bhw_plan.h, shared with the HIP-free planner)

__device__ __forceinline__ int2 tab_predict(const int4 rec, uint32_t f, uint32_t d)
{
    return make_int2(rec.x + (__mul24(rec.z, (int32_t)f) >> d), rec.y + (__mul24(rec.w, (int32_t)f) >> d));   // |dc|, |ds| < 2^17, f < 2^9
}

// Nibble formats (3, 5): the records hold the slopes DOUBLED and the start values less the fields' bias (see "nibble" above), so
// that the tile kernel evaluates the line as one v_mul_hi_i32 per coordinate -- hi32(2 dc * (f << (31 - d))) = (dc * f) >> d
// exactly -- and reads the fields as unsigned (an and and a shift, VOP2 at full rate, instead of two v_bfe_i32).  Everyone else
// evaluates the same line here.
__device__ __forceinline__ int2 tab_predict_nib(const int4 rec, uint32_t f, uint32_t d)
{
    return make_int2(rec.x + (__mul24(rec.z, (int32_t)f) >> (d + 1u)), rec.y + (__mul24(rec.w, (int32_t)f) >> (d + 1u)));   // |2 dc|, |2 ds| < 2^18, f < 2^9
}
__device__ __forceinline__ int2 nib_fields(uint32_t e) { return make_int2((int32_t)(e & 15u), (int32_t)((e >> 4) & 15u)); }

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
struct ResidK {
    uint32_t d;       // log2 of the cell size
    uint32_t fmask;   // 2^d - 1
};

// Nibble + escapes: the entry whose low field holds the marker -8 is listed exactly.  Entry u of [0, E) was stored by the build
// workgroup that owns its source min(u, E - u) (the middle entry E/2 by the last one); that workgroup's list is a hash table of
// kEscSlots slots { u or -1, c, s, - }, open addressing from esc_slot(u).  A few entries per 100 000 (the cpp model at 2^26 /
// 32 bits: 547 of 2^24), at most kEscFill per workgroup: one or two probes, and the branch that leads here is cold.
constexpr uint32_t kEscMarker = 0u;                               // low field (biased by 8) of a listed entry: the deviation -8
__device__ __forceinline__ uint32_t esc_slot(uint32_t u) { return (u * 0x9E3779B1u) >> 25; }
static_assert(kEscSlots == 128, "esc_slot: seven bits");
__device__ __forceinline__ int2 esc_lookup(const void *__restrict__ esc, uint32_t esc_wg_log, uint32_t log2_entries, uint32_t u)
{
    const uint32_t E = 1u << log2_entries, half = E >> 1;
    uint32_t own = u <= half ? u : E - u;
    if (own > half - 1u) own = half - 1u;
    const int4 *tab = reinterpret_cast<const int4 *>(esc) + (size_t)(own >> esc_wg_log) * kEscSlots;
    uint32_t h = esc_slot(u);
    int2 r = make_int2(0, 0);
    for (uint32_t i = 0; i < kEscSlots; ++i) {                      // (bounded: a marker without its entry cannot happen)
        const int4 e = tab[h];
        if ((uint32_t)e.x == u) { r = make_int2(e.y, e.z); break; }
        h = (h + 1u) & (kEscSlots - 1u);
    }
    return r;
}

// The same for the lanes of a wave that hold a marked entry (`marked`, u = the entry), on the scalar unit: one lane at a time, its
// index read into an SGPR, the probes as scalar loads, the pair written back into that lane.  No vector register beyond the two
// of the caller -- the tile kernel has none to spare (62 of 64 at eight waves per SIMD) -- and the path is cold.
typedef int bhw_v4i __attribute__((ext_vector_type(4)));
typedef const bhw_v4i __attribute__((address_space(4))) *BhwEscConstPtr;
__device__ __forceinline__ void esc_fix_wave(const void *__restrict__ esc, uint32_t esc_wg_log, uint32_t log2_entries, uint32_t u, bool marked, int2 &cs)
{
    const uint32_t E = 1u << log2_entries, half = E >> 1;
    uint64_t m = __ballot(marked);
    while (m) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        const uint32_t us = __builtin_amdgcn_readlane(u, l);
        uint32_t own = us <= half ? us : E - us;
        if (own > half - 1u) own = half - 1u;
        BhwEscConstPtr tab = (BhwEscConstPtr)(uintptr_t)(reinterpret_cast<const int4 *>(esc) + (size_t)(own >> esc_wg_log) * kEscSlots);
        uint32_t h = esc_slot(us);
        int32_t rc = 0, rs = 0;
#pragma unroll 1
        for (uint32_t i = 0; i < kEscSlots; ++i) {
            const bhw_v4i e = tab[h];
            if ((uint32_t)e.x == us) { rc = e.y; rs = e.z; break; }
            h = (h + 1u) & (kEscSlots - 1u);
        }
        // (one SGPR per vector instruction on gfx9: the lane select goes through M0)
        asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0" : "+v"(cs.x), "+v"(cs.y) : "s"(rc), "s"(rs), "s"(l) : "m0");
    }
}

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
    const int4 rec = ld_off<int4>(cfg.tab_coarse, (u >> d) << 4);
    if (FMT == 3 || FMT == 5 || (FMT < 0 && cfg.tab_dlog >= kNibbleFlag)) {
        const int2 p = tab_predict_nib(rec, u & ((1u << d) - 1u), d);
        const uint32_t e = ld_off<uint8_t>(table, idx);
        if (FMT == 5 || (FMT < 0 && cfg.tab_dlog >= kEscFlag)) {
            if ((e & 0xFu) == kEscMarker) return esc_lookup(cfg.tab_esc, cfg.esc_wg_log, cfg.phi_width - 2u, u);
        }
        const int2 n = nib_fields(e);
        return make_int2(p.x + n.x, p.y + n.y);
    }
    const int2 p = tab_predict(rec, u & ((1u << d) - 1u), d);
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
    static_assert(FMT == 2 || FMT == 3 || FMT == 5, "residual / nibble entries");
    int4 rec;
    if constexpr (LDS) rec = *reinterpret_cast<const int4 *>(lrec + (((theta >> rk.d) << 4) + bias));
    else rec = ld_off<int4>(cfg.tab_coarse, ((theta & emask) >> rk.d) << 4);
    // (FMT 5: the caller looks for the escape marker, once per harmonic -- BHW_TILE_HARMONIC)
    if constexpr (FMT == 3 || FMT == 5) {
        const int2 p = tab_predict_nib(rec, theta & rk.fmask, rk.d), n = nib_fields(e);
        return make_int2(p.x + n.x, p.y + n.y);
    } else {
        const int2 p = tab_predict(rec, theta & rk.fmask, rk.d);
        return make_int2(p.x + (int32_t)(int8_t)(e & 0xFFu), p.y + (int32_t)(int8_t)(e >> 8));
    }
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
    if constexpr (FMT == 2 || FMT == 3 || FMT == 5) return ((r & 3u) ? e : 0u) | (s - 1u);      // (FMT 3, 5: unused, natural layout)
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
    static_assert(FMT == 2 || FMT == 3 || FMT == 5, "residual / nibble entries");
    if constexpr (FMT == 3 || FMT == 5) {                           // nibble tables keep the natural layout (table_layout)
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
        if (FMT == 3 || (FMT < 0 && dlog >= kNibbleFlag)) {
            const int2 p = tab_predict_nib(rec, u & ((1u << d) - 1u), d);
            const uint32_t nc = (uint32_t)(c - p.x), ns = (uint32_t)(s - p.y);        // the biased fields
            if (check_flag && (nc | ns) > 15u) atomicOr(check_flag, 1u);
            reinterpret_cast<uint8_t *>(table)[idx] = (uint8_t)((nc & 0xFu) | ((ns & 0xFu) << 4));
        } else {
            const int2 p = tab_predict(rec, u & ((1u << d) - 1u), d);
            const int32_t dc = c - p.x, ds = s - p.y;
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
    if (win.apply_x) w = (int32_t)(((int64_t)__builtin_nontemporal_load(&win.apply_x[idx]) * (int64_t)w) >> win.apply_shift);   // (x is read once: streamed past the caches)
    out[idx] = w;
}

template <typename T>
__device__ __forceinline__ void stage_lut(const BhwCordicCfg &cfg, T *lut_s)
{
    if (threadIdx.x < 32) lut_s[threadIdx.x] = (T)cfg.lut[threadIdx.x];
    __syncthreads();
}
constexpr int kPrefixMax = 20;   // deepest rotation a 64-leaf group is followed to in phase 1
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
    void *tab_esc;            // nibble + escapes: the per-workgroup escape lists (BhwCordicCfg::tab_esc)
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
__device__ __forceinline__ int32_t wrap32(int32_t v, uint32_t bits)
{
    const uint32_t sh = 32u - bits;
    return (int32_t)((uint32_t)v << sh) >> sh;
}


// W-bit sums of the cosine-sum rules in 32-bit registers.  HLS rule: everything modulo 2^32, wrapped to W bits at the end.
// VHDL rule: the sum S of the terms b_k needs W+2 bits, so two 32-bit words are carried (both modulo 2^32):
//   b_k = wrap_W((P >> (W-1)) + ((P >> (W-2)) & 1))   == the slice-and-round of bh_win_7term.vhd:353-402 on the 2W-bit product P
//   hi  = sum of +/- (b_k >> 2),   sum = sum of +/- b_k = S mod 2^32
// and lo = sum - 4*hi = the sum of the +/- (b_k & 3), a small integer (|lo| <= 3 * 7) that 32-bit arithmetic returns exactly:
//   S = 4*hi + lo,  S>>2 = hi + (lo>>2),  (S>>1)&1 = (lo>>1)&1,  S>>1 = 2*hi + (lo>>1),  S&1 = lo&1
// (one shift and two adds per term; carrying lo itself costs a mask more per term)
struct Sum32 {
    int32_t hi, sum;
    __device__ __forceinline__ Sum32 &operator+=(const Sum32 &o) { hi += o.hi; sum += o.sum; return *this; }
};
__device__ __forceinline__ Sum32 sum32_first(int32_t a0) { return Sum32{a0 >> 2, a0}; }      // the a_0 term of the VHDL rule

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
        if (k & 1u) { acc.hi -= b >> 2; acc.sum -= b; }
        else        { acc.hi += b >> 2; acc.sum += b; }
    }
}

template <uint32_t COMBINE>
__device__ __forceinline__ int32_t w32_final(const Sum32 &acc, uint32_t W, uint32_t n_terms)
{
    if constexpr (COMBINE == BHW_COMBINE_HLS) return wrap32(acc.hi, W);
    const int32_t lo = (int32_t)((uint32_t)acc.sum - 4u * (uint32_t)acc.hi);
    if (n_terms == 2) return wrap32(2 * acc.hi + (lo >> 1) + (lo & 1), W);                        // hamming_win.vhd:214-228
    return wrap32(acc.hi + (lo >> 2) + ((lo >> 1) & 1), W);                                       // bh_win_7term.vhd:409-435
}

// VHDL rule when the exact sum S of the terms fits one 32-bit word (the caller checks the weights): the same roundings on S itself
__device__ __forceinline__ int32_t w32_final_exact(int32_t S, uint32_t W, uint32_t n_terms)
{
    const uint32_t sh = n_terms == 2 ? 1u : 2u;                  // hamming_win.vhd:214-228: (S >> 1) + (S & 1); bh_win_7term.vhd:409-435: (S >> 2) + ((S >> 1) & 1)
    return wrap32((S >> sh) + (int32_t)(((uint32_t)S >> (sh - 1u)) & 1u), W);
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
    } else if constexpr (FAST && MODE == 2) {
        // VHDL rule on the pre-shifted weight: q = (a * v) >> (W-2) is one v_mul_hi_i32, and the slice-and-round of
        // bh_win_7term.vhd:353-402,  (P >> (W-1)) + ((P >> (W-2)) & 1) = (q >> 1) + (q & 1) = (q + 1) >> 1  (floor shifts), wrapped
        // to W bits = bits 1 .. W of q + 1 sign-extended: v_mul_hi_i32, v_add, v_bfe_i32 per candidate (at W = 32 the field is
        // bits 1 .. 31, i.e. the arithmetic shift) instead of the 64-bit product, two shifts, mask, add and the two wrap shifts
        const uint32_t wb = W < 31u ? W : 31u;
        auto round_half = [&](int32_t q) -> int32_t { return __builtin_amdgcn_sbfe(q + 1, 1u, wb); };
        p0 = round_half(__mulhi(a, cs.x));
        p3 = round_half(__mulhi(a, cs.y));
        // either quadrant map without a branch per gather: two's complement a * (-v) == (-a) * v (the callers keep |a| < 2^(W-3)),
        // one's complement a * ~v; the mask and the weight are scalars
        const int32_t flip = cfg.ones_neg ? -1 : 0, na = cfg.ones_neg ? a : -a;
        p1 = round_half(__mulhi(na, cs.y ^ flip));
        p2 = round_half(__mulhi(na, cs.x ^ flip));
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
// VHDL rule: the W+2-bit sums are carried as (hi, sum) (Sum32): a term b adds b >> 2 to hi and b itself to sum -- the high parts of
// the four candidates once, then the same scalar-branched accumulate for each word.
template <int K, int OFF, int QBASE, int QBITS>
__device__ __forceinline__ void tile_accumulate_uniform(uint32_t q, const int32_t (&sv)[4], Sum32 (&acc)[4])
{
    int32_t svh[4], h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { svh[i] = sv[i] >> 2; h[i] = acc[i].hi; l[i] = acc[i].sum; }
    if constexpr (QBITS == 1) {
        // both words behind ONE scalar compare-and-branch (the scalar unit is what this kernel runs short of)
        auto S = [&](int j, int Q) -> int32_t { return sv[(j * K + OFF + Q) & 3]; };
        auto H = [&](int j, int Q) -> int32_t { return svh[(j * K + OFF + Q) & 3]; };
#define BHW_UNI2W(OP)                                                                                                  \
        asm("s_cmp_eq_u32 %24, %25\n\ts_cbranch_scc0 1f\n\t"                                                          \
            OP " %0, %0, %8\n\t" OP " %1, %1, %9\n\t" OP " %2, %2, %10\n\t" OP " %3, %3, %11\n\t"                      \
            OP " %4, %4, %12\n\t" OP " %5, %5, %13\n\t" OP " %6, %6, %14\n\t" OP " %7, %7, %15\n\ts_branch 2f\n1:\n\t" \
            OP " %0, %0, %16\n\t" OP " %1, %1, %17\n\t" OP " %2, %2, %18\n\t" OP " %3, %3, %19\n\t"                    \
            OP " %4, %4, %20\n\t" OP " %5, %5, %21\n\t" OP " %6, %6, %22\n\t" OP " %7, %7, %23\n2:"                     \
            : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3])           \
            : "v"(H(0, QBASE)), "v"(H(1, QBASE)), "v"(H(2, QBASE)), "v"(H(3, QBASE)),                                  \
              "v"(S(0, QBASE)), "v"(S(1, QBASE)), "v"(S(2, QBASE)), "v"(S(3, QBASE)),                                  \
              "v"(H(0, QBASE + 1)), "v"(H(1, QBASE + 1)), "v"(H(2, QBASE + 1)), "v"(H(3, QBASE + 1)),                  \
              "v"(S(0, QBASE + 1)), "v"(S(1, QBASE + 1)), "v"(S(2, QBASE + 1)), "v"(S(3, QBASE + 1)), "s"(q), "n"(QBASE) : "scc")
        if constexpr (K & 1) BHW_UNI2W("v_sub_u32"); else BHW_UNI2W("v_add_u32");
#undef BHW_UNI2W
    } else {
        tile_accumulate_uniform<K, OFF, QBASE, QBITS>(q, svh, h);
        tile_accumulate_uniform<K, OFF, QBASE, QBITS>(q, sv, l);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i].hi = h[i]; acc[i].sum = l[i]; }
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
        if (K & 1) { acc[j].hi -= b >> 2; acc[j].sum -= b; }
        else       { acc[j].hi += b >> 2; acc[j].sum += b; }
    }
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
