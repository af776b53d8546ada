// bhw_tile9.hip -- table strategy, pass 2: the 15-run tile kernel of the long seven-term windows, on an instruction diet
//
// Part of the hand-written HIP kernels for gfx950 (MI355X, CDNA4) behind include/bhw.h.  Hot path of the reference: phase
// accumulator -> CORDIC rotation chain (or Taylor LUT) -> weighted N-term cosine sum -> int32 coefficient (SURVEY section 8a
// rows a1-a11).  Integer semantics follow SURVEY App. A; reference lines are cited at each step.
//
// k_table_combine_tile<15, ...> (bhw_combine.hip) is bound by vector-instruction ISSUE: with every memory operation removed it
// still takes 48 of its 61 us (profiles/r04_ab_combine_floor.txt), and priced with the measured issue costs of gfx950
// (profiles/r02_ubench_gfx950.txt: a VOP2 on VGPR / inline-constant operands 2.5 - 2.7 cycles per wave-instruction per SIMD, any
// VOP3 form or ANY instruction with an SGPR or literal operand 4.1 - 4.5) its hot path adds up to ~2 600 cycles per wave x 42.7
// waves per SIMD = 46 us.  Three quarters of that is not the cosine sum but the table decode around it.  This kernel is the same
// algorithm -- the same tile plan, lane -> r mapping, eight coefficients per lane and run, records of the wave's cells staged in
// shared memory, wave-uniform quadrants, one-instruction products -- with the decode rebuilt around the cheap encodings:
//   * cell size 2^9 as a compile-time fact (launched only for such tables: every 2^26-point / 32-bit window), so the per-gather shifts
//     take immediates instead of an SGPR;
//   * the record line as ONE v_mul_hi_i32 per coordinate (doubled slopes in the records, the in-cell position as a left-aligned
//     word: two immediate shifts shared by both coordinates) instead of v_and (SGPR) + 2 x (v_mul_i32_i24 + v_ashrrev (SGPR));
//   * unsigned nibble fields (bias folded into the records): v_and 15 / v_lshrrev 4 instead of two v_bfe_i32;
//   * the staged records in a four-slot RING per set, indexed by the low two bits of the cell number: the address is shift,
//     and-or, an immediate offset -- no per-set bias word, which also removes the ~110 scalar instructions per wave that computed
//     the 27 of them;
//   * the 27 wave-uniform quadrants as two ballot words computed by 27 lanes once, tested with s_bitcmp1_b64 in the
//     scalar-branched accumulates: no extraction shifts / masks / compares (the CU's scalar unit serves four SIMDs);
//   * a_0 in a vector register (the first harmonic's 24 subtractions otherwise each carry an SGPR operand).
// ~2 050 issue cycles per wave instead of ~2 600, ~190 scalar instructions instead of 343.
// Everything else (MASKED image subsets, caller-scaled weights beyond the one-instruction products, other cell sizes and table
// formats, one- and three-run tiles) stays with k_table_combine_tile.
#include "bhw_device.h"

namespace {

#ifndef BHW_T9_WAVES
#define BHW_T9_WAVES 8                                              // waves per SIMD the register allocation aims at
#endif
constexpr int kT9Threads = kTileLanes;                              // one part (192 lanes, three waves) of a tile per workgroup
constexpr int kT9Runs = 3;                                          // inv3-siblings per thread
constexpr uint32_t kT9D = 9;                                        // log2 of the cell size this kernel is compiled for
constexpr int kT9Slots = 4;                                         // ring slots per set: K * 191 + 511 < 4 * 512 for K <= 6
constexpr int kT9SetsPerRun = 9;                                    // (K, g): (1,0) (1,1) (2) (3,0) (3,1) (4) (5,0) (5,1) (6)
constexpr int kT9Sets = kT9Runs * kT9SetsPerRun;                    // 27 per wave
constexpr int kT9SlotLog = 9;                                       // slot stride in bytes: 2^9, the cell size in entries -- slot offset = angle & 0x600, no shift
constexpr int kT9WaveLog = 11;                                      // ring bytes per wave: 4 slots x 512 (27 sets x 16 = 432 used of each), the wave's base rides in an and-or
static_assert(kT9Sets * 16 <= (1 << kT9SlotLog) && (kT9Slots << kT9SlotLog) <= (1 << kT9WaveLog) && kT9SlotLog == (int)kT9D, "ring of one wave");
static_assert(6 * (kTileLanes - 1) + (1 << kT9D) - 1 < kT9Slots << kT9D, "a run's entries of one harmonic touch at most kT9Slots cells");
__host__ __device__ constexpr int t9_set(int K, int g) { return K == 1 ? g : K == 2 ? 2 : K == 3 ? 3 + g : K == 4 ? 5 : K == 5 ? 6 + g : 8; }

// One nibble-table entry for harmonic K at byte offset boff (= the entry index: natural layout).  K = 2 / 4: the same byte as
// the low byte of a unit-stride short / dword load (4.7 instead of 16.4 cycles of the CU's address path, profiles/r02_ubench_vmem.txt;
// boff is a multiple of K and boff + K <= E).
template <int K>
__device__ __forceinline__ uint32_t t9_load(const void *__restrict__ table, uint32_t boff)
{
    if constexpr (K == 2) return (uint32_t)ld_off<uint16_t>(table, boff);
    else if constexpr (K == 4) return ld_off<uint32_t>(table, boff);
    else return (uint32_t)ld_off<uint8_t>(table, boff);
}

// acc[j] -/+= sv[(j*K + OFF + q) & 3] with the wave-uniform quadrant q of set BIT read from the two ballot words (bit BIT of qm0 /
// qm1 = bit 0 / 1 of q): one s_bitcmp + s_cbranch per level, the adds of every case inside ONE statement (as C++ control flow the
// compiler sinks them below the join and leaves a register move per slot).  QBASE / QBITS: ring_qbase / ring_qbits (bhw_device.h).
template <int K, int OFF, int QBASE, int QBITS, int BIT>
__device__ __forceinline__ void t9_accumulate(uint64_t qm0, uint64_t qm1, const int32_t (&sv)[4], int32_t (&acc)[4])
{
    auto S = [&](int j, int Q) -> int32_t { return sv[(j * K + OFF + Q) & 3]; };
#ifdef BHW_X_NOBRANCH
    constexpr bool kFixed = true;                                   // (timing experiment: every quadrant taken as QBASE, no scalar branch)
#else
    constexpr bool kFixed = false;
#endif
    if constexpr (QBITS == 0 || kFixed) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = (K & 1) ? acc[j] - S(j, QBASE) : acc[j] + S(j, QBASE);
    } else if constexpr (QBITS == 1) {
        // q is QBASE or QBASE + 1: its low bit tells which
#define BHW_T9_UNI2(OP, CMP)                                                                                           \
        asm(CMP " %[qm], %[bit]\n\ts_cbranch_scc0 1f\n\t"                                                             \
            OP " %0, %0, %4\n\t" OP " %1, %1, %5\n\t" OP " %2, %2, %6\n\t" OP " %3, %3, %7\n\ts_branch 2f\n1:\n\t"      \
            OP " %0, %0, %8\n\t" OP " %1, %1, %9\n\t" OP " %2, %2, %10\n\t" OP " %3, %3, %11\n2:"                        \
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                                   \
            : "v"(S(0, QBASE)), "v"(S(1, QBASE)), "v"(S(2, QBASE)), "v"(S(3, QBASE)),                                  \
              "v"(S(0, QBASE + 1)), "v"(S(1, QBASE + 1)), "v"(S(2, QBASE + 1)), "v"(S(3, QBASE + 1)), [qm] "s"(qm0), [bit] "n"(BIT) : "scc")
        if constexpr ((K & 1) && (QBASE & 1)) BHW_T9_UNI2("v_sub_u32", "s_bitcmp1_b64");
        else if constexpr (K & 1) BHW_T9_UNI2("v_sub_u32", "s_bitcmp0_b64");
        else if constexpr (QBASE & 1) BHW_T9_UNI2("v_add_u32", "s_bitcmp1_b64");
        else BHW_T9_UNI2("v_add_u32", "s_bitcmp0_b64");
#undef BHW_T9_UNI2
    } else {
#define BHW_T9_UNI4(OP)                                                                                                \
        asm("s_bitcmp1_b64 %[q1], %[bit]\n\ts_cbranch_scc1 2f\n\ts_bitcmp1_b64 %[q0], %[bit]\n\ts_cbranch_scc1 1f\n\t"   \
            OP " %0, %0, %4\n\t" OP " %1, %1, %5\n\t" OP " %2, %2, %6\n\t" OP " %3, %3, %7\n\ts_branch 4f\n1:\n\t"      \
            OP " %0, %0, %8\n\t" OP " %1, %1, %9\n\t" OP " %2, %2, %10\n\t" OP " %3, %3, %11\n\ts_branch 4f\n2:\n\t"   \
            "s_bitcmp1_b64 %[q0], %[bit]\n\ts_cbranch_scc1 3f\n\t"                                                     \
            OP " %0, %0, %12\n\t" OP " %1, %1, %13\n\t" OP " %2, %2, %14\n\t" OP " %3, %3, %15\n\ts_branch 4f\n3:\n\t" \
            OP " %0, %0, %16\n\t" OP " %1, %1, %17\n\t" OP " %2, %2, %18\n\t" OP " %3, %3, %19\n4:"                      \
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])                                                   \
            : "v"(S(0, 0)), "v"(S(1, 0)), "v"(S(2, 0)), "v"(S(3, 0)), "v"(S(0, 1)), "v"(S(1, 1)), "v"(S(2, 1)), "v"(S(3, 1)), \
              "v"(S(0, 2)), "v"(S(1, 2)), "v"(S(2, 2)), "v"(S(3, 2)), "v"(S(0, 3)), "v"(S(1, 3)), "v"(S(2, 3)), "v"(S(3, 3)), \
              [q0] "s"(qm0), [q1] "s"(qm1), [bit] "n"(BIT) : "scc")
        if constexpr (K & 1) BHW_T9_UNI4("v_sub_u32"); else BHW_T9_UNI4("v_add_u32");
#undef BHW_T9_UNI4
    }
}

// set = 9 b + si (< 27): harmonic K and unmasked angle of the first lane of run b (starts st0 .. st2) in set si.
// (A plain function of scalars on purpose: as a lambda the captured starts form an aggregate, a select among its members becomes a
// dynamically indexed load, and the aggregate then lives in scratch or shared memory.)
__device__ __forceinline__ void t9_set_of(uint32_t set, uint32_t st0, uint32_t st1, uint32_t st2, uint32_t H, uint32_t &K, uint32_t &th0)
{
    const uint32_t b = (set * 57u) >> 9, si = set - 9u * b;
    uint32_t st = st2;
    st = b == 1u ? st1 : st;
    st = b == 0u ? st0 : st;
    K = (uint32_t)(0x655433211ull >> (4u * si)) & 15u;                        // 1 1 2 3 3 4 5 5 6
    const uint32_t g = (0x92u >> si) & 1u;                                    // sets 1, 4, 7
    th0 = K * (st + g * H);
}

// what every harmonic of the hot path shares (all wave-uniform except rr, wbase and the sums)
struct T9Ctx {
    const void *table;
    const char *lring;        // the workgroup's ring (LDS)
    uint32_t H, emask;
    uint64_t qm0, qm1;        // bit 9 b + t9_set(K, g): bit 0 / 1 of the quadrant of harmonic K, image g, run b
    uint32_t W;
    uint32_t st[kT9Runs];     // first ring index of each run
};

// Run B's share of harmonic K: decode of its one or two gathers (words already loaded), products, accumulate.
// Run B's share of harmonic K: decode of its one or two gathers (words already loaded).  For an odd harmonic the second
// half-period image g = 1 reads entry K (r + E/2) = K r + K E/2: the same low bits (K E/2 is a multiple of 2^11), so the position
// inside the cell and the ring slot are those of g = 0 -- only the set (an immediate offset) differs.
template <int K, int MODE, int FMT, int B>
__device__ __forceinline__ void t9_run(const T9Ctx &cx, const uint32_t wbase, const uint32_t (&rr)[kT9Runs], const uint32_t (&e)[kT9Runs][2],
                                       int2 (&cs)[kT9Runs][2], uint32_t &esc_min)
{
    constexpr int NG = (K & 1) ? 2 : 1;
    const uint32_t t = (uint32_t)K * rr[B];       // the unmasked angle: whole turns only move the quadrant
    // position inside the cell, left-aligned below the sign bit: hi32(2 dc * xs) = (dc * f) >> 9 (tab_predict_nib)
    uint32_t xs = t << (32u - kT9D);
    asm("" : "+v"(xs));                           // (two immediate shifts, VOP2: folded, the second becomes an and with a 32-bit literal)
    xs >>= 1;
    // the records: slot (cell & 3) of the sets' rings -- slots 512 bytes apart, so the slot's offset is bits 9, 10 of the angle as
    // they are; the wave's base rides in the and-or, the set's offset in the instruction
    const uint32_t la = (t & (uint32_t)((kT9Slots - 1) << kT9SlotLog)) | wbase;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#ifdef BHW_X_NOLDS
        const int4 rec = make_int4((int32_t)la, (int32_t)(t >> 1), 3, 5 + g);    // (timing experiment: no LDS read)
#else
        const int4 rec = *reinterpret_cast<const int4 *>(cx.lring + la + (uint32_t)((B * kT9SetsPerRun + t9_set(K, g)) * 16));
#endif
        const int32_t pc = __mulhi(rec.z, (int32_t)xs), ps = __mulhi(rec.w, (int32_t)xs);
        const uint32_t w = e[B][g];
        const uint32_t nc = w & 15u, ns = (K == 2 || K == 4) ? (w >> 4) & 15u : w >> 4;      // (byte loads: nothing above bit 7)
        cs[B][g] = make_int2(rec.x + pc + (int32_t)nc, rec.y + ps + (int32_t)ns);
        if constexpr (FMT == 5) esc_min = nc < esc_min ? nc : esc_min;
    }
}

// The four candidates c, -s, -c, s times the weight (hls/windows/win_function.cpp:368-373 | src/bh_win_7term.vhd:353-402); the quadrant
// picks among them in the accumulate.  `aK` is the weight shifted left by 34 - W (tile_harmonic FAST).
// VHDL32: the VHDL rule at W = 32.  There the slice-and-round of bh_win_7term.vhd:353-402 on the 2W-bit product P = a_k v,
//     b = (P >> (W-1)) + ((P >> (W-2)) & 1) = (q + 1) >> 1,  q = P >> (W-2) = hi32((a_k << 2) v)          (floor shifts; tile_harmonic),
// is  floor((a2 v / 2^31 + 1) / 2) = hi32(a2 v + 2^31)  with  a2 = a_k << 1  -- ONE v_mad_i64_i32 per candidate instead of v_mul_hi_i32,
// v_add, v_bfe_i32 (nothing to wrap: b fits the 32-bit word).
template <int K, int MODE, bool VHDL32>
__device__ __forceinline__ void t9_products(const BhwCordicCfg &cfg, const T9Ctx &cx, const int32_t aK, const int2 cs, int32_t (&sv)[4])
{
    if constexpr (MODE == 2 && VHDL32) {
        const int32_t a2 = aK >> 1;                                           // (aK = a_k << 2 at W = 32)
        auto rp = [](int32_t w, int32_t v) -> int32_t { return (int32_t)(((int64_t)w * (int64_t)v + (int64_t)0x80000000ll) >> 32); };
        // either quadrant map: two's complement a (-v) == (-a) v, one's complement a ~v (the mask and the weight are scalars)
        const int32_t flip = cfg.ones_neg ? -1 : 0, na2 = cfg.ones_neg ? a2 : -a2;
        sv[0] = rp(a2, cs.x);
        sv[3] = rp(a2, cs.y);
        sv[1] = rp(na2, cs.y ^ flip);
        sv[2] = rp(na2, cs.x ^ flip);
    } else if constexpr (MODE == 1) {
        // HLS rule on the cpp model's one's-complement map (cpp/cordic_sincos.cpp:70-86): the candidates are a c, a ~s, a ~c, a s, and
        // a ~v = a (-v - 1) = (-a) v + (-a) exactly -- ONE v_mad_i64_i32 (high word) instead of v_not + v_mul_hi_i32
        const int32_t na = -aK;
        auto cp = [](int32_t w, int32_t v) -> int32_t { return (int32_t)(((int64_t)w * (int64_t)v + (int64_t)w) >> 32); };
        sv[0] = __mulhi(aK, cs.x);
        sv[3] = __mulhi(aK, cs.y);
        sv[1] = cp(na, cs.y);
        sv[2] = cp(na, cs.x);
    } else tile_harmonic<K, MODE, 0, 0, true>(cfg, aK, cx.W, cs, 0u, sv);
}

template <int K, int MODE, bool VHDL32, int B>
__device__ __forceinline__ void t9_sum(const BhwCordicCfg &cfg, const T9Ctx &cx, const int32_t aK, const int2 (&cs)[kT9Runs][2], int32_t (&acc)[kT9Runs][2][4])
{
    constexpr int NG = (K & 1) ? 2 : 1;
    int32_t sv[4];
    t9_products<K, MODE, VHDL32>(cfg, cx, aK, cs[B][0], sv);
    t9_accumulate<K, 0, ring_qbase(K, 0), ring_qbits(K, 0), B * kT9SetsPerRun + t9_set(K, 0)>(cx.qm0, cx.qm1, sv, acc[B][0]);
    if constexpr (NG == 1) {
        // even K: the second half-period image reads the same entry K/2 quadrants further on
        t9_accumulate<K, K / 2, ring_qbase(K, 0), ring_qbits(K, 0), B * kT9SetsPerRun + t9_set(K, 0)>(cx.qm0, cx.qm1, sv, acc[B][1]);
    } else {
        t9_products<K, MODE, VHDL32>(cfg, cx, aK, cs[B][1], sv);
        t9_accumulate<K, 0, ring_qbase(K, 1), ring_qbits(K, 1), B * kT9SetsPerRun + t9_set(K, 1)>(cx.qm0, cx.qm1, sv, acc[B][1]);
    }
}

// Nibble + escapes, cold path: harmonic K of this wave read at least one marked entry.  Every gather of the harmonic is formed
// again from the table (record from the record array, the byte itself), once as the hot path saw it -- the marker's fields taken at
// face value -- and once with the listed pair in the marked lanes (esc_fix_wave: scalar unit, no extra vector registers); the
// difference of the two contributions goes into the sums.  Unmarked lanes add zero.  Quadrants per lane (theta >> lq: the
// wave-uniform value of the hot path, which only runs where no harmonic crosses a quarter turn inside a run).
template <int K, int MODE, bool VHDL32>
__device__ __forceinline__ void t9_esc_repair(const BhwCordicCfg &cfg, const T9Ctx &cx, const uint32_t (&rr)[kT9Runs], const int32_t aK, int32_t (&acc)[kT9Runs][2][4])
{
    constexpr int NG = (K & 1) ? 2 : 1;
    const uint32_t lq = cfg.phi_width - 2u;
    // (unrolled: the sums are registers)
#pragma unroll
    for (int b = 0; b < kT9Runs; ++b) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint32_t th = (uint32_t)K * (rr[b] + (uint32_t)g * cx.H), u = th & cx.emask;
            const uint32_t byte = ld_off<uint8_t>(cx.table, u);
            const bool marked = (byte & 0xFu) == kEscMarker;
            if (__builtin_amdgcn_ballot_w64(marked) == 0ull) continue;                 // wave-uniform
            const int2 p = tab_predict_nib(ld_off<int4>(cfg.tab_coarse, (u >> kT9D) << 4), u & ((1u << kT9D) - 1u), kT9D), n = nib_fields(byte);
            const int2 wrong = make_int2(p.x + n.x, p.y + n.y);
            int2 right = wrong;
            esc_fix_wave(cfg.tab_esc, cfg.esc_wg_log, lq, u, marked, right);
            int32_t sw[4], sr[4], d[4];
            t9_products<K, MODE, VHDL32>(cfg, cx, aK, wrong, sw);
            t9_products<K, MODE, VHDL32>(cfg, cx, aK, right, sr);
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = (K & 1) ? sw[i] - sr[i] : sr[i] - sw[i];   // odd harmonics are subtracted
            const uint32_t q = th >> lq;
            auto pick = [&](uint32_t i) -> int32_t { i &= 3u; return i == 0u ? d[0] : i == 1u ? d[1] : i == 2u ? d[2] : d[3]; };
            // image j sits K * j quadrants after image 0; even K: the h = 1 image K / 2 quadrants further on (t9_sum)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (NG == 2) acc[b][g][j] += pick(q + (uint32_t)(j * K));
                else {
                    acc[b][0][j] += pick(q + (uint32_t)(j * K));
                    acc[b][1][j] += pick(q + (uint32_t)(j * K + K / 2));
                }
            }
        }
    }
}

// The gathers of harmonic K: one table byte per (run, image g) -- requested here, decoded in t9_finish.  Image g = 1 of an odd
// harmonic sits a wave-uniform distance from image g = 0 (no run of this path crosses a quarter turn): the same vector offset
// against a scalar base moved by that distance -- no vector instruction for its address at all.
template <int K>
__device__ __forceinline__ void t9_issue(const T9Ctx &cx, const uint32_t (&rr)[kT9Runs], uint32_t (&e)[kT9Runs][2])
{
#pragma unroll
    for (int b = 0; b < kT9Runs; ++b) {
        const uint32_t th = (uint32_t)K * rr[b];
#ifdef BHW_X_NOLOAD
        e[b][0] = th >> 3;                                                    // (timing experiment: no table load at all)
        e[b][1] = th >> 5;
#else
#ifdef BHW_X_HOT
        const uint32_t voff = th & 0xFFFCu;                                   // (timing experiment: every gather inside 64 KiB)
#else
#ifdef BHW_X_NOWRAPS
        const uint32_t voff = th & cx.emask;                                  // (timing experiment: waves that wrap stay on this path -- every offset masked)
#else
        const uint32_t voff = K <= 2 ? th : th & cx.emask;                    // K (r + g E/2) < E for K <= 2: nothing to wrap
#endif
#endif
        e[b][0] = t9_load<K>(cx.table, voff);
        if constexpr ((K & 1) != 0) {
            const uint32_t s0 = (uint32_t)K * cx.st[b], s1 = s0 + (uint32_t)K * cx.H;                   // scalars: the run's first lane
#ifdef BHW_X_HOT
            const int32_t dist = 64;
#else
            const int32_t dist = (int32_t)(s1 & cx.emask) - (int32_t)(s0 & cx.emask);
#endif
#ifdef BHW_X_NOWRAPS
            e[b][1] = t9_load<K>(cx.table, (voff + (uint32_t)dist) & cx.emask & ~3u);
#else
            e[b][1] = t9_load<K>(reinterpret_cast<const char *>(cx.table) + dist, voff);
#endif
        }
#endif
    }
}

// Nibble + escapes: the marker is looked for AFTER the harmonic has been summed (one v_min per gather, one vote and one scalar branch
// per harmonic), so that decode and products stay one scheduling region; a wave that met one -- a few per cent of them: the listed
// entries are a few per 100 000 -- repairs the sums of the marked lanes (t9_esc_repair).
template <int K, int MODE, int FMT, bool VHDL32>
__device__ __forceinline__ void t9_finish(const BhwCordicCfg &cfg, const T9Ctx &cx, const uint32_t wbase, const uint32_t (&rr)[kT9Runs], const uint32_t (&e)[kT9Runs][2],
                                          const int32_t aK, int32_t (&acc)[kT9Runs][2][4])
{
#ifdef BHW_X_NOALU
    {   // (timing experiment: the pass's memory operations -- the same gathers, the same 24 stores -- without its arithmetic: every sum
        // takes one add per gather; no records, no products)
#pragma unroll
        for (int b = 0; b < kT9Runs; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[b][h][j] += (int32_t)e[b][(K & 1) ? h : 0];
        return;
    }
#endif
    int2 cs[kT9Runs][2];
    uint32_t esc_min = 15u;
    t9_run<K, MODE, FMT, 0>(cx, wbase, rr, e, cs, esc_min);
    t9_run<K, MODE, FMT, 1>(cx, wbase, rr, e, cs, esc_min);
    t9_run<K, MODE, FMT, 2>(cx, wbase, rr, e, cs, esc_min);
    // (VHDL32: one run at a time -- the 64-bit results of its v_mad_i64_i32 products take register pairs, and left alone the scheduler
    // forms all twelve candidates of the three runs before the first accumulate)
    t9_sum<K, MODE, VHDL32, 0>(cfg, cx, aK, cs, acc);
    if constexpr (VHDL32) __builtin_amdgcn_sched_barrier(0);
    t9_sum<K, MODE, VHDL32, 1>(cfg, cx, aK, cs, acc);
    if constexpr (VHDL32) __builtin_amdgcn_sched_barrier(0);
    t9_sum<K, MODE, VHDL32, 2>(cfg, cx, aK, cs, acc);
    if constexpr (FMT == 5) {
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(esc_min == kEscMarker) != 0ull, 0)) t9_esc_repair<K, MODE, VHDL32>(cfg, cx, rr, aK, acc);
    }
#ifdef BHW_X_PAD
    {   // (timing experiment: BHW_X_PAD dummy vector instructions per harmonic, independent of everything else; BHW_X_PADOP 0: VOP2 on
        // registers, 1: a VOP3 form)
        uint32_t d0 = rr[0], d1 = rr[1];
#pragma unroll
        for (int i = 0; i < BHW_X_PAD / 2; ++i) {
#if BHW_X_PADOP
            asm volatile("v_add3_u32 %0, %0, %1, %1\n\tv_add3_u32 %1, %1, %0, %0" : "+v"(d0), "+v"(d1));
#else
            asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %0" : "+v"(d0), "+v"(d1));
#endif
        }
    }
#endif
}

// A wave one of whose runs wraps around the ring, or one of whose harmonics crosses a quarter turn inside a run (1.6 % of the
// waves of a 2^26-point window): records from the table's record array, per-lane quadrants -- the general arithmetic of
// k_table_combine_tile's global path.  Nibble + escapes: every entry of the harmonic is read as plain nibbles first and the listed
// pairs are fetched afterwards, NOT by a test per gather (tab_fetch): that test is control flow between a lane's loads, it
// serialises the 54 of them, and these few waves then live so long that they hold the whole pass back -- 71.6 against 65.2 us for
// the VHDL product window with every wave forced onto the hot path (profiles/r05_ab_tile9_slow_path.txt).
template <int K, int MODE, int FMT>
__device__ __forceinline__ void t9_harmonic_slow(const BhwCordicCfg &cfg, const void *__restrict__ table, const uint32_t (&rr)[kT9Runs], uint32_t lq,
                                                 const int32_t aK, uint32_t W, int32_t (&acc)[kT9Runs][2][4])
{
    constexpr int NG = (K & 1) ? 2 : 1;
    const uint32_t E = 1u << lq, emask = E - 1u, H = E >> 1;
    int2 cs[kT9Runs][2];
    uint32_t marked = 0u;                                              // bit 2 b + g: that gather read a marker
#pragma unroll
    for (int b = 0; b < kT9Runs; ++b)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const uint32_t u = ((uint32_t)K * (rr[b] + (uint32_t)g * H)) & emask;
            const uint32_t e = ld_off<uint8_t>(table, u);
            const int2 p = tab_predict_nib(ld_off<int4>(cfg.tab_coarse, (u >> kT9D) << 4), u & ((1u << kT9D) - 1u), kT9D), n = nib_fields(e);
            cs[b][g] = make_int2(p.x + n.x, p.y + n.y);
            if constexpr (FMT == 5) marked |= ((e & 0xFu) == kEscMarker ? 1u : 0u) << (2 * b + g);
        }
    if constexpr (FMT == 5) {
        if (__builtin_expect(marked != 0u, 0)) {                       // (per lane: the few that hold a listed entry)
#pragma unroll
            for (int b = 0; b < kT9Runs; ++b)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    if ((marked >> (2 * b + g)) & 1u)
                        cs[b][g] = esc_lookup(cfg.tab_esc, cfg.esc_wg_log, lq, ((uint32_t)K * (rr[b] + (uint32_t)g * H)) & emask);
        }
    }
#pragma unroll
    for (int b = 0; b < kT9Runs; ++b) {
        int32_t sv[4];
        const uint32_t t0 = (uint32_t)K * rr[b];
        tile_harmonic<K, MODE, ring_qbase(K, 0), ring_qbits(K, 0), true>(cfg, aK, W, cs[b][0], t0 >> lq, sv);
        tile_accumulate<K, 0, true>(sv, acc[b][0]);
        if constexpr (NG == 2) {
            const uint32_t t1 = (uint32_t)K * (rr[b] + H);
            tile_harmonic<K, MODE, ring_qbase(K, 1), ring_qbits(K, 1), true>(cfg, aK, W, cs[b][1], t1 >> lq, sv);
            tile_accumulate<K, 0, true>(sv, acc[b][1]);
        } else tile_accumulate<K, K / 2, true>(sv, acc[b][1]);
    }
}

// VHDL32: MODE 2 at dat_width 32 (t9_products); false for the HLS rule.
template <int MODE, int FMT, bool APPLY, bool VHDL32 = false>
__global__ __launch_bounds__(kT9Threads) __attribute__((amdgpu_waves_per_eu(BHW_T9_WAVES))) void k_tile9(BhwCordicCfg cfg, BhwWinCfg win, BhwTilePlan tp,
                                                                                           const void *__restrict__ table, int32_t *__restrict__ out)
{
    static_assert(FMT == 3 || FMT == 5, "one-byte entries");

    const uint32_t lq = cfg.phi_width - 2;
    const uint32_t E = 1u << lq, emask = E - 1u, H = E >> 1, hmask = H - 1u;
    const uint32_t W = cfg.dat_width;
    // workgroup -> (tile, part).  Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share an L2); the tiles are
    // renumbered so that each XCD sweeps a contiguous eighth of the ring and the five parts of a tile follow each other on one XCD.
    constexpr uint32_t kParts = kTileThreads / kT9Threads;
    uint32_t tile = blockIdx.x / kParts, part = blockIdx.x % kParts;
    {
        const uint32_t per = (gridDim.x / kParts) >> 3, main = (per << 3) * kParts;
        if (blockIdx.x < main) {
            const uint32_t j = blockIdx.x >> 3;
            tile = (blockIdx.x & 7u) * per + j / kParts;
            part = j % kParts;
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // lane -> r inside a run, rotated by the start's offset inside a 64-element block so that every wave stores aligned 256-byte chunks
    uint32_t starts[kT9Runs], rr[kT9Runs];
#pragma unroll
    for (int b = 0; b < kT9Runs; ++b) {
        const uint32_t start = ((tile + tp.tile0) * (uint32_t)kT9Threads + tp.offs[part * kT9Runs + b]) & hmask;
        starts[b] = start;
        rr[b] = (start + (threadIdx.x + (uint32_t)kT9Threads - (start & 63u)) % (uint32_t)kT9Threads) & hmask;
    }
    // One lane per (run, set): does its harmonic cross a quarter turn inside the run (or, lanes 27 .. 29, the run wrap around the
    // ring)?  And which quadrant is it in: two ballots give the 27 wave-uniform quadrants as bit vectors for s_bitcmp1_b64.
    bool wraps;
    uint64_t qm0, qm1;
    const uint32_t cellmask = (E >> kT9D) - 1u;
    const uint32_t st0 = starts[0], st1 = starts[1], st2 = starts[2];
    {
        uint32_t K, th0;
        t9_set_of(lane < 27u ? lane : 0u, st0, st1, st2, H, K, th0);
        uint32_t st = st2;
        st = lane == 28u ? st1 : st;
        st = lane == 27u ? st0 : st;
        const bool w = lane < 27u ? (th0 & emask) + K * (uint32_t)(kT9Threads - 1) > emask : lane < 30u ? st + (uint32_t)kT9Threads > H : false;
        wraps = __builtin_amdgcn_ballot_w64(w) != 0ull;
#ifdef BHW_X_NOWRAPS
        wraps = false;                                                        // (timing experiment: every wave on the hot path)
#endif
        const uint32_t q = th0 >> lq;
        qm0 = __builtin_amdgcn_ballot_w64((q & 1u) != 0u);
        qm1 = __builtin_amdgcn_ballot_w64((q & 2u) != 0u);
    }
    __shared__ int4 ring[(kT9Threads / 64) << (kT9WaveLog - 4)];
#ifdef BHW_X_LDSPAD
    __shared__ int4 lds_pad[BHW_X_LDSPAD / 16];                      // (timing experiment: fewer workgroups per CU)
    if (cfg.phi_width == 77u) lds_pad[threadIdx.x] = make_int4(1, 2, 3, 4);
#endif
    if (!wraps) {
        // the records of the 27 sets' cells: slot (cell & 3) of the set's ring holds the cell's record, four cells from the first lane's on
        int4 rec[2];
        uint32_t where[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t idx = lane + 64u * (uint32_t)h;                    // (set, slot)
            const uint32_t set = idx < (uint32_t)(kT9Sets * kT9Slots) ? idx >> 2 : 0u, j = idx & 3u;
            uint32_t K, th0;
            t9_set_of(set, st0, st1, st2, H, K, th0);
            const uint32_t cell0 = th0 >> kT9D, cell = cell0 + ((j - cell0) & 3u);
            rec[h] = ld_off<int4>(cfg.tab_coarse, (cell & cellmask) << 4);
            where[h] = (wave << (kT9WaveLog - 4)) + (j << (kT9SlotLog - 4)) + set;
        }
        ring[where[0]] = rec[0];
        if (lane + 64u < (uint32_t)(kT9Sets * kT9Slots)) ring[where[1]] = rec[1];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                                      // the wave's own LDS writes are ordered before its reads
    }

    int32_t acc[kT9Runs][2][4];
    {
        int32_t a0v;                                                          // a_0 in a vector register: the first subtractions carry no SGPR
        asm volatile("v_mov_b32 %0, %1" : "=v"(a0v) : "s"(win.aa[0]));
#pragma unroll
        for (int b = 0; b < kT9Runs; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[b][h][j] = a0v;
    }
    // pre-shifted weights: (a_k * v) >> (W-2) is the high half of (a_k << (34-W)) * v (tile_harmonic FAST; the launcher checks the bound)
    auto weight = [&](int k) -> int32_t { return (int32_t)((uint32_t)win.aa[k] << (34u - W)); };
    if (!wraps) {
        const T9Ctx cx{table, reinterpret_cast<const char *>(ring), H, emask, qm0, qm1, W, {st0, st1, st2}};
        uint32_t wbase;
        asm volatile("v_mov_b32 %0, %1" : "=v"(wbase) : "s"(wave << kT9WaveLog));
        // The harmonics are kept apart by real branches on the term count (7 here: always true), as in k_table_combine_tile: merged
        // into one region their 27 loads are hoisted to the top and the kernel runs out of registers.  (Scheduling fences alone --
        // __builtin_amdgcn_sched_barrier -- do hold the loads, but the register allocation is worse with them: 79 against 62 registers
        // for the VHDL-rule instance.  A software pipeline -- harmonic K + 1's bytes requested before harmonic K is worked on --
        // measured: nothing.)
        uint32_t e[kT9Runs][2];
#define BHW_T9_HARMONIC(K) if (win.n_terms > K) { t9_issue<K>(cx, rr, e); t9_finish<K, MODE, FMT, VHDL32>(cfg, cx, wbase, rr, e, weight(K), acc); __builtin_amdgcn_sched_barrier(0); }
        BHW_T9_HARMONIC(1) BHW_T9_HARMONIC(2) BHW_T9_HARMONIC(3) BHW_T9_HARMONIC(4) BHW_T9_HARMONIC(5) BHW_T9_HARMONIC(6)
#undef BHW_T9_HARMONIC
    } else {
        // (real branches between the harmonics -- the term count is 7 here, the tests are always true: they keep the compiler from
        // forming every harmonic's addresses up front, which costs this rare path registers the whole kernel then pays for)
        if (win.n_terms > 1u) t9_harmonic_slow<1, MODE, FMT>(cfg, table, rr, lq, weight(1), W, acc);
        if (win.n_terms > 2u) t9_harmonic_slow<2, MODE, FMT>(cfg, table, rr, lq, weight(2), W, acc);
        if (win.n_terms > 3u) t9_harmonic_slow<3, MODE, FMT>(cfg, table, rr, lq, weight(3), W, acc);
        if (win.n_terms > 4u) t9_harmonic_slow<4, MODE, FMT>(cfg, table, rr, lq, weight(4), W, acc);
        if (win.n_terms > 5u) t9_harmonic_slow<5, MODE, FMT>(cfg, table, rr, lq, weight(5), W, acc);
        if (win.n_terms > 6u) t9_harmonic_slow<6, MODE, FMT>(cfg, table, rr, lq, weight(6), W, acc);
    }

    auto final_value = [&](int b, int h, int j) -> int32_t {
        if constexpr (MODE == 2) return w32_final_exact(acc[b][h][j], W, 7u);                    // bh_win_7term.vhd:409-435 (one-word sums: the launcher checks the bound)
        else return (int32_t)((uint32_t)acc[b][h][j] << (32u - W)) >> (32u - W);                // (win_t)(...) wrap to W bits, win_function.cpp:375
    };
#ifdef BHW_X_NOSTORE
    {   // (timing experiment: the sums are formed, nothing is stored -- the test fails for every lane of every wave)
        int32_t x = 0;
#pragma unroll
        for (int b = 0; b < kT9Runs; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) x ^= acc[b][h][j];
        if (x != (int32_t)0x5EED1234 || rr[0] != 0xFFFFFFFFu) return;
    }
#endif
    if constexpr (!APPLY) {
        // image (h, j) starts at out + h*H + j*E, a scalar address the lane adds its 32-bit byte offset r * 4 to (saddr stores)
        auto store_all = [&](auto full_width, auto through) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint64_t img_off = (uint64_t)h * H + (uint64_t)j * E;
                    asm volatile("" : "+s"(img_off));
                    int32_t *img = out + img_off;
#pragma unroll
                    for (int b = 0; b < kT9Runs; ++b) {
                        int32_t v;
                        if constexpr (MODE != 2 && decltype(full_width)::value) v = acc[b][h][j];   // W == 32: nothing to wrap
                        else v = final_value(b, h, j);
                        if constexpr (decltype(through)::value) asm volatile("global_store_dword %0, %1, %2 sc1" :: "v"(rr[b] << 2), "v"(v), "s"(img) : "memory");
#ifdef BHW_X_NTSTORE
                        else __builtin_nontemporal_store(v, reinterpret_cast<int32_t *>(reinterpret_cast<char *>(img) + (rr[b] << 2)));   // (timing experiment)
#else
                        else *reinterpret_cast<int32_t *>(reinterpret_cast<char *>(img) + (rr[b] << 2)) = v;
#endif
                    }
                }
        };
        // windows of up to 2^24 coefficients are stored at agent scope (written through as they are produced; they would otherwise
        // sit dirty in the L2s until the kernel ends), longer ones keep write-back stores (profiles/r04_ab_store_scope.txt)
        const bool wt = cfg.phi_width <= 24u;
        if (wt) { if (W == 32u) store_all(std::true_type{}, std::true_type{}); else store_all(std::false_type{}, std::true_type{}); }
        else if (W == 32u) store_all(std::true_type{}, std::false_type{});
        else store_all(std::false_type{}, std::false_type{});
    } else {
        // Fused apply (emit()): y = (x * w) >> shift, exact 64-bit product like int_multNxN_dsp48.vhd:102.  The 24 samples are requested
        // together before the first product, nontemporal (read once: streamed past the caches the table lives in).  (Requested at the
        // very start of the wave instead and held in 24 more registers, five waves per SIMD: 0.1569 against 0.1520 ms per window,
        // profiles/r05_ab_apply_early.txt -- the pass moves 650 MB, it does not wait for its own loads.)
        int32_t xv[kT9Runs][2][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint64_t img_off = (uint64_t)h * H + (uint64_t)j * E;
                asm volatile("" : "+s"(img_off));
                const int32_t *ximg = win.apply_x + img_off;
#pragma unroll
                for (int b = 0; b < kT9Runs; ++b)
                    xv[b][h][j] = __builtin_nontemporal_load(reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(ximg) + (rr[b] << 2)));
            }
#pragma unroll
        for (int b = 0; b < kT9Runs; ++b)     // all of them in registers before the first store (otherwise each load is sunk next to its use)
            asm volatile("" : "+v"(xv[b][0][0]), "+v"(xv[b][0][1]), "+v"(xv[b][0][2]), "+v"(xv[b][0][3]),
                              "+v"(xv[b][1][0]), "+v"(xv[b][1][1]), "+v"(xv[b][1][2]), "+v"(xv[b][1][3]));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint64_t img_off = (uint64_t)h * H + (uint64_t)j * E;
                asm volatile("" : "+s"(img_off));
                int32_t *img = out + img_off;
#pragma unroll
                for (int b = 0; b < kT9Runs; ++b)
                    *reinterpret_cast<int32_t *>(reinterpret_cast<char *>(img) + (rr[b] << 2)) =
                        (int32_t)(((int64_t)xv[b][h][j] * (int64_t)final_value(b, h, j)) >> win.apply_shift);
            }
    }
}

} // namespace

// (bhwk_tile9_applicable, the configurations this kernel is compiled for: bhw_plan.cpp)
static_assert(kT9D == kTile9CellLog, "the planner names the cell size");

int bhwk_tile9(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const BhwTilePlan &tp, uint32_t tile_count, const int32_t *d_table, int32_t *d_out)
{
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    const int mode = (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0);
    const dim3 grid(tile_count * (uint32_t)(kTileThreads / kT9Threads)), block(kT9Threads);
    const int fmt = fmt_of(c.tab_dlog);
    // instances: plain nibbles for every cosine-sum rule; nibble + escapes for the VHDL rule at 32 bits only (bhwk_tile9_applicable)
#define BHW_T9(M, F, V)                                                                                                   \
    do {                                                                                                                 \
        if (w.apply_x) BHW_LAUNCH((k_tile9<M, F, true, V>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out);  \
        else           BHW_LAUNCH((k_tile9<M, F, false, V>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out); \
    } while (0)
    const bool v32 = c.dat_width == 32u;
#ifdef BHW_T9_ALLFMT5
    if (fmt == 5 && mode == 0) BHW_T9(0, 5, false);
    else if (fmt == 5 && mode == 1) BHW_T9(1, 5, false);
    else
#endif
    if (fmt == 5) { if (mode == 2 && v32) BHW_T9(2, 5, true); else return (int)hipErrorInvalidValue; }
    else if (mode == 0) BHW_T9(0, 3, false);
    else if (mode == 1) BHW_T9(1, 3, false);
    else if (v32) BHW_T9(2, 3, true);
    else BHW_T9(2, 3, false);
#undef BHW_T9
    return finish(hipSuccess);
}
