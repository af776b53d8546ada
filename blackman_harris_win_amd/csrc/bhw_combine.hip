// bhw_combine.hip -- table strategy, pass 2: cosine-sum over the table (general, quadrant fold, 15-run tiles, run-length)
//
// Part of the hand-written HIP kernels for gfx950 (MI355X, CDNA4) behind include/bhw.h.  Hot path of the reference: phase
// accumulator -> CORDIC rotation chain (or Taylor LUT) -> weighted N-term cosine sum -> int32 coefficient (SURVEY section 8a
// rows a1-a11).  Integer semantics follow SURVEY App. A; reference lines are cited at each step.
#include "bhw_device.h"

namespace {

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
// (kTileThreads = 960, kTileLanes = 192 -- the tile's 15 runs are split over kTileThreads / kTileLanes thread groups -- and
// BhwTilePlan: bhw_plan.h; the plan itself is made by bhwp_tile_plan, bhw_plan.cpp)

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
        if constexpr (MODE == 2) acc[j] = sum32_first(win.aa[0]);
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

// Residual-format records through LDS.  On gfx950 a vector load whose lanes do not read consecutive elements
// costs the CU's address path ~16 cycles per wave-instruction whatever its width, a unit-stride one ~4.7
// (profiles/r02_ubench_vmem.txt), and the tile kernel issues 54 such loads per thread: the address path, not the vector ALU, is
// what bounds it.  The records (16 bytes per cell of 2^d >= 128 entries) of the cells a wave's three runs touch are few -- for
// harmonic K a run of 192 lanes spans K * 191 + 1 entries -- so every wave copies them into shared memory once (three load
// instructions) and its 27 gathers read them with ds_read_b128 instead.  Set (K, g) of run b holds rec_slots(K) consecutive cells
// from the cell of the run's first lane; tiles in which a run wraps around the ring or a harmonic's entries wrap around the table
// (about 1.6 % of the 2 913 x 5 thread groups of a 2^26-point window: every wave decides for its own three runs) take the global loads.
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
//   run-major (kRunMajor; the VHDL cosine-sum and the 64-bit-product form): one run at a time -- its six harmonics, then its eight
//     stores -- with the nine residual words of the next run requested before the current one is worked on (kPrefetch,
//     one register each).  8 sums live instead of 24: the VHDL-rule instance drops from 93 to 66 registers (5 -> 7 waves per
//     SIMD) and runs 3.7 % faster; the HLS-rule instance (64 registers either way) ties with harmonic-major (0.1479 against
//     0.1470 ms) and, without the prefetch, loses 10 % (fewer independent gathers in flight per harmonic).
// Requesting all 27 residual words of a thread up front in harmonic-major order is slower (0.1562 ms: registers), as a
// 320-thread / 64-lane tile shape with five waves per SIMD it ties (0.1467 ms): the kernel's remaining stall is not the latency
// of its own loads.  (Two gathers per register through global_load_ubyte_d16 / _d16_hi is not available: with SRAM ECC a d16
// load clears the other half.)
__host__ __device__ constexpr int gather_order(int K, int b, int g)     // consumption order of the tile kernel's gathers, NR = 3
{
    return K == 1 ? b * 2 + g : K == 2 ? 6 + b : K == 3 ? 9 + b * 2 + g : K == 4 ? 15 + b : K == 5 ? 18 + b * 2 + g : 24 + b;
}

// One nibble-table entry (one byte at offset boff) for harmonic K.  The byte offsets of a wave's lanes are K apart, and the
// CU's address path spends ~16 cycles on a load whose lanes are not consecutive elements against ~4.7 on one whose are
// (profiles/r02_ubench_vmem.txt).  For K = 2 the entry is the low byte of 16-bit element r, for K = 4 of 32-bit element r: the
// same bytes fetched as a unit-stride short / dword load (the unpack reads bits 0..7 only; boff is a multiple of K and
// boff + K <= E, so the wider load stays inside the table).
template <int K>
__device__ __forceinline__ uint32_t ld_nibble(const void *__restrict__ table, uint32_t boff)
{
    if constexpr (K == 2) return (uint32_t)ld_off<uint16_t>(table, boff);
    else if constexpr (K == 4) return ld_off<uint32_t>(table, boff);
    else return (uint32_t)ld_off<uint8_t>(table, boff);
}

// Lane r in [0, E/2) owns the eight coefficients n = r + h*E/2 + j*E (h = 0,1; j = 0..3).  For even k the
// two h-images share one gather (k*E/2 is a whole number of quadrants); for odd k the second image reads
// entry t + E/2, another dense span of the same tile.
// MASKED: the launch produces only some of the eight images (a contiguous index range of the window that is a whole number of
// eighths -- one device's contiguous shard of a window split over 2, 4 or 8): a gather is skipped when no wanted image reads it
// (odd harmonics: the h = 0 / h = 1 gathers), sums of an unwanted half are not formed, unwanted images are not stored.
// Timeline instrumentation (development builds only, -DBHW_COMBINE_STAMPS; tools/combine_timeline.py): 8 words per workgroup.
#ifdef BHW_COMBINE_STAMPS
__device__ unsigned long long *g_combine_stamps = nullptr;
// one lane per wave, by narrowing EXEC inside one asm statement: a C++ `if (lane == 0)` around an atomic makes the compiler treat
// scalars computed after it as divergent, and the kernel's scalar-operand asm statements no longer compile
__device__ __forceinline__ void cstamp(unsigned long long *slot, bool is_max)
{
    const unsigned long long now = wall_clock64();
    unsigned long long sv;
    if (is_max) asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_umax_x2 %1, %2, off\n\ts_mov_b64 exec, %0" : "=&s"(sv) : "v"(slot), "v"(now) : "memory");
    else        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_umin_x2 %1, %2, off\n\ts_mov_b64 exec, %0" : "=&s"(sv) : "v"(slot), "v"(now) : "memory");
}
#define BHW_CSTAMP_MAX(i) do { if (g_combine_stamps) cstamp(g_combine_stamps + blockIdx.x * 8u + (i), true); } while (0)
#define BHW_CSTAMP_MIN(i) do { if (g_combine_stamps) cstamp(g_combine_stamps + blockIdx.x * 8u + (i), false); } while (0)
extern "C" int bhw_dbg_combine_stamps(void *d_words)      // 8 x uint64 per workgroup; words 0, 2 preset to ~0 (minima), the others to 0; NULL = off
{
    unsigned long long *p = (unsigned long long *)d_words;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_combine_stamps), &p, sizeof p);
}
#else
#define BHW_CSTAMP_MAX(i) do { } while (0)
#define BHW_CSTAMP_MIN(i) do { } while (0)
#endif

// Workgroup of the 15-run tiles: ONE part (192 lanes, three waves), five workgroups per tile.  The tile's five thread groups
// share nothing but table lines -- every wave stages its own records, there is no barrier -- and a workgroup gives its wave
// slots back only when its last wave is done: as one 960-thread workgroup (rounds 1 - 3) the waves that were served first (the
// arbiter prefers the oldest) reached their stores 3 us before the last ones of an ~10 us life (profiles/r04_combine_timeline.txt),
// and the last of the 5.7 rounds of 512 such workgroups ran a third empty.  0.1002 -> 0.0967 ms per window, combine pass 67.7 ->
// 63.5 us (profiles/r04_ab_tile_wg.txt; one wave per workgroup ties: 0.0972).  The five workgroups of a tile are dealt to the same
// XCD one after the other, so the sibling runs still meet in one L2.
// (The one- and three-run tiles of windows of up to five terms keep ONE 960-thread workgroup per tile -- kWgPerTile = kWgPerPart = 1
// there: five 192-thread workgroups measured a tie, 0.0821 / 0.0816 ms for BH-4 2^26 at 24 bits, and were not adopted.)
constexpr int kTileWg = kTileLanes;
constexpr int tile_wg_of(int nb) { return nb >= 15 ? kTileWg : kTileThreads; }
template <int NB, int MODE, int FMT, bool FAST = false, bool MASKED = false>
__global__ __launch_bounds__(tile_wg_of(NB)) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? 4 : 8))) void k_table_combine_tile(BhwCordicCfg cfg, BhwWinCfg win, BhwTilePlan tp,
                                                                      const void *__restrict__ table, int32_t *__restrict__ out)
{
    // VHDL rule: the W+2-bit sum as two words (Sum32) in general; FAST instances are launched only when the sum of the |a_k| stays
    // below 2^31 (every term b_k is at most |a_k| + 1 in magnitude), so the exact sum is one 32-bit word
    constexpr bool kWideSum = MODE == 2 && !FAST;
    using acc_t = typename std::conditional<kWideSum, Sum32, int32_t>::type;
    BHW_CSTAMP_MIN(0);                                               // first wave of the workgroup starts
    BHW_CSTAMP_MAX(1);                                               // last wave starts
#ifdef BHW_COMBINE_STAMPS
    if (g_combine_stamps) {                                          // (every lane stores the same two words: no divergence)
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        g_combine_stamps[blockIdx.x * 8u + 6u] = hw;
        g_combine_stamps[blockIdx.x * 8u + 7u] = xcc;
    }
#endif
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
    // Workgroups are dealt round-robin over the 8 XCDs (block b and b + 8 share an L2).  The tiles are renumbered so
    // that each XCD sweeps a contiguous eighth of the ring (neighbouring tiles share table lines at their run boundaries and the
    // 16-byte records of the residual format).
    constexpr int kWgThreads = tile_wg_of(NB);
    constexpr uint32_t kWgPerTile = kTileThreads / kWgThreads;      // workgroups per tile
    constexpr uint32_t kWgPerPart = kLanes / kWgThreads;            // ... per part: 1 either way -- a 15-run tile is five 192-thread workgroups of one part each, a one- or three-run tile ONE 960-thread workgroup that is its only part
    static_assert(kWgPerPart * kWgThreads == kLanes && kWgPerPart * kParts == kWgPerTile, "a workgroup lies inside one part");
    uint32_t tile_of_block = blockIdx.x / kWgPerTile, part_of_block = blockIdx.x % kWgPerTile;
    {
        const uint32_t per = (gridDim.x / kWgPerTile) >> 3, main = (per << 3) * kWgPerTile;   // tiles per XCD in the evenly divisible part
        if (blockIdx.x < main) {
            const uint32_t j = blockIdx.x >> 3;                     // position in this XCD's sequence of workgroups
            tile_of_block = (blockIdx.x & 7u) * per + j / kWgPerTile;
            part_of_block = j % kWgPerTile;
        }
    }
    const uint32_t part = kWgPerTile > 1 ? part_of_block / kWgPerPart : __builtin_amdgcn_readfirstlane(threadIdx.x / kLanes);   // wave-uniform: kLanes is a multiple of 64
    const uint32_t lane_in_part = kWgPerTile > 1 ? (part_of_block % kWgPerPart) * kWgThreads + threadIdx.x : threadIdx.x % kLanes;
    constexpr bool kLdsRec = fmt_is_resid(FMT) && NB >= 15 && kLanes == kTileLanes;
    uint32_t rec_meta = 0;                                           // slot -> (K, g, j) of the record staging below, fetched first
    if constexpr (kLdsRec) rec_meta = kRecMeta.v[threadIdx.x & 63u];
    uint32_t rr[NR], starts[NR];
#pragma unroll
    for (int b = 0; b < NR; ++b) {
        const uint32_t start = ((tile_of_block + tp.tile0) * kLanes + tp.offs[part * NR + b]) & hmask;   // scalar; offs padded with copies of the last run
        starts[b] = start;
        rr[b] = (start + (lane_in_part + kLanes - (start & 63u)) % kLanes) & hmask;
    }
    // records of the cells this wave's runs touch, staged in shared memory (see above)
    constexpr int kWavesWg = kWgThreads / 64;
    __shared__ int4 rec_s[kLdsRec ? kWavesWg * NR * kRecPerRun : 1];
    uint32_t rbias[kRecSets][NR];                                    // scalar: byte offset of "cell 0" of set si, run b in rec_s
    uint32_t qpack[NR];                                              // scalar: quadrant of set si, run b in bits 2 si, 2 si + 1
    bool wraps = false;
    uint32_t cls[NR];                                                // residue class of r in the split layout (odd harmonics)
#pragma unroll
    for (int b = 0; b < NR; ++b) cls[b] = split_class<FMT>(rr[b], lq);
    // (run-major for the FAST instances too, measured again with one part per workgroup: 0.0966 -> 0.0993 ms)
    constexpr bool kRunMajor = NB >= 15 && !FAST;   // also the 64-bit-product form (caller-scaled weights): no registers to spare otherwise
    constexpr bool kPrefetch = kRunMajor && kLdsRec && NR == 3;
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
                    if constexpr (fmt_is_resid(FMT)) {                                             \
                        const uint32_t rg = rr[b] + (uint32_t)g * H;                                     \
                        const uint32_t boff = resid_offset<FMT, K>(rg, (uint32_t)K * rg, cls[b], lq, emask); \
                        land[gather_order(K, b, g)] = fmt_is_nibble(FMT) ? ld_nibble<K>(table, boff) : (uint32_t)ld_off<uint16_t>(table, boff); \
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
        // Does a run wrap around the ring, or a harmonic's entries around the table?  One lane per test (lane 9 b + si: set si of run b;
        // lanes 27 .. 29: the runs themselves) and one vote, instead of 30 scalar compare / select / or chains per wave: the CU's one
        // scalar unit serves four SIMDs, and this kernel issues as many scalar as vector cycles (profiles/r03_pmc_summary.json).
        {
            static_assert(NR == 3 && kRecSets == 9, "lane -> (run, set) below");
            const uint32_t li = threadIdx.x & 63u;
            const uint32_t bq = li < 27u ? li / 9u : li - 27u, si = li - 9u * bq;       // (si meaningful for li < 27)
            const uint32_t st = bq == 0u ? starts[0] : bq == 1u ? starts[1] : starts[2];
            const uint32_t K = (uint32_t)(0x655433211ull >> (4u * (si < 9u ? si : 0u))) & 15u;   // rec_set_K(si): 1 1 2 3 3 4 5 5 6
            const uint32_t g = (0x92u >> (si < 9u ? si : 0u)) & 1u;                               // rec_set_g(si): sets 1, 4, 7
            const uint32_t u0 = (K * (st + g * H)) & emask;
            const bool w = li < 27u ? u0 + K * (uint32_t)(kLanes - 1) > emask : li < 30u ? st + (uint32_t)kLanes > H : false;
            wraps = __builtin_amdgcn_ballot_w64(w) != 0ull;
        }
#pragma unroll
        for (int b = 0; b < NR; ++b) {
            qpack[b] = 0u;
#pragma unroll
            for (int si = 0; si < kRecSets; ++si) {
                const uint32_t K = rec_set_K(si);
                const uint32_t th0 = K * (starts[b] + (uint32_t)rec_set_g(si) * H);
                qpack[b] |= ((th0 >> lq) & 3u) << (2 * si);
                rbias[si][b] = (((wave * NR + b) * kRecPerRun + (uint32_t)rec_set_base(si)) << 4) - ((th0 >> d) << 4);   // cell of the unmasked angle (resid_value)
            }
        }
        if constexpr (kPrefetch) {
            if (!wraps) {
                if constexpr (kRunMajor) issue_runs(std::integral_constant<int, 0>{});   // run 0 now, run b + 1 while run b is worked on
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
    ResidK rk{fmt_cell_log(cfg.tab_dlog), (1u << fmt_cell_log(cfg.tab_dlog)) - 1u};
    uint32_t emask_v = emask, lq_v = lq;                            // per-gather shift / mask operands: VGPR copies (see ResidK)
    // Run-major order (kRunMajor): one run at a time -- its six harmonics, then its eight stores -- instead of every
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
                    if constexpr (kWideSum) acc[b][h][j] = sum32_first(win.aa[0]);
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
        uint32_t esc_min = ~0u;                                                                          \
        _Pragma("unroll") for (int b = B0; b < B1; ++b) {                                                 \
            _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                             \
                if (MASKED && NG == 2 && !(g ? want1 : want0)) continue;                                 \
                const uint32_t rg = rr[b] + (uint32_t)g * H;                                             \
                const uint32_t theta = (uint32_t)K * rg;                                                 \
                if constexpr (NB > 1 && fmt_is_resid(FMT)) {                                             \
                    uint32_t bias = rbias[rec_set_index(K, g)][b];                                       \
                    if constexpr (LDS) asm("" : "+s"(bias));    /* one scalar: the record address is shift, shift-add */ \
                    uint32_t e;                                                                          \
                    if constexpr (kPrefetch && LDS) e = land[gather_order(K, b, g)];                     \
                    else {                                                                               \
                        const uint32_t boff = resid_offset<FMT, K>(rg, theta, cls[b], lq, emask_v);     \
                        e = fmt_is_nibble(FMT) ? ld_nibble<K>(table, boff) : (uint32_t)ld_off<uint16_t>(table, boff); \
                    }                                                                                    \
                    cs[b][g] = resid_value<FMT, LDS>(cfg, theta, emask_v, rk, lrec, bias, e);            \
                    if constexpr (FMT == 5) {                                                            \
                        const uint32_t dcv = e << 28;               /* the low field, zero = the marker */ \
                        esc_min = dcv < esc_min ? dcv : esc_min;                                         \
                    }                                                                                    \
                } else if constexpr (NB > 1 && (K & 1)) cs[b][g] = tab_load_class<FMT>(cfg, table, theta & emask_v, cls[b]); \
                else if constexpr (NB > 1) {                                                             \
                    const uint32_t u = theta & emask_v;                                                  \
                    cs[b][g] = tab_fetch<FMT>(cfg, table, u, tab_index<KC, 1>(u, lq, 1u));              \
                } else cs[b][g] = tab_load<KC, FMT, -1>(cfg, table, (theta & emask) >> cfg.z_shr, lq - cfg.z_shr); \
            }                                                                                            \
        }                                                                                                \
        if constexpr (NB > 1 && FMT == 5) {                                                              \
            /* nibble + escapes: one test per harmonic (the minimum of its low fields is the marker); the listed entries are a few   \
               per million, so the branch is almost never taken and the loads of the next harmonic are not held up by six of them */ \
            if (__builtin_expect(esc_min == (kEscMarker << 28), 0)) {                                    \
                _Pragma("unroll") for (int b = B0; b < B1; ++b) {                                         \
                    _Pragma("unroll") for (int g = 0; g < NG; ++g) {                                     \
                        if (MASKED && NG == 2 && !(g ? want1 : want0)) continue;                         \
                        const uint32_t ue = ((uint32_t)K * (rr[b] + (uint32_t)g * H)) & emask_v;       /* (natural layout: the word again) */ \
                        esc_fix_wave(cfg.tab_esc, cfg.esc_wg_log, lq, ue, (ld_off<uint8_t>(table, ue) & 0xFu) == kEscMarker, cs[b][g]); \
                    }                                                                                    \
                }                                                                                        \
            }                                                                                            \
        }                                                                                                \
        _Pragma("unroll") for (int b = B0; b < B1; ++b) {                                                 \
            int32_t sv[4];                                                                               \
            /* only quadrant bits 0,1 of theta >> lq are used */                                         \
            const int32_t aK = FAST ? (int32_t)((uint32_t)win.aa[K] << (34u - W)) : win.aa[K];          \
            if constexpr (LDS && (FAST || MODE == 2)) {                                 \
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
        if constexpr (kWideSum) return w32_final<BHW_COMBINE_VHDL>(acc[b][h][j], W, win.n_terms);
        else if constexpr (MODE == 2) return w32_final_exact(acc[b][h][j], W, win.n_terms);
        else return (int32_t)((uint32_t)acc[b][h][j] << (32u - W)) >> (32u - W);   // (win_t)(...) wrap to W bits
    };
    auto store_runs = [&](auto run_tag) {
    constexpr int B0 = decltype(run_tag)::value < 0 ? 0 : decltype(run_tag)::value, B1 = decltype(run_tag)::value < 0 ? NR : B0 + 1;
    if (win.apply_x == nullptr) {                                    // wave-uniform
        // image (h, j) starts at out + h*H + j*E, a scalar address the lane adds its 32-bit byte offset r * 4 to (saddr stores;
        // the empty asm keeps the compiler from folding the image offset back into a 64-bit vector add per store)
        auto store_all = [&](auto full_width, auto through) {
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
                        // (plain write-back stores: at agent or system scope -- sc1, sc0 sc1 -- the pass takes 91 us instead of 67, profiles/r04_ab_store_scope.txt)
                        if constexpr (decltype(through)::value) asm volatile("global_store_dword %0, %1, %2 sc1" :: "v"(rr[b] << 2), "v"(v), "s"(img) : "memory");
                        else *reinterpret_cast<int32_t *>(reinterpret_cast<char *>(img) + (rr[b] << 2)) = v;
                    }
                }
        };
        // Store scope.  A short window stays dirty in the eight L2s (32 MB together) until the end-of-kernel release writes it back
        // with nothing else running: windows of up to 2^24 coefficients are stored at agent scope (written through as they are
        // produced), like the table of the build pass -- 0.0246 -> 0.0239 ms at 2^22, 0.0463 -> 0.0456 at 2^24, a loss from 2^25 on
        // (0.0576 -> 0.0584).  Long windows keep write-back stores: the L2s evict as they go, and at agent scope the 268 MB of
        // the headline window take 91 us instead of 67 (profiles/r04_ab_store_scope.txt; through-stores for only the last 1/16 ..
        // 1/4 of the tiles: +0.6 .. +2.8 us, profiles/r04_ab_tail_wt.txt).
        const bool wt = cfg.phi_width <= 24u;
        if (wt) { if (W == 32u) store_all(std::true_type{}, std::true_type{}); else store_all(std::false_type{}, std::true_type{}); }
        else if (W == 32u) store_all(std::true_type{}, std::false_type{});
        else store_all(std::false_type{}, std::false_type{});
    } else {
        // Fused apply (emit()): y = (x * w) >> shift.  The x samples of every run of this pass are requested together before the
        // first product (harmonic-major: 24 unit-stride loads in flight per thread; fetched eight at a time, run by run, the pass
        // waited three times for HBM: 0.208 -> ms in profiles/r03_*), and x and y address as scalar image base + 32-bit lane offset
        // like the plain stores.
        auto apply_runs = [&](const int b0, const int b1) {
            int32_t xv[NR][2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint64_t img_off = (uint64_t)h * H + (uint64_t)j * E;
                    asm volatile("" : "+s"(img_off));
                    const int32_t *ximg = win.apply_x + img_off;
#pragma unroll
                    for (int b = 0; b < NR; ++b)
                        // nontemporal: every sample is read once, and streaming 268 MB of them through the L2s / the memory-side cache
                        // evicts the table lines the gathers live on (0.1698 -> 0.1556 ms for 2^26 samples; nontemporal y stores on
                        // top: 0.178, profiles/r04_ab_apply_nt.txt)
                        if (b >= b0 && b < b1)
                            xv[b][h][j] = __builtin_nontemporal_load(reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(ximg) + (rr[b] << 2)));
                }
#pragma unroll
            for (int b = 0; b < NR; ++b)     // all of them in registers before the first store (otherwise each load is sunk next to its use)
                if (b >= b0 && b < b1)
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
                    for (int b = 0; b < NR; ++b)
                        if (b >= b0 && b < b1)
                            *reinterpret_cast<int32_t *>(reinterpret_cast<char *>(img) + (rr[b] << 2)) =
                                (int32_t)(((int64_t)xv[b][h][j] * (int64_t)final_value(b, h, j)) >> win.apply_shift);
                }
        };
        if constexpr (NB >= 15) apply_runs(B0, B1);                  // 64 registers hold the 24 sums and the 24 samples
        else {                                                       // one-part tiles (n_terms <= 5) carry more state: run by run
#pragma unroll
            for (int b = B0; b < B1; ++b) {
                int32_t xv[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[h][j] = __builtin_nontemporal_load(&win.apply_x[(uint64_t)(rr[b] + (uint32_t)h * H + (uint32_t)j * E)]);
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
    }
    };
    if constexpr (kRunMajor && NR == 3) {
        if constexpr (kPrefetch) { if (!wraps) issue_runs(std::integral_constant<int, 1>{}); }
        init_acc(std::integral_constant<int, 0>{}); run_harmonics(std::integral_constant<int, 0>{}); store_runs(std::integral_constant<int, 0>{});
        if constexpr (kPrefetch) { if (!wraps) issue_runs(std::integral_constant<int, 2>{}); }
        init_acc(std::integral_constant<int, 1>{}); run_harmonics(std::integral_constant<int, 1>{}); store_runs(std::integral_constant<int, 1>{});
        init_acc(std::integral_constant<int, 2>{}); run_harmonics(std::integral_constant<int, 2>{}); store_runs(std::integral_constant<int, 2>{});
    } else {
        init_acc(std::integral_constant<int, -1>{}); run_harmonics(std::integral_constant<int, -1>{});
        BHW_CSTAMP_MIN(2);                                           // first wave reaches its stores
        BHW_CSTAMP_MAX(3);                                           // last wave reaches its stores
        // (s_setprio 3 for the store phase, or for the set-up and first loads: +-0.3 us, nothing)
        store_runs(std::integral_constant<int, -1>{});
    }
    BHW_CSTAMP_MAX(4);                                               // last wave has issued its stores
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
// kRlRun = 16 consecutive ring lanes per thread (8 or 16: the swizzles below assume a multiple of 4 granule rows), kRlBlock = 64: bhw_plan.h

// LDS tile of one (wave, image): 1024 values as 256 granules of 16 bytes, granule index XOR-swizzled inside rows of eight
// so that both the producer pattern (granule 4*lane + c) and the consumer pattern (granule 64*s + lane) are conflict-free.
__device__ __forceinline__ uint32_t rl_swizzle(uint32_t g) { return (g & ~7u) | ((g ^ (g >> 3)) & 7u); }
// the same for 8-byte granules (rows of sixteen): dat_width <= 16 keeps the tile as int16, half the LDS, twice the waves per CU
__device__ __forceinline__ uint32_t rl_swizzle16(uint32_t g) { return (g & ~15u) | ((g ^ (g >> 4)) & 15u); }

template <int NTERMS, int MODE, bool NARROW>
__global__ __launch_bounds__(kRlBlock) void k_runlength_window(BhwCordicCfg cfg, BhwWinCfg win, const int2 *__restrict__ table,
                                                              int32_t *__restrict__ out)
{
    using gran_t = typename std::conditional<NARROW, uint2, int4>::type;  // four coefficients: 4 x int16 or 4 x int32
    __shared__ gran_t tile[kRlBlock / 64][4][16 * kRlRun];                        // [wave][image j][granule]: 8 / 16 KiB per wave
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
                        uint32_t bk = brk[K];                                                                           \
                        asm volatile("" : "+v"(bk));   /* the vote stays here: hoisted, the 6 x 15 masks of a sweep spill */ \
                        if (__builtin_amdgcn_ballot_w64(bk == i)) {                /* wave-uniform: usually nobody */   \
                            const bool mine = bk == i;                                                                  \
                            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j] += mine ? dlt[K][j] : 0;               \
                        }                                                                                               \
                    }
                    BHW_RL_STEP(1) BHW_RL_STEP(2) BHW_RL_STEP(3) BHW_RL_STEP(4) BHW_RL_STEP(5) BHW_RL_STEP(6)
#undef BHW_RL_STEP
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j][e] = finish(acc[j]);
            }
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
    }
}

} // namespace

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
    bhwp_tile_plan(c, w, tp, nb, lanes);
    if (tile_count == 0) { tile0 = 0; tile_count = tp.n_tiles; }
    if (tile0 + tile_count > tp.n_tiles) return (int)hipErrorInvalidValue;
    tp.tile0 = tile0;
    tp.img_mask = img_mask & 0xFFu;
    tp.n0mod = n0mod;
    const bool masked = tp.img_mask != 0xFFu;                       // some of the eight images only (bhwk_tile_images_applicable)
    if (masked && (nb != 15 || w.apply_x != nullptr || tp.img_mask == 0u)) return (int)hipErrorInvalidValue;
    const int mode = (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0);
    const uint32_t wg_threads = (uint32_t)tile_wg_of(nb);
    const dim3 grid(tile_count * ((uint32_t)kTileThreads / wg_threads)), block(wg_threads);
    const bool fast = bhwp_tile_fast(c, w, nb);                     // one-instruction products (and, VHDL rule, one-word sums)
    if (bhwk_tile9_applicable(c, w, nb, fast, masked)) return bhwk_tile9(l, c, w, tp, tile_count, d_table, d_out);   // bhw_tile9.hip
#define BHW_LAUNCH_TILE_MFK(NB, M, F, K)                                                                                 \
    do {                                                                                                                 \
        if (c.tab_dlog == 0)             BHW_LAUNCH((k_table_combine_tile<NB, M, 0, F, K>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out); \
        else if (c.tab_dlog == kPackLog) BHW_LAUNCH((k_table_combine_tile<NB, M, 1, F, K>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out); \
        else if (c.tab_dlog < kNibbleFlag) BHW_LAUNCH((k_table_combine_tile<NB, M, 2, F, K>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out); \
        else if (c.tab_dlog < kEscFlag)  BHW_LAUNCH((k_table_combine_tile<NB, M, 3, F, K>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out); \
        else                             BHW_LAUNCH((k_table_combine_tile<NB, M, 5, F, K>), grid, block, 0, st, c, w, tp, (const void *)d_table, d_out); \
    } while (0)
#define BHW_LAUNCH_TILE_MF(NB, M, F)                                                                                     \
    do {                                                                                                                 \
        if (NB == 15 && masked) BHW_LAUNCH_TILE_MFK(NB, M, F, (NB == 15));                                               \
        else                    BHW_LAUNCH_TILE_MFK(NB, M, F, false);                                                    \
    } while (0)
#define BHW_LAUNCH_TILE_M(NB, M)                                                                                         \
    do {                                                                                                                 \
        if (NB == 15 && fast) BHW_LAUNCH_TILE_MF(NB, M, (NB == 15));                                                    \
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

int bhwk_runlength_window(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_table, int32_t *d_out)
{
    if (!bhwk_runlength_applicable(c, w, d_out)) return (int)hipErrorInvalidValue;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    const uint32_t H = 1u << (c.phi_width - 3);
    const dim3 grid(H / (kRlBlock * kRlRun)), block(kRlBlock);
    const int mode = (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0);
    const bool narrow = c.dat_width <= 16;                       // coefficients fit int16: half-size LDS tile
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

