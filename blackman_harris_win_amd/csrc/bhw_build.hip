// bhw_build.hip -- table strategy, pass 1: the first-quadrant (c, s) table
//
// Part of the hand-written HIP kernels for gfx950 (MI355X, CDNA4) behind include/bhw.h.  Hot path of the reference: phase
// accumulator -> CORDIC rotation chain (or Taylor LUT) -> weighted N-term cosine sum -> int32 coefficient (SURVEY section 8a
// rows a1-a11).  Integer semantics follow SURVEY App. A; reference lines are cited at each step.
#include "bhw_device.h"

namespace {

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
#pragma unroll
            for (int r = 1; r < n_iter; ++r) rot_step(x, y, z, r, plan.lut[r]);
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

// Table strategy, pass 1, octant-mirror form (residual / nibble formats).
//
// After rotation 0 the state is the 45-degree vector (x0, x0) and 2 * lut[0] is exactly a quarter turn (checked by the launcher),
// so the chain of u' = E - u is the chain of u with x and y swapped, z negated and every decision flipped -- bit for bit, floors
// included -- as long as no z_k on the way is exactly 0 (z >= 0 rotates forward in both).  Zero events are common only in the last
// rotations (|z| is a few units there), so:
//   * only u in [0, E/2) run a chain of their own; u' in (E/2, E) are images of u in [1, E/2 - 1]; the middle entry E/2 is a
//     deferred chain of the last workgroup;
//   * at rotation KS = NITER - kMirrorTail the image state is taken as (y, x, -z) and the last rotations are evaluated for both;
//   * a lane that met z_k == 0 before KS (one v_cmp per shared rotation, collected in a scalar mask; about 1 % of the lanes, but
//     every second wave has one) does not store its image: it appends u to a worklist in shared memory, and after the groups
//     the workgroup's first lanes run those images as chains of their own (a wave-wide replay in place costs more than the
//     symmetry saves);
//   * a group whose shared prefix itself met z == 0 (only its leaf 0 can) recomputes that one image from scratch;
//   * the records of the image cells come from chains of their own (third wave), never from the symmetry.
//
// Round 3: two exact short cuts inside a group, both established on the host by tools/sim_build.cpp (every entry of the tables
// of the BASELINE configurations against the plain chain) and covered on the device by the parity tests:
//   NARROW STATE.  From its split rotation k0 on a group's leaves move x and y by less than D = 2^(NITER - k0 + 1) + 2 in total,
//     so when the parked group state satisfies D <= X, Y and X + D, Y + D < 2^NITER (a scalar test in phase 1; 99 % of the
//     groups: all but the first degree of the octant) every x_k, y_k of every leaf lies in [0, 2^NITER), NITER <= 32: the
//     rotations run on 32-bit unsigned words with logical shifts, and the decision is applied by masking EXEC -- one compare,
//     then add / sub / add on the lanes with z < 0 and sub / add / sub on the others: 9 plain VOP2 instructions (+ the zero test)
//     instead of 2 x v_mad_i64_i32, 2 x v_alignbit_b32 and v_mad_i32_i24 among 9 (those issue in 4.2 - 4.8 cycles, plain
//     VOP2 in 2.5: profiles/r02_ubench_gfx950.txt).  The other groups run the 64-bit rotation as a rolled loop.
//   TABLE TAIL.  In such a group x_k >> k for k >= KS is (x_KS >> KS) >> (k - KS) as long as the low KS bits of x_KS keep a
//     margin of 128 to both ends (the tail moves x by at most 120): the last kMirrorTail rotations collapse to
//         x_end = x_KS - D[p][y_KS >> KS],   y_end = y_KS + D[p][x_KS >> KS],   D[p][v] = sum_j sigma_j(p) (v >> j),
//     p = the tail's decision pattern, a function of z_KS alone (|z_KS| < 256 at every width: table tail_p, rebuilt from the
//     ROM by every workgroup), D a 64 x 64 byte table (kTailD).  The image reads the same tables with z -> -z and x, y swapped.
//     Two table reads per coordinate instead of six rotations; a wave with a lane outside the margins (50 of 131 072 waves at
//     2^26 / 32 bits) runs the six rotations instead (wave-uniform branch).
// Per pair of entries: 9 shared rotations of 10 instructions + ~25 for the two tails, instead of 9 + 2 x 6 rotations of 9.
constexpr int kMirrorTail = 6;         // rotations of the table tail (D's row index has kMirrorTail bits, its column 32 - KS <= 6 bits)
// Workgroup shape.  gfx950 starts about one workgroup per 10 ns chip-wide whatever its size (tools/ubench_turnaround.hip,
// profiles/r04_ubench_turnaround.txt: one round of 2 048 x 256 threads needs ~20 us before its last workgroup runs, 512 x 1 024
// threads ~5 us), and this pass is ONE round of workgroups: with 256 threads the chip was still filling during the first half of the
// 46 us pass.  So the workgroup is 1 024 threads (16 waves, two per CU: the same eight waves per SIMD) wherever the table gives
// at least one workgroup per CU; every wave still walks 16 groups, so a workgroup owns THREADS / 4 groups.
constexpr int mirror_gpw(int threads) { return threads / 4; }                       // own groups per workgroup
constexpr int mirror_heads_max(int threads) { return mirror_gpw(threads) / 2 + 8; } // cells of one range (>= 128 entries each) + 3
constexpr int mirror_heads_pad(int threads) { return (mirror_heads_max(threads) + 63) / 64 * 64; }   // head-chain lanes per range
constexpr int kTailZ = 256;            // tail_p covers z_KS in (-kTailZ, kTailZ)

constexpr int kTailRows = 65;          // x >> KS of a biased-narrow group reaches 64 (x up to 2^NITER (1 + eps)); 64 + 32 + .. + 2 = 126 fits the byte
struct TailD { int8_t v[kTailRows * 64 + 16]; };
constexpr TailD make_tail_d()
{
    TailD t{};
    for (int p = 0; p < 64; ++p)
        for (int v = 0; v < kTailRows; ++v) {
            int d = 0;
            for (int j = 0; j < kMirrorTail; ++j) d += ((p >> j) & 1) ? -(v >> j) : (v >> j);   // bit j set: z < 0 at rotation KS + j
            // stored [v][p]: the 64 leaves of a group share x >> KS and y >> KS (they differ by less than 2^18), so a wave reads one
            // 64-byte row at 64 patterns -- 16 consecutive dwords, no bank conflict ([p][v] put all 64 lanes on two banks)
            t.v[v * 64 + p] = (int8_t)d;
        }
    return t;
}
__device__ const TailD kTailD = make_tail_d();

// One rotation on the narrow state (see above); K is an immediate, lutk a scalar.  Every lane of the wave is active on entry and
// on exit (phase 2 runs whole groups: EXEC is all ones), so the decision is: v_cmpx (EXEC = lanes with z < 0), three updates,
// s_not EXEC, three updates, EXEC = -1.  ZERO: zacc = min(zacc, (unsigned)z) -- zero exactly when some z_k was 0 (one VOP2
// instruction instead of a compare and a scalar OR: the CU's one scalar unit serves four SIMDs, 4.2 cycles per scalar
// instruction per SIMD against 2.7 for a VOP2, profiles/r03_ubench_issue.txt).
template <int K, bool ZERO>
__device__ __forceinline__ void rot_narrow(uint32_t &x, uint32_t &y, int32_t &z, uint32_t &zacc, uint32_t lutk)
{
    uint32_t a, b;
    if constexpr (ZERO) {
        asm volatile("v_lshrrev_b32 %[a], %[k], %[y]\n\t"
                     "v_lshrrev_b32 %[b], %[k], %[x]\n\t"
                     "v_min_u32 %[za], %[za], %[z]\n\t"
                     "v_cmpx_gt_i32 vcc, 0, %[z]\n\t"
                     "v_add_u32 %[x], %[x], %[a]\n\t"
                     "v_sub_u32 %[y], %[y], %[b]\n\t"
                     "v_add_u32 %[z], %[z], %[l]\n\t"
                     "s_not_b64 exec, exec\n\t"
                     "v_sub_u32 %[x], %[x], %[a]\n\t"
                     "v_add_u32 %[y], %[y], %[b]\n\t"
                     "v_sub_u32 %[z], %[z], %[l]\n\t"
                     "s_mov_b64 exec, -1"
                     : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [za] "+v"(zacc), [a] "=&v"(a), [b] "=&v"(b)
                     : [l] "s"(lutk), [k] "n"(K)
                     : "vcc", "scc");
    } else {
        asm volatile("v_lshrrev_b32 %[a], %[k], %[y]\n\t"
                     "v_lshrrev_b32 %[b], %[k], %[x]\n\t"
                     "v_cmpx_gt_i32 vcc, 0, %[z]\n\t"
                     "v_add_u32 %[x], %[x], %[a]\n\t"
                     "v_sub_u32 %[y], %[y], %[b]\n\t"
                     "v_add_u32 %[z], %[z], %[l]\n\t"
                     "s_not_b64 exec, exec\n\t"
                     "v_sub_u32 %[x], %[x], %[a]\n\t"
                     "v_add_u32 %[y], %[y], %[b]\n\t"
                     "v_sub_u32 %[z], %[z], %[l]\n\t"
                     "s_mov_b64 exec, -1"
                     : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [a] "=&v"(a), [b] "=&v"(b)
                     : [l] "s"(lutk), [k] "n"(K)
                     : "vcc", "scc");
    }
}

// The last ten narrow rotations before the hand-over, KA = KS - 10 .. KS - 1, as ONE statement: entry at rotation KA + e (e <= 0: from the
// top; the group's split rotation usually lies one or two rotations into the block), no scalar guard and no hazard no-op
// between the rotations (the compiler separates adjacent asm statements that end / begin with EXEC accesses by an s_nop).
// HI: the biased narrow state -- the word holds x - B, B = 2^(NITER-1), for groups whose x reaches 2^NITER (the first third of a
// degree of the octant; every x_k then lies in [B, B + 2^32)).  B is a multiple of 2^k, so x >> k = ((x - B) >> k) + (B >> k)
// exactly: one more VOP2 per rotation (a literal operand), everything else is the same code.  (Those 0.7 % of the groups all sit
// in the first four workgroups; on the 64-bit path they kept the whole pass waiting 6 us for those four,
// profiles/r04_build_timeline.txt.)
template <int KA, int NITER>
__device__ __forceinline__ void rot_narrow_block(uint32_t &x, uint32_t &y, int32_t &z, uint32_t &zacc, const uint32_t *lut, int e, uint32_t hi)
{
    // Both forms in ONE statement behind a scalar branch on `hi` (labels 0 .. 10 plain, 20 .. 30 biased): as two statements in an
    // if / else the register allocator gave them different registers and joined them with four v_mov per group.
    uint32_t a, b;
#define BHW_ROT_I(lbl, i, HI_ADD)                                                                                       \
    #lbl ":\n\t"                                                                                                        \
    "v_lshrrev_b32 %[a], %[k" #i "], %[y]\n\t"                                                                          \
    "v_lshrrev_b32 %[b], %[k" #i "], %[x]\n\t"                                                                          \
    HI_ADD                                                                                                             \
    "v_min_u32 %[za], %[za], %[z]\n\t"                                                                                  \
    "v_cmpx_gt_i32 vcc, 0, %[z]\n\t"                                                                                    \
    "v_add_u32 %[x], %[x], %[a]\n\t"                                                                                    \
    "v_sub_u32 %[y], %[y], %[b]\n\t"                                                                                    \
    "v_add_u32 %[z], %[z], %[l" #i "]\n\t"                                                                              \
    "s_not_b64 exec, exec\n\t"                                                                                          \
    "v_sub_u32 %[x], %[x], %[a]\n\t"                                                                                    \
    "v_add_u32 %[y], %[y], %[b]\n\t"                                                                                    \
    "v_sub_u32 %[z], %[z], %[l" #i "]\n\t"                                                                              \
    "s_mov_b64 exec, -1\n"
#define BHW_ROT_PLAIN(i) BHW_ROT_I(i, i, "")
#define BHW_ROT_HI(i) BHW_ROT_I(2##i, i, "v_add_u32 %[b], %[c" #i "], %[b]\n\t")
#define BHW_ROT_DISPATCH(P)                                                                                            \
                 "s_cmp_lt_i32 %[e], 1\n\ts_cbranch_scc1 " #P "0f\n\t"                                                 \
                 "s_cmp_eq_u32 %[e], 1\n\ts_cbranch_scc1 " #P "1f\n\t"                                                 \
                 "s_cmp_eq_u32 %[e], 2\n\ts_cbranch_scc1 " #P "2f\n\t"                                                 \
                 "s_cmp_eq_u32 %[e], 3\n\ts_cbranch_scc1 " #P "3f\n\t"                                                 \
                 "s_cmp_eq_u32 %[e], 4\n\ts_cbranch_scc1 " #P "4f\n\t"                                                 \
                 "s_cmp_eq_u32 %[e], 5\n\ts_cbranch_scc1 " #P "5f\n\t"                                                 \
                 "s_cmp_eq_u32 %[e], 6\n\ts_cbranch_scc1 " #P "6f\n\t"                                                 \
                 "s_cmp_eq_u32 %[e], 7\n\ts_cbranch_scc1 " #P "7f\n\t"                                                 \
                 "s_cmp_eq_u32 %[e], 8\n\ts_cbranch_scc1 " #P "8f\n\t"                                                 \
                 "s_cmp_eq_u32 %[e], 9\n\ts_cbranch_scc1 " #P "9f\n\t"
    constexpr int S = NITER - 1 - KA;                      // B >> (KA + i) = 2^(S - i)
    static_assert(S >= 9 && S <= 30, "bias shifts");
    asm volatile("s_cmp_lg_u32 %[hi], 0\n\ts_cbranch_scc1 100f\n\t"
                 BHW_ROT_DISPATCH()
                 "s_branch 10f\n"
                 BHW_ROT_PLAIN(0) BHW_ROT_PLAIN(1) BHW_ROT_PLAIN(2) BHW_ROT_PLAIN(3) BHW_ROT_PLAIN(4) BHW_ROT_PLAIN(5) BHW_ROT_PLAIN(6) BHW_ROT_PLAIN(7)
                 BHW_ROT_PLAIN(8) BHW_ROT_PLAIN(9)
                 "10:\n\ts_branch 200f\n"
                 "100:\n\t"
                 BHW_ROT_DISPATCH(2)
                 "s_branch 200f\n"
                 BHW_ROT_HI(0) BHW_ROT_HI(1) BHW_ROT_HI(2) BHW_ROT_HI(3) BHW_ROT_HI(4) BHW_ROT_HI(5) BHW_ROT_HI(6) BHW_ROT_HI(7) BHW_ROT_HI(8) BHW_ROT_HI(9)
                 "200:"
                 : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [za] "+v"(zacc), [a] "=&v"(a), [b] "=&v"(b)
                 : [e] "s"(e), [hi] "s"(hi), [l0] "s"(lut[KA]), [l1] "s"(lut[KA + 1]), [l2] "s"(lut[KA + 2]), [l3] "s"(lut[KA + 3]), [l4] "s"(lut[KA + 4]),
                   [l5] "s"(lut[KA + 5]), [l6] "s"(lut[KA + 6]), [l7] "s"(lut[KA + 7]), [l8] "s"(lut[KA + 8]), [l9] "s"(lut[KA + 9]),
                   [k0] "n"(KA), [k1] "n"(KA + 1), [k2] "n"(KA + 2), [k3] "n"(KA + 3), [k4] "n"(KA + 4), [k5] "n"(KA + 5), [k6] "n"(KA + 6),
                   [k7] "n"(KA + 7), [k8] "n"(KA + 8), [k9] "n"(KA + 9),
                   [c0] "n"(1 << S), [c1] "n"(1 << (S - 1)), [c2] "n"(1 << (S - 2)), [c3] "n"(1 << (S - 3)), [c4] "n"(1 << (S - 4)),
                   [c5] "n"(1 << (S - 5)), [c6] "n"(1 << (S - 6)), [c7] "n"(1 << (S - 7)), [c8] "n"(1 << (S - 8)), [c9] "n"(1 << (S - 9))
                 : "vcc", "scc");
#undef BHW_ROT_I
#undef BHW_ROT_PLAIN
#undef BHW_ROT_HI
#undef BHW_ROT_DISPATCH
}

// Timeline instrumentation (development builds only, -DBHW_BUILD_STAMPS; tools/build_timeline.py): every workgroup records the
// 100 MHz wall clock at its phase boundaries -- 16 words per workgroup at g_build_stamps.
#ifdef BHW_BUILD_STAMPS
__device__ unsigned long long *g_build_stamps = nullptr;
#define BHW_STAMP(i) do { if (g_build_stamps && (threadIdx.x & 63u) == 0u) atomicMax(&g_build_stamps[blockIdx.x * 16u + (i)], (unsigned long long)wall_clock64()); } while (0)
#define BHW_STAMP_MIN(i) do { if (g_build_stamps && threadIdx.x == 0u) g_build_stamps[blockIdx.x * 16u + (i)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define BHW_STAMP(i) do { } while (0)
#define BHW_STAMP_MIN(i) do { } while (0)
#endif

// One packed table entry, stored at agent scope (sc1): written through the XCD's L2 to memory as it is produced.  With plain
// (write-back) stores the whole table -- 17 MB, less than the eight L2s hold together -- stayed dirty in the L2s until the
// end-of-kernel release wrote it back: 5 us during which nothing else ran (profiles/r04_ab_store_policy.txt: 0.1069 -> 0.1022 ms
// per window; sc0 alone changes nothing; nontemporal stores make the build as fast but leave the table out of the memory-side
// cache, and the combine pass then takes 94 us instead of 67).  The combine pass runs on other XCDs and reads the table through
// their own L2s, so device scope is also what the data needs.
// (scalar table base + 32-bit byte offset: the saddr form, no 64-bit vector add per store; the value's upper bits are ignored by
// the narrow store, so the caller need not mask them)
template <int BYTES>
__device__ __forceinline__ void table_store(void *__restrict__ table, uint32_t idx, uint32_t v)
{
    if constexpr (BYTES == 1) asm volatile("global_store_byte %0, %1, %2 sc1" :: "v"(idx), "v"(v), "s"(table) : "memory");
    else asm volatile("global_store_short %0, %1, %2 sc1" :: "v"(idx << 1), "v"(v), "s"(table) : "memory");
}

template <int NITER, int FMT, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_table_build_mirror(BhwBuildPlan plan, void *__restrict__ table)
{
    BHW_STAMP_MIN(0);
#ifdef BHW_BUILD_STAMPS
    if (g_build_stamps && threadIdx.x == 0u) {                       // where this workgroup runs: HW_ID (CU / SH / SE) and the XCD
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        g_build_stamps[blockIdx.x * 16u + 6u] = hw;
        g_build_stamps[blockIdx.x * 16u + 7u] = xcc;
    }
#endif
    static_assert(FMT == 2 || FMT == 3 || FMT == 5, "residual / nibble entries");
    static_assert(NITER >= 21 && NITER <= 32, "narrow state: 32-bit words");
    static_assert(THREADS == 256 || THREADS == 1024, "workgroup shapes");
    constexpr uint32_t gpw = mirror_gpw(THREADS);               // own groups per workgroup
    constexpr uint32_t kHeads = mirror_heads_max(THREADS), kHeadsPad = mirror_heads_pad(THREADS);
    static_assert(gpw + 2 * kHeadsPad + 64 <= THREADS, "prefix lanes, two ranges of head chains, the tail-table wave");
    __shared__ int64_t gx[gpw];
    __shared__ int64_t gy[gpw];
    __shared__ int32_t gz[gpw];
    __shared__ int32_t gk[gpw];
    __shared__ uint32_t gflag[gpw];                             // bit 0: leaf 0 met z == 0 inside the shared prefix; bit 1: narrow state; bit 2: ... with x biased
    __shared__ uint32_t lut_s[32];
    __shared__ uint8_t tail_p[2 * kTailZ];                      // z_KS + kTailZ -> decision pattern of the tail
    __shared__ __attribute__((aligned(16))) int8_t tail_d[kTailRows * 64 + 16];   // 65 rows of 64 patterns (+ padding to whole 16-byte packets)
    constexpr uint32_t kWorkMax = 8 * gpw;           // images to run as chains of their own (expected ~0.6 per group)
    __shared__ uint32_t work_n;
    __shared__ uint32_t work_u[kWorkMax];
    __shared__ uint32_t esc_n;                       // nibble + escapes: entries this workgroup has listed
    __shared__ int4 esc_list[FMT == 5 ? kEscFill : 1];   // ... in arrival order; hashed into esc_tab (esc_lookup) and written out at the end
    __shared__ int4 esc_tab[FMT == 5 ? kEscSlots : 1];
    // (all kernel arguments requested in one batch of scalar loads up front: measured, no gain -- 0.0966 / 0.0977 ms; the serial start is
    // bound by the dependent 64-bit rotations of the prefix and head chains, profiles/r04_build_timeline_final.txt)
    if (threadIdx.x < 32) lut_s[threadIdx.x] = plan.lut[threadIdx.x];
    if constexpr (FMT == 5) {
        if (threadIdx.x >= 64u && threadIdx.x < 64u + kEscSlots) esc_tab[threadIdx.x - 64u] = make_int4(-1, 0, 0, 0);   // (first used behind the barriers below)
    }
#ifdef BHW_BUILD_STAMPS
    if (g_build_stamps && threadIdx.x == 0u) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); g_build_stamps[blockIdx.x * 16u + 8u] = (unsigned long long)wall_clock64() | (plan.entries & 0u); }   // kernel arguments have arrived
#endif
    constexpr int n_iter = NITER;
    constexpr int KS = NITER - kMirrorTail;                     // image state taken at this rotation
    const uint32_t s = plan.z_shl;
    const uint32_t group0 = blockIdx.x * gpw;
    const uint32_t E = plan.entries;
    const uint32_t n_groups = E >> 7;   // groups that run chains of their own
    const uint32_t u_lo = group0 << 6;
    const uint32_t u_hi = ((group0 + gpw) << 6) < (n_groups << 6) ? ((group0 + gpw) << 6) : (n_groups << 6);   // own entries [u_lo, u_hi)
    const uint32_t m_last = (E >> 1) - 1u;    // u in [1, m_last] also produce the image E - u
    if (threadIdx.x == 0) {
        work_n = 0u;
        esc_n = 0u;
        if (u_hi == (E >> 1)) {                  // the middle entry E/2 (its own image): one more deferred chain
            work_u[0] = E >> 1;
            work_n = 1u;
        }
    }
    // tail tables, by the last wave alone (the first ones run the prefix and head chains below, and every workgroup starts
    // with this serial phase): D copied from its constant image, the pattern of every z_KS from the ROM words
    if (threadIdx.x >= (uint32_t)THREADS - 64u) {
        const uint32_t t3 = threadIdx.x - ((uint32_t)THREADS - 64u);
#pragma unroll
        for (int q = 0; q < 5; ++q)
            if (t3 + 64u * q < (uint32_t)(kTailRows * 64 + 16) / 16u) reinterpret_cast<int4 *>(tail_d)[t3 + 64u * q] = reinterpret_cast<const int4 *>(kTailD.v)[t3 + 64u * q];
#pragma unroll
        for (int q = 0; q < 2 * kTailZ / 64; ++q) {
            const uint32_t zi = t3 + 64u * (uint32_t)q;
            int32_t z = (int32_t)zi - kTailZ;
            uint32_t p = 0u;
#pragma unroll
            for (int j = 0; j < kMirrorTail; ++j) {
                if (z < 0) { p |= 1u << j; z += (int32_t)plan.lut[KS + j]; } else z -= (int32_t)plan.lut[KS + j];
            }
            tail_p[zi] = (uint8_t)p;
        }
        BHW_STAMP(9);                                                // tail tables filled
    }
    // (no barrier here: the tail tables and lut_s are first read in phase 2, behind the two barriers below)


    // records {c, s, dc, ds} of the cells this workgroup stores into: w = 0 its own range, w = 1 the image range.  Heads
    // lo .. lo + n, plus head lo - 1 for the table's last cell (see k_table_build_shared), by full chains of the lanes behind the prefix lanes.
    __shared__ int32_t hc[2][kHeads], hs[2][kHeads];
    const uint32_t d = fmt_cell_log(plan.tab_dlog);
    const uint32_t cells_total = E >> d;
    uint32_t cell_lo[2] = {0u, 0u}, n_cell[2] = {0u, 0u}, r_lo[2] = {u_lo, 0u}, r_hi[2] = {u_hi, 0u};   // entry ranges [r_lo, r_hi)
    {
        const uint32_t a = u_lo > 1u ? u_lo : 1u, b = (u_hi - 1u) < m_last ? (u_hi - 1u) : m_last;        // sources a .. b
        if (a <= b) { r_lo[1] = E - b; r_hi[1] = E - a + 1u; }
        if (a <= b && b == (E >> 1) - 1u) r_lo[1] = E >> 1;      // the middle entry belongs to this image range
    }
#pragma unroll
    for (int w = 0; w < 2; ++w)
        if (r_hi[w] > r_lo[w]) {
            cell_lo[w] = r_lo[w] >> d;
            n_cell[w] = ((r_hi[w] - 1u) >> d) - cell_lo[w] + 1u;
        }
    if (threadIdx.x >= gpw && threadIdx.x < gpw + 2u * kHeadsPad) {
        const uint32_t w = (threadIdx.x - gpw) / kHeadsPad;          // lanes [gpw, gpw + pad) -> own range, the next pad lanes -> image range
        const uint32_t t = (threadIdx.x - gpw) % kHeadsPad;
        if (n_cell[w] && t < n_cell[w] + 2u) {
            const int64_t cell = (t <= n_cell[w]) ? (int64_t)cell_lo[w] + t : (int64_t)cell_lo[w] - 1;
            if (cell >= 0 && cell < (int64_t)cells_total) {
                int64_t x = plan.x0, y = plan.x0;
                int32_t z = (int32_t)((((uint32_t)cell << d) << s) - plan.lut[0]);
#pragma unroll
                for (int r = 1; r < n_iter; ++r) rot_step(x, y, z, r, plan.lut[r]);
                hc[w][t] = (int32_t)(x >> plan.out_shr);
                hs[w][t] = (int32_t)(y >> plan.out_shr);
            }
        }
        BHW_STAMP(10);                                               // head chains done
    }

    // ---- phase 1: shared prefix of each 64-leaf group (never past KS: the image state is taken there) ----
    constexpr int kcap = KS < kPrefixMax ? KS : kPrefixMax;
    if (threadIdx.x < gpw) {
        const uint32_t g = group0 + threadIdx.x;
        const uint32_t u_first = g << 6;
        int64_t x = plan.x0, y = plan.x0;                                        // after rotation 0
        int32_t zf = (int32_t)((u_first << s) - plan.lut[0]);
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
        // narrow state: every later x_k, y_k of every leaf stays inside [0, 2^NITER) (see the head comment)
        const int64_t drift = ((int64_t)1 << (NITER - k + 1)) + 2, lim = (int64_t)1 << NITER;
        const bool narrow = x >= drift && y >= drift && x + drift < lim && y + drift < lim;
        // biased narrow state (rot_narrow_block<.., true>): x - B fits the word and x stays below row 65 of the tail table; only groups
        // that enter the ten-rotation block (tools/sim_build.cpp checks every entry of such tables against the plain chain)
        constexpr int64_t kBias = (int64_t)1 << (NITER - 1);
        const bool hi = !narrow && k >= KS - 10 && y >= drift && y + drift < lim && x - kBias >= drift && x - kBias + drift < ((int64_t)1 << 32) &&
                        x + drift < lim + ((int64_t)1 << KS);
        gx[threadIdx.x] = hi ? x - kBias : x;
        gy[threadIdx.x] = y;
        gz[threadIdx.x] = zf;
        gk[threadIdx.x] = k;
        gflag[threadIdx.x] = zero0 | (narrow || hi ? 2u : 0u) | (hi ? 4u : 0u);
        BHW_STAMP(11);                                               // prefixes done
    }
    BHW_STAMP(1);                                                    // last wave to reach the first barrier
    __syncthreads();

    // the records themselves, once per workgroup: the groups read them back with one ds_read_b128 each
    __shared__ int4 hrec[2][kHeads];
    {
        const uint32_t w = threadIdx.x / kHeadsPad, t = threadIdx.x % kHeadsPad;
        if (w < 2u && t < n_cell[w]) {
            const uint32_t cell = cell_lo[w] + t;
            int4 r;
            if (cell + 1u < cells_total) r = make_int4(hc[w][t], hs[w][t], hc[w][t + 1] - hc[w][t], hs[w][t + 1] - hs[w][t]);
            else {
                const uint32_t tp = t ? t - 1u : n_cell[w] + 1u;     // last cell of the table: slope of the cell before it
                r = make_int4(hc[w][t], hs[w][t], hc[w][t] - hc[w][tp], hs[w][t] - hs[w][tp]);
            }
            if constexpr (FMT == 5) { r.x += (int32_t)kEscBias; r.y += (int32_t)kEscBias; }   // the line the deviations are centred on
            if constexpr (fmt_is_nibble(FMT)) { r.x -= 8; r.y -= 8; }                         // the fields are stored unsigned: deviation + 8
            hrec[w][t] = r;
            // (the table's copy of a nibble record carries the slopes doubled -- tab_predict_nib: the tile kernel evaluates the line with
            // one v_mul_hi_i32 per coordinate; this kernel keeps the plain slopes and the shift by d it has in a register anyway)
            if constexpr (fmt_is_nibble(FMT)) { r.z *= 2; r.w *= 2; }
            if ((cell << d) >= r_lo[w])                              // its first entry is stored by this workgroup
                reinterpret_cast<int4 *>(const_cast<void *>(plan.tab_coarse))[cell] = r;
        }
    }
    __syncthreads();
    BHW_STAMP_MIN(2);
    auto record_of = [&](int w, uint32_t cell) -> int4 { return hrec[w][cell - cell_lo[w]]; };   // cell in [cell_lo[w], cell_lo[w] + n_cell[w])
    auto record = [&](int w, uint32_t cell) -> int4 {              // the same for a wave-uniform cell (a broadcast read)
        return record_of(w, __builtin_amdgcn_readfirstlane(cell));
    };

    // one entry: deviation from the record's straight line, checked and packed
    auto store_entry = [&](uint32_t idx, int32_t c, int32_t sn, const int4 rec, uint32_t pos) {
        if constexpr (FMT == 5) {
            // the deviation that does not fit (or collides with the marker) goes to this workgroup's list, the marker into the table
            const int2 p = tab_predict(rec, pos, d);                                            // (hrec: plain slopes)
            const uint32_t nc = (uint32_t)(c - p.x), ns = (uint32_t)(sn - p.y);                 // the biased fields: deviation + 8
            uint32_t v = nc | (ns << 4);
            // (both tests evaluated, then OR-ed: a short-circuit && made the compiler fetch the record's second half and form ds inside a branch)
            if (__builtin_expect((int)(nc - 1u > 14u) | (int)(ns > 15u), 0)) {                   // dev_c in -7 .. 7 (-8 is the marker), dev_s in -8 .. 7
                const uint32_t slot = atomicAdd(&esc_n, 1u);
                if (slot < kEscFill) esc_list[slot] = make_int4((int32_t)idx, c, sn, 0);    // (beyond: the format is refused below)
                v = kEscMarker;
            }
            table_store<1>(table, idx, v);
        } else if constexpr (FMT == 3) {
            const int2 p = tab_predict(rec, pos, d);                                            // (hrec: plain slopes)
            const uint32_t nc = (uint32_t)(c - p.x), ns = (uint32_t)(sn - p.y);
            if (plan.check_flag && (nc | ns) > 15u) atomicOr(plan.check_flag, 1u);
            table_store<1>(table, idx, nc | (ns << 4));                                           // bits 8.. are not stored
        } else {
            const int2 p = tab_predict(rec, pos, d);
            const int32_t dc = c - p.x, ds = sn - p.y;
            if (plan.check_flag && !(fits_bits(dc, 8) && fits_bits(ds, 8))) atomicOr(plan.check_flag, 1u);
            table_store<2>(table, idx, ((uint32_t)dc & 0xFFu) | ((uint32_t)ds << 8));           // bits 16.. are not stored
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
    // (groups handed out dynamically -- an LDS counter, the next index requested a group ahead -- instead of this fixed interleave:
    // 0.0966 -> 0.0988 ms, profiles/HISTORY.md)
    for (uint32_t gi = wave; gi < gpw; gi += THREADS / 64) {
        const uint32_t g = __builtin_amdgcn_readfirstlane(group0 + gi);   // (kept in a vector register otherwise, and the cell arithmetic with it)
        if (g >= n_groups) break;
        // Issue priority by progress.  The SIMD's arbiter serves the oldest wave first, so of the two workgroups of a CU the one
        // that was started first runs ahead, finishes its 16 rounds of groups ~12 us before the other (profiles/r04_build_timeline.txt:
        // 21 against 33 us) and leaves the CU half empty for the rest of the pass -- and there is no second round of workgroups to
        // refill it.  A wave lowers its own priority every four groups: whoever is behind is served first, the waves of both
        // workgroups stay within four groups of each other and the CU stays full until the end (0.1076 -> 0.1047 ms per window,
        // profiles/r04_ab_prio.txt).
        {
            const uint32_t it = (gi - wave) / (uint32_t)(THREADS / 64);
            if (it == 0u) __builtin_amdgcn_s_setprio(3);
            else if (it == 4u) __builtin_amdgcn_s_setprio(2);
            else if (it == 8u) __builtin_amdgcn_s_setprio(1);
            else if (it == 12u) __builtin_amdgcn_s_setprio(0);
        }
        const int k0 = __builtin_amdgcn_readfirstlane(gk[gi]);
        const uint32_t gf = __builtin_amdgcn_readfirstlane(gflag[gi]);
        int32_t z = (int32_t)((uint32_t)gz[gi] + (lane << s));
        uint64_t zmask = 0ull;                                       // lanes whose z_k was exactly 0 at a rotation before KS
        int32_t c1, s1, c2, s2;                                      // the entry and its image
        bool replay = false;                                         // a lane whose image cannot come from the hand-over state
        if (gf & 2u) {
            // ---- narrow state: 32-bit words, EXEC-masked add / sub, table tail ----
            uint32_t x = (uint32_t)gx[gi], y = (uint32_t)gy[gi];
            uint32_t zacc = ~0u;                                      // min over the rotations of (unsigned)z_k: 0 = a zero event
            // entry at the group's split rotation k0 (17 or 18 at 2^26 / 32 bits): rotations below KA = KS - 10 behind one scalar guard
            // each (rare: a group that splits that early), the last ten as one block entered at k0 (a switch with fall-through is
            // lowered to a flag machine that costs 10 us of the pass)
            constexpr int KA = KS - 10;
            static_assert(KA >= 5 && KA < kcap && KS - KA == 10, "ten-rotation block");
            if (k0 < KA) {
#define BHW_NARROW(K) if constexpr (K < KA) { if (K >= k0) rot_narrow<K, true>(x, y, z, zacc, plan.lut[K]); }
                BHW_NARROW(1) BHW_NARROW(2) BHW_NARROW(3) BHW_NARROW(4) BHW_NARROW(5) BHW_NARROW(6) BHW_NARROW(7) BHW_NARROW(8)
                BHW_NARROW(9) BHW_NARROW(10) BHW_NARROW(11) BHW_NARROW(12) BHW_NARROW(13) BHW_NARROW(14) BHW_NARROW(15)
#undef BHW_NARROW
            }
            const bool hi = (gf & 4u) != 0u;                          // scalar: x holds x - 2^(NITER-1)
            rot_narrow_block<KA, NITER>(x, y, z, zacc, plan.lut, k0 - KA, hi ? 1u : 0u);
            zmask = __builtin_amdgcn_ballot_w64(zacc == 0u);
            const uint32_t xrow6 = hi ? (1u << (NITER - 1 - KS + 6)) : 0u;   // (B >> KS) << 6: the bias in the tail table's row index
            const uint32_t bias_out = hi ? ((1u << (NITER - 1)) >> plan.out_shr) : 0u;   // B >> out_shr
            static_assert(KS <= 26, "unrolled to rotation 25");
            constexpr uint32_t lowm = (1u << KS) - 1u;
            const uint32_t zi = (uint32_t)(z + kTailZ);              // own pattern at zi, the image's (z -> -z) at 2 kTailZ - zi
            const uint32_t mx = (x + 128u) & lowm, my = (y + 128u) & lowm;      // margin of 128 to both ends of the low KS bits
            const uint32_t unsafe = (uint32_t)((mx < my ? mx : my) < 256u) | (uint32_t)((zi - 1u) >= (uint32_t)(2 * kTailZ - 1));
            if (__builtin_amdgcn_ballot_w64(unsafe != 0u) == 0ull) {
                const uint32_t xx = ((x >> KS) << 6) + xrow6, yy = (y >> KS) << 6;   // row of D: y < 2^NITER, so y >> KS < 64; x >> KS <= 64
                const uint32_t p1 = tail_p[zi], p2 = tail_p[2u * kTailZ - zi];
                const int32_t dx1 = tail_d[p1 + yy], dy1 = tail_d[p1 + xx], dx2 = tail_d[p2 + xx], dy2 = tail_d[p2 + yy];
                c1 = (int32_t)(((x - (uint32_t)dx1) >> plan.out_shr) + bias_out);
                s1 = (int32_t)((y + (uint32_t)dy1) >> plan.out_shr);
                c2 = (int32_t)((y - (uint32_t)dx2) >> plan.out_shr);
                s2 = (int32_t)(((x + (uint32_t)dy2) >> plan.out_shr) + bias_out);
            } else if (hi) {
                // (rare: a lane outside the tail table's margins in a biased group) the six rotations on the 64-bit state
                int64_t xw = (int64_t)x + ((int64_t)1 << (NITER - 1)), yw = (int64_t)y;
                int64_t x2 = yw, y2 = xw;
                int32_t z2 = -z;
#pragma unroll 1
                for (int k = KS; k < NITER; ++k) {
                    rot_step_dyn(xw, yw, z, k, lut_s[k], true);
                    rot_step_dyn(x2, y2, z2, k, lut_s[k], true);
                }
                c1 = (int32_t)(xw >> plan.out_shr); s1 = (int32_t)(yw >> plan.out_shr);
                c2 = (int32_t)(x2 >> plan.out_shr); s2 = (int32_t)(y2 >> plan.out_shr);
            } else {
                uint32_t x2 = y, y2 = x;
                int32_t z2 = -z;
                uint32_t unused = 0u;
#define BHW_NARROW(K) if constexpr (K >= KS && K < NITER) { rot_narrow<K, false>(x, y, z, unused, plan.lut[K]); rot_narrow<K, false>(x2, y2, z2, unused, plan.lut[K]); }
                BHW_NARROW(15) BHW_NARROW(16) BHW_NARROW(17) BHW_NARROW(18) BHW_NARROW(19) BHW_NARROW(20) BHW_NARROW(21) BHW_NARROW(22)
                BHW_NARROW(23) BHW_NARROW(24) BHW_NARROW(25) BHW_NARROW(26) BHW_NARROW(27) BHW_NARROW(28) BHW_NARROW(29) BHW_NARROW(30)
                BHW_NARROW(31)
#undef BHW_NARROW
                c1 = (int32_t)(x >> plan.out_shr); s1 = (int32_t)(y >> plan.out_shr);
                c2 = (int32_t)(x2 >> plan.out_shr); s2 = (int32_t)(y2 >> plan.out_shr);
            }
        } else {
            // ---- wide state (the first degree of the octant, groups that split early): the 64-bit rotation, rolled ----
            int64_t x = gx[gi], y = gy[gi];
#pragma unroll 1
            for (int k = k0; k < KS; ++k) {
                zmask |= __builtin_amdgcn_ballot_w64(z == 0);
                rot_step_dyn(x, y, z, k, lut_s[k], k >= kMad24From);
            }
            int64_t x2 = y, y2 = x;
            int32_t z2 = -z;
#pragma unroll 1
            for (int k = KS; k < NITER; ++k) {
                rot_step_dyn(x, y, z, k, lut_s[k], true);
                rot_step_dyn(x2, y2, z2, k, lut_s[k], true);
            }
            c1 = (int32_t)(x >> plan.out_shr); s1 = (int32_t)(y >> plan.out_shr);
            c2 = (int32_t)(x2 >> plan.out_shr); s2 = (int32_t)(y2 >> plan.out_shr);
        }
        const uint32_t g6 = __builtin_amdgcn_readfirstlane(g << 6);   // back in a scalar register (merged with the branches' copies it
                                                                      // lands in a vector one, and the wave-uniform cell arithmetic below with it)
        const uint32_t u = g6 + lane;
        const bool has_image = u >= 1u && u <= m_last;
        bool deferred = false;                                       // this lane's image goes to the worklist
        if (zmask != 0ull || gf & 1u) {                              // scalar
            // z_k == 0 before the hand-over (or, leaf 0, inside the shared prefix): the image is a chain of its own
            replay = has_image && ((((zmask >> lane) & 1ull) != 0ull) || ((gf & 1u) && lane == 0u));
            if (replay) {
                const uint32_t slot = atomicAdd(&work_n, 1u);
                if (slot < kWorkMax) { work_u[slot] = u; deferred = true; }
                else {                                               // list full (never seen): the image chain from scratch, in place
                    int64_t xf = plan.x0, yf = plan.x0;
                    int32_t zf = (int32_t)(((E - u) << s) - lut_s[0]);
#pragma unroll 1
                    for (int r = 1; r < n_iter; ++r) rot_step_dyn(xf, yf, zf, r, lut_s[r], r >= kMad24From);
                    c2 = (int32_t)(xf >> plan.out_shr);
                    s2 = (int32_t)(yf >> plan.out_shr);
                }
            }
        }
        // (nibble tables are always in the natural order: scalar shift + one add instead of the 64-bit multiply-add of the general form)
        const uint32_t idx = fmt_is_nibble(FMT) ? (g << 6) + lane : idx_a + g * idx_m;
        store_entry(idx, c1, s1, record(0, g6 >> d), (g6 & fmask) + lane);
        if (has_image && !deferred) {
            // images E - 64g - 63 .. E - 64g, descending with the lane: one cell, or two when lane 0's image opens the next one
            const uint32_t um = E - u;
            const uint32_t top = E - g6, cell_a = (top - 63u) >> d, cell_b = top >> d;             // wave-uniform
            const uint32_t cell_c = cell_a >= cell_lo[1] ? cell_a : cell_lo[1];                      // (sources above m_last are masked off)
            int4 rec2 = record(1, cell_c);
            if (__builtin_amdgcn_readfirstlane(cell_b) != __builtin_amdgcn_readfirstlane(cell_c)) { // scalar branch, 1 group in 2^(d-6)
                const int4 rb = record(1, cell_b);
                if (lane == 0u) rec2 = rb;
            }
            store_entry(fmt_is_nibble(FMT) ? E - idx : idx_i - idx, c2, s2, rec2, um & fmask);
        }
    }
    BHW_STAMP_MIN(3);                                                // first wave done with its groups
    BHW_STAMP(4);                                                    // last wave done
    // ---- the deferred images: one lane each.  The chain is unrolled on the scalar ROM words (the rolled loop on the LDS copy kept
    // every workgroup 1.5 us longer at the end of the kernel, where nothing else is left to overlap it) ----
    __syncthreads();
    const uint32_t n_work = work_n < kWorkMax ? work_n : kWorkMax;
    for (uint32_t i = threadIdx.x; i < n_work; i += THREADS) {
        const uint32_t um = E - work_u[i];
        int64_t xf = plan.x0, yf = plan.x0;
        int32_t zf = (int32_t)((um << s) - plan.lut[0]);
#pragma unroll
        for (int r = 1; r < n_iter; ++r) rot_step(xf, yf, zf, r, plan.lut[r]);     // (restarting from the group's parked state, per-lane start rotation: +0.3 us)
        store_entry(tab_index(um, plan.log2_entries, plan.tab_split), (int32_t)(xf >> plan.out_shr), (int32_t)(yf >> plan.out_shr), record_of(1, um >> d), um & fmask);
    }
    if constexpr (FMT == 5) {                                        // the list's length; a list that overflowed fails the format
        __syncthreads();
        const uint32_t n_esc = esc_n < kEscFill ? esc_n : kEscFill;
        // (the thread index re-read behind an empty asm: compared as threadIdx.x, the lane masks of the kernel's first lines were kept
        // alive across the whole kernel for these three tests -- in spilled scalar registers)
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (tid < n_esc) {                                   // one lane per listed entry: its slot by open addressing
            const int4 it = esc_list[tid];
            uint32_t h = esc_slot((uint32_t)it.x);
#pragma unroll 1
            for (uint32_t i = 0; i < kEscSlots; ++i) {
                const uint32_t was = atomicCAS(reinterpret_cast<uint32_t *>(&esc_tab[h].x), ~0u, (uint32_t)it.x);
                if (was == ~0u || was == (uint32_t)it.x) { esc_tab[h].y = it.y; esc_tab[h].z = it.z; break; }
                h = (h + 1u) & (kEscSlots - 1u);
            }
        }
        __syncthreads();
        if (tid < kEscSlots) reinterpret_cast<int4 *>(plan.tab_esc)[(size_t)blockIdx.x * kEscSlots + tid] = esc_tab[tid];
        if (tid == 0u && esc_n > kEscFill && plan.check_flag) atomicOr(plan.check_flag, 1u);
    }
    BHW_STAMP(5);
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

} // namespace

// (bhwk_packed_ok, bhwk_resid_dlog, bhwk_build_mirror_applies, bhwk_build_mirror_threads: bhw_plan.cpp)

// One whole period of cordic() (bhwk_sincos): the shared-prefix chains of the table build with the four quadrant images written
// straight out (k_table_build_shared<N, 4>).
int bhwk_sincos_sweep(const BhwLaunch &l, const BhwCordicCfg &c, uint64_t theta0, int32_t *d_sin, int32_t *d_cos)
{
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
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
    plan.tab_esc = nullptr;
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

#ifdef BHW_BUILD_STAMPS
extern "C" int bhw_dbg_build_stamps(void *d_words)      // 8 x uint64 per workgroup of the mirror kernel, zeroed by the caller; NULL = off
{
    unsigned long long *p = (unsigned long long *)d_words;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_build_stamps), &p, sizeof p);
}
#endif

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
        plan.tab_esc = const_cast<void *>(c.tab_esc);
        const unsigned groups = entries >> 6;
        plan.groups_per_wg = groups >= 64u * 1024u ? 64u : groups >= 16u * 1024u ? 16u : 4u;
        plan.pad = 0;
        const dim3 grid((groups + plan.groups_per_wg - 1) / plan.groups_per_wg), block(kBuildThreads);
        // plain tables for every rotation count; the packed formats (whole-period tile calls at z_shr == 0, i.e. PW >= 22 and
        // therefore at least 21 rotations: the VHDL model at PW == W runs W - 1 of them) from 21 rotations on
        const int fmt = fmt_of(c.tab_dlog);
        if (c.n_iter < 21 && fmt != 0) return (int)hipErrorInvalidValue;
        if (bhwk_build_mirror_applies(c, entries)) {
            const unsigned own_groups = entries >> 7;
            // 1 024-thread workgroups once they still give every CU two (tables of 2^24 entries and more), 256 below
            const unsigned threads = bhwk_build_mirror_threads(entries);
            const unsigned mgpw = (unsigned)mirror_gpw((int)threads);
            const dim3 mgrid((own_groups + mgpw - 1) / mgpw), mblock(threads);
            plan.groups_per_wg = mgpw;
            switch (c.n_iter) {
#define BHW_CASE_MT(N, F) do { if (threads == 1024u) BHW_LAUNCH((k_table_build_mirror<N, F, 1024>), mgrid, mblock, 0, st, plan, (void *)d_table); \
                               else                  BHW_LAUNCH((k_table_build_mirror<N, F, 256>), mgrid, mblock, 0, st, plan, (void *)d_table); } while (0)
#define BHW_CASE_M(N) case N: if (fmt == 2) BHW_CASE_MT(N, 2); else if (fmt == 3) BHW_CASE_MT(N, 3); else BHW_CASE_MT(N, 5); break;
                BHW_CASE_M(21) BHW_CASE_M(22) BHW_CASE_M(23) BHW_CASE_M(24) BHW_CASE_M(25) BHW_CASE_M(26) BHW_CASE_M(27) BHW_CASE_M(28)
                BHW_CASE_M(29) BHW_CASE_M(30) BHW_CASE_M(31) BHW_CASE_M(32)
#undef BHW_CASE_M
#undef BHW_CASE_MT
            default: return (int)hipErrorInvalidValue;
            }
            return finish(hipSuccess);
        }
#define BHW_LAUNCH_BUILD(N, F) BHW_LAUNCH((k_table_build_shared<N, F>), grid, block, 0, st, plan, (void *)d_table)
#define BHW_CASE(N) case N: BHW_LAUNCH_BUILD(N, 0); break;
#define BHW_CASE_T(N) case N: if (fmt == 0) BHW_LAUNCH_BUILD(N, 0); else if (fmt == 1) BHW_LAUNCH_BUILD(N, 1); else return (int)hipErrorInvalidValue; break;   /* residual / nibble tables: the mirror kernel above (tables of 2^20 entries and more, the only ones that use them) */
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

