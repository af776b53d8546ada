// bhw_plan.cpp -- the HIP-free half of the host side (see bhw_plan.h): validation, resolution of (model, widths) into kernel
// constants, strategy / format / shape decisions, ownership segments, scratch sizing, bhw_describe_plan.  Plain C++: no hip*
// include, no device state; swept under AddressSanitizer + UBSan by tests/test_sanitizers.py.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "bhw_plan.h"
#include "bhw_tables.inc"

namespace {

thread_local std::string g_last_error;

// Built-in float weights: hls/windows/win_function.cpp:173-174,191-192,206-208,253-256,306-310,341-347.
const double kHamming[2] = {0.5434783, 1 - 0.5434783};
const double kHann[2] = {0.5, 0.5};
const double kBh3[3] = {0.21, 0.25, 0.04};
const double kBh4[4] = {0.35875, 0.48829, 0.14128, 0.01168};
const double kBh5[5] = {0.3232153788877343, 0.4714921439576260, 0.1755341299601972, 0.0284969901061499,
                        0.0012613570882927};
const double kBh7[7] = {0.271220360585039, 0.433444612327442, 0.218004122892930, 0.065785343295606,
                        0.010761867305342, 0.000770012710581, 0.000013680883060};

const uint32_t kSelSize[25] = {15, 15, 15, 18, 21, 22, 23, 26, 30, 31, 32, 33,           // src/cordic_dds_scaled.vhd:102-107
                               38, 38, 38, 42, 42, 45, 47, 47, 47, 48, 48, 48, 48};

std::mutex g_fmt_mu;
std::map<uint64_t, int> g_fmt_verdict;

uint64_t fmt_key(const bhw_params *p, uint32_t dlog)
{
    return ((uint64_t)p->model << 40) | ((uint64_t)p->phi_width << 32) | ((uint64_t)p->dat_width << 24) |
           ((uint64_t)(p->model == BHW_MODEL_VHDL ? p->precision : 0u) << 16) | dlog;
}

uint64_t align256(uint64_t v) { return (v + 255ull) & ~255ull; }

// Whole periods up to this length go through the fused kernel under AUTO: one launch of 5/8 .. 9/8 chains per coefficient beats
// two dependent launches around a table of 1/4 chain per coefficient while the call is launch- and latency-bound.  Measured per
// call (profiles/r02_small_windows.json): BH-4/24-bit fused 8.0 / 12.1 / 17.0 us at 2^20 / 2^21 / 2^22 against 11.7 / 14.6 /
// 19.9 us for the table strategy; BH-7/32-bit 9.0 (2^16) / 13.3 / 21.8 / 33.5 us against 11.9 / 11.9 / 16.8 / 27.5 us.
uint32_t fused_max_pw(uint32_t n_terms) { return n_terms <= 5 ? 22u : 19u; }

uint32_t inv_mod_pow2(uint32_t a, uint32_t log2m)
{
    uint32_t x = a;                      // Newton iteration: x <- x (2 - a x), doubles the correct bits
    for (int i = 0; i < 6; ++i) x *= 2u - a * x;
    return log2m >= 32 ? x : (x & ((1u << log2m) - 1u));
}

int mode_of(const BhwCordicCfg &c, const BhwWinCfg &w) { return (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0); }

} // namespace

int bhwp_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int bhwp_terms_of(uint32_t win_type)
{
    switch (win_type) {
    case BHW_WIN_HAMMING: case BHW_WIN_HANN: return 2;
    case BHW_WIN_BH3: return 3;
    case BHW_WIN_BH4: return 4;
    case BHW_WIN_BH5: return 5;
    case BHW_WIN_BH7: return 7;
    default: return 0;
    }
}

int bhwp_validate(const bhw_params *p, bool sincos_only)
{
    if (!p) return bhwp_fail(BHW_ERR_BADARG, "params is NULL");
    if (p->struct_size != sizeof(bhw_params))
        return bhwp_fail(BHW_ERR_BADARG, "struct_size %u != %zu", p->struct_size, sizeof(bhw_params));
    if (p->model > BHW_MODEL_SCALED) return bhwp_fail(BHW_ERR_BADARG, "model %u", p->model);
    if (p->model > BHW_MODEL_VHDL && !sincos_only)
        return bhwp_fail(BHW_ERR_UNSUPPORTED, "cordic_dds48 / cordic_dds_scaled feed no window entity: bhw_sincos_* only");
    if (p->combine > BHW_COMBINE_VHDL) return bhwp_fail(BHW_ERR_BADARG, "combine %u", p->combine);
    if (p->sin_type > BHW_SIN_TAYLOR_ALL) return bhwp_fail(BHW_ERR_BADARG, "sin_type %u", p->sin_type);
    const uint32_t K = p->n_terms;
    if (!(K == 2 || K == 3 || K == 4 || K == 5 || K == 7)) return bhwp_fail(BHW_ERR_BADARG, "n_terms %u (2,3,4,5,7)", K);
    const uint32_t PW = p->phi_width, W = p->dat_width;
    if (PW < 4 || PW > 30) return bhwp_fail(BHW_ERR_BADARG, "phi_width %u outside 4..30", PW);
    if (W < 8 || W > 32) return bhwp_fail(BHW_ERR_BADARG, "dat_width %u outside 8..32", W);
    if (p->sin_type != BHW_SIN_CORDIC) {
        // win_selector wires the Taylor source only to HAMMING and BH3TERM: src/win_selector.vhd:93-135
        if (K > 3 && p->sin_type == BHW_SIN_TAYLOR)
            return bhwp_fail(BHW_ERR_UNSUPPORTED, "Taylor source exists only for 2- and 3-term windows (BHW_SIN_TAYLOR_ALL is the extension)");
        const uint32_t L = p->lut_size;
        if (L < 1 || L > 16) return bhwp_fail(BHW_ERR_BADARG, "lut_size %u outside 1..16", L);
        // generators in use: PHASE_WIDTH - v, v = 0 .. vmax  (bh_win_3term.vhd:221-226; k = 4 needs v = 2)
        const uint32_t vmax = K > 4 ? 2u : K > 2 ? 1u : 0u;
        if (PW < 3 + vmax) return bhwp_fail(BHW_ERR_UNSUPPORTED, "phi_width %u too short for the PHASE_WIDTH-%u generator", PW, vmax);
        const uint32_t pw_min = PW - vmax;
        for (uint32_t pw = pw_min; pw <= PW; ++pw) {
            const int d = (int)pw - (int)L;
            if (d > 2) {
                if (d - 3 > 15) return bhwp_fail(BHW_ERR_UNSUPPORTED, "Taylor STAGE %d > 15 (tay1_order cnt_exp is 16 bits)", d - 3);
                if (W < 19 && 19 + L + W > 48) return bhwp_fail(BHW_ERR_UNSUPPORTED, "Taylor narrow path: 19+L+W > 48 DSP bits");
                if (W > 18 && 19 + L + W > 62) return bhwp_fail(BHW_ERR_UNSUPPORTED, "Taylor wide path: 19+L+W > 62 product bits");
            }
        }
        return BHW_OK;
    }
    if (p->model == BHW_MODEL_HLS && PW > W + 2)
        return bhwp_fail(BHW_ERR_UNSUPPORTED, "HLS model is ill-defined for phi_width > dat_width + 2 (init_t truncation)");
    if (p->model == BHW_MODEL_VHDL && (p->precision < 1 || p->precision > 7))
        return bhwp_fail(BHW_ERR_BADARG, "precision %u outside 1..7", p->precision);
    return BHW_OK;
}

int bhwp_validate_atan2(const bhw_atan2_params *p)
{
    if (!p) return bhwp_fail(BHW_ERR_BADARG, "params is NULL");
    if (p->struct_size != sizeof(bhw_atan2_params))
        return bhwp_fail(BHW_ERR_BADARG, "struct_size %u != %zu", p->struct_size, sizeof(bhw_atan2_params));
    if (p->precision < 1 || p->precision > 7) return bhwp_fail(BHW_ERR_BADARG, "precision %u outside 1..7", p->precision);
    if (p->angle_width < 4 || p->angle_width > 32) return bhwp_fail(BHW_ERR_BADARG, "angle_width %u outside 4..32", p->angle_width);
    if (p->input_width > 32) return bhwp_fail(BHW_ERR_BADARG, "input_width %u > 32", p->input_width);
    if (p->input_width + 1 < p->angle_width)   // VEC_DX(ii) for ii = 0 .. ANGLE_WIDTH-2: src/cordic_atan2.vhd:142-145
        return bhwp_fail(BHW_ERR_UNSUPPORTED, "input_width %u < angle_width-1: the entity indexes input bits 0..ANGLE_WIDTH-2", p->input_width);
    return BHW_OK;
}

// Resolve the CORDIC constants.
void bhwp_resolve_cordic(const bhw_params *p, BhwCordicCfg &c)
{
    memset(&c, 0, sizeof c);
    const uint32_t PW = p->phi_width, W = p->dat_width;
    c.phi_width = PW;
    c.dat_width = W;
    uint32_t n_lut = W - 1;
    switch (p->model) {
    case BHW_MODEL_HLS:  // hls/windows/win_function.cpp:77-96
        for (uint32_t i = 0; i < n_lut; ++i) c.lut[i] = kAtanT4[i] >> (47 - W);
        c.x0 = kGain46 >> (46 - W);
        c.n_iter = W;
        if (PW - 1 < W) { c.z_shr = 0; c.z_shl = W - PW + 2; } else { c.z_shr = PW - W; c.z_shl = 2; }
        c.out_shr = 2;
        c.ones_neg = 0;
        c.wide = (W + 2 > 32);
        break;
    case BHW_MODEL_CPP:  // cpp/cordic_sincos.cpp:15-36
        for (uint32_t i = 0; i < n_lut; ++i) c.lut[i] = kAtanT2[i] >> (47 - W);
        c.x0 = kGain46 >> (46 - W);
        c.n_iter = W;
        if (PW - 1 < W) { c.z_shr = 0; c.z_shl = W - PW + 1; } else { c.z_shr = PW - W; c.z_shl = 1; }
        c.out_shr = 2;
        c.ones_neg = 1;
        c.wide = (W + 2 > 32);
        break;
    default: {           // src/cordic_dds.vhd:97-131,159-166
        const uint32_t P = p->precision, Wi = W + P;
        for (uint32_t i = 0; i < n_lut; ++i) c.lut[i] = kAtanT4[i] >> (49 - Wi);
        c.x0 = kGain47 >> (49 - Wi);
        c.n_iter = W - 1;
        if (PW >= W) { c.z_shr = PW - W; c.z_shl = P; } else { c.z_shr = 0; c.z_shl = W - PW + P; }
        c.out_shr = P;
        c.ones_neg = 0;
        c.wide = (Wi > 32);
        break;
    }
    }
}

void bhwp_resolve_window(const bhw_params *p, BhwWinCfg &w)
{
    memset(&w, 0, sizeof w);
    for (int k = 0; k < 7; ++k) w.aa[k] = p->aa[k];
    w.n_terms = p->n_terms;
    w.combine = p->combine;
}

// cordic_dds48: SIZE = DWPH = 48 (src/cordic_dds48.vhd:143-153); cordic_dds_scaled: SIZE = SEL_SIZE(DATA_WIDTH-8),
// DWPH = max(SIZE, PHASE_WIDTH) (src/cordic_dds_scaled.vhd:109,133-143)
void bhwp_resolve_prerot(const bhw_params *p, BhwPrerotCfg &c)
{
    memset(&c, 0, sizeof c);
    c.phi_width = p->phi_width;
    c.dat_width = p->dat_width;
    c.size = p->model == BHW_MODEL_DDS48 ? 48u : kSelSize[p->dat_width - 8];
    c.dwph = c.size < p->phi_width ? p->phi_width : c.size;
    c.gain = kGain46 >> (48 - c.size);                                          // GAIN48(47 downto 48-SIZE)
    for (uint32_t i = 0; i + 1 < p->dat_width; ++i) c.lut[i] = kAtanT2[i] >> (48 - c.dwph);   // ROM_LUT(ii)(47 downto 48-DWPH)
}

void bhwp_resolve_atan2(const bhw_atan2_params *p, BhwAtan2Cfg &c)
{
    memset(&c, 0, sizeof c);
    c.precision = p->precision;
    c.input_width = p->input_width;
    c.angle_width = p->angle_width;
    const uint32_t B = p->angle_width + p->precision;
    for (uint32_t i = 0; i + 1 < p->angle_width; ++i) c.lut[i] = kAtanT4[i] >> (49 - B);   // src/cordic_atan2.vhd:100-103
}

bool bhwp_has_whole_period(const bhw_params *p, uint64_t n0, uint64_t count)
{
    const uint64_t N = 1ull << p->phi_width;
    return count >= (N - n0 % N) % N + N;
}

// AUTO: the fused kernel for short whole periods; else build the shared table when it replaces clearly more CORDIC chains
// than it costs; else one chain per harmonic per coefficient.
uint32_t bhwp_pick_algo(const bhw_params *p, const BhwCordicCfg &c, const BhwWinCfg &w, uint64_t n0, uint64_t count, uint32_t requested)
{
    if (p->sin_type != BHW_SIN_CORDIC) return BHW_ALGO_DIRECT;
    const bool fused_ok = bhwk_fold_direct_applicable(c) && bhwp_has_whole_period(p, n0, count);
    if (requested == BHW_ALGO_FUSED) return fused_ok ? BHW_ALGO_FUSED : BHW_ALGO_TABLE;
    if (requested == BHW_ALGO_DIRECT || requested == BHW_ALGO_TABLE) return requested;
    // With dropped phase bits the table has only 2^(W-2) entries and the run-length kernel runs at the store rate: the crossover
    // above was measured at z_shr == 0 only, so windows that kernel takes keep the table strategy.  Where it does not apply
    // (small z_shr: fewer than (K-1) * 16 coefficients per entry; the VHDL sum beyond 28 bits) the table strategy would fall to
    // build + quadrant fold, which the fused kernel beats at these lengths (11.7 against 8.0 us at 2^20, round 2): fused.
    if (fused_ok && p->phi_width <= fused_max_pw(p->n_terms) &&
        (c.z_shr == 0 || p->phi_width < 15 || !bhwk_runlength_applicable(c, w, nullptr)))
        return BHW_ALGO_FUSED;
    const uint64_t chains_direct = count * (p->n_terms - 1);
    return chains_direct >= 2 * bhwp_table_entries(c) ? BHW_ALGO_TABLE : BHW_ALGO_DIRECT;
}

uint32_t bhwp_exec_table_format(const bhw_exec *ex)
{
    return (ex && ex->struct_size >= sizeof(bhw_exec)) ? ex->table_format : (uint32_t)BHW_TABLE_BEST;
}

int bhwp_check_exec(const bhw_exec *ex)
{
    if (!ex) return BHW_OK;
    if (ex->struct_size != sizeof(bhw_exec) && ex->struct_size != 32u)       // 32 = the ABI-1 layout (no table_format)
        return bhwp_fail(BHW_ERR_BADARG, "bhw_exec.struct_size %u", ex->struct_size);
    if (ex->struct_size >= sizeof(bhw_exec) && (ex->table_format > BHW_TABLE_NIBBLE_ESC || ex->reserved != 0))
        return bhwp_fail(BHW_ERR_BADARG, "bhw_exec.table_format %u / reserved %u", ex->table_format, ex->reserved);
    return BHW_OK;
}

// ---- table formats ----------------------------------------------------------------------------------------------------------------

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

// octant mirror (k_table_build_mirror): residual / nibble entries, tables of 2^20 entries and more, and the
// exact quarter turn 2 * lut[0] == E << z_shl the symmetry rests on (true for every model at z_shr == 0; checked, not assumed)
bool bhwk_build_mirror_applies(const BhwCordicCfg &c, uint32_t entries)
{
    const int fmt = fmt_of(c.tab_dlog);
    return (fmt == 2 || fmt == 3 || fmt == 5) && (c.tab_split || fmt != 2) && c.z_shr == 0 && entries >= (1u << 20) && c.n_iter >= 21 &&
           c.dat_width + c.out_shr <= 34 && 2ull * (uint64_t)(uint32_t)c.lut[0] == ((uint64_t)entries << c.z_shl);
}

// Workgroup size of the mirror kernel: 1 024 threads (16 waves, every wave walks 16 groups) where that still gives every CU of
// a 256-CU device its two workgroups (tables of 2^24 entries and more), 256 threads below (profiles/r04_ab_build_wg.txt)
unsigned bhwk_build_mirror_threads(uint32_t entries)
{
    return entries >= (1u << 24) ? 1024u : 256u;
}

BhwTableLayout bhwp_table_layout(uint64_t E, uint32_t tab_dlog)
{
    const int fmt = fmt_of(tab_dlog);
    const uint64_t entry_bytes = fmt == 0 ? 8ull : fmt == 1 ? 4ull : fmt == 2 ? 2ull : 1ull;
    const uint64_t coarse_bytes = fmt == 0 ? 0ull : fmt == 1 ? (E >> kPackLog) * 8ull : (E >> fmt_cell_log(tab_dlog)) * 16ull;
    BhwTableLayout l;
    l.coarse_off = align256(E * entry_bytes);
    l.esc_off = 0;
    l.esc_wg_log = 0;
    l.check_off = l.coarse_off + align256(coarse_bytes);
    if (fmt == 5) {
        // one list per workgroup of the mirror build kernel: it owns 64 x (threads / 4) entries of [0, E/2) and their images
        const uint32_t gpw = bhwk_build_mirror_threads((uint32_t)E) / 4u;
        uint32_t lg = 6;
        while ((1u << (lg - 6)) < gpw) ++lg;
        l.esc_wg_log = lg;
        const uint64_t n_wg = ((E >> 1) + (1ull << lg) - 1ull) >> lg;
        l.esc_off = l.check_off;
        l.check_off = l.esc_off + align256(n_wg * kEscSlots * 16ull);
    }
    l.bytes = fmt == 0 ? E * 8ull : l.check_off + 256ull;          // plain tables carry neither records nor a check word
    return l;
}

// Table formats a tiled whole-period call may use, narrowest first (tab_dlog values: 16 + d nibble, 48 + d nibble + escapes,
// d = 7..9 residual, 6 delta16, 0 plain).  The packed build variants exist from 21 rotations on (always true at PW >= 22).  Residual / nibble tables are built
// by the octant-mirror kernel only, so they are proposed only where it applies.
int bhwp_table_format_candidates(const BhwCordicCfg &c, bool tiled, uint32_t limit, uint32_t out[kMaxFormats])
{
    int n = 0;
    if (tiled && c.n_iter >= 21) {
        uint32_t d = bhwk_resid_dlog(c);
        if (d) {
            BhwCordicCfg probe = c;
            probe.tab_dlog = d;
            probe.tab_split = 1u;
            if (!bhwk_build_mirror_applies(probe, (uint32_t)bhwp_table_entries(c))) d = 0;
        }
        if (d && (limit == BHW_TABLE_BEST || limit == BHW_TABLE_NIBBLE)) out[n++] = kNibbleFlag + d;
        // the same one-byte entries with the rare deviation that does not fit listed exactly (models whose CORDIC noise is wider than
        // the 4-bit fields: cpp, VHDL at 32 bits)
        if (d && (limit == BHW_TABLE_BEST || limit == BHW_TABLE_NIBBLE || limit == BHW_TABLE_NIBBLE_ESC)) out[n++] = kEscFlag + kNibbleFlag + d;
        if (d && (limit == BHW_TABLE_BEST || limit == BHW_TABLE_NIBBLE || limit == BHW_TABLE_NIBBLE_ESC || limit == BHW_TABLE_RESIDUAL)) out[n++] = d;
        if (bhwk_packed_ok(c) && limit != BHW_TABLE_PLAIN) out[n++] = kPackLog;
    }
    out[n++] = 0u;
    return n;
}

int bhwp_fmt_verdict(const bhw_params *p, uint32_t dlog)
{
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    auto it = g_fmt_verdict.find(fmt_key(p, dlog));
    return it == g_fmt_verdict.end() ? kFmtUnknown : it->second;
}

void bhwp_fmt_set_verdict(const bhw_params *p, uint32_t dlog, int v)
{
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    if (v == kFmtOk || v == kFmtBad) g_fmt_verdict[fmt_key(p, dlog)] = v;
    else g_fmt_verdict.erase(fmt_key(p, dlog));                  // anything else: forget it (unknown again)
}

uint64_t bhwp_table_scratch_bytes(const bhw_params *p, const BhwCordicCfg &c, bool tiled, uint32_t limit, bool capturing)
{
    uint32_t cand[kMaxFormats];
    const int n = bhwp_table_format_candidates(c, tiled, limit, cand);
    const uint64_t E = bhwp_table_entries(c);
    uint64_t need = 0;
    for (int i = 0; i < n; ++i) {
        const int v = cand[i] ? bhwp_fmt_verdict(p, cand[i]) : (int)kFmtOk;
        if (v == kFmtBad || (v == kFmtUnknown && capturing)) continue;
        const uint64_t b = bhwp_table_layout(E, cand[i]).bytes;
        if (b > need) need = b;
        if (v == kFmtOk) break;
    }
    return need;
}

// ---- combine pass: tile plan ------------------------------------------------------------------------------------------------------

bool bhwk_tile_applicable(const BhwCordicCfg &c, const BhwWinCfg &w)
{
    // With dropped phase bits (z_shr > 0) consecutive lanes share table entries, so the gathers are dense on their
    // own: such tables take the one-run form of the kernel over the natural layout.
    (void)w;
    // below 2^22 coefficients a grid of 960-thread tiles leaves CUs idle; the one-lane-per-four fold kernel has many more,
    // smaller workgroups and wins there (2^20: 15.0 vs 18.7 us, 2^21: 20.5 vs 21.2, 2^22: 36.0 vs 25.8; BH-7)
    return c.phi_width >= 22 && c.phi_width <= 30;
}

void bhwp_tile_plan(const BhwCordicCfg &c, const BhwWinCfg &w, BhwTilePlan &tp, int &nb, uint32_t &lanes)
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
            // 5 thread groups (kTileThreads = 5 * kTileLanes): group p holds the three inv3-siblings of i5 = p
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
    tp.img_mask = 0xFFu;
    tp.n0mod = 0u;
}

// one-instruction products (tile_harmonic FAST): 15-run tiles, every harmonic weight below 2^(W-3) in magnitude (the built-in
// weights are: a_k <= 0.49 * 2^(W-1 or W-2)); caller-scaled weights beyond that take the 64-bit products.  VHDL rule, one-word
// sums: |sum of the terms| <= sum of (|a_k| + 1) must also stay below 2^31 (the built-in weights: < 2^(W-1))
bool bhwp_tile_fast(const BhwCordicCfg &c, const BhwWinCfg &w, int nb)
{
    bool fast = nb == 15 && c.dat_width >= 3;
    for (uint32_t k = 1; k < w.n_terms && fast; ++k) {
        const int64_t lim = (int64_t)1 << (c.dat_width - 3);
        fast = (int64_t)w.aa[k] < lim && (int64_t)w.aa[k] > -lim;       // (> : the kernel also multiplies by the negated pre-shifted weight)
    }
    if (w.combine != BHW_COMBINE_HLS && fast) {
        int64_t bound = 0;
        for (uint32_t k = 0; k < w.n_terms; ++k) bound += (w.aa[k] < 0 ? -(int64_t)w.aa[k] : (int64_t)w.aa[k]) + 1;
        fast = bound < ((int64_t)1 << 31);
    }
    return fast;
}

// Measured per instance (profiles/r05_kernel_stats_all_legs_tile9_everywhere.csv against round 4's): plain nibbles 60.6 us (HLS rule,
// 61.3 before) and 61.3 us (VHDL rule at 32 bits, 71.7 before); nibble + escapes 69.7 us with the VHDL rule at 32 bits (74.2 before)
// but 68.9 - 70.8 us with the HLS rule, against 64 - 67 in k_table_combine_tile: those stay where they were (-DBHW_T9_ALLFMT5 sends
// them here for the A/B harness).  What those tables cost is not the escape test -- with the marker never looked for, or the very
// code of the plain-nibble instance run over them, the pass stays 6 us slower than over an HLS-model table
// (profiles/r05_ab_tile9_cpp_model_fmt5.txt, r05_ab_tile9_fmt5_code_vs_table.txt): unexplained, recorded as such.
bool bhwk_tile9_applicable(const BhwCordicCfg &c, const BhwWinCfg &w, int nb, bool fast, bool masked)
{
    const int fmt = fmt_of(c.tab_dlog);
    if (!(nb == 15 && fast && !masked && fmt_cell_log(c.tab_dlog) == kTile9CellLog && c.z_shr == 0)) return false;
#ifdef BHW_T9_ALLFMT5
    return fmt == 3 || fmt == 5;                                     // (development: every nibble + escapes instance, for the A/B harness)
#else
    return fmt == 3 || (fmt == 5 && w.combine != BHW_COMBINE_HLS && c.dat_width == 32u);
#endif
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

BhwTableCall bhwp_table_call(const bhw_params *p, const BhwCordicCfg &c, const BhwWinCfg &w, uint64_t n0, uint64_t count, bool apply)
{
    BhwTableCall t;
    t.has_period = bhwp_has_whole_period(p, n0, count);
    t.img_mask = 0xFFu;
    t.n0mod = 0u;
    BhwWinCfg probe = w;
    if (apply && !probe.apply_x) probe.apply_x = reinterpret_cast<const int32_t *>(uintptr_t(1));   // "an input exists"
    t.images = !t.has_period && bhwk_tile_images_applicable(c, probe, n0, count, &t.img_mask, &t.n0mod);
    t.tiled = (t.has_period && bhwk_tile_applicable(c, w)) || t.images;
    return t;
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
    return (((uintptr_t)d_out) & 15u) == 0;                                       // (NULL: the caller asks about the configuration only)
}

// ---- fused kernel -------------------------------------------------------------------------------------------------------------------

bool bhwk_fold_direct_applicable(const BhwCordicCfg &c)
{
    // rot_step's forms: |x| < 2^33 and a quarter circle <= 2^32; ring of at least one wave
    return c.dat_width + c.out_shr <= 34 && c.phi_width >= 9 && c.phi_width <= 30 && c.n_iter >= 2;
}

int bhwp_fold_form(const BhwCordicCfg &c, const BhwWinCfg &w, uint64_t total)
{
    // on 32-bit state with in-wave prefixes when x, y (|.| < 2^(W + out_shr - 1)) and z (the quarter circle) fit signed words -- at
    // every launch size: where the chip is full the form still wins a little over the 64-bit one with its own split level per chain
    // (BH-4 2^22 / 24-bit 16.1 -> 15.2 us, BH-5 17.7 -> 17.4)
    const bool narrow = c.dat_width + c.out_shr <= 30u && c.phi_width - 2u - c.z_shr + c.z_shl <= 30u;
    // short launches, form of the kernel: one chain per wave over the same 64 lanes (k_fold_split), lockstep, or sequential.
    // measured per call (profiles/r02_ab_fused_lockstep.txt): split 7.8 / lockstep 9.0 us at 2^13 lanes (BH-7 2^16), 6.9 / 7.7 at
    // 2^15 (BH-5 2^18), 9.7 / 9.7 at 2^16, 9.0 / 8.2 at 2^17 (BH-4 2^20): split up to 2^15 lanes, lockstep up to 2^18
    // (round 3, tools/bench_short_graph.py: where the narrow form applies it beats the split one at every size for windows of up to
    // five terms -- 4.4 / 4.5 / 4.8 against 5.1 / 5.2 / 5.5 us at 2^14 / 2^16 / 2^18 points of BH-4 -- and loses to it with the nine
    // chains of a 7-term window, 6.8 against 6.6 us at 2^16)
    // (round 5, one chain per wave in the split form -- profiles/r05_short_windows_split*.txt: 7.1 -> 5.0 us for BH-7 2^16 at 32 bits; it
    // now also wins one size up for windows of up to five terms, 2^16 lanes: BH-4 2^19 / 32-bit 7.07 -> 6.23 us, BH-3 5.96 -> 5.00, BH-5
    // 7.35 -> 6.85, while the nine chains of a 7-term window lose there, 9.01 -> 9.50, and everything loses at 2^17 lanes)
    const uint64_t split_max = w.n_terms <= 5 ? (1u << 16) : (1u << 15);
    if (total <= split_max && !(narrow && w.n_terms <= 5)) return BHWP_FOLD_SPLIT;
    if (narrow) return BHWP_FOLD_NARROW;
    // fewer than ~4 waves per SIMD in the whole launch: latency-bound, walk the chains in lockstep
    return total <= (1u << 18) ? BHWP_FOLD_LOCKSTEP : BHWP_FOLD_SEQUENTIAL;
}

uint32_t bhwp_fold_k24(const BhwCordicCfg &c)
{
    // |x|, |y| < 2^B, B = W + out_shr - 1: (x >> k) fits 24 signed bits from k = B - 23 on; twice the ROM word (the kernel carries the
    // angle doubled: rot_mad24) from the first lut[k] < 2^22 on
    const int B = (int)(c.dat_width + c.out_shr) - 1;
    uint32_t k24 = B > 23 ? (uint32_t)(B - 23) : 1u;
    while (k24 < c.n_iter && k24 < 32u && (uint32_t)c.lut[k24] >= (1u << 22)) ++k24;
    return k24;
}

// ---- ownership parts ----------------------------------------------------------------------------------------------------------------

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
    bhwp_tile_plan(c, w, tp, nb, lanes);
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

int bhwp_part_checks(const bhw_params *p, uint32_t part, uint32_t n_parts)
{
    int rc = bhwp_validate(p);
    if (rc) return rc;
    if (p->sin_type != BHW_SIN_CORDIC) return bhwp_fail(BHW_ERR_UNSUPPORTED, "interleaved parts exist for the CORDIC source only");
    if (n_parts < 1 || n_parts > 64 || part >= n_parts) return bhwp_fail(BHW_ERR_BADARG, "part %u of %u (1..64 parts)", part, n_parts);
    if (p->phi_width < 9) return bhwp_fail(BHW_ERR_UNSUPPORTED, "interleaved parts need phi_width >= 9 (a ring of 64 lanes)");
    // a part is produced by the fused kernel (CORDIC state within 34 bits) or by the tile kernel over the full table (N >= 2^22):
    // configurations with neither (e.g. VHDL model, W = 32, PRECISION >= 3 below 2^22) have no part kernel, and the segment
    // arithmetic must not promise what bhw_generate_part_device cannot deliver
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    BhwWinCfg w;
    bhwp_resolve_window(p, w);
    if (!bhwk_fold_direct_applicable(c) && !bhwk_tile_applicable(c, w))
        return bhwp_fail(BHW_ERR_UNSUPPORTED, "no kernel produces ownership parts of this configuration (CORDIC state beyond 34 bits and no tile plan)");
    return BHW_OK;
}

// Strategy of one ownership part.  Fused: chains = lanes x (chains per lane), no table.  Table: the full first-quadrant table (it
// does not shrink with the part) + this part's tiles.  Measured per part (BH-7 2^26 / 32-bit, profiles/r04_small_windows_and_parts.json):
// table 0.068 / 0.055 / 0.049 ms at 2 / 4 / 8 parts, fused 0.092 / 0.051 ms at 4 / 8 -- the build pass got 20 % faster this round and the
// crossover moved out: the fused kernel is taken once the part's own chains (9/8 per owned coefficient) are at most HALF the
// table's (more than 8 parts of this window).  (Round 3's file showed 0.1195 ms for AUTO at one part against 0.1123 for the same
// table plan: the first timing after a run of short kernels -- clocks, not the plan; the tool now ramps again before that section.)
bool bhwp_part_fused(const bhw_params *p, const BhwCordicCfg &c, const BhwFoldRun *runs, int n_runs, uint32_t tile_count, uint32_t requested, int *rc)
{
    *rc = BHW_OK;
    uint64_t lanes = 0;
    for (int i = 0; i < n_runs; ++i) lanes += runs[i].r_end - runs[i].r0;
    static const int kChains[8] = {0, 0, 2, 3, 5, 6, 0, 9};
    const uint64_t chains_fused = lanes * (uint64_t)kChains[p->n_terms];
    const bool fused_ok = bhwk_fold_direct_applicable(c);
    const bool table_ok = tile_count != 0;                     // tile-aligned ownership: the tile kernel can produce exactly this part
    bool fused;
    if (requested == BHW_ALGO_FUSED) fused = fused_ok;
    else if (requested == BHW_ALGO_TABLE) fused = !table_ok;
    else fused = fused_ok && (!table_ok || 2 * chains_fused <= bhwp_table_entries(c));
    if (fused && !fused_ok) *rc = bhwp_fail(BHW_ERR_UNSUPPORTED, "no kernel produces this part (CORDIC state beyond 34 bits and no tile plan)");
    if (!fused && !table_ok) *rc = bhwp_fail(BHW_ERR_UNSUPPORTED, "the table strategy produces whole tiles only and this window has no tile plan");
    return fused;
}

// Kernel names of the table strategy's two passes for a resolved configuration (bhw_describe_plan: profilers, bench labels).
// Mirrors the dispatch in bhwk_table_build / bhwk_table_combine_tile_range / bhwk_table_combine_fold.
void bhwk_describe_table(const BhwCordicCfg &c_in, const BhwWinCfg &w, bool tiled, bool images, char *build, char *combine, size_t len)
{
    const BhwCordicCfg c = table_layout(c_in);
    const uint32_t entries = 1u << (c.phi_width - 2 - c.z_shr);
    const bool fits = (c.dat_width + c.out_shr <= 34);
    const int fmt = fmt_of(c.tab_dlog);
    if (fits && c.n_iter >= 7 && entries < (1u << 20) && c.tab_dlog == 0 && !c.tab_split) snprintf(build, len, "k_table_build_plain<%u>", c.n_iter);
    else if (entries >= 64 && fits && c.n_iter >= 2) {
        if (bhwk_build_mirror_applies(c, entries)) snprintf(build, len, "k_table_build_mirror<%u,%d,%u>", c.n_iter, fmt, bhwk_build_mirror_threads(entries));
        else snprintf(build, len, "k_table_build_shared<%u,%d>", c.n_iter, fmt);
    } else snprintf(build, len, "k_table_build<%s>", c.wide ? "int64_t" : "int32_t");
    const int mode = mode_of(c, w);
    if (tiled) {
        const int nb3 = (c.z_shr == 0 && w.n_terms > 3) ? 3 : 1, nb5 = (c.z_shr == 0 && w.n_terms > 5) ? 5 : 1;
        // (`images`: a subset of the eight images, which k_table_combine_tile's MASKED instances produce)
        if (bhwk_tile9_applicable(c, w, nb3 * nb5, bhwp_tile_fast(c, w, nb3 * nb5), images)) snprintf(combine, len, "k_tile9<%d,%d>", mode, fmt);
        else snprintf(combine, len, "k_table_combine_tile<%d,%d,%d>", nb3 * nb5, mode, fmt);
    } else if (c.tab_dlog == 0 && !c.tab_split) snprintf(combine, len, "k_table_combine_fold_t<%u,%d>", w.n_terms, mode);
    else snprintf(combine, len, "k_table_combine_fold");
}

// ---- the pure entry points of the C ABI (include/bhw.h) -------------------------------------------------------------------------------
extern "C" {

uint32_t bhw_abi_version(void) { return BHW_ABI_VERSION; }

const char *bhw_strerror(int code)
{
    switch (code) {
    case BHW_OK: return "ok";
    case BHW_ERR_BADARG: return "bad argument";
    case BHW_ERR_UNSUPPORTED: return "unsupported parameter combination";
    case BHW_ERR_HIP: return "HIP runtime error or no device";
    case BHW_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
    }
}

const char *bhw_last_error(void) { return g_last_error.c_str(); }

int bhw_coeffs_from_float(uint32_t win_type, uint32_t dat_width, const double *a, int32_t aa[7])
{
    const int K = bhwp_terms_of(win_type);
    if (!K) return bhwp_fail(BHW_ERR_BADARG, "win_type %u", win_type);
    if (dat_width < 8 || dat_width > 32) return bhwp_fail(BHW_ERR_BADARG, "dat_width %u outside 8..32", dat_width);
    if (!aa) return bhwp_fail(BHW_ERR_BADARG, "aa is NULL");
    if (!a) {
        switch (win_type) {
        case BHW_WIN_HAMMING: a = kHamming; break;
        case BHW_WIN_HANN: a = kHann; break;
        case BHW_WIN_BH3: a = kBh3; break;
        case BHW_WIN_BH4: a = kBh4; break;
        case BHW_WIN_BH5: a = kBh5; break;
        default: a = kBh7; break;
        }
    }
    // s = 1: win_function.cpp:176-177,210-212,258-261; s = 2: :312-316,349-355.  C round(): half away from zero.
    const unsigned s = (K >= 5) ? 2 : 1;
    const double scale = std::pow(2.0, (double)(dat_width - s)) - 1.0;
    for (int k = 0; k < 7; ++k) aa[k] = 0;
    for (int k = 0; k < K; ++k) {
        const double v = std::round(a[k] * scale);
        if (!(v >= -2147483648.0 && v <= 2147483647.0)) return bhwp_fail(BHW_ERR_BADARG, "weight %d (%g) does not fit int32 at dat_width %u", k, a[k], dat_width);
        aa[k] = (int32_t)(int64_t)v;
    }
    return BHW_OK;
}

int bhw_coeffs_preset(uint32_t preset, uint32_t dat_width, uint32_t *win_type, double a[7], int32_t aa[7])
{
    // hls/windows/win_function.cpp:241-250 (Nuttall, Blackman-Nuttall), :292-303 (flat-top 1 / 2), README.md:30-51
    static const struct { uint32_t win; double a[7]; } kPresets[] = {
        {0, {0}},
        {BHW_WIN_BH4, {0.355768, 0.487396, 0.144232, 0.012604}},
        {BHW_WIN_BH4, {0.3635819, 0.4891775, 0.1365995, 0.0106411}},
        {BHW_WIN_BH5, {0.25, 0.4925, 0.3225, 0.097, 0.0075}},
        {BHW_WIN_BH5, {0.215578950, 0.416631580, 0.277263158, 0.083578947, 0.006947368}},
        {BHW_WIN_BH7, {0.27105140069342, 0.43329793923448, 0.21812299954311, 0.06592544638803, 0.01081174209837,
                       0.00077658482522, 0.00001388721735}},
        {BHW_WIN_BH3, {0.42, 0.5, 0.08}},
        {BHW_WIN_BH3, {0.42323, 0.49755, 0.07922}},
    };
    if (preset < 1 || preset >= sizeof kPresets / sizeof kPresets[0]) return bhwp_fail(BHW_ERR_BADARG, "preset %u", preset);
    if (win_type) *win_type = kPresets[preset].win;
    if (a) memcpy(a, kPresets[preset].a, 7 * sizeof(double));
    if (aa) return bhw_coeffs_from_float(kPresets[preset].win, dat_width, kPresets[preset].a, aa);
    return BHW_OK;
}

int bhw_params_init(bhw_params *p, uint32_t win_type, uint32_t phi_width, uint32_t dat_width)
{
    if (!p) return bhwp_fail(BHW_ERR_BADARG, "params is NULL");
    memset(p, 0, sizeof *p);
    p->struct_size = sizeof *p;
    p->model = BHW_MODEL_HLS;
    p->combine = BHW_COMBINE_HLS;
    p->sin_type = BHW_SIN_CORDIC;
    p->win_type = win_type;
    p->n_terms = (uint32_t)bhwp_terms_of(win_type);
    p->phi_width = phi_width;
    p->dat_width = dat_width;
    p->precision = 1;
    p->lut_size = 9;
    if (!p->n_terms) return bhwp_fail(BHW_ERR_BADARG, "win_type %u", win_type);
    int rc = bhw_coeffs_from_float(win_type, dat_width, nullptr, p->aa);
    if (rc) return rc;
    return bhwp_validate(p);
}

int bhw_params_validate(const bhw_params *p) { return bhwp_validate(p); }

int bhw_constant_tables(uint32_t which, int64_t table[48], int64_t gains[2])
{
    if (which > 1) return bhwp_fail(BHW_ERR_BADARG, "which %u", which);
    if (table) memcpy(table, which ? kAtanT4 : kAtanT2, 48 * sizeof(int64_t));
    if (gains) { gains[0] = kGain46; gains[1] = kGain47; }
    return BHW_OK;
}

// Upper bound over every table format the call may use (8 bytes per table entry: the plain format).  bhw_workspace_bytes_ex
// gives the figure for the format the call would use right now.
uint64_t bhw_workspace_bytes(const bhw_params *p, uint64_t n0, uint64_t count, uint32_t algo)
{
    if (bhwp_validate(p)) return 0;
    if (p->sin_type != BHW_SIN_CORDIC) return 0;
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    BhwWinCfg w;
    bhwp_resolve_window(p, w);
    return bhwp_pick_algo(p, c, w, n0, count, algo) == BHW_ALGO_TABLE ? bhwp_table_entries(c) * 8ull : 0;
}

uint64_t bhw_workspace_bytes_ex(const bhw_params *p, uint64_t n0, uint64_t count, const bhw_exec *ex)
{
    if (bhwp_validate(p) || bhwp_check_exec(ex)) return 0;
    if (p->sin_type != BHW_SIN_CORDIC) return 0;
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    BhwWinCfg w;
    bhwp_resolve_window(p, w);
    if (bhwp_pick_algo(p, c, w, n0, count, ex ? ex->algo : (uint32_t)BHW_ALGO_AUTO) != BHW_ALGO_TABLE) return 0;
    const BhwTableCall t = bhwp_table_call(p, c, w, n0, count, false);
    return bhwp_table_scratch_bytes(p, c, t.tiled, bhwp_exec_table_format(ex), false);
}

int bhw_describe_plan(const bhw_params *p, uint64_t n0, uint64_t count, const bhw_exec *ex, char *buf, uint64_t len)
{
    int rc = bhwp_validate(p);
    if (rc) return rc;
    rc = bhwp_check_exec(ex);
    if (rc) return rc;
    if (!buf || !len) return bhwp_fail(BHW_ERR_BADARG, "buf is NULL or empty");
    const bool period = bhwp_has_whole_period(p, n0, count);
    if (p->sin_type != BHW_SIN_CORDIC) {
        snprintf(buf, len, "taylor: %s", period && p->phi_width >= 5 ? "k_taylor_window_fold (+ k_taylor_window on ragged ends)" : "k_taylor_window");
        return BHW_OK;
    }
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    BhwWinCfg w;
    bhwp_resolve_window(p, w);
    const uint32_t algo = bhwp_pick_algo(p, c, w, n0, count, ex ? ex->algo : (uint32_t)BHW_ALGO_AUTO);
    if (algo == BHW_ALGO_DIRECT) {
        snprintf(buf, len, "direct: %s", (c.dat_width + c.out_shr <= 34 && c.n_iter >= 7) ? "k_direct_fast" : "k_direct");
        return BHW_OK;
    }
    if (algo == BHW_ALGO_FUSED) {
        // one launch over the whole ring [0, N/8) per period: the form bhwk_fold_direct picks for that many lanes
        const int form = bhwp_fold_form(c, w, 1ull << (p->phi_width - 3));
        if (form == BHWP_FOLD_SPLIT) snprintf(buf, len, "fused: k_fold_split<%u,%d,%u> (+ k_direct_fast on ragged ends)", p->n_terms, mode_of(c, w), p->n_terms == 2 ? 2u : p->n_terms == 3 ? 3u : p->n_terms == 4 ? 5u : p->n_terms == 5 ? 6u : 9u);
        else snprintf(buf, len, "fused: k_fold_direct<%u,%d,%d> (+ k_direct_fast on ragged ends)", p->n_terms, mode_of(c, w), form);
        return BHW_OK;
    }
    const BhwTableCall t = bhwp_table_call(p, c, w, n0, count, false);
    c.tab_split = (t.tiled && c.z_shr == 0) ? 1u : 0u;
    uint32_t cand[kMaxFormats];
    const int n_cand = bhwp_table_format_candidates(c, t.tiled, bhwp_exec_table_format(ex), cand);
    const char *state = "";
    c.tab_dlog = 0;
    for (int i = 0; i < n_cand; ++i) {
        const int v = cand[i] ? bhwp_fmt_verdict(p, cand[i]) : (int)kFmtOk;
        if (v == kFmtBad) continue;
        c.tab_dlog = cand[i];
        if (v == kFmtUnknown) state = ", unverified";
        break;
    }
    char build[64], combine[96];
    bhwk_describe_table(c, w, t.tiled, t.images, build, combine, sizeof build);
    if (period && c.tab_dlog == 0 && bhwk_runlength_applicable(c, w, nullptr))     // generate_impl's period(): dropped phase bits
        snprintf(combine, sizeof combine, "k_runlength_window<%u,%d,%s> (16-byte aligned output; else k_table_combine_fold_t)", p->n_terms,
                 mode_of(c, w), c.dat_width <= 16 ? "true" : "false");
    const char *fmt = c.tab_dlog == 0 ? "plain" : c.tab_dlog == 6 ? "delta16" : c.tab_dlog >= kEscFlag ? "nibble+esc" : c.tab_dlog >= 16 ? "nibble" : "residual";
    snprintf(buf, len, "table[%s%s]: %s + %s%s", fmt, state, build, (period || t.images) ? combine : "k_table_combine",
             t.images ? " (image subset)" : period && count != (1ull << p->phi_width) ? " (+ k_table_combine / k_replicate on the rest)" : "");
    return BHW_OK;
}

int bhw_part_segments(const bhw_params *p, uint32_t part, uint32_t n_parts, bhw_segment *segs, uint32_t capacity, uint32_t *n_segs)
{
    int rc = bhwp_part_checks(p, part, n_parts);
    if (rc) return rc;
    if (!n_segs) return bhwp_fail(BHW_ERR_BADARG, "n_segs is NULL");
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    BhwWinCfg w;
    bhwp_resolve_window(p, w);
    BhwFoldRun runs[32];
    uint32_t t0, tc;
    const int n_runs = bhwk_part_runs(c, w, part, n_parts, runs, &t0, &tc);
    std::vector<bhw_segment> all;
    const uint64_t H = 1ull << (p->phi_width - 3);
    for (int i = 0; i < n_runs; ++i)
        for (uint64_t img = 0; img < 8; ++img)
            all.push_back(bhw_segment{runs[i].r0 + img * H, (uint64_t)(runs[i].r_end - runs[i].r0)});
    // sorted, touching or overlapping segments merged
    for (size_t i = 1; i < all.size(); ++i)
        for (size_t j = i; j > 0 && all[j - 1].n0 > all[j].n0; --j) std::swap(all[j - 1], all[j]);
    std::vector<bhw_segment> merged;
    for (const bhw_segment &sg : all) {
        if (!merged.empty() && sg.n0 <= merged.back().n0 + merged.back().count) {
            const uint64_t end = sg.n0 + sg.count;
            if (end > merged.back().n0 + merged.back().count) merged.back().count = end - merged.back().n0;
        } else merged.push_back(sg);
    }
    *n_segs = (uint32_t)merged.size();
    if (segs) {
        if (capacity < merged.size()) return bhwp_fail(BHW_ERR_BADARG, "capacity %u < %zu segments", capacity, merged.size());
        for (size_t i = 0; i < merged.size(); ++i) segs[i] = merged[i];
    }
    return BHW_OK;
}

// Verdict cache of the packed formats (0 unknown, 1 exact, 2 overflows); set != 0 overrides it (tests of the fallback).
int bhw_dbg_table_format_verdict(const bhw_params *p, uint32_t dlog, int set)
{
    if (bhwp_validate(p)) return BHW_ERR_BADARG;
    if (set) bhwp_fmt_set_verdict(p, dlog, set);
    return bhwp_fmt_verdict(p, dlog);
}

// tab_dlog the residual format would use for `p` (0: not applicable) and whether delta16 applies
int bhw_dbg_table_format_info(const bhw_params *p, uint32_t *resid_dlog, uint32_t *delta16_ok)
{
    if (bhwp_validate(p)) return BHW_ERR_BADARG;
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    if (resid_dlog) *resid_dlog = c.n_iter >= 21 ? bhwk_resid_dlog(c) : 0u;
    if (delta16_ok) *delta16_ok = (c.n_iter >= 21 && bhwk_packed_ok(c)) ? 1u : 0u;
    return BHW_OK;
}

} // extern "C"
