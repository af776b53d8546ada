// bhw_api.cpp -- host side of the C ABI (include/bhw.h): parameter validation, resolution of a
// (model, widths) tuple into kernel constants, strategy choice, per-device scratch, launches.
//
// Host mirror of the reference's own host code: cpp/cordic_sincos.cpp:12-36 derives the rescaled
// ROM, gain and z scaling per call; hls/windows/win_function.cpp:74-96 does the same for the HLS
// model; src/cordic_dds.vhd:97-131,159-166 at elaboration.  Here that derivation runs once per call
// on the host and is handed to the kernels as a kernel-argument struct.  No per-sample arithmetic
// happens on the host: every compute entry point fails with BHW_ERR_HIP when no device is usable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "bhw_internal.h"
#include "bhw_tables.inc"

extern "C" void bhw_taylor_rom(uint32_t dat_width, uint32_t lut_size, int32_t *rom_sin_cos);
extern "C" uint32_t bhw_taylor_pi_word(int e);

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int fail_hip(int hip_code, const char *what)
{
    return fail(BHW_ERR_HIP, "%s: %s (hipError %d)", what, hipGetErrorString((hipError_t)hip_code), hip_code);
}

int terms_of(uint32_t win_type)
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

int validate(const bhw_params *p, bool sincos_only = false)
{
    if (!p) return fail(BHW_ERR_BADARG, "params is NULL");
    if (p->struct_size != sizeof(bhw_params))
        return fail(BHW_ERR_BADARG, "struct_size %u != %zu", p->struct_size, sizeof(bhw_params));
    if (p->model > BHW_MODEL_SCALED) return fail(BHW_ERR_BADARG, "model %u", p->model);
    if (p->model > BHW_MODEL_VHDL && !sincos_only)
        return fail(BHW_ERR_UNSUPPORTED, "cordic_dds48 / cordic_dds_scaled feed no window entity: bhw_sincos_* only");
    if (p->combine > BHW_COMBINE_VHDL) return fail(BHW_ERR_BADARG, "combine %u", p->combine);
    if (p->sin_type > BHW_SIN_TAYLOR_ALL) return fail(BHW_ERR_BADARG, "sin_type %u", p->sin_type);
    const uint32_t K = p->n_terms;
    if (!(K == 2 || K == 3 || K == 4 || K == 5 || K == 7)) return fail(BHW_ERR_BADARG, "n_terms %u (2,3,4,5,7)", K);
    const uint32_t PW = p->phi_width, W = p->dat_width;
    if (PW < 4 || PW > 30) return fail(BHW_ERR_BADARG, "phi_width %u outside 4..30", PW);
    if (W < 8 || W > 32) return fail(BHW_ERR_BADARG, "dat_width %u outside 8..32", W);
    if (p->sin_type != BHW_SIN_CORDIC) {
        // win_selector wires the Taylor source only to HAMMING and BH3TERM: src/win_selector.vhd:93-135
        if (K > 3 && p->sin_type == BHW_SIN_TAYLOR)
            return fail(BHW_ERR_UNSUPPORTED, "Taylor source exists only for 2- and 3-term windows (BHW_SIN_TAYLOR_ALL is the extension)");
        const uint32_t L = p->lut_size;
        if (L < 1 || L > 16) return fail(BHW_ERR_BADARG, "lut_size %u outside 1..16", L);
        // generators in use: PHASE_WIDTH - v, v = 0 .. vmax  (bh_win_3term.vhd:221-226; k = 4 needs v = 2)
        const uint32_t vmax = K > 4 ? 2u : K > 2 ? 1u : 0u;
        if (PW < 3 + vmax) return fail(BHW_ERR_UNSUPPORTED, "phi_width %u too short for the PHASE_WIDTH-%u generator", PW, vmax);
        const uint32_t pw_min = PW - vmax;
        for (uint32_t pw = pw_min; pw <= PW; ++pw) {
            const int d = (int)pw - (int)L;
            if (d > 2) {
                if (d - 3 > 15) return fail(BHW_ERR_UNSUPPORTED, "Taylor STAGE %d > 15 (tay1_order cnt_exp is 16 bits)", d - 3);
                if (W < 19 && 19 + L + W > 48) return fail(BHW_ERR_UNSUPPORTED, "Taylor narrow path: 19+L+W > 48 DSP bits");
                if (W > 18 && 19 + L + W > 62) return fail(BHW_ERR_UNSUPPORTED, "Taylor wide path: 19+L+W > 62 product bits");
            }
        }
        return BHW_OK;
    }
    if (p->model == BHW_MODEL_HLS && PW > W + 2)
        return fail(BHW_ERR_UNSUPPORTED, "HLS model is ill-defined for phi_width > dat_width + 2 (init_t truncation)");
    if (p->model == BHW_MODEL_VHDL && (p->precision < 1 || p->precision > 7))
        return fail(BHW_ERR_BADARG, "precision %u outside 1..7", p->precision);
    return BHW_OK;
}

// Resolve the CORDIC constants.
void resolve_cordic(const bhw_params *p, BhwCordicCfg &c)
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

void resolve_window(const bhw_params *p, BhwWinCfg &w)
{
    memset(&w, 0, sizeof w);
    for (int k = 0; k < 7; ++k) w.aa[k] = p->aa[k];
    w.n_terms = p->n_terms;
    w.combine = p->combine;
}

// The calling thread's current device is switched for the duration of an entry point and put back afterwards.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) {
            err = hipSetDevice(device);
            switched = (err == hipSuccess);
        }
    }
    ~DeviceGuard()
    {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

bool stream_is_capturing(void *stream)
{
    if (!stream) return false;                               // the legacy default stream cannot be captured
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &st) != hipSuccess) return false;
    return st != hipStreamCaptureStatusNone;
}

// ---- library-owned scratch, one buffer per (device, stream) ------------------------------------
// The table is rebuilt by every call, so two calls that share a buffer must not interleave their launches
// (A.build, B.build, A.combine would combine A from B's table).  Calls on different streams get different
// buffers; calls on one (device, stream) -- from any number of host threads, the *_to_host helpers on the
// NULL stream included -- hold the slot's mutex from before the build launch until after the last launch,
// and the stream then orders the kernels.  Growing a buffer happens under the same mutex.
// Callers that want no allocation in the launch path (graph capture) pass their own workspace through
// bhw_exec, or call bhw_prepare_device first.
struct Slot {
    std::mutex mu;
    void *buf = nullptr;
    uint64_t bytes = 0;
};
struct DeviceScratch {
    std::map<void *, std::shared_ptr<Slot>> bufs;            // stream -> slot
    std::map<std::pair<uint32_t, uint32_t>, int32_t *> roms;  // Taylor ROM cache keyed by (W, L)
};
std::mutex g_mu;                                              // guards g_scratch's maps (never held across a launch)
std::map<int, DeviceScratch> g_scratch;

std::shared_ptr<Slot> slot_of(int device, void *stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto &sp = g_scratch[device].bufs[stream];
    if (!sp) sp = std::make_shared<Slot>();
    return sp;
}

// slot.mu is held by the caller; the current device is `device`
int ensure_slot_bytes(Slot &slot, void *stream, uint64_t bytes)
{
    if (slot.bytes >= bytes) return BHW_OK;
    if (stream_is_capturing(stream))
        return fail(BHW_ERR_HIP, "library scratch of this stream must grow to %llu bytes during stream capture: call "
                    "bhw_prepare_device first or pass bhw_exec.workspace", (unsigned long long)bytes);
    if (slot.buf) {
        (void)hipStreamSynchronize((hipStream_t)stream);      // earlier launches on this stream may still read it
        (void)hipFree(slot.buf);
        slot.buf = nullptr;
        slot.bytes = 0;
    }
    void *b = nullptr;
    const hipError_t e = hipMalloc(&b, bytes);
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(scratch)");
    slot.buf = b;
    slot.bytes = bytes;
    return BHW_OK;
}

// ---- packed table formats: verified once per configuration ------------------------------------------
// Whether every (c, s) difference fits its int8 / int16 field is a property of (model, PW, W, PRECISION, format) alone
// -- the table does not depend on the weights or on the call's range.  The first table build of a configuration in a
// packed format runs with the kernels' overflow check on and is read back once; the verdict is cached for the process.
// A configuration that fails falls back to the next wider format, so an overflow can never reach the coefficients.
enum { kFmtUnknown = 0, kFmtOk = 1, kFmtBad = 2 };
std::mutex g_fmt_mu;
std::map<uint64_t, int> g_fmt_verdict;

uint64_t fmt_key(const bhw_params *p, uint32_t dlog)
{
    return ((uint64_t)p->model << 40) | ((uint64_t)p->phi_width << 32) | ((uint64_t)p->dat_width << 24) |
           ((uint64_t)(p->model == BHW_MODEL_VHDL ? p->precision : 0u) << 16) | dlog;
}
int fmt_verdict(const bhw_params *p, uint32_t dlog)
{
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    auto it = g_fmt_verdict.find(fmt_key(p, dlog));
    return it == g_fmt_verdict.end() ? kFmtUnknown : it->second;
}
void fmt_set_verdict(const bhw_params *p, uint32_t dlog, int v)
{
    std::lock_guard<std::mutex> lk(g_fmt_mu);
    g_fmt_verdict[fmt_key(p, dlog)] = v;
}

int get_taylor_rom(int device, void *stream, uint32_t W, uint32_t L, const int32_t **rom)
{
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceScratch &s = g_scratch[device];
    auto key = std::make_pair(W, L);
    auto it = s.roms.find(key);
    if (it != s.roms.end()) { *rom = it->second; return BHW_OK; }
    if (stream_is_capturing(stream))
        return fail(BHW_ERR_HIP, "the Taylor ROM for dat_width %u / lut_size %u is not on the device yet and the stream is "
                    "capturing: call bhw_prepare_device first", W, L);
    std::vector<int32_t> host(2u << L);
    bhw_taylor_rom(W, L, host.data());
    int32_t *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, host.size() * sizeof(int32_t));
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(rom)");
    e = hipMemcpy(d, host.data(), host.size() * sizeof(int32_t), hipMemcpyHostToDevice);   // synchronous: once per (device, W, L)
    if (e != hipSuccess) { (void)hipFree(d); return fail_hip(e, "hipMemcpy(rom)"); }
    s.roms[key] = d;
    *rom = d;
    return BHW_OK;
}

int resolve_taylor(const bhw_params *p, int device, void *stream, BhwTaylorCfg &t)
{
    memset(&t, 0, sizeof t);
    t.phi_width = p->phi_width;
    t.dat_width = p->dat_width;
    t.lut_size = p->lut_size;
    const int d = (int)p->phi_width - (int)p->lut_size;
    t.mode = d < 2 ? 0u : d == 2 ? 1u : 2u;
    t.xshift = 19 + p->lut_size;
    t.pi_word = d > 2 ? bhw_taylor_pi_word(17 - (d - 3)) : 0;          // tay1_order.vhd:133, STAGE = PW-L-3
    t.pad[0] = (d - 1) > 2 ? bhw_taylor_pi_word(17 - (d - 4)) : 0;      // generator at PHASE_WIDTH-1 (harmonics 2, 6)
    t.pad[1] = (d - 2) > 2 ? bhw_taylor_pi_word(17 - (d - 5)) : 0;      // generator at PHASE_WIDTH-2 (harmonic 4)
    return get_taylor_rom(device, stream, p->dat_width, p->lut_size, &t.rom);
}

bool device_ok(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return false;
    return device >= 0 && device < n;
}

uint64_t table_entries(const BhwCordicCfg &c) { return 1ull << (c.phi_width - 2 - c.z_shr); }

bool has_whole_period(const bhw_params *p, uint64_t n0, uint64_t count)
{
    const uint64_t N = 1ull << p->phi_width;
    return count >= (N - n0 % N) % N + N;
}

// Whole periods up to this length go through the fused kernel under AUTO: one launch of 5/8 .. 9/8 chains per coefficient beats
// two dependent launches around a table of 1/4 chain per coefficient while the call is launch- and latency-bound.  Measured per
// call (profiles/r02_small_windows.json): BH-4/24-bit fused 8.0 / 12.1 / 17.0 us at 2^20 / 2^21 / 2^22 against 11.7 / 14.6 /
// 19.9 us for the table strategy; BH-7/32-bit 9.0 (2^16) / 13.3 / 21.8 / 33.5 us against 11.9 / 11.9 / 16.8 / 27.5 us.
uint32_t fused_max_pw(uint32_t n_terms) { return n_terms <= 5 ? 22u : 19u; }

// AUTO: the fused kernel for short whole periods; else build the shared table when it replaces clearly more CORDIC chains
// than it costs; else one chain per harmonic per coefficient.
uint32_t pick_algo(const bhw_params *p, const BhwCordicCfg &c, uint64_t n0, uint64_t count, uint32_t requested)
{
    if (p->sin_type != BHW_SIN_CORDIC) return BHW_ALGO_DIRECT;
    const bool fused_ok = bhwk_fold_direct_applicable(c) && has_whole_period(p, n0, count);
    if (requested == BHW_ALGO_FUSED) return fused_ok ? BHW_ALGO_FUSED : BHW_ALGO_TABLE;
    if (requested == BHW_ALGO_DIRECT || requested == BHW_ALGO_TABLE) return requested;
    // (with dropped phase bits the table has only 2^(W-2) entries and the run-length kernel runs at the store rate: the crossover
    // above was measured at z_shr == 0 only, so such windows keep the table strategy once they are long enough for that kernel)
    if (fused_ok && p->phi_width <= fused_max_pw(p->n_terms) && (c.z_shr == 0 || p->phi_width < 15)) return BHW_ALGO_FUSED;
    const uint64_t chains_direct = count * (p->n_terms - 1);
    return chains_direct >= 2 * table_entries(c) ? BHW_ALGO_TABLE : BHW_ALGO_DIRECT;
}

// Runs [n0, n0+count) as  head | whole periods | tail.  `ragged(off, len)` handles an arbitrary sub-range,
// `period(off)` one whole period starting at a multiple of N.  Without the fused apply only the first period is
// computed and the others are store-only replicas; with it every period has its own x.
template <typename Ragged, typename Period>
int run_split(const BhwLaunch &l, uint64_t n0, uint64_t count, uint64_t N, bool per_period_input, int32_t *d_out,
              Ragged ragged, Period period)
{
    const uint64_t head_len = (N - n0 % N) % N;
    int e;
    if (count < head_len + N) {
        e = ragged(0, count);
        return e ? fail_hip(e, "launch") : BHW_OK;
    }
    const uint64_t head = head_len, periods = (count - head) / N, tail = count - head - periods * N;
    e = head ? ragged(0, head) : 0;
    if (e) return fail_hip(e, "head launch");
    const uint64_t computed = per_period_input ? periods : 1;
    for (uint64_t f = 0; f < computed; ++f) {
        e = period(head + f * N);
        if (e) return fail_hip(e, "whole-period launch");
    }
    if (!per_period_input && periods > 1) {
        e = bhwk_replicate(l, d_out + head, N, (uint32_t)(periods - 1), d_out + head + N);
        if (e) return fail_hip(e, "replicate launch");
    }
    e = tail ? ragged(head + periods * N, tail) : 0;
    return e ? fail_hip(e, "tail launch") : BHW_OK;
}

uint32_t exec_table_format(const bhw_exec *ex)
{
    return (ex && ex->struct_size >= sizeof(bhw_exec)) ? ex->table_format : (uint32_t)BHW_TABLE_BEST;
}

int check_exec(const bhw_exec *ex)
{
    if (!ex) return BHW_OK;
    if (ex->struct_size != sizeof(bhw_exec) && ex->struct_size != 32u)       // 32 = the ABI-1 layout (no table_format)
        return fail(BHW_ERR_BADARG, "bhw_exec.struct_size %u", ex->struct_size);
    if (ex->struct_size >= sizeof(bhw_exec) && (ex->table_format > BHW_TABLE_NIBBLE || ex->reserved != 0))
        return fail(BHW_ERR_BADARG, "bhw_exec.table_format %u / reserved %u", ex->table_format, ex->reserved);
    return BHW_OK;
}

// Table formats a tiled whole-period call may use, narrowest first (tab_dlog values: 16 + d nibble, d = 7..9 residual, 6 delta16,
// 0 plain).  The packed build variants exist from 21 rotations on (always true at PW >= 22).
constexpr int kMaxFormats = 4;
int table_format_candidates(const BhwCordicCfg &c, bool tiled, uint32_t limit, uint32_t out[kMaxFormats])
{
    int n = 0;
    if (tiled && c.n_iter >= 21) {
        const uint32_t d = bhwk_resid_dlog(c);
        if (d && (limit == BHW_TABLE_BEST || limit == BHW_TABLE_NIBBLE)) out[n++] = 16u + d;
        if (d && (limit == BHW_TABLE_BEST || limit == BHW_TABLE_NIBBLE || limit == BHW_TABLE_RESIDUAL)) out[n++] = d;
        if (bhwk_packed_ok(c) && limit != BHW_TABLE_PLAIN) out[n++] = 6u;
    }
    out[n++] = 0u;
    return n;
}

// Build the table in the narrowest format that is exact for this configuration.  A packed format whose exactness has not
// been established yet for the configuration is built with the kernels' overflow check on and read back here (once per
// process and configuration; during stream capture the plain format is used instead).  `ws` holds E*8 bytes:
// [ entries | ... | records / block heads at byte offset E*4 | ... | check word in the last 8 bytes ].
int build_table(const bhw_params *p, const BhwLaunch &l, BhwCordicCfg &c, bool tiled, uint32_t limit, void *ws)
{
    uint32_t cand[kMaxFormats];
    const int n_cand = table_format_candidates(c, tiled, limit, cand);
    const uint64_t E = table_entries(c);
    for (int i = 0; i < n_cand; ++i) {
        const uint32_t dlog = cand[i];
        int verdict = dlog ? fmt_verdict(p, dlog) : (int)kFmtOk;
        if (verdict == kFmtBad) continue;
        if (verdict == kFmtUnknown && stream_is_capturing(l.stream)) continue;   // no read-back inside a capture
        c.tab_dlog = dlog;
        c.tab_coarse = dlog ? (const void *)((const char *)ws + E * 4ull) : nullptr;
        c.tab_check = nullptr;
        if (verdict == kFmtUnknown) {
            c.tab_check = (uint32_t *)((char *)ws + E * 8ull - 8ull);
            const hipError_t he = hipMemsetAsync(c.tab_check, 0, 8, (hipStream_t)l.stream);
            if (he != hipSuccess) return fail_hip(he, "hipMemsetAsync(check word)");
        }
        const int e = bhwk_table_build(l, c, (int32_t *)ws);
        if (e) return fail_hip(e, "table build launch");
        if (verdict == kFmtOk) return BHW_OK;
        uint32_t flag = 1;
        hipError_t he = hipMemcpyAsync(&flag, c.tab_check, sizeof flag, hipMemcpyDeviceToHost, (hipStream_t)l.stream);
        if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)l.stream);
        if (he != hipSuccess) return fail_hip(he, "read-back of the table format check");
        c.tab_check = nullptr;
        fmt_set_verdict(p, dlog, flag ? kFmtBad : kFmtOk);
        if (!flag) return BHW_OK;                                                // exact: keep the table just built
    }
    return fail(BHW_ERR_HIP, "no table format applies");                         // unreachable: plain is always a candidate
}

int generate_impl(const bhw_params *p, int device, void *stream, uint64_t n0, uint64_t count, int32_t *d_out,
                  const bhw_exec *ex, const int32_t *apply_x = nullptr, uint32_t apply_shift = 0)
{
    int rc = validate(p);
    if (rc) return rc;
    if (count && !d_out) return fail(BHW_ERR_BADARG, "d_out is NULL");
    rc = check_exec(ex);
    if (rc) return rc;
    if (!count) return BHW_OK;
    if (count > (1ull << 34)) return fail(BHW_ERR_BADARG, "count %llu > 2^34 per call", (unsigned long long)count);
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    BhwLaunch l{device, stream};
    BhwWinCfg w;
    resolve_window(p, w);
    w.apply_x = apply_x;
    w.apply_shift = apply_shift;
    if (p->sin_type != BHW_SIN_CORDIC) {
        BhwTaylorCfg t;
        rc = resolve_taylor(p, device, stream, t);
        if (rc) return rc;
        const uint64_t NT = 1ull << p->phi_width;
        auto ragged = [&](uint64_t off, uint64_t len) -> int {
            if (apply_x) w.apply_x = apply_x + off;
            return bhwk_taylor_window(l, t, w, n0 + off, len, d_out + off);
        };
        if (p->phi_width < 5) {
            int e = ragged(0, count);
            return e ? fail_hip(e, "taylor window launch") : BHW_OK;
        }
        auto period = [&](uint64_t off) -> int {
            if (apply_x) w.apply_x = apply_x + off;
            return bhwk_taylor_window_fold(l, t, w, d_out + off);
        };
        return run_split(l, n0, count, NT, apply_x != nullptr, d_out, ragged, period);
    }
    BhwCordicCfg c;
    resolve_cordic(p, c);
    const uint32_t algo = pick_algo(p, c, n0, count, ex ? ex->algo : BHW_ALGO_AUTO);
    if (algo == BHW_ALGO_DIRECT) {
        int e = bhwk_direct(l, c, w, n0, count, d_out);
        return e ? fail_hip(e, "direct launch") : BHW_OK;
    }
    if (algo == BHW_ALGO_FUSED) {
        // head | whole periods | tail: each whole period is one launch of the fused kernel over the full ring, the ragged
        // ends take the direct kernel; nothing is allocated and no table exists
        const uint64_t NF = 1ull << p->phi_width;
        const BhwFoldRun ring{0u, 1u << (p->phi_width - 3)};
        auto ragged = [&](uint64_t off, uint64_t len) -> int {
            if (apply_x) w.apply_x = apply_x + off;
            return bhwk_direct(l, c, w, n0 + off, len, d_out + off);
        };
        auto period = [&](uint64_t off) -> int {
            if (apply_x) w.apply_x = apply_x + off;
            return bhwk_fold_direct(l, c, w, &ring, 1, d_out + off);
        };
        return run_split(l, n0, count, NF, apply_x != nullptr, d_out, ragged, period);
    }
    const uint64_t need = table_entries(c) * 8ull;
    void *ws = nullptr;
    std::shared_ptr<Slot> slot;
    std::unique_lock<std::mutex> slot_lock;                  // held until every launch of this call is enqueued
    if (ex && ex->workspace) {
        if (ex->workspace_bytes < need)
            return fail(BHW_ERR_WORKSPACE, "workspace %llu < %llu bytes", (unsigned long long)ex->workspace_bytes,
                        (unsigned long long)need);
        ws = ex->workspace;
    } else {
        slot = slot_of(device, stream);
        slot_lock = std::unique_lock<std::mutex>(slot->mu);
        rc = ensure_slot_bytes(*slot, stream, need);
        if (rc) return rc;
        ws = slot->buf;
    }
    // head | whole periods | tail over the one table built here: the whole periods take the fold / tile kernels, the
    // ragged ends the general gather kernel
    const uint64_t N = 1ull << p->phi_width;
    const bool has_period = count >= (N - n0 % N) % N + N;
    // a contiguous range of whole eighths of one window (one device's contiguous shard of a window split over 2, 4 or 8): the
    // tile kernel over the images it covers
    uint32_t img_mask = 0xFFu, n0mod = 0u;
    BhwWinCfg w_probe = w;
    w_probe.apply_x = apply_x;
    const bool images = !has_period && bhwk_tile_images_applicable(c, w_probe, n0, count, &img_mask, &n0mod);
    const bool tiled = (has_period && bhwk_tile_applicable(c, w)) || images;
    c.tab_split = (tiled && c.z_shr == 0) ? 1u : 0u;
    // whole-period tile tables are stored packed when the widths allow it (formats in bhw_device.h): "residual" = 2 bytes per
    // entry + one int4 record per 2^d entries, else "delta16" = 4 bytes per entry + one int2 head per 64 entries, else the plain
    // 8 bytes per entry.  The combine pass is bound by table + output traffic as much as by arithmetic.
    rc = build_table(p, l, c, tiled, exec_table_format(ex), ws);
    if (rc) return rc;
    if (ex && ex->event_after_build) {
        hipError_t he = hipEventRecord((hipEvent_t)ex->event_after_build, (hipStream_t)stream);
        if (he != hipSuccess) return fail_hip(he, "hipEventRecord(event_after_build)");
    }
    auto ragged = [&](uint64_t off, uint64_t len) -> int {
        if (apply_x) w.apply_x = apply_x + off;
        return bhwk_table_combine(l, c, w, (const int32_t *)ws, n0 + off, len, d_out + off);
    };
    auto period = [&](uint64_t off) -> int {
        if (apply_x) w.apply_x = apply_x + off;
        // dropped phase bits: consecutive coefficients repeat table entries -- the run-length kernel works per breakpoint
        if (bhwk_runlength_applicable(c, w, d_out + off)) return bhwk_runlength_window(l, c, w, (const int32_t *)ws, d_out + off);
        return tiled ? bhwk_table_combine_tile(l, c, w, (const int32_t *)ws, d_out + off)
                     : bhwk_table_combine_fold(l, c, w, (const int32_t *)ws, d_out + off);
    };
    if (images) {
        const int e = bhwk_table_combine_tile_range(l, c, w, (const int32_t *)ws, d_out, 0, 0, img_mask, n0mod);
        return e ? fail_hip(e, "tile launch (image subset)") : BHW_OK;
    }
    return run_split(l, n0, count, N, apply_x != nullptr, d_out, ragged, period);
}

} // namespace

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
    const int K = terms_of(win_type);
    if (!K) return fail(BHW_ERR_BADARG, "win_type %u", win_type);
    if (dat_width < 8 || dat_width > 32) return fail(BHW_ERR_BADARG, "dat_width %u outside 8..32", dat_width);
    if (!aa) return fail(BHW_ERR_BADARG, "aa is NULL");
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
    for (int k = 0; k < K; ++k) aa[k] = (int32_t)(int64_t)std::round(a[k] * scale);
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
    if (preset < 1 || preset >= sizeof kPresets / sizeof kPresets[0]) return fail(BHW_ERR_BADARG, "preset %u", preset);
    if (win_type) *win_type = kPresets[preset].win;
    if (a) memcpy(a, kPresets[preset].a, 7 * sizeof(double));
    if (aa) return bhw_coeffs_from_float(kPresets[preset].win, dat_width, kPresets[preset].a, aa);
    return BHW_OK;
}

int bhw_params_init(bhw_params *p, uint32_t win_type, uint32_t phi_width, uint32_t dat_width)
{
    if (!p) return fail(BHW_ERR_BADARG, "params is NULL");
    memset(p, 0, sizeof *p);
    p->struct_size = sizeof *p;
    p->model = BHW_MODEL_HLS;
    p->combine = BHW_COMBINE_HLS;
    p->sin_type = BHW_SIN_CORDIC;
    p->win_type = win_type;
    p->n_terms = (uint32_t)terms_of(win_type);
    p->phi_width = phi_width;
    p->dat_width = dat_width;
    p->precision = 1;
    p->lut_size = 9;
    if (!p->n_terms) return fail(BHW_ERR_BADARG, "win_type %u", win_type);
    int rc = bhw_coeffs_from_float(win_type, dat_width, nullptr, p->aa);
    if (rc) return rc;
    return validate(p);
}

int bhw_params_validate(const bhw_params *p) { return validate(p); }

int bhw_constant_tables(uint32_t which, int64_t table[48], int64_t gains[2])
{
    if (which > 1) return fail(BHW_ERR_BADARG, "which %u", which);
    if (table) memcpy(table, which ? kAtanT4 : kAtanT2, 48 * sizeof(int64_t));
    if (gains) { gains[0] = kGain46; gains[1] = kGain47; }
    return BHW_OK;
}

int bhw_generate_device(const bhw_params *p, int device, void *hip_stream, uint64_t n0, uint64_t count, int32_t *d_out)
{
    return generate_impl(p, device, hip_stream, n0, count, d_out, nullptr);
}

int bhw_generate_device_ex(const bhw_params *p, int device, void *hip_stream, uint64_t n0, uint64_t count,
                           int32_t *d_out, const bhw_exec *ex)
{
    return generate_impl(p, device, hip_stream, n0, count, d_out, ex);
}

int bhw_apply_device(const bhw_params *p, int device, void *hip_stream, uint64_t n0, uint64_t count,
                     const int32_t *d_x, int32_t *d_y, uint32_t shift)
{
    if (count && (!d_x || !d_y)) return fail(BHW_ERR_BADARG, "d_x / d_y is NULL");
    if (shift > 62) return fail(BHW_ERR_BADARG, "shift %u > 62", shift);
    const uintptr_t xa = (uintptr_t)d_x, ya = (uintptr_t)d_y, bytes = (uintptr_t)count * 4u;
    if (count && xa < ya + bytes && ya < xa + bytes)
        return fail(BHW_ERR_BADARG, "d_y must not overlap d_x (tile seams recompute a few samples)");
    return generate_impl(p, device, hip_stream, n0, count, d_y, nullptr, d_x, shift);
}

uint64_t bhw_workspace_bytes(const bhw_params *p, uint64_t n0, uint64_t count, uint32_t algo)
{
    if (validate(p)) return 0;
    if (p->sin_type != BHW_SIN_CORDIC) return 0;
    BhwCordicCfg c;
    resolve_cordic(p, c);
    return pick_algo(p, c, n0, count, algo) == BHW_ALGO_TABLE ? table_entries(c) * 8ull : 0;
}

int bhw_describe_plan(const bhw_params *p, uint64_t n0, uint64_t count, const bhw_exec *ex, char *buf, uint64_t len)
{
    int rc = validate(p);
    if (rc) return rc;
    rc = check_exec(ex);
    if (rc) return rc;
    if (!buf || !len) return fail(BHW_ERR_BADARG, "buf is NULL or empty");
    const bool period = has_whole_period(p, n0, count);
    if (p->sin_type != BHW_SIN_CORDIC) {
        snprintf(buf, len, "taylor: %s", period && p->phi_width >= 5 ? "k_taylor_window_fold (+ k_taylor_window on ragged ends)" : "k_taylor_window");
        return BHW_OK;
    }
    BhwCordicCfg c;
    resolve_cordic(p, c);
    BhwWinCfg w;
    resolve_window(p, w);
    const uint32_t algo = pick_algo(p, c, n0, count, ex ? ex->algo : BHW_ALGO_AUTO);
    if (algo == BHW_ALGO_DIRECT) {
        snprintf(buf, len, "direct: %s", (c.dat_width + c.out_shr <= 34 && c.n_iter >= 7) ? "k_direct_fast" : "k_direct");
        return BHW_OK;
    }
    if (algo == BHW_ALGO_FUSED) {
        snprintf(buf, len, "fused: k_fold_direct<%u,%d> (+ k_direct_fast on ragged ends)", p->n_terms,
                 (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0));
        return BHW_OK;
    }
    uint32_t img_mask = 0xFFu, n0mod = 0u;
    const bool images = !period && bhwk_tile_images_applicable(c, w, n0, count, &img_mask, &n0mod);
    const bool tiled = (period && bhwk_tile_applicable(c, w)) || images;
    c.tab_split = (tiled && c.z_shr == 0) ? 1u : 0u;
    uint32_t cand[kMaxFormats];
    const int n_cand = table_format_candidates(c, tiled, exec_table_format(ex), cand);
    const char *state = "";
    c.tab_dlog = 0;
    for (int i = 0; i < n_cand; ++i) {
        const int v = cand[i] ? fmt_verdict(p, cand[i]) : (int)kFmtOk;
        if (v == kFmtBad) continue;
        c.tab_dlog = cand[i];
        if (v == kFmtUnknown) state = ", unverified";
        break;
    }
    char build[64], combine[64];
    bhwk_describe_table(c, w, tiled, build, combine, sizeof build);
    if (period && c.tab_dlog == 0 && bhwk_runlength_applicable(c, w, nullptr))     // generate_impl's period(): dropped phase bits
        snprintf(combine, sizeof combine, "k_runlength_window<%u,%d,%s>", p->n_terms, (w.combine != BHW_COMBINE_HLS) ? 2 : (c.ones_neg ? 1 : 0),
                 c.dat_width <= 16 ? "true" : "false");
    const char *fmt = c.tab_dlog == 0 ? "plain" : c.tab_dlog == 6 ? "delta16" : c.tab_dlog >= 16 ? "nibble" : "residual";
    snprintf(buf, len, "table[%s%s]: %s + %s%s", fmt, state, build, (period || images) ? combine : "k_table_combine",
             images ? " (image subset)" : period && count != (1ull << p->phi_width) ? " (+ k_table_combine / k_replicate on the rest)" : "");
    return BHW_OK;
}

int bhw_generate_batched_device(const bhw_params *p, int device, void *hip_stream, uint32_t frames, int32_t *d_out)
{
    int rc = validate(p);
    if (rc) return rc;
    if (!frames) return BHW_OK;
    if (!d_out) return fail(BHW_ERR_BADARG, "d_out is NULL");
    const uint64_t N = 1ull << p->phi_width;
    // frame 0 is generated in place, then replicated into frames 1..frames-1
    rc = generate_impl(p, device, hip_stream, 0, N, d_out, nullptr);
    if (rc || frames == 1) return rc;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    BhwLaunch l{device, hip_stream};
    int e = bhwk_replicate(l, d_out, N, frames - 1, d_out + N);
    return e ? fail_hip(e, "replicate launch") : BHW_OK;
}

int bhw_sincos_device(const bhw_params *p, int device, void *hip_stream, uint64_t theta0, uint64_t count,
                      int32_t *d_sin, int32_t *d_cos)
{
    int rc = validate(p, true);
    if (rc) return rc;
    if (!count) return BHW_OK;
    if (!d_sin && !d_cos) return fail(BHW_ERR_BADARG, "both outputs NULL");
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    BhwLaunch l{device, hip_stream};
    if (p->sin_type != BHW_SIN_CORDIC) {
        BhwTaylorCfg t;
        rc = resolve_taylor(p, device, hip_stream, t);
        if (rc) return rc;
        int e = bhwk_taylor_sincos(l, t, theta0, count, d_sin, d_cos);
        return e ? fail_hip(e, "taylor sincos launch") : BHW_OK;
    }
    if (p->model > BHW_MODEL_VHDL) {
        // cordic_dds48: SIZE = DWPH = 48 (src/cordic_dds48.vhd:143-153); cordic_dds_scaled: SIZE = SEL_SIZE(DATA_WIDTH-8),
        // DWPH = max(SIZE, PHASE_WIDTH) (src/cordic_dds_scaled.vhd:109,133-143)
        BhwPrerotCfg c;
        memset(&c, 0, sizeof c);
        c.phi_width = p->phi_width;
        c.dat_width = p->dat_width;
        c.size = p->model == BHW_MODEL_DDS48 ? 48u : kSelSize[p->dat_width - 8];
        c.dwph = c.size < p->phi_width ? p->phi_width : c.size;
        c.gain = kGain46 >> (48 - c.size);                                          // GAIN48(47 downto 48-SIZE)
        for (uint32_t i = 0; i + 1 < p->dat_width; ++i) c.lut[i] = kAtanT2[i] >> (48 - c.dwph);   // ROM_LUT(ii)(47 downto 48-DWPH)
        int e = bhwk_sincos_prerot(l, c, theta0, count, d_sin, d_cos);
        return e ? fail_hip(e, "sincos (pre-rotated) launch") : BHW_OK;
    }
    BhwCordicCfg c;
    resolve_cordic(p, c);
    int e = bhwk_sincos(l, c, theta0, count, d_sin, d_cos);
    return e ? fail_hip(e, "sincos launch") : BHW_OK;
}

int bhw_generate_to_host(const bhw_params *p, int device, uint64_t n0, uint64_t count, int32_t *h_out)
{
    int rc = validate(p);
    if (rc) return rc;
    if (!count) return BHW_OK;
    if (!h_out) return fail(BHW_ERR_BADARG, "h_out is NULL");
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    int32_t *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, count * sizeof(int32_t));
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(out)");
    rc = generate_impl(p, device, nullptr, n0, count, d, nullptr);
    if (!rc) {
        e = hipMemcpy(h_out, d, count * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail_hip(e, "hipMemcpy(D2H)");
    }
    (void)hipFree(d);
    return rc;
}

int bhw_sincos_to_host(const bhw_params *p, int device, uint64_t theta0, uint64_t count, int32_t *h_sin, int32_t *h_cos)
{
    int rc = validate(p, true);
    if (rc) return rc;
    if (!count) return BHW_OK;
    if (!h_sin && !h_cos) return fail(BHW_ERR_BADARG, "both outputs NULL");
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    int32_t *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, 2 * count * sizeof(int32_t));
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(out)");
    rc = bhw_sincos_device(p, device, nullptr, theta0, count, d, d + count);
    if (!rc && h_sin) {
        e = hipMemcpy(h_sin, d, count * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail_hip(e, "hipMemcpy(D2H)");
    }
    if (!rc && h_cos) {
        e = hipMemcpy(h_cos, d + count, count * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail_hip(e, "hipMemcpy(D2H)");
    }
    (void)hipFree(d);
    return rc;
}

// Development hooks (not part of the ABI in include/bhw.h): the two passes of the table strategy on their own,
// for overlap experiments (tools/overlap_probe.py).  They use the narrowest table format already verified for the
// configuration (a bhw_generate_device call of the same parameters settles it), plain otherwise.
static void dbg_verified_format(const bhw_params *p, BhwCordicCfg &c, bool tiled, const void *ws)
{
    uint32_t cand[kMaxFormats];
    const int n = table_format_candidates(c, tiled, BHW_TABLE_BEST, cand);
    c.tab_dlog = 0;
    for (int i = 0; i < n; ++i)
        if (cand[i] && fmt_verdict(p, cand[i]) == kFmtOk) { c.tab_dlog = cand[i]; break; }
    c.tab_coarse = c.tab_dlog ? (const void *)((const char *)ws + table_entries(c) * 4ull) : nullptr;
    c.tab_check = nullptr;
}

int bhw_dbg_table_build(const bhw_params *p, int device, void *stream, void *ws)
{
    if (validate(p)) return BHW_ERR_BADARG;
    BhwCordicCfg c;
    resolve_cordic(p, c);
    BhwWinCfg w;
    resolve_window(p, w);
    c.tab_split = (bhwk_tile_applicable(c, w) && c.z_shr == 0) ? 1u : 0u;
    dbg_verified_format(p, c, bhwk_tile_applicable(c, w), ws);
    DeviceGuard guard(device);
    BhwLaunch l{device, stream};
    return bhwk_table_build(l, c, (int32_t *)ws);
}

int bhw_dbg_table_combine(const bhw_params *p, int device, void *stream, const void *ws, int32_t *d_out)
{
    if (validate(p)) return BHW_ERR_BADARG;
    BhwCordicCfg c;
    resolve_cordic(p, c);
    BhwWinCfg w;
    resolve_window(p, w);
    DeviceGuard guard(device);
    BhwLaunch l{device, stream};
    if (bhwk_tile_applicable(c, w)) {
        c.tab_split = c.z_shr == 0 ? 1u : 0u;
        dbg_verified_format(p, c, true, ws);
        return bhwk_table_combine_tile(l, c, w, (const int32_t *)ws, d_out);
    }
    return bhwk_table_combine_fold(l, c, w, (const int32_t *)ws, d_out);
}

// Builds the table of `p` in the packed format `dlog` (6 delta16, 7..9 residual, 23..25 nibble) with the overflow check on, whether or not
// the format would be chosen for this configuration, and returns the check word.  `ws`: bhw_workspace_bytes(TABLE) bytes.
int bhw_dbg_check_table_format(const bhw_params *p, int device, void *stream, uint32_t dlog, void *ws, uint32_t *flag_out)
{
    if (validate(p) || !ws || !flag_out || dlog < 6 || (dlog > 9 && (dlog < 16u + 7u || dlog > 16u + 9u))) return BHW_ERR_BADARG;
    BhwCordicCfg c;
    resolve_cordic(p, c);
    if (c.z_shr != 0 || c.n_iter < 21 || c.dat_width + c.out_shr > 34 || table_entries(c) < (1ull << 20)) return BHW_ERR_UNSUPPORTED;   // packed tables exist for tiled windows (PW >= 22) only
    DeviceGuard guard(device);
    const uint64_t E = table_entries(c);
    c.tab_split = 1u;
    c.tab_dlog = dlog;
    c.tab_coarse = (const char *)ws + E * 4ull;
    c.tab_check = (uint32_t *)((char *)ws + E * 8ull - 8ull);
    hipError_t he = hipMemsetAsync(c.tab_check, 0, 8, (hipStream_t)stream);
    if (he != hipSuccess) return BHW_ERR_HIP;
    BhwLaunch l{device, stream};
    if (bhwk_table_build(l, c, (int32_t *)ws)) return BHW_ERR_HIP;
    he = hipMemcpyAsync(flag_out, c.tab_check, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);
    return he == hipSuccess ? BHW_OK : BHW_ERR_HIP;
}

// Verdict cache of the packed formats (0 unknown, 1 exact, 2 overflows); set != 0 overrides it (tests of the fallback).
int bhw_dbg_table_format_verdict(const bhw_params *p, uint32_t dlog, int set)
{
    if (validate(p)) return BHW_ERR_BADARG;
    if (set) fmt_set_verdict(p, dlog, set);
    return fmt_verdict(p, dlog);
}

// tab_dlog the residual format would use for `p` (0: not applicable) and whether delta16 applies
int bhw_dbg_table_format_info(const bhw_params *p, uint32_t *resid_dlog, uint32_t *delta16_ok)
{
    if (validate(p)) return BHW_ERR_BADARG;
    BhwCordicCfg c;
    resolve_cordic(p, c);
    if (resid_dlog) *resid_dlog = c.n_iter >= 21 ? bhwk_resid_dlog(c) : 0u;
    if (delta16_ok) *delta16_ok = (c.n_iter >= 21 && bhwk_packed_ok(c)) ? 1u : 0u;
    return BHW_OK;
}

static int validate_atan2(const bhw_atan2_params *p)
{
    if (!p) return fail(BHW_ERR_BADARG, "params is NULL");
    if (p->struct_size != sizeof(bhw_atan2_params))
        return fail(BHW_ERR_BADARG, "struct_size %u != %zu", p->struct_size, sizeof(bhw_atan2_params));
    if (p->precision < 1 || p->precision > 7) return fail(BHW_ERR_BADARG, "precision %u outside 1..7", p->precision);
    if (p->angle_width < 4 || p->angle_width > 32) return fail(BHW_ERR_BADARG, "angle_width %u outside 4..32", p->angle_width);
    if (p->input_width > 32) return fail(BHW_ERR_BADARG, "input_width %u > 32", p->input_width);
    if (p->input_width + 1 < p->angle_width)   // VEC_DX(ii) for ii = 0 .. ANGLE_WIDTH-2: src/cordic_atan2.vhd:142-145
        return fail(BHW_ERR_UNSUPPORTED, "input_width %u < angle_width-1: the entity indexes input bits 0..ANGLE_WIDTH-2", p->input_width);
    return BHW_OK;
}

int bhw_atan2_device(const bhw_atan2_params *p, int device, void *hip_stream, uint64_t count,
                     const int32_t *d_x, const int32_t *d_y, int32_t *d_phi)
{
    int rc = validate_atan2(p);
    if (rc) return rc;
    if (!count) return BHW_OK;
    if (!d_x || !d_y || !d_phi) return fail(BHW_ERR_BADARG, "d_x / d_y / d_phi is NULL");
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    BhwAtan2Cfg c;
    memset(&c, 0, sizeof c);
    c.precision = p->precision;
    c.input_width = p->input_width;
    c.angle_width = p->angle_width;
    const uint32_t B = p->angle_width + p->precision;
    for (uint32_t i = 0; i + 1 < p->angle_width; ++i) c.lut[i] = kAtanT4[i] >> (49 - B);   // src/cordic_atan2.vhd:100-103
    BhwLaunch l{device, hip_stream};
    int e = bhwk_atan2(l, c, count, d_x, d_y, d_phi);
    return e ? fail_hip(e, "atan2 launch") : BHW_OK;
}

int bhw_atan2_to_host(const bhw_atan2_params *p, int device, uint64_t count, const int32_t *h_x, const int32_t *h_y, int32_t *h_phi)
{
    int rc = validate_atan2(p);
    if (rc) return rc;
    if (!count) return BHW_OK;
    if (!h_x || !h_y || !h_phi) return fail(BHW_ERR_BADARG, "h_x / h_y / h_phi is NULL");
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    int32_t *d = nullptr;
    hipError_t e = hipMalloc((void **)&d, 3 * count * sizeof(int32_t));
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(atan2)");
    e = hipMemcpy(d, h_x, count * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + count, h_y, count * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail_hip(e, "hipMemcpy(H2D)");
    if (!rc) rc = bhw_atan2_device(p, device, nullptr, count, d, d + count, d + 2 * count);
    if (!rc) {
        e = hipMemcpy(h_phi, d + 2 * count, count * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail_hip(e, "hipMemcpy(D2H)");
    }
    (void)hipFree(d);
    return rc;
}

// ---- interleaved ownership parts (include/bhw.h) ------------------------------------------------------------------------
namespace {
int part_checks(const bhw_params *p, uint32_t part, uint32_t n_parts)
{
    int rc = validate(p);
    if (rc) return rc;
    if (p->sin_type != BHW_SIN_CORDIC) return fail(BHW_ERR_UNSUPPORTED, "interleaved parts exist for the CORDIC source only");
    if (n_parts < 1 || n_parts > 64 || part >= n_parts) return fail(BHW_ERR_BADARG, "part %u of %u (1..64 parts)", part, n_parts);
    if (p->phi_width < 9) return fail(BHW_ERR_UNSUPPORTED, "interleaved parts need phi_width >= 9 (a ring of 64 lanes)");
    // a part is produced by the fused kernel (CORDIC state within 34 bits) or by the tile kernel over the full table (N >= 2^22):
    // configurations with neither (e.g. VHDL model, W = 32, PRECISION >= 3 below 2^22) have no part kernel, and the segment
    // arithmetic must not promise what bhw_generate_part_device cannot deliver
    BhwCordicCfg c;
    resolve_cordic(p, c);
    BhwWinCfg w;
    resolve_window(p, w);
    if (!bhwk_fold_direct_applicable(c) && !bhwk_tile_applicable(c, w))
        return fail(BHW_ERR_UNSUPPORTED, "no kernel produces ownership parts of this configuration (CORDIC state beyond 34 bits and no tile plan)");
    return BHW_OK;
}
} // namespace

int bhw_part_segments(const bhw_params *p, uint32_t part, uint32_t n_parts, bhw_segment *segs, uint32_t capacity, uint32_t *n_segs)
{
    int rc = part_checks(p, part, n_parts);
    if (rc) return rc;
    if (!n_segs) return fail(BHW_ERR_BADARG, "n_segs is NULL");
    BhwCordicCfg c;
    resolve_cordic(p, c);
    BhwWinCfg w;
    resolve_window(p, w);
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
        if (capacity < merged.size()) return fail(BHW_ERR_BADARG, "capacity %u < %zu segments", capacity, merged.size());
        for (size_t i = 0; i < merged.size(); ++i) segs[i] = merged[i];
    }
    return BHW_OK;
}

int bhw_generate_part_device(const bhw_params *p, int device, void *hip_stream, uint32_t part, uint32_t n_parts,
                             int32_t *d_window, const bhw_exec *ex)
{
    int rc = part_checks(p, part, n_parts);
    if (rc) return rc;
    if (!d_window) return fail(BHW_ERR_BADARG, "d_window is NULL");
    rc = check_exec(ex);
    if (rc) return rc;
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    BhwLaunch l{device, hip_stream};
    BhwCordicCfg c;
    resolve_cordic(p, c);
    BhwWinCfg w;
    resolve_window(p, w);
    BhwFoldRun runs[32];
    uint32_t tile0 = 0, tile_count = 0;
    const int n_runs = bhwk_part_runs(c, w, part, n_parts, runs, &tile0, &tile_count);
    if (n_runs == 0) return BHW_OK;
    uint64_t lanes = 0;
    for (int i = 0; i < n_runs; ++i) lanes += runs[i].r_end - runs[i].r0;
    // Strategy.  Fused: chains = lanes x (chains per lane), no table.  Table: the full first-quadrant table (it does not
    // shrink with the part) + this part's tiles.  The fused kernel wins once the part is small enough.
    static const int kChains[8] = {0, 0, 2, 3, 5, 6, 0, 9};
    const uint64_t chains_fused = lanes * (uint64_t)kChains[p->n_terms];
    const uint32_t req = ex ? ex->algo : (uint32_t)BHW_ALGO_AUTO;
    const bool fused_ok = bhwk_fold_direct_applicable(c);
    const bool table_ok = tile_count != 0;                     // tile-aligned ownership: the tile kernel can produce exactly this part
    bool fused;
    if (req == BHW_ALGO_FUSED) fused = fused_ok;
    else if (req == BHW_ALGO_TABLE) fused = !table_ok;
    else fused = fused_ok && (!table_ok || chains_fused <= table_entries(c));
    // measured (BH-7 2^26 / 32-bit, profiles/r02_small_windows_and_parts.json): fused = 0.085 ms x chains_fused / entries, table =
    // 0.073 ms + 0.063 ms / n_parts -> they cross at 1.04 x the table's chains (between 4 and 5 parts)
    if (fused) {
        if (!fused_ok) return fail(BHW_ERR_UNSUPPORTED, "no kernel produces this part (CORDIC state beyond 34 bits and no tile plan)");
        const int e = bhwk_fold_direct(l, c, w, runs, (uint32_t)n_runs, d_window);
        return e ? fail_hip(e, "fused part launch") : BHW_OK;
    }
    if (!table_ok) return fail(BHW_ERR_UNSUPPORTED, "the table strategy produces whole tiles only and this window has no tile plan");
    const uint64_t need = table_entries(c) * 8ull;
    void *ws = nullptr;
    std::shared_ptr<Slot> slot;
    std::unique_lock<std::mutex> slot_lock;
    if (ex && ex->workspace) {
        if (ex->workspace_bytes < need)
            return fail(BHW_ERR_WORKSPACE, "workspace %llu < %llu bytes", (unsigned long long)ex->workspace_bytes, (unsigned long long)need);
        ws = ex->workspace;
    } else {
        slot = slot_of(device, hip_stream);
        slot_lock = std::unique_lock<std::mutex>(slot->mu);
        rc = ensure_slot_bytes(*slot, hip_stream, need);
        if (rc) return rc;
        ws = slot->buf;
    }
    c.tab_split = c.z_shr == 0 ? 1u : 0u;
    rc = build_table(p, l, c, true, exec_table_format(ex), ws);
    if (rc) return rc;
    if (ex && ex->event_after_build) {
        hipError_t he = hipEventRecord((hipEvent_t)ex->event_after_build, (hipStream_t)hip_stream);
        if (he != hipSuccess) return fail_hip(he, "hipEventRecord(event_after_build)");
    }
    const int e = bhwk_table_combine_tile_range(l, c, w, (const int32_t *)ws, d_window, tile0, tile_count);
    return e ? fail_hip(e, "tile part launch") : BHW_OK;
}

int bhw_gather_parts_device(const bhw_params *p, uint32_t n_parts, const int *src_devices, const int32_t *const *d_windows,
                            int dst_device, void *dst_stream, int32_t *d_dst)
{
    int rc = part_checks(p, 0, n_parts);
    if (rc) return rc;
    if (!src_devices || !d_windows || !d_dst) return fail(BHW_ERR_BADARG, "src_devices / d_windows / d_dst is NULL");
    if (!device_ok(dst_device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", dst_device);
    for (uint32_t g = 0; g < n_parts; ++g) {
        if (!d_windows[g]) return fail(BHW_ERR_BADARG, "d_windows[%u] is NULL", g);
        if (!device_ok(src_devices[g])) return fail(BHW_ERR_HIP, "no usable HIP device %d", src_devices[g]);
    }
    DeviceGuard guard(dst_device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    std::vector<bhw_segment> segs(256);
    for (uint32_t g = 0; g < n_parts; ++g) {
        if (d_windows[g] == d_dst) continue;                      // this part was generated in place
        uint32_t n = 0;
        rc = bhw_part_segments(p, g, n_parts, segs.data(), (uint32_t)segs.size(), &n);
        if (rc) return rc;
        for (uint32_t i = 0; i < n; ++i) {
            const hipError_t e = hipMemcpyPeerAsync(d_dst + segs[i].n0, dst_device, d_windows[g] + segs[i].n0, src_devices[g],
                                                    segs[i].count * sizeof(int32_t), (hipStream_t)dst_stream);
            if (e != hipSuccess) return fail_hip(e, "hipMemcpyPeerAsync(segment)");
        }
    }
    return BHW_OK;
}

int bhw_release_device(int device)
{
    DeviceScratch taken;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_scratch.find(device);
        if (it == g_scratch.end()) return BHW_OK;
        taken = std::move(it->second);
        g_scratch.erase(it);
    }
    DeviceGuard guard(device);
    if (guard.err == hipSuccess) {
        (void)hipDeviceSynchronize();
        for (auto &kv : taken.bufs) {
            std::lock_guard<std::mutex> lk(kv.second->mu);   // a call still enqueueing on this slot finishes first
            if (kv.second->buf) (void)hipFree(kv.second->buf);
            kv.second->buf = nullptr;
            kv.second->bytes = 0;
        }
        for (auto &kv : taken.roms) (void)hipFree(kv.second);
    }
    return BHW_OK;
}

int bhw_prepare_device(const bhw_params *p, int device, void *hip_stream)
{
    int rc = validate(p, true);
    if (rc) return rc;
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    if (p->sin_type != BHW_SIN_CORDIC) {
        BhwTaylorCfg t;
        return resolve_taylor(p, device, hip_stream, t);        // uploads the ROM on first use
    }
    if (p->model > BHW_MODEL_VHDL) return BHW_OK;               // variant generators: nothing lazy
    BhwCordicCfg c;
    resolve_cordic(p, c);
    BhwWinCfg w;
    resolve_window(p, w);
    // whole periods are what prepared callers replay; a configuration AUTO sends to the fused or direct kernel needs no scratch
    const uint64_t N = 1ull << p->phi_width;
    if (pick_algo(p, c, 0, N, BHW_ALGO_AUTO) != BHW_ALGO_TABLE) return BHW_OK;
    const uint64_t need = table_entries(c) * 8ull;
    auto slot = slot_of(device, hip_stream);
    std::unique_lock<std::mutex> lk(slot->mu);
    rc = ensure_slot_bytes(*slot, hip_stream, need);
    if (rc) return rc;
    // settle the packed-format verdicts of this configuration (build_table reads the check word back when one is open)
    if (bhwk_tile_applicable(c, w)) {
        BhwLaunch l{device, hip_stream};
        c.tab_split = c.z_shr == 0 ? 1u : 0u;
        rc = build_table(p, l, c, true, BHW_TABLE_BEST, slot->buf);
        if (rc) return rc;
    }
    const hipError_t he = hipStreamSynchronize((hipStream_t)hip_stream);
    return he == hipSuccess ? BHW_OK : fail_hip(he, "hipStreamSynchronize");
}

} // extern "C"
