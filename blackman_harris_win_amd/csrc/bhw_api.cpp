// bhw_api.cpp -- the device side of the C ABI (include/bhw.h): per-device scratch, table-format verification on the device,
// launches.  Everything that is decided from parameters alone -- validation, resolution of a (model, widths) tuple into kernel
// constants, strategy / format / shape choice, ownership segments, scratch sizing, bhw_describe_plan -- lives in the HIP-free
// bhw_plan.cpp (sanitised on the CPU build).  No per-sample arithmetic happens on the host: every compute entry point fails
// with BHW_ERR_HIP when no device is usable.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "bhw_plan.h"

extern "C" void bhw_taylor_rom(uint32_t dat_width, uint32_t lut_size, int32_t *rom_sin_cos);
extern "C" uint32_t bhw_taylor_pi_word(int e);

namespace {

#define fail bhwp_fail

int fail_hip(int hip_code, const char *what)
{
    return fail(BHW_ERR_HIP, "%s: %s (hipError %d)", what, hipGetErrorString((hipError_t)hip_code), hip_code);
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
    bool oversized = false;   // sized while some packed-format verdict of the configuration was still open (the maximum over
                              // every candidate up to plain): re-sized once the verdicts are known, see acquire_scratch
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

// slot.mu is held by the caller.  Gives back what a slot holds beyond `bytes` (never inside a capture: it synchronises).
int shrink_slot_to(Slot &slot, void *stream, uint64_t bytes)
{
    if (slot.bytes <= bytes || stream_is_capturing(stream)) return BHW_OK;
    (void)hipStreamSynchronize((hipStream_t)stream);          // earlier launches on this stream may still read it
    (void)hipFree(slot.buf);
    slot.buf = nullptr;
    slot.bytes = 0;
    if (!bytes) return BHW_OK;
    void *b = nullptr;
    const hipError_t e = hipMalloc(&b, bytes);
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(scratch)");
    slot.buf = b;
    slot.bytes = bytes;
    return BHW_OK;
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

// Build the table in the narrowest format that is exact for this configuration.  A packed format whose exactness has not
// been established yet for the configuration is built with the kernels' overflow check on and read back here (once per
// process and configuration; during stream capture the plain format is used instead).  `ws` holds
// bhwp_table_scratch_bytes(...) bytes, laid out per format by bhwp_table_layout.
int build_table(const bhw_params *p, const BhwLaunch &l, BhwCordicCfg &c, bool tiled, uint32_t limit, void *ws)
{
    uint32_t cand[kMaxFormats];
    const int n_cand = bhwp_table_format_candidates(c, tiled, limit, cand);
    const uint64_t E = bhwp_table_entries(c);
    for (int i = 0; i < n_cand; ++i) {
        const uint32_t dlog = cand[i];
        int verdict = dlog ? bhwp_fmt_verdict(p, dlog) : (int)kFmtOk;
        if (verdict == kFmtBad) continue;
        if (verdict == kFmtUnknown && stream_is_capturing(l.stream)) continue;   // no read-back inside a capture
        const BhwTableLayout lay = bhwp_table_layout(E, dlog);
        c.tab_dlog = dlog;
        c.tab_coarse = dlog ? (const void *)((const char *)ws + lay.coarse_off) : nullptr;
        c.tab_esc = lay.esc_off ? (const void *)((const char *)ws + lay.esc_off) : nullptr;
        c.esc_wg_log = lay.esc_wg_log;
        c.tab_check = nullptr;
        if (verdict == kFmtUnknown) {
            c.tab_check = (uint32_t *)((char *)ws + lay.check_off);
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
        bhwp_fmt_set_verdict(p, dlog, flag ? kFmtBad : kFmtOk);
        if (!flag) return BHW_OK;                                                // exact: keep the table just built
    }
    return fail(BHW_ERR_HIP, "no table format applies");                         // unreachable: plain is always a candidate
}

// Scratch of a table-strategy call: the caller's workspace when it passed one, else the library-owned buffer of this stream
// (locked until every launch of the call is enqueued), holding at least `need` bytes.
struct TableScratch {
    void *ws = nullptr;
    std::shared_ptr<Slot> slot;
    std::unique_lock<std::mutex> lock;
};
// `open_verdicts`: `need` is the maximum over formats whose verdict is not known yet (the first call of a configuration without
// bhw_prepare_device).  The slot remembers that, and the next call that finds the verdicts settled gives the excess back once
// (128 MiB -> 16.5 MiB for a 2^26-point window at 32 bits); a slot is never re-sized downwards otherwise, so streams that alternate
// between configurations of different sizes do not re-allocate per call.
int acquire_scratch(const bhw_exec *ex, int device, void *stream, uint64_t need, bool open_verdicts, TableScratch &t)
{
    if (ex && ex->workspace) {
        if (ex->workspace_bytes < need)
            return fail(BHW_ERR_WORKSPACE, "workspace %llu < %llu bytes", (unsigned long long)ex->workspace_bytes, (unsigned long long)need);
        t.ws = ex->workspace;
        return BHW_OK;
    }
    t.slot = slot_of(device, stream);
    t.lock = std::unique_lock<std::mutex>(t.slot->mu);
    if (t.slot->oversized && !open_verdicts && t.slot->bytes > need && !stream_is_capturing(stream)) {   // (a capture keeps the excess for a later call)
        t.slot->oversized = false;
        const int rs = shrink_slot_to(*t.slot, stream, need);
        if (rs) return rs;
    }
    const bool grows = t.slot->bytes < need;
    const int rc = ensure_slot_bytes(*t.slot, stream, need);
    if (rc) return rc;
    if (grows && open_verdicts) t.slot->oversized = true;
    t.ws = t.slot->buf;
    return BHW_OK;
}

// true while some packed format this call may try has no verdict yet (its scratch is then sized for every candidate)
bool verdicts_open(const bhw_params *p, const BhwCordicCfg &c, bool tiled, uint32_t limit)
{
    uint32_t cand[kMaxFormats];
    const int n = bhwp_table_format_candidates(c, tiled, limit, cand);
    for (int i = 0; i < n; ++i) {
        const int v = cand[i] ? bhwp_fmt_verdict(p, cand[i]) : (int)kFmtOk;
        if (v == kFmtOk) return false;
        if (v == kFmtUnknown) return true;
    }
    return false;
}

int generate_impl(const bhw_params *p, int device, void *stream, uint64_t n0, uint64_t count, int32_t *d_out,
                  const bhw_exec *ex, const int32_t *apply_x = nullptr, uint32_t apply_shift = 0)
{
    int rc = bhwp_validate(p);
    if (rc) return rc;
    if (count && !d_out) return fail(BHW_ERR_BADARG, "d_out is NULL");
    rc = bhwp_check_exec(ex);
    if (rc) return rc;
    if (!count) return BHW_OK;
    if (count > (1ull << 34)) return fail(BHW_ERR_BADARG, "count %llu > 2^34 per call", (unsigned long long)count);
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    BhwLaunch l{device, stream};
    BhwWinCfg w;
    bhwp_resolve_window(p, w);
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
    bhwp_resolve_cordic(p, c);
    const uint32_t algo = bhwp_pick_algo(p, c, w, n0, count, ex ? ex->algo : (uint32_t)BHW_ALGO_AUTO);
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
    // head | whole periods | tail over the one table built here: the whole periods take the fold / tile kernels (a contiguous
    // range of whole eighths of one window -- one device's contiguous shard of a window split over 2, 4 or 8 -- the tile kernel
    // over the images it covers), the ragged ends the general gather kernel
    const uint64_t N = 1ull << p->phi_width;
    const BhwTableCall tc = bhwp_table_call(p, c, w, n0, count, apply_x != nullptr);
    const bool tiled = tc.tiled, images = tc.images;
    const uint32_t img_mask = tc.img_mask, n0mod = tc.n0mod;
    c.tab_split = (tiled && c.z_shr == 0) ? 1u : 0u;
    // whole-period tile tables are stored packed when the widths allow it (formats in bhw_device.h): "nibble" = 1 byte per entry,
    // "residual" = 2 bytes + one int4 record per 2^d entries, else "delta16" = 4 bytes per entry + one int2 head per 64 entries, else
    // the plain 8 bytes per entry.  The scratch is sized for the format(s) this call may use.
    const uint32_t limit = bhwp_exec_table_format(ex);
    TableScratch scratch;
    const bool capturing = stream_is_capturing(stream);
    rc = acquire_scratch(ex, device, stream, bhwp_table_scratch_bytes(p, c, tiled, limit, capturing), !capturing && verdicts_open(p, c, tiled, limit), scratch);
    if (rc) return rc;
    void *ws = scratch.ws;
    rc = build_table(p, l, c, tiled, limit, ws);
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

int bhw_generate_batched_device(const bhw_params *p, int device, void *hip_stream, uint32_t frames, int32_t *d_out)
{
    int rc = bhwp_validate(p);
    if (rc) return rc;
    if (!frames) return BHW_OK;
    if (!d_out) return fail(BHW_ERR_BADARG, "d_out is NULL");
    const uint64_t N = 1ull << p->phi_width;
    if (frames > 1 && p->sin_type == BHW_SIN_CORDIC) {
        // a period the fused kernel takes, up to 2^19 coefficients: ONE launch computes it and writes every frame -- no second kernel
        // reading frame 0 back.  1024 x 2^16 (BASELINE configs[3]) 0.0430 -> 0.0408 ms; 0.0410 - 0.0419 against 0.0427 - 0.0439 for
        // periods of 2^14 .. 2^19; at 2^20 the rows' repeated computation is no longer hidden (0.0468 against 0.0441) and the old
        // path stays (profiles/r04_ab_batched_one_launch.txt)
        BhwCordicCfg c;
        bhwp_resolve_cordic(p, c);
        BhwWinCfg w;
        bhwp_resolve_window(p, w);
        if (p->phi_width <= 19 && bhwp_pick_algo(p, c, w, 0, N, BHW_ALGO_AUTO) == BHW_ALGO_FUSED && bhwp_fold_form(c, w, N >> 3) != BHWP_FOLD_SPLIT) {
            if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
            DeviceGuard guard(device);
            if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
            BhwLaunch l{device, hip_stream};
            const BhwFoldRun ring{0u, 1u << (p->phi_width - 3)};
            const int e = bhwk_fold_direct(l, c, w, &ring, 1, d_out, frames);
            return e ? fail_hip(e, "fused batched launch") : BHW_OK;
        }
    }
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
    int rc = bhwp_validate(p, true);
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
        BhwPrerotCfg c;
        bhwp_resolve_prerot(p, c);                                                  // cordic_dds48 / cordic_dds_scaled constants
        int e = bhwk_sincos_prerot(l, c, theta0, count, d_sin, d_cos);
        return e ? fail_hip(e, "sincos (pre-rotated) launch") : BHW_OK;
    }
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    int e = bhwk_sincos(l, c, theta0, count, d_sin, d_cos);
    return e ? fail_hip(e, "sincos launch") : BHW_OK;
}

int bhw_generate_to_host(const bhw_params *p, int device, uint64_t n0, uint64_t count, int32_t *h_out)
{
    int rc = bhwp_validate(p);
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
    int rc = bhwp_validate(p, true);
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
    const int n = bhwp_table_format_candidates(c, tiled, BHW_TABLE_BEST, cand);
    c.tab_dlog = 0;
    for (int i = 0; i < n; ++i)
        if (cand[i] && bhwp_fmt_verdict(p, cand[i]) == kFmtOk) { c.tab_dlog = cand[i]; break; }
    const BhwTableLayout lay = bhwp_table_layout(bhwp_table_entries(c), c.tab_dlog);
    c.tab_coarse = c.tab_dlog ? (const void *)((const char *)ws + lay.coarse_off) : nullptr;
    c.tab_esc = lay.esc_off ? (const void *)((const char *)ws + lay.esc_off) : nullptr;
    c.esc_wg_log = lay.esc_wg_log;
    c.tab_check = nullptr;
}

int bhw_dbg_table_build(const bhw_params *p, int device, void *stream, void *ws)
{
    if (bhwp_validate(p)) return BHW_ERR_BADARG;
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    BhwWinCfg w;
    bhwp_resolve_window(p, w);
    c.tab_split = (bhwk_tile_applicable(c, w) && c.z_shr == 0) ? 1u : 0u;
    dbg_verified_format(p, c, bhwk_tile_applicable(c, w), ws);
    DeviceGuard guard(device);
    BhwLaunch l{device, stream};
    return bhwk_table_build(l, c, (int32_t *)ws);
}

int bhw_dbg_table_combine(const bhw_params *p, int device, void *stream, const void *ws, int32_t *d_out)
{
    if (bhwp_validate(p)) return BHW_ERR_BADARG;
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    BhwWinCfg w;
    bhwp_resolve_window(p, w);
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
    if (bhwp_validate(p) || !ws || !flag_out || dlog < 6 || (dlog > 9 && (dlog < 16u + 7u || dlog > 16u + 9u) && (dlog < 48u + 7u || dlog > 48u + 9u))) return BHW_ERR_BADARG;
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    if (c.z_shr != 0 || c.n_iter < 21 || c.dat_width + c.out_shr > 34 || bhwp_table_entries(c) < (1ull << 20)) return BHW_ERR_UNSUPPORTED;   // packed tables exist for tiled windows (PW >= 22) only
    DeviceGuard guard(device);
    const BhwTableLayout lay = bhwp_table_layout(bhwp_table_entries(c), dlog);
    c.tab_split = 1u;
    c.tab_dlog = dlog;
    c.tab_coarse = (const char *)ws + lay.coarse_off;
    c.tab_esc = lay.esc_off ? (const void *)((const char *)ws + lay.esc_off) : nullptr;
    c.esc_wg_log = lay.esc_wg_log;
    c.tab_check = (uint32_t *)((char *)ws + lay.check_off);
    hipError_t he = hipMemsetAsync(c.tab_check, 0, 8, (hipStream_t)stream);
    if (he != hipSuccess) return BHW_ERR_HIP;
    BhwLaunch l{device, stream};
    if (bhwk_table_build(l, c, (int32_t *)ws)) return BHW_ERR_HIP;
    he = hipMemcpyAsync(flag_out, c.tab_check, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);
    return he == hipSuccess ? BHW_OK : BHW_ERR_HIP;
}

int bhw_atan2_device(const bhw_atan2_params *p, int device, void *hip_stream, uint64_t count,
                     const int32_t *d_x, const int32_t *d_y, int32_t *d_phi)
{
    int rc = bhwp_validate_atan2(p);
    if (rc) return rc;
    if (!count) return BHW_OK;
    if (!d_x || !d_y || !d_phi) return fail(BHW_ERR_BADARG, "d_x / d_y / d_phi is NULL");
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    BhwAtan2Cfg c;
    bhwp_resolve_atan2(p, c);
    BhwLaunch l{device, hip_stream};
    int e = bhwk_atan2(l, c, count, d_x, d_y, d_phi);
    return e ? fail_hip(e, "atan2 launch") : BHW_OK;
}

int bhw_atan2_to_host(const bhw_atan2_params *p, int device, uint64_t count, const int32_t *h_x, const int32_t *h_y, int32_t *h_phi)
{
    int rc = bhwp_validate_atan2(p);
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

// ---- interleaved ownership parts (include/bhw.h; bhw_part_segments and the ownership arithmetic: bhw_plan.cpp) ----------------------
int bhw_generate_part_device(const bhw_params *p, int device, void *hip_stream, uint32_t part, uint32_t n_parts,
                             int32_t *d_window, const bhw_exec *ex)
{
    int rc = bhwp_part_checks(p, part, n_parts);
    if (rc) return rc;
    if (!d_window) return fail(BHW_ERR_BADARG, "d_window is NULL");
    rc = bhwp_check_exec(ex);
    if (rc) return rc;
    if (!device_ok(device)) return fail(BHW_ERR_HIP, "no usable HIP device %d (this library has no CPU path)", device);
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return fail_hip(guard.err, "hipSetDevice");
    BhwLaunch l{device, hip_stream};
    BhwCordicCfg c;
    bhwp_resolve_cordic(p, c);
    BhwWinCfg w;
    bhwp_resolve_window(p, w);
    BhwFoldRun runs[32];
    uint32_t tile0 = 0, tile_count = 0;
    const int n_runs = bhwk_part_runs(c, w, part, n_parts, runs, &tile0, &tile_count);
    if (n_runs == 0) return BHW_OK;
    const bool fused = bhwp_part_fused(p, c, runs, n_runs, tile_count, ex ? ex->algo : (uint32_t)BHW_ALGO_AUTO, &rc);
    if (rc) return rc;
    if (fused) {
        const int e = bhwk_fold_direct(l, c, w, runs, (uint32_t)n_runs, d_window);
        return e ? fail_hip(e, "fused part launch") : BHW_OK;
    }
    c.tab_split = c.z_shr == 0 ? 1u : 0u;
    const uint32_t limit = bhwp_exec_table_format(ex);
    TableScratch scratch;
    const bool capturing = stream_is_capturing(hip_stream);
    rc = acquire_scratch(ex, device, hip_stream, bhwp_table_scratch_bytes(p, c, true, limit, capturing), !capturing && verdicts_open(p, c, true, limit), scratch);
    if (rc) return rc;
    void *ws = scratch.ws;
    rc = build_table(p, l, c, true, limit, ws);
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
    int rc = bhwp_part_checks(p, 0, n_parts);
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
    int rc = bhwp_validate(p, true);
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
    bhwp_resolve_cordic(p, c);
    BhwWinCfg w;
    bhwp_resolve_window(p, w);
    // Reserve the scratch every later table-strategy call with these widths can need -- also for configurations AUTO sends to the
    // fused kernel as whole periods: a partial range of such a window, or an explicit BHW_ALGO_TABLE, still builds a table, and
    // must not allocate inside a stream capture.  Two shapes: the plain table of ranges without a whole period (8 bytes per entry;
    // reserved up to 64 MiB -- beyond that a ragged call after prepare may still grow the scratch once, outside a capture) and the
    // packed table of whole-period tile calls, whose format verdicts are settled here first.
    const bool tile = bhwk_tile_applicable(c, w);
    const uint64_t plain = bhwp_table_layout(bhwp_table_entries(c), 0).bytes;
    uint64_t need = plain <= (64ull << 20) || !tile ? plain : 0;
    const uint32_t limits[] = {BHW_TABLE_BEST, BHW_TABLE_NIBBLE_ESC, BHW_TABLE_RESIDUAL, BHW_TABLE_DELTA16};
    if (tile) {
        c.tab_split = c.z_shr == 0 ? 1u : 0u;
        for (uint32_t limit : limits) {
            const uint64_t first = bhwp_table_scratch_bytes(p, c, true, limit, false);   // while formats are unverified: the widest that may be tried
            if ((limit == BHW_TABLE_BEST || verdicts_open(p, c, true, limit)) && first > need) need = first;
        }
    }
    auto slot = slot_of(device, hip_stream);
    std::unique_lock<std::mutex> lk(slot->mu);
    rc = ensure_slot_bytes(*slot, hip_stream, need);
    if (rc) return rc;
    // Settle the packed-format verdict of EVERY format a later call may name (build_table reads the check word back when one is
    // open): the chain table_format BEST walks, and each explicit limit -- a captured call with an explicit bhw_exec.table_format
    // must not meet an open verdict (it would fall back to the plain table, which the scratch below no longer holds).
    if (tile) {
        BhwLaunch l{device, hip_stream};
        for (uint32_t limit : limits) {
            if (limit != BHW_TABLE_BEST && !verdicts_open(p, c, true, limit)) continue;
            rc = build_table(p, l, c, true, limit, slot->buf);
            if (rc) return rc;
        }
    }
    hipError_t he = hipStreamSynchronize((hipStream_t)hip_stream);
    if (he != hipSuccess) return fail_hip(he, "hipStreamSynchronize");
    // ... and with the verdicts known the scratch is what table_format BEST needs from now on (16.5 MiB instead of 128 MiB for a
    // 2^26-point window at 32 bits), plus the plain table of partial ranges where that was reserved above
    if (tile) {
        const uint64_t settled = bhwp_table_scratch_bytes(p, c, true, BHW_TABLE_BEST, false);
        const uint64_t keep = plain <= (64ull << 20) ? (plain > settled ? plain : settled) : settled;
        slot->oversized = false;
        rc = shrink_slot_to(*slot, hip_stream, keep);
        if (rc) return rc;
    }
    return BHW_OK;
}

// Bytes the library-owned scratch of (device, hip_stream) holds right now (0: none yet) -- what bench.py reports beside the size of
// the workspace it passes itself.  Not part of the ABI in include/bhw.h.
uint64_t bhw_dbg_library_scratch_bytes(int device, void *hip_stream)
{
    std::shared_ptr<Slot> sp;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_scratch.find(device);
        if (it == g_scratch.end()) return 0;
        auto jt = it->second.bufs.find(hip_stream);
        if (jt == it->second.bufs.end()) return 0;
        sp = jt->second;
    }
    std::lock_guard<std::mutex> lk(sp->mu);
    return sp->bytes;
}

} // extern "C"
