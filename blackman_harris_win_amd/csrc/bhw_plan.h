// bhw_plan.h -- the HIP-free half of the host side: everything the library decides from parameters alone.
//
// Parameter validation, resolution of a (model, widths) tuple into kernel constants, strategy and table-format choice, the
// tile plan of the combine pass, ownership parts and their segments, scratch sizing, and the text of bhw_describe_plan.  No
// hip* include and no device state: bhw_plan.cpp compiles with a plain C++ compiler and runs under AddressSanitizer / UBSan over
// the whole parameter lattice (tests/test_sanitizers.py), which the launch code in bhw_api.cpp and the kernel units cannot.
// The kernel units include this header (through bhw_device.h) for the shapes they share with the planner.
//
// Host mirror of the reference's own host code: cpp/cordic_sincos.cpp:12-36 derives the rescaled ROM, gain and z scaling per
// call; hls/windows/win_function.cpp:74-96 does the same for the HLS model; src/cordic_dds.vhd:97-131,159-166 at elaboration.
#pragma once
#include "bhw_internal.h"

#if defined(__HIPCC__)
#define BHW_HD __host__ __device__
#else
#define BHW_HD
#endif

// ---- table formats (bhw_device.h documents the encodings) -----------------------------------------------------------------------
constexpr uint32_t kPackLog = 6;                // cfg.tab_dlog = 6: delta16
constexpr uint32_t kNibbleFlag = 16;            // cfg.tab_dlog = kNibbleFlag + d: nibble; d = 7..9 alone: residual; 0: plain
constexpr uint32_t kEscFlag = 32;               // cfg.tab_dlog = kEscFlag + kNibbleFlag + d: nibble with escapes (format 5)
constexpr uint32_t kEscSlots = 128;             // escape table of one build workgroup: open addressing, one int4 {entry, c, s, -} per slot
constexpr uint32_t kEscBias = 1;                // ... records carry c + 1, s + 1: the chord of a concave arc lies below it, and the deviations lean positive
constexpr uint32_t kEscFill = 96;               // ... entries it may hold before the format is refused (cpp at 2^26 / 32 bits: 547 in all, at most 36 in one)
BHW_HD constexpr uint32_t fmt_cell_log(uint32_t tab_dlog) { return tab_dlog & (kNibbleFlag - 1u); }
BHW_HD constexpr int fmt_of(uint32_t tab_dlog) { return tab_dlog == 0 ? 0 : tab_dlog == kPackLog ? 1 : tab_dlog >= kEscFlag ? 5 : tab_dlog >= kNibbleFlag ? 3 : 2; }

BHW_HD constexpr bool fmt_is_resid(int fmt) { return fmt == 2 || fmt == 3 || fmt == 5; }     // straight-line records + a deviation per entry
BHW_HD constexpr bool fmt_is_nibble(int fmt) { return fmt == 3 || fmt == 5; }               // ... in one byte, natural layout

// The layout goes with the format: nibble tables are always in the natural order (resid_offset), whatever the caller asked for.
inline BhwCordicCfg table_layout(const BhwCordicCfg &c)
{
    BhwCordicCfg n = c;
    if (fmt_of(c.tab_dlog) == 3 || fmt_of(c.tab_dlog) == 5) n.tab_split = 0u;
    return n;
}

// One table inside a scratch buffer: [ entries | records or block heads at coarse_off | check word at check_off ], each part
// 256-byte aligned.  bytes = what a call in this format needs.
struct BhwTableLayout {
    uint64_t coarse_off, check_off, bytes;
    uint64_t esc_off;       // nibble + escapes: the per-workgroup escape lists (0 otherwise)
    uint32_t esc_wg_log;    // ... log2 of the entries one build workgroup owns
};
BhwTableLayout bhwp_table_layout(uint64_t entries, uint32_t tab_dlog);

// ---- shapes shared with the kernels ---------------------------------------------------------------------------------------------
constexpr int kTileThreads = 960;   // combine pass, 15-run tiles: 5 thread groups x 192 lanes (bhw_combine.hip)
constexpr int kTileLanes = 192;
constexpr int kRlRun = 16;          // run-length kernel: consecutive ring lanes per thread, threads per workgroup
constexpr int kRlBlock = 64;
constexpr int kFoldRunsMax = 32;    // fused kernel: runs per launch, threads per workgroup
constexpr int kFoldBlock = 256;

struct BhwTilePlan {
    uint32_t offs[16];   // (i3*inv3 + i5*inv5) mod ring, index i3 + 3*i5; padded by repeating the last run
    uint32_t n_tiles;    // tiles that cover the ring once
    uint32_t tile0;      // first tile of this launch (interleaved ownership parts launch a sub-range of the tiles)
    uint32_t img_mask;   // MASKED instances: bit 2j + h set = image (h, j), i.e. stream indices [(2j + h) N/8, +N/8), is wanted
    uint32_t n0mod;      // MASKED instances: stream index (mod N) that `out` points at; image m lands at ((m N/8 - n0mod) mod N)
};
// The tile plan of a configuration: run offsets on the ring [0, N/8), runs per tile (1, 3 or 15), lanes per run, tiles that cover the ring.
void bhwp_tile_plan(const BhwCordicCfg &c, const BhwWinCfg &w, BhwTilePlan &tp, int &nb, uint32_t &lanes);
// one-instruction products in the 15-run tile kernel (tile_harmonic FAST) for these weights and this cosine-sum rule
bool bhwp_tile_fast(const BhwCordicCfg &c, const BhwWinCfg &w, int nb);
// k_tile9 (bhw_tile9.hip), the 15-run tile kernel compiled for one-byte tables with cells of 2^9 entries: applies to whole 15-run
// tiles with one-instruction products and all eight images, where it measured faster (everything else: k_table_combine_tile)
constexpr uint32_t kTile9CellLog = 9;
bool bhwk_tile9_applicable(const BhwCordicCfg &c, const BhwWinCfg &w, int nb, bool fast, bool masked);
int bhwk_tile9(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const BhwTilePlan &tp, uint32_t tile_count, const int32_t *d_table, int32_t *d_out);

// Form of the fused kernel for a launch of `total` ring lanes: which kernel bhwk_fold_direct starts (and bhw_describe_plan names).
enum { BHWP_FOLD_SEQUENTIAL = 0, BHWP_FOLD_LOCKSTEP = 1, BHWP_FOLD_NARROW = 2, BHWP_FOLD_SPLIT = 3 };
int bhwp_fold_form(const BhwCordicCfg &c, const BhwWinCfg &w, uint64_t total);
// first rotation from which x >> k and the ROM word fit the 24-bit factors of v_mad_i32_i24 (narrow form)
uint32_t bhwp_fold_k24(const BhwCordicCfg &c);

// workgroup size of the octant-mirror build kernel for a table of `entries`
unsigned bhwk_build_mirror_threads(uint32_t entries);

// ---- resolution, validation, strategy -------------------------------------------------------------------------------------------
int  bhwp_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));   // sets the thread's bhw_last_error() text, returns code
int  bhwp_terms_of(uint32_t win_type);
int  bhwp_validate(const bhw_params *p, bool sincos_only = false);
int  bhwp_validate_atan2(const bhw_atan2_params *p);
void bhwp_resolve_cordic(const bhw_params *p, BhwCordicCfg &c);
void bhwp_resolve_window(const bhw_params *p, BhwWinCfg &w);
void bhwp_resolve_prerot(const bhw_params *p, BhwPrerotCfg &c);
void bhwp_resolve_atan2(const bhw_atan2_params *p, BhwAtan2Cfg &c);
inline uint64_t bhwp_table_entries(const BhwCordicCfg &c) { return 1ull << (c.phi_width - 2 - c.z_shr); }
bool bhwp_has_whole_period(const bhw_params *p, uint64_t n0, uint64_t count);
uint32_t bhwp_pick_algo(const bhw_params *p, const BhwCordicCfg &c, const BhwWinCfg &w, uint64_t n0, uint64_t count, uint32_t requested);
int  bhwp_check_exec(const bhw_exec *ex);
uint32_t bhwp_exec_table_format(const bhw_exec *ex);

// Table formats a tiled whole-period call may use, narrowest first (tab_dlog values); plain (0) is always the last one.
constexpr int kMaxFormats = 5;
int bhwp_table_format_candidates(const BhwCordicCfg &c, bool tiled, uint32_t limit, uint32_t out[kMaxFormats]);
// verdict cache of the packed formats: a property of (model, PW, W, PRECISION, format), settled on the device once per process
enum { kFmtUnknown = 0, kFmtOk = 1, kFmtBad = 2 };
int  bhwp_fmt_verdict(const bhw_params *p, uint32_t dlog);
void bhwp_fmt_set_verdict(const bhw_params *p, uint32_t dlog, int v);
// scratch bytes a table-strategy call needs right now: the first candidate that is known to be exact, or -- while a narrower one is
// still unverified and may fall back -- the largest of those that may be tried (`capturing`: unverified formats are skipped)
uint64_t bhwp_table_scratch_bytes(const bhw_params *p, const BhwCordicCfg &c, bool tiled, uint32_t limit, bool capturing);

// What a table-strategy call over [n0, n0 + count) does with its whole periods.
struct BhwTableCall {
    bool has_period;      // the range holds at least one whole period
    bool images;          // a contiguous range of whole eighths of one window: the tile kernel over those images
    bool tiled;           // tile kernel (else quadrant fold / run-length / general gather)
    uint32_t img_mask, n0mod;
};
BhwTableCall bhwp_table_call(const bhw_params *p, const BhwCordicCfg &c, const BhwWinCfg &w, uint64_t n0, uint64_t count, bool apply);

// ownership parts
int  bhwp_part_checks(const bhw_params *p, uint32_t part, uint32_t n_parts);
// strategy of bhw_generate_part_device: true = fused kernel over the part's runs, false = full table + the part's tiles; rc != 0: neither applies
bool bhwp_part_fused(const bhw_params *p, const BhwCordicCfg &c, const BhwFoldRun *runs, int n_runs, uint32_t tile_count, uint32_t requested, int *rc);
