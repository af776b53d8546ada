// bhw_internal.h -- structures shared by the host-side ABI (bhw_api.cpp) and the HIP kernels.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/bhw.h"

// One CORDIC bit-model resolved to plain numbers for a (model, PW, W, PRECISION) tuple.
//   model HLS : hls/windows/win_function.cpp:74-154   
//   model CPP : cpp/cordic_sincos.cpp:12-90           
//   model VHDL: src/cordic_dds.vhd:97-249             
// All three run   x' = x -/+ (y >> k), y' = y +/- (x >> k), z' = z -/+ lut[k]   and differ only in
// the constants below.
struct BhwCordicCfg {
    int64_t  lut[32];     // rescaled arctangent ROM; entries >= n_lut are 0
    int64_t  x0;          // gain-compensated start value
    uint32_t phi_width;   // PW
    uint32_t dat_width;   // W
    uint32_t n_iter;      // W (HLS, CPP) or W-1 (VHDL)
    uint32_t z_shr;       // z0 = (t >> z_shr) << z_shl,  t = theta mod 2^(PW-2)
    uint32_t z_shl;
    uint32_t out_shr;     // 2 (HLS, CPP) or PRECISION (VHDL)
    uint32_t ones_neg;    // 1: quadrant map negates with ~v (CPP); 0: -v
    uint32_t wide;        // 1: state needs more than 32 bits
    uint32_t tab_split;   // table layout: 0 natural index u; 1 split by residue class (u%4==0 | u%4==2 | u odd)
    uint32_t tab_dlog;    // table format (tab_load() in bhw_device.h): 0 plain int2 (c, s) entries; 6 "delta16" -- one dword of
                          // two int16 differences to the first entry of the 64-entry block, block heads as int2 records at
                          // tab_coarse; 7..9 "residual" -- two bytes per entry against a linear predictor, int4 {c, s, dc, ds}
                          // records every 2^tab_dlog entries at tab_coarse
                          // 23..25 (16 + d) "nibble": the residual format in 4-bit fields; 55..57 (48 + d) "nibble + escapes": the same with a
                          // reserved marker value for the rare entry whose deviation does not fit, listed exactly at tab_esc
    const void *tab_coarse;
    uint32_t *tab_check;  // build pass, packed formats: device word set to 1 when an entry does not fit its field (NULL: no check)
    const void *tab_esc;  // nibble + escapes: per build workgroup a hash table of kEscSlots x { entry index or -1, c, s, - }
    uint32_t esc_wg_log;  // ... log2 of the table entries one build workgroup owns (its own range; the images mirror it)
    uint32_t pad_esc;
};

// Cosine-sum stage.
struct BhwWinCfg {
    int32_t  aa[8];
    uint32_t n_terms;
    uint32_t combine;     // BHW_COMBINE_*
    // fused apply: when apply_x != NULL the kernels store (x[i] * w[i]) >> apply_shift instead of w[i]
    uint32_t apply_shift;
    uint32_t pad;
    const int32_t *apply_x;
};

// Taylor feeder (src/taylor_sincos.vhd + src/tay1_order.vhd).
struct BhwTaylorCfg {
    const int32_t *rom;   // device pointer: 2^lut_size (sin, cos) pairs, interleaved
    uint32_t phi_width, dat_width, lut_size;
    uint32_t mode;        // 0: PW-L < 2, 1: PW-L == 2, 2: PW-L > 2 (1st-order correction)
    uint32_t pi_word;     // round(pi * 2^(17-STAGE))
    uint32_t xshift;      // 19 + L
    uint32_t pad[2];
};

// cordic_dds48 / cordic_dds_scaled (quadrant folded into the start vector) and cordic_atan2
struct BhwPrerotCfg {
    int64_t  lut[32];     // T2[i] >> (48 - DWPH)
    int64_t  gain;        // GAIN48 >> (48 - SIZE)
    uint32_t phi_width, dat_width, size, dwph;
};
struct BhwAtan2Cfg {
    int64_t  lut[32];     // T4[i] >> (49 - (ANGLE_WIDTH + PRECISION))
    uint32_t precision, input_width, angle_width, pad;
};

struct BhwLaunch {
    int   device;
    void *stream;
};

// kernel launchers (bhw_direct / bhw_build / bhw_combine / bhw_fused / bhw_taylor / bhw_variants .hip) -- each returns a hipError_t cast to int
int bhwk_direct(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w,
                uint64_t n0, uint64_t count, int32_t *d_out);
int bhwk_sincos(const BhwLaunch &l, const BhwCordicCfg &c, uint64_t theta0, uint64_t count,
                int32_t *d_sin, int32_t *d_cos);
int bhwk_sincos_prerot(const BhwLaunch &l, const BhwPrerotCfg &c, uint64_t theta0, uint64_t count,
                       int32_t *d_sin, int32_t *d_cos);
int bhwk_atan2(const BhwLaunch &l, const BhwAtan2Cfg &c, uint64_t count, const int32_t *d_x, const int32_t *d_y, int32_t *d_phi);
int bhwk_replicate(const BhwLaunch &l, const int32_t *d_frame, uint64_t frame_len, uint32_t frames, int32_t *d_out);
int bhwk_table_build(const BhwLaunch &l, const BhwCordicCfg &c, int32_t *d_table /* (c,s) pairs, 2^(PW-2) */);
int bhwk_table_combine(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_table,
                       uint64_t n0, uint64_t count, int32_t *d_out);
// whole-period cordic() sweep through the shared-prefix chains of the table build (bhwk_sincos picks it for periods >= 2^16)
int bhwk_sincos_sweep(const BhwLaunch &l, const BhwCordicCfg &c, uint64_t theta0, int32_t *d_sin, int32_t *d_cos);
// Predicates and shapes below that take no BhwLaunch are host arithmetic only and live in the HIP-free bhw_plan.cpp.
// the octant-mirror build kernel applies to this table (bhw_build.hip; bhwk_describe_table names the kernel)
bool bhwk_build_mirror_applies(const BhwCordicCfg &c, uint32_t entries);
// which packed table formats a configuration admits (delta16; residual cell size, 0 = not applicable)
bool bhwk_packed_ok(const BhwCordicCfg &c);
uint32_t bhwk_resid_dlog(const BhwCordicCfg &c);
// whole period [0, 2^PW) via the quadrant fold (one lane per four coefficients)
int bhwk_table_combine_fold(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_table, int32_t *d_out);
// whole period, 15-block super-tiles over the residue-split table (z_shr == 0 only)
int bhwk_table_combine_tile(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_table, int32_t *d_out);
bool bhwk_tile_applicable(const BhwCordicCfg &c, const BhwWinCfg &w);
void bhwk_describe_table(const BhwCordicCfg &c, const BhwWinCfg &w, bool tiled, bool images, char *build, char *combine, size_t len);
// tiles [tile0, tile0 + tile_count) of the tile plan only (tile_count 0: the whole ring)
int bhwk_table_combine_tile_range(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_table, int32_t *d_out,
                                  uint32_t tile0, uint32_t tile_count, uint32_t img_mask = 0xFFu, uint32_t n0mod = 0u);
bool bhwk_tile_images_applicable(const BhwCordicCfg &c, const BhwWinCfg &w, uint64_t n0, uint64_t count, uint32_t *img_mask, uint32_t *n0mod);
// Run-length kernel: whole period of a configuration that drops phase bits (z_shr > 0), over the plain natural table
bool bhwk_runlength_applicable(const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_out);
int bhwk_runlength_window(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const int32_t *d_table, int32_t *d_out);
// Fused fold kernel (no table): for every ring lane r of the runs, the eight coefficients r + h*N/8 + j*N/4 into the
// full-window buffer d_out.  At most 32 runs.
struct BhwFoldRun { uint32_t r0, r_end; };
bool bhwk_fold_direct_applicable(const BhwCordicCfg &c);
// frames > 1: `frames` identical periods back to back from d_out (whole ring, not the split form, no fused apply)
int bhwk_fold_direct(const BhwLaunch &l, const BhwCordicCfg &c, const BhwWinCfg &w, const BhwFoldRun *runs, uint32_t n_runs, int32_t *d_out, uint32_t frames = 1);
// ring lanes of interleaved-ownership part `part` of `n_parts` as runs (returns the count, <= 32); tile0 / tile_count: the same
// part as a range of the tile plan's tiles when the tile kernel applies (tile_count 0 otherwise)
int bhwk_part_runs(const BhwCordicCfg &c, const BhwWinCfg &w, uint32_t part, uint32_t n_parts, BhwFoldRun *runs, uint32_t *tile0, uint32_t *tile_count);
int bhwk_taylor_window(const BhwLaunch &l, const BhwTaylorCfg &t, const BhwWinCfg &w,
                       uint64_t n0, uint64_t count, int32_t *d_out);
// one whole period [0, 2^PW) with the quadrant fold (PW >= 5)
int bhwk_taylor_window_fold(const BhwLaunch &l, const BhwTaylorCfg &t, const BhwWinCfg &w, int32_t *d_out);
int bhwk_taylor_sincos(const BhwLaunch &l, const BhwTaylorCfg &t, uint64_t theta0, uint64_t count,
                       int32_t *d_sin, int32_t *d_cos);
