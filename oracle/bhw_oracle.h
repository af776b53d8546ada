/*
 * bhw_oracle.h -- CPU restatement of the reference's fixed-point window path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (the C-ABI in
 * include/bhw.h) never links or calls it and has no CPU fallback.
 *
 * Every function cites the reference file:line (relative to the upstream
 * repository hukenovs/blackman_harris_win) whose arithmetic it restates.
 *
 * Parity pin status (see DESIGN.md section "Oracle"):
 *   model A (cpp/cordic_sincos.cpp) : PINNED  -- checked bit-for-bit against
 *       oracle/_ref (the reference's own cordic() compiled from its source)
 *       and against its coe.dat output (md5 in tests/golden).
 *   model B (hls/...)               : pinned by known answers only (SURVEY
 *       App. B, produced from the unmodified HLS sources during the survey) +
 *       the reference's own tolerance tests; ap_int.h is absent here, so the
 *       HLS sources are unbuildable in this image.
 *   model C (src/ .vhd files), Taylor    : PARITY UNPINNED -- no VHDL simulator;
 *       restated from source, checked only against the ideal-window tolerance.
 */
#ifndef BHW_ORACLE_H
#define BHW_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { BHWO_MODEL_HLS = 0, BHWO_MODEL_CPP = 1, BHWO_MODEL_VHDL = 2,
       BHWO_MODEL_DDS48 = 3,  /* src/cordic_dds48.vhd       (sin/cos only: no window instantiates it) */
       BHWO_MODEL_SCALED = 4  /* src/cordic_dds_scaled.vhd  (sin/cos only)                            */ };
enum { BHWO_COMBINE_HLS = 0, BHWO_COMBINE_VHDL = 1 };
enum { BHWO_SIN_CORDIC = 0, BHWO_SIN_TAYLOR = 1, BHWO_SIN_TAYLOR_ALL = 2 /* extension: see include/bhw.h */ };

typedef struct {
    uint32_t model;      /* CORDIC bit-model: HLS (B), CPP (A), VHDL (C)     */
    uint32_t combine;    /* cosine-sum rule: HLS or VHDL                      */
    uint32_t sin_type;   /* CORDIC or TAYLOR feeder                           */
    uint32_t n_terms;    /* 2,3,4,5,7                                         */
    uint32_t phi_width;  /* N = 2^phi_width                                   */
    uint32_t dat_width;  /* output bits                                       */
    uint32_t precision;  /* model C only (cordic_dds generic PRECISION)       */
    uint32_t lut_size;   /* Taylor only (LUT_SIZE)                            */
    int32_t  aa[7];      /* integer weights AA0..AA6                          */
} bhwo_params;

/* 48-entry arctangent tables and gains, derived from closed form in binary128. */
const int64_t *bhwo_table_t2(void);   /* round(atan(2^-i) * 2^47 / pi) */
const int64_t *bhwo_table_t4(void);   /* round(atan(2^-i) * 2^48 / pi), [47] forced to 0 as in the reference */
int64_t bhwo_gain46(void);            /* round(2^46 / K) */
int64_t bhwo_gain47(void);            /* round(2^47 / K) */

/* One CORDIC evaluation; returns 0 or -1 on bad widths.  wrap_events (may be
 * NULL) is incremented whenever a typed store actually changed a value. */
int bhwo_cordic(uint32_t model, uint32_t phi_width, uint32_t dat_width, uint32_t precision,
                uint64_t theta, int32_t *out_cos, int32_t *out_sin, uint64_t *wrap_events);

/* Vectoring CORDIC src/cordic_atan2.vhd: x, y are INPUT_WIDTH-bit two's-complement words, *out_phi the ANGLE_WIDTH-bit
 * word PHI_DT (sign-extended).  PARITY UNPINNED like every VHDL-only item. */
int bhwo_atan2(uint32_t precision, uint32_t input_width, uint32_t angle_width, int64_t x, int64_t y, int32_t *out_phi);

/* Taylor feeder (src/taylor_sincos.vhd + src/tay1_order.vhd). */
int bhwo_taylor(uint32_t phi_width, uint32_t dat_width, uint32_t lut_size,
                uint64_t cnt, int32_t *out_cos, int32_t *out_sin);

/* a_k derivation of the HLS model (round half away from zero). */
int bhwo_coeffs_from_float(uint32_t win_type, uint32_t dat_width, const double *a, int32_t aa[7]);

/* Window coefficients w[n0 .. n0+count). */
int bhwo_generate(const bhwo_params *p, uint64_t n0, uint64_t count, int32_t *out);

/* sin/cos sweep theta0 .. theta0+count (phase wraps mod 2^phi_width). */
int bhwo_sincos(const bhwo_params *p, uint64_t theta0, uint64_t count, int32_t *out_sin, int32_t *out_cos);

/* FNV-1a-64 over a little-endian int32 vector (the checksum used in tests/golden). */
uint64_t bhwo_fnv1a64(const int32_t *v, uint64_t count, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
