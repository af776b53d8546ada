/*
 * bhw.h -- C ABI of the MI355X fixed-point window-coefficient generator.
 *
 * Drop-in boundary for the reference's coefficient source.  The reference
 * (hukenovs/blackman_harris_win) has no FFI layer; what a consumer binds is
 *   - the `win_selector` entity, src/win_selector.vhd:60-87
 *       generics PHI_WIDTH, DAT_WIDTH, WIN_TYPE, SIN_TYPE, LUT_SIZE, XSERIES
 *       ports    AA0..AA6 (integer weights), ENABLE, DT_WIN, DT_VLD
 *   - the HLS top   void win_function(char win_type, phi_t i, win_t *out)
 *                                             hls/windows/win_function.h:65-69
 *   - the CORDIC    void cordic(phi_t, win_t *cos, win_t *sin)
 *                                             hls/windows/win_function.cpp:47-51
 *                   void cordic(int theta, long long *lut, int *s, int *c)
 *                                             cpp/cordic_sincos.cpp:10
 * Each entry point below names the reference interface it replaces.
 *
 * Plain C types only: pointers, sizes, PODs.  No exceptions cross the ABI.
 * All compute runs in hand-written HIP kernels on the selected device; there is
 * no CPU fallback -- without a usable HIP device every compute entry point
 * returns BHW_ERR_HIP.
 */
#ifndef BHW_H
#define BHW_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BHW_ABI_VERSION 4u   /* 3: bhw_coeffs_preset, bhw_gather_parts_device; 4: bhw_workspace_bytes_ex (nothing removed or changed) */

/* CORDIC bit-model (the reference holds three that are not bit-identical). */
enum {
    BHW_MODEL_HLS  = 0,  /* hls/windows/win_function.cpp:47-156 (= hls/cordic/cordic.cpp)   */
    BHW_MODEL_CPP  = 1,  /* cpp/cordic_sincos.cpp:10-92                                      */
    BHW_MODEL_VHDL = 2,  /* src/cordic_dds.vhd:94-249                                        */
    /* The two variant generators of the repository: accepted by bhw_sincos_* only (no window entity instantiates them,
     * so window generation with them is BHW_ERR_UNSUPPORTED).  As written upstream they deliver DT_COS = +cos, DT_SIN = -sin
     * at amplitude 2^(DATA_WIDTH-2) (x/y update sense of src/cordic_dds48.vhd:234-242). */
    BHW_MODEL_DDS48  = 3, /* src/cordic_dds48.vhd:94-260: 48-bit data path, quadrant folded into the start vector */
    BHW_MODEL_SCALED = 4  /* src/cordic_dds_scaled.vhd:98-286: data path SEL_SIZE(DATA_WIDTH-8) bits (:102-107)  */
};
/* Cosine-sum rule. */
enum {
    BHW_COMBINE_HLS  = 0, /* truncating shift, no rounding: hls/windows/win_function.cpp:168-377 */
    BHW_COMBINE_VHDL = 1  /* per-product round + final round: src/bh_win_7term.vhd:353-438 etc.  */
};
/* SIN_TYPE generic of win_selector (src/win_selector.vhd:66).  The reference wires the Taylor source only into the
 * 2- and 3-term windows (src/win_selector.vhd:93-135; the BH4/5/7 entities take no SIN_TYPE, :137-199), so
 * BHW_SIN_TAYLOR with more terms is BHW_ERR_UNSUPPORTED here and the win_selector mirrors fall back to CORDIC exactly as
 * the reference's selector does.  BHW_SIN_TAYLOR_ALL is this library's extension: harmonic
 * k = m * 2^v (m odd) is read from a taylor_sincos generator of PHASE_WIDTH - v at phase (m * n) mod 2^(PHASE_WIDTH - v)
 * -- the rule bh_win_3term.vhd:221-226 applies to its 2nd harmonic, continued to k = 3..6.  For 2 and 3 terms it is
 * identical to BHW_SIN_TAYLOR; for more terms no reference output exists (the oracle defines it). */
enum { BHW_SIN_CORDIC = 0, BHW_SIN_TAYLOR = 1, BHW_SIN_TAYLOR_ALL = 2 };
/* win_type codes of win_function() (hls/windows/win_function.cpp:391-420). */
enum { BHW_WIN_HAMMING = 1, BHW_WIN_HANN = 2, BHW_WIN_BH3 = 3, BHW_WIN_BH4 = 4, BHW_WIN_BH5 = 5, BHW_WIN_BH7 = 7 };
/* Execution strategy (results are bit-identical across strategies). */
enum {
    BHW_ALGO_AUTO   = 0,
    BHW_ALGO_DIRECT = 1, /* one lane per coefficient, K-1 CORDIC chains per lane              */
    BHW_ALGO_TABLE  = 2, /* first-quadrant CORDIC table built once per call, then gather-combine */
    BHW_ALGO_FUSED  = 3  /* whole periods in one launch, no table: one lane per eight coefficients (quadrant + half-period
                            fold), shared rotation prefixes per wave; ragged ends take BHW_ALGO_DIRECT.  Falls back to
                            BHW_ALGO_TABLE where it does not apply (34-bit+ CORDIC state, phi_width < 9)            */
};

enum {
    BHW_OK              = 0,
    BHW_ERR_BADARG      = -1,
    BHW_ERR_UNSUPPORTED = -2,
    BHW_ERR_HIP         = -3,
    BHW_ERR_WORKSPACE   = -4
};

/* The win_selector parameter surface as one POD (src/win_selector.vhd:61-81). */
typedef struct bhw_params {
    uint32_t struct_size;  /* sizeof(bhw_params), for ABI evolution                       */
    uint32_t model;        /* BHW_MODEL_*                                                 */
    uint32_t combine;      /* BHW_COMBINE_*                                               */
    uint32_t sin_type;     /* BHW_SIN_*                      SIN_TYPE                     */
    uint32_t win_type;     /* BHW_WIN_* (informational; n_terms + aa define the window)   */
    uint32_t n_terms;      /* 2,3,4,5,7                      WIN_TYPE                     */
    uint32_t phi_width;    /* 4..30, N = 2^phi_width         PHI_WIDTH (the reference documents up to 26,
                              README.md:2; 27..30 are the same arithmetic on a longer counter)          */
    uint32_t dat_width;    /* 8..32                          DAT_WIDTH                    */
    uint32_t precision;    /* model VHDL only, 1..7          cordic_dds PRECISION         */
    uint32_t lut_size;     /* Taylor only                    LUT_SIZE                     */
    int32_t  aa[7];        /* AA0..AA6, caller-scaled integer weights                     */
} bhw_params;

/* Storage format of the first-quadrant (c, s) table of BHW_ALGO_TABLE (results are bit-identical across formats;
 * a format that does not apply to a configuration falls back to the next wider one). */
enum {
    BHW_TABLE_BEST     = 0, /* narrowest format that is exact for the configuration                 */
    BHW_TABLE_PLAIN    = 1, /* int2 (c, s) per entry, 8 bytes                                       */
    BHW_TABLE_DELTA16  = 2, /* at most: 4 bytes per entry (int16 differences to a 64-entry block head) */
    BHW_TABLE_RESIDUAL = 3, /* at most: 2 bytes per entry against a linear predictor (8-bit fields) */
    BHW_TABLE_NIBBLE   = 4, /* at most: 1 byte per entry, the same predictor with 4-bit fields (what BEST tries first) */
    BHW_TABLE_NIBBLE_ESC = 5 /* at most: the same one-byte entries with a marker for the rare deviation beyond the fields, listed exactly
                                beside the table (second choice of BEST: models whose CORDIC noise is wider, cpp / VHDL at 32 bits)  */
};

/* Optional execution controls for the *_ex entry points. */
typedef struct bhw_exec {
    uint32_t struct_size;     /* sizeof(bhw_exec) (the 32-byte ABI-1 layout without table_format is accepted) */
    uint32_t algo;            /* BHW_ALGO_*                                                */
    void    *workspace;       /* device scratch (NULL: library-owned scratch of this stream) */
    uint64_t workspace_bytes;
    void    *event_after_build; /* optional hipEvent_t recorded on the stream between the table build and the
                                   combine pass (per-kernel timing for profilers); NULL = none      */
    uint32_t table_format;    /* BHW_TABLE_* (diagnostics / A-B runs; 0 = best)            */
    uint32_t reserved;        /* must be 0                                                 */
} bhw_exec;

uint32_t    bhw_abi_version(void);
const char *bhw_strerror(int code);
const char *bhw_last_error(void); /* thread-local detail of the last failure */

/* Defaults: model HLS, combine HLS, CORDIC source, precision 1, lut_size 9, built-in a_k
 * (hls/windows/win_function.cpp:173-355 constants and scaling). */
int bhw_params_init(bhw_params *p, uint32_t win_type, uint32_t phi_width, uint32_t dat_width);
int bhw_params_validate(const bhw_params *p);

/* a_k = round(coe_k * (2^(W-s)-1)), s = 1 (2/3/4-term) or 2 (5/7-term): the HLS derivation
 * (hls/windows/win_function.cpp:176-177,210-212,258-261,312-316,349-355).  a == NULL selects
 * the built-in constants of `win_type`.  Host arithmetic only (doubles -> 7 integers). */
int bhw_coeffs_from_float(uint32_t win_type, uint32_t dat_width, const double *a, int32_t aa[7]);

/* The named coefficient sets the reference lists beside its built-in ones (comments in hls/windows/win_function.cpp:241-250 and
 * :292-303, README.md:30-51): the float weights a[0..6] (unused terms 0), the window type that takes them, and -- when aa != NULL
 * -- the integer weights of the HLS derivation above for dat_width (bhw_coeffs_from_float(*win_type, dat_width, a, aa)).
 * Two upstream slips are not reproduced: the Nuttall a2 is the published 0.144232 (win_function.cpp:244 prints 0.144323, with
 * which the weights no longer sum to one), and flat-top (2) lists its fifth weight as a second "a3" (:302-303). */
enum {
    BHW_PRESET_NUTTALL          = 1,  /* 4-term: 0.355768, 0.487396, 0.144232, 0.012604                 -93 dB (README.md:35) */
    BHW_PRESET_BLACKMAN_NUTTALL = 2,  /* 4-term: 0.3635819, 0.4891775, 0.1365995, 0.0106411             -98 dB (README.md:37) */
    BHW_PRESET_FLATTOP_1        = 3,  /* 5-term: 0.25, 0.4925, 0.3225, 0.097, 0.0075  (win_function.cpp:292-297)             */
    BHW_PRESET_FLATTOP_2        = 4,  /* 5-term: 0.21557895, 0.41663158, 0.277263158, 0.083578947, 0.006947368 (:298-303)   */
    BHW_PRESET_BH7_README       = 5,  /* 7-term set of README.md:45-51 ("up to 180 dB")                                      */
    BHW_PRESET_BLACKMAN         = 6,  /* 3-term: 0.42, 0.5, 0.08                                         -58 dB (README.md:32) */
    BHW_PRESET_BH3              = 7   /* 3-term Blackman-Harris: 0.42323, 0.49755, 0.07922               -71 dB (README.md:33) */
};
int bhw_coeffs_preset(uint32_t preset, uint32_t dat_width, uint32_t *win_type, double a[7], int32_t aa[7]);

/* The 48-entry arctangent ROMs and gains the kernels use (which: 0 = T2 of the cpp model,
 * 1 = T4 of the HLS/VHDL models); gains[0] = G46, gains[1] = G47. */
int bhw_constant_tables(uint32_t which, int64_t table[48], int64_t gains[2]);

/* Replaces: ENABLE held high for `count` clocks on win_selector (src/win_selector.vhd:83-86)
 * after the phase counter reached n0; equivalently `for i in n0..n0+count: win_function(sel, i, &w)`
 * (hls/windows/window_test.cpp:93,193).  Writes count sign-extended int32 coefficients to d_out
 * (device memory of `device`), asynchronously on `hip_stream` (hipStream_t; NULL = default
 * stream).  n wraps modulo 2^phi_width like the hardware counter (src/bh_win_7term.vhd:92-97). */
int bhw_generate_device(const bhw_params *p, int device, void *hip_stream,
                        uint64_t n0, uint64_t count, int32_t *d_out);
int bhw_generate_device_ex(const bhw_params *p, int device, void *hip_stream,
                           uint64_t n0, uint64_t count, int32_t *d_out, const bhw_exec *ex);
/* Device scratch that is enough for the given call with `algo` whatever table format it ends up using (0 for the direct and
 * fused strategies): 8 bytes per first-quadrant table entry.  An upper bound that never changes for a configuration. */
uint64_t bhw_workspace_bytes(const bhw_params *p, uint64_t n0, uint64_t count, uint32_t algo);
/* The same for the call as `ex` describes it (algo, table_format), counting the table format(s) the call would use right now:
 * after bhw_prepare_device (or a first call) has settled the packed formats of the configuration this is the size of the one
 * format in use -- 16.5 MiB instead of 128 MiB for a 2^26-point window at 32 bits -- and it never grows afterwards.  A
 * workspace of at least this size is accepted by bhw_generate_device_ex / bhw_generate_part_device (bhw_apply_device takes no
 * bhw_exec: it always uses the library-owned scratch).  The library-owned scratch of a stream is sized by the same rule:
 * bhw_prepare_device leaves it at this size; a stream that was never prepared holds the 8-bytes-per-entry bound from its first
 * call (which has to try the formats) until the next call of that configuration, which gives the excess back once. */
uint64_t bhw_workspace_bytes_ex(const bhw_params *p, uint64_t n0, uint64_t count, const bhw_exec *ex);

/* What bhw_generate_device_ex(p, ..., n0, count, ..., ex) would launch right now, as one line of text into buf (NUL-terminated,
 * truncated to len): strategy, table format and the kernel names a profiler will show.  The table format of a configuration
 * is settled on its first use (or by bhw_prepare_device); before that the line says "unverified".  Host arithmetic only. */
int bhw_describe_plan(const bhw_params *p, uint64_t n0, uint64_t count, const bhw_exec *ex, char *buf, uint64_t len);

/* Does every lazy step of later calls with `p` on (device, hip_stream) now: uploads the Taylor quarter-wave ROM
 * (taylor_sincos.vhd:91-111 builds it at elaboration), allocates the library-owned table scratch of this stream, and verifies
 * once, on the device, that the packed table formats are exact for this (model, phi_width, dat_width, precision) -- a property
 * of the configuration, not of the weights.  Synchronous.  After it, bhw_generate_* / bhw_apply_* / bhw_sincos_* calls with
 * these widths on this stream neither allocate nor synchronise, so they can be captured into a HIP graph -- whole periods,
 * partial ranges and explicit bhw_exec.algo alike (the scratch is reserved also for configurations whose whole periods take the
 * table-free fused kernel).  The verdict of every packed format is settled, so an explicit bhw_exec.table_format never meets
 * an open one; the scratch is sized for table_format BEST (the narrowest exact format).  Two exceptions, both answered by
 * passing bhw_exec.workspace (bhw_workspace_bytes_ex bytes) inside a capture: a range WITHOUT a whole period of a window whose
 * plain table exceeds 64 MiB (phi_width >= 26 at z_shr = 0), and an explicit table_format wider than the one BEST resolves to --
 * either would have to grow the library scratch, which a capturing stream refuses (BHW_ERR_HIP).
 * Without prepare the first call does the same work inline (one synchronisation); during stream capture an unprepared Taylor
 * call fails with BHW_ERR_HIP and an unprepared table call uses the plain table format. */
int bhw_prepare_device(const bhw_params *p, int device, void *hip_stream);

/* Interleaved ownership of ONE window over several devices (every coefficient is an independent function of its
 * index, src/bh_win_7term.vhd:176-197 | hls/windows/win_function.cpp:361-375).  Contiguous index shards -- n0 = g * N / G with
 * bhw_generate_device -- cannot share CORDIC work between the quadrant images of a coefficient; an interleaved part can: part
 * `part` of `n_parts` owns a set of lanes r of the ring [0, N/8) together with the eight coefficients r + h*N/8 + j*N/4 of each,
 * so one first-quadrant CORDIC result still serves up to eight coefficients.  The ownership is a deterministic function of
 * (phi_width, dat_width, model, n_terms, n_parts): bhw_part_segments lists it as sorted contiguous index segments (at most
 * 256); the parts of one window cover [0, N) -- neighbouring parts may both own a few hundred coefficients at the seams of
 * the internal tiling, with identical values.
 * bhw_generate_part_device writes exactly the owned coefficients into d_window, the base of a full-length (2^phi_width)
 * int32 buffer on `device`; all other elements are left untouched.  CORDIC source only.  No collective is involved: a
 * consumer that wants the whole window on one device copies the segments (hipMemcpyPeerAsync). */
typedef struct bhw_segment {
    uint64_t n0;     /* first coefficient index */
    uint64_t count;  /* coefficients            */
} bhw_segment;
int bhw_part_segments(const bhw_params *p, uint32_t part, uint32_t n_parts, bhw_segment *segs, uint32_t capacity, uint32_t *n_segs);
int bhw_generate_part_device(const bhw_params *p, int device, void *hip_stream, uint32_t part, uint32_t n_parts,
                             int32_t *d_window, const bhw_exec *ex);

/* One window on one device from its `n_parts` interleaved ownership parts (optional, outside the metric).  Part g
 * was produced by bhw_generate_part_device(p, src_devices[g], ..., g, n_parts, d_windows[g]) into a full-length buffer on
 * src_devices[g]; every segment part g owns (bhw_part_segments) is copied into d_dst, a full-length buffer on dst_device, with
 * hipMemcpyPeerAsync on dst_stream (xGMI between devices; a plain device copy where source and destination coincide; nothing
 * is copied when d_windows[g] == d_dst).  The caller orders dst_stream after the producing streams (an event per part, or a
 * device synchronise); no collective and no host staging are involved. */
int bhw_gather_parts_device(const bhw_params *p, uint32_t n_parts, const int *src_devices, const int32_t *const *d_windows,
                            int dst_device, void *dst_stream, int32_t *d_dst);

/* Fused apply (the step after the path in every consumer -- the window multiplies the samples in
 * front of an FFT).  d_y[i] = (d_x[i] * w[n0+i]) >> shift with the exact 64-bit product (as int_multNxN_dsp48,
 * src/int_multNxN_dsp48.vhd:102), floor shift (0..62) and the low 32 bits stored; the coefficients are generated
 * on the fly and never written to HBM.  d_y must not overlap d_x. */
int bhw_apply_device(const bhw_params *p, int device, void *hip_stream, uint64_t n0, uint64_t count,
                     const int32_t *d_x, int32_t *d_y, uint32_t shift);

/* `frames` back-to-back periods of the coefficient stream (the streaming-frame workload:
 * ENABLE held for frames * 2^phi_width clocks).  One period is computed, then replicated by a
 * store-only kernel: d_out holds frames * 2^phi_width int32. */
int bhw_generate_batched_device(const bhw_params *p, int device, void *hip_stream,
                                uint32_t frames, int32_t *d_out);

/* Replaces cordic() alone: cpp/cordic_sincos.cpp:10 (model CPP), hls/cordic/cordic.cpp:45
 * (model HLS), the cordic_dds entity src/cordic_dds.vhd:77-92 (model VHDL), or taylor_sincos
 * src/taylor_sincos.vhd:64-80 (sin_type TAYLOR), cordic_dds48 src/cordic_dds48.vhd:98-112 (model DDS48) or
 * cordic_dds_scaled src/cordic_dds_scaled.vhd:81-96 (model SCALED).  Either output pointer may be NULL. */
int bhw_sincos_device(const bhw_params *p, int device, void *hip_stream,
                      uint64_t theta0, uint64_t count, int32_t *d_sin, int32_t *d_cos);

/* Convenience for host consumers (file writers, testbenches): runs bhw_generate_device into
 * library scratch on `device`, then copies the result to host memory.  Synchronous. */
int bhw_generate_to_host(const bhw_params *p, int device, uint64_t n0, uint64_t count, int32_t *h_out);
int bhw_sincos_to_host(const bhw_params *p, int device, uint64_t theta0, uint64_t count,
                       int32_t *h_sin, int32_t *h_cos);

/* Replaces the cordic_atan2 entity (src/cordic_atan2.vhd:64-76): vectoring CORDIC, one angle per (x, y) pair.
 * VEC_DX / VEC_DY are INPUT_WIDTH-bit two's-complement words carried in int32; PHI_DT is the ANGLE_WIDTH-bit output word,
 * sign-extended to int32 (full circle = 2^angle_width; quadrant fix-ups exactly as written at :126-128,:207-213).
 * The entity reads bits 0..ANGLE_WIDTH-2 of its inputs (:142-145), so input_width >= angle_width - 1 is required
 * (upstream's default generics 20/24 do not elaborate).  d_x, d_y, d_phi: `count` int32 each; d_phi may alias neither. */
typedef struct bhw_atan2_params {
    uint32_t struct_size;   /* sizeof(bhw_atan2_params)                                  */
    uint32_t precision;     /* 1..7              PRECISION                               */
    uint32_t input_width;   /* <= 32             INPUT_WIDTH                             */
    uint32_t angle_width;   /* 4..32             ANGLE_WIDTH                             */
} bhw_atan2_params;
int bhw_atan2_device(const bhw_atan2_params *p, int device, void *hip_stream, uint64_t count,
                     const int32_t *d_x, const int32_t *d_y, int32_t *d_phi);
int bhw_atan2_to_host(const bhw_atan2_params *p, int device, uint64_t count,
                      const int32_t *h_x, const int32_t *h_y, int32_t *h_phi);

/* Threading: every entry point may be called from any host thread.  Calls that use the library-owned scratch of one
 * (device, stream) are serialised against each other for the duration of their launches (the table is rebuilt per call);
 * callers that pass their own bhw_exec.workspace must not share one workspace between concurrent calls.  The calling
 * thread's current HIP device is restored before every entry point returns, and no entry point reads or clears the
 * thread's hipGetLastError() state.
 *
 * Releases the library-owned per-device scratch. */
int bhw_release_device(int device);

#ifdef __cplusplus
}
#endif
#endif /* BHW_H */
