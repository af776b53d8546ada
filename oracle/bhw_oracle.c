/*
 * bhw_oracle.c -- CPU restatement of the reference's integer window path.
 *
 * TEST INFRASTRUCTURE ONLY (see bhw_oracle.h).  Plain C, int64/__int128
 * arithmetic, one function per reference item; each cites the upstream
 * file:line it follows.  Written from the arithmetic description, not from
 * the reference text: typed stores are explicit wrap() calls, constant tables
 * are recomputed from closed form in binary128.
 */
#include "bhw_oracle.h"

#include <math.h>
#include <quadmath.h>
#include <stddef.h>
#include <string.h>

/* ------------------------------------------------------------------ helpers */

/* two's-complement wrap of v to `bits` bits (what a store into ap_int<bits> or
 * a std_logic_vector(bits-1 downto 0) does). */
static int64_t wrap(int64_t v, unsigned bits, uint64_t *events)
{
    if (bits >= 64) return v;
    uint64_t m = (1ull << bits) - 1ull;
    uint64_t u = (uint64_t)v & m;
    int64_t r = (u >> (bits - 1)) ? (int64_t)(u | ~m) : (int64_t)u;
    if (events && r != v) ++*events;
    return r;
}

/* arithmetic shift right = floor division by 2^k, also for k >= 63 */
static int64_t asr(int64_t v, unsigned k)
{
    if (k >= 63) return v < 0 ? -1 : 0;
    return v >> k;
}

/* ------------------------------------------------------- constants (A.1) */
/*
 * T2[i] = round(atan(2^-i) * 2^47 / pi)   -- the 48 literals of cpp/cordic_sincos.cpp:97-110
 * T4[i] = round(atan(2^-i) * 2^48 / pi)   -- hls/windows/win_function.cpp:59-72,
 *                                            src/cordic_dds.vhd:104-117 (their [47] is 0, formula gives 1)
 * G46   = round(2^46 / K), G47 = round(2^47 / K), K = prod sqrt(1 + 2^-2i)
 *                                         -- cpp/cordic_sincos.cpp:21, src/cordic_dds.vhd:97
 */
static int64_t g_t2[48], g_t4[48], g_g46, g_g47;
static int g_tables_ready;

static void build_tables(void)
{
    if (g_tables_ready) return;
    const __float128 pi = M_PIq;
    __float128 k = 1.0Q;
    for (int i = 0; i < 48; ++i) {
        __float128 a = atanq(ldexpq(1.0Q, -i)) / pi;
        g_t2[i] = (int64_t)floorq(ldexpq(a, 47) + 0.5Q);
        g_t4[i] = (int64_t)floorq(ldexpq(a, 48) + 0.5Q);
    }
    g_t4[47] = 0; /* reference literal; never reached for dat_width <= 46 */
    for (int i = 0; i < 120; ++i) k *= sqrtq(1.0Q + ldexpq(1.0Q, -2 * i));
    g_g46 = (int64_t)floorq(ldexpq(1.0Q, 46) / k + 0.5Q);
    g_g47 = (int64_t)floorq(ldexpq(1.0Q, 47) / k + 0.5Q);
    g_tables_ready = 1;
}

const int64_t *bhwo_table_t2(void) { build_tables(); return g_t2; }
const int64_t *bhwo_table_t4(void) { build_tables(); return g_t4; }
int64_t bhwo_gain46(void) { build_tables(); return g_g46; }
int64_t bhwo_gain47(void) { build_tables(); return g_g47; }

/* ------------------------------------------------------------ CORDIC models */

/* Model B: hls/windows/win_function.cpp:47-156 (= hls/cordic/cordic.cpp:45-153). */
static void cordic_hls(unsigned PW, unsigned W, uint64_t theta, int64_t *oc, int64_t *os, uint64_t *ev)
{
    const unsigned DW = W + 2;                       /* dat_t = ap_int<NWIDTH+2>, win_function.h:61 */
    int64_t lut[64];
    for (unsigned i = 0; i + 1 < W; ++i)             /* :77-80  lut_table[i] >> (48-NWIDTH-2+1) */
        lut[i] = wrap(g_t4[i] >> (47 - W), DW, ev);
    int64_t x = wrap(g_g46 >> (46 - W), DW, ev);     /* :83 */
    int64_t y = 0;
    unsigned q = (unsigned)((theta >> (PW - 2)) & 3u);            /* :86 */
    /* :88  phi_t is signed, so the AND keeps sign-extension bits above PW; they are
     * shifted out below whenever PW <= W+2 (the only widths accepted). */
    int64_t phi = wrap((int64_t)theta, PW, NULL);
    int64_t t = wrap(phi & ~((int64_t)3 << (PW - 2)), DW, NULL);
    int64_t z;
    if (PW - 1 < W) z = wrap((int64_t)((uint64_t)t << (W - PW + 2)), DW, NULL);     /* :91-93 */
    else            z = wrap((int64_t)((uint64_t)asr(t, PW - W) << 2), DW, NULL);   /* :94-96 */
    for (unsigned k = 0; k < W; ++k) {               /* :110-125 */
        int64_t xs = asr(x, k), ys = asr(y, k);
        if (z < 0) {
            x = wrap(x + ys, DW, ev); y = wrap(y - xs, DW, ev);
            if (k + 1 < W) z = wrap(z + lut[k], DW, ev);   /* z[W] is never read; lut[W-1] is out of bounds upstream */
        } else {
            x = wrap(x - ys, DW, ev); y = wrap(y + xs, DW, ev);
            if (k + 1 < W) z = wrap(z - lut[k], DW, ev);
        }
    }
    int64_t c = asr(x, 2), s = asr(y, 2);            /* :128-129 */
    int64_t dc, ds;
    switch (q) {                                     /* :135-150  ~v + 1 in dat_t */
    case 0: ds = s; dc = c; break;
    case 1: ds = c; dc = wrap(-s, DW, ev); break;
    case 2: ds = wrap(-s, DW, ev); dc = wrap(-c, DW, ev); break;
    default: ds = wrap(-c, DW, ev); dc = s; break;
    }
    *oc = wrap(dc, W, ev);                           /* :153-154  store into win_t */
    *os = wrap(ds, W, ev);
}

/* Model A: cpp/cordic_sincos.cpp:10-92 (PRECISION = 1, long long state, no wrap). */
static void cordic_cpp(unsigned PW, unsigned W, uint64_t theta, int64_t *oc, int64_t *os)
{
    int64_t lut[64];
    for (unsigned i = 0; i + 1 < W; ++i) lut[i] = g_t2[i] >> (47 - W);   /* :15-18 */
    int64_t x = g_g46 >> (46 - W);                   /* :21-22 */
    int64_t y = 0;
    unsigned q = (unsigned)(theta >> (PW - 2));      /* :25 */
    int64_t t = (int64_t)(theta & ~(3ull << (PW - 2)));                  /* :27 */
    int64_t z = (PW - 1 < W) ? t << (W - PW + 1) : (t >> (PW - W)) << 1; /* :30-36 */
    for (unsigned k = 0; k < W; ++k) {               /* :49-63 */
        int64_t xs = x >> k, ys = y >> k;
        if (z < 0) { x = x + ys; y = y - xs; if (k + 1 < W) z += lut[k]; }
        else       { x = x - ys; y = y + xs; if (k + 1 < W) z -= lut[k]; }
    }
    int64_t c = x >> 2, s = y >> 2;                  /* :64-65 */
    int64_t dc, ds;
    switch (q) {                                     /* :70-86  one's complement */
    case 0: ds = s; dc = c; break;
    case 1: ds = c; dc = ~s; break;
    case 2: ds = ~s; dc = ~c; break;
    default: ds = ~c; dc = s; break;
    }
    *oc = (int32_t)dc;                               /* :89-90 */
    *os = (int32_t)ds;
}

/* Model C: src/cordic_dds.vhd:94-249. */
static void cordic_vhdl(unsigned PW, unsigned W, unsigned P, uint64_t theta, int64_t *oc, int64_t *os, uint64_t *ev)
{
    const unsigned Wi = W + P;                       /* DATA_WIDTH+PRECISION bits everywhere */
    int64_t lut[64];
    for (unsigned i = 0; i + 1 < W; ++i) lut[i] = g_t4[i] >> (49 - Wi);  /* :119-131 top Wi-1 bits, MSB 0 */
    int64_t x = g_g47 >> (49 - Wi);                  /* :97-98 */
    int64_t y = 0;
    unsigned q = (unsigned)((theta >> (PW - 2)) & 3u);                   /* :170-172 */
    int64_t t = (int64_t)(theta & ((1ull << (PW - 2)) - 1ull));          /* :179 */
    int64_t z = (PW >= W) ? (t >> (PW - W)) << P : t << (W - PW + P);    /* :159-166 */
    for (unsigned k = 0; k + 1 < W; ++k) {           /* :197-213  DATA_WIDTH-1 stages */
        int64_t xs = asr(x, k), ys = asr(y, k);      /* slice (Wi-1 downto k), sign-extended by std_logic_signed */
        if (z < 0) { x = wrap(x + ys, Wi, ev); y = wrap(y - xs, Wi, ev); z = wrap(z + lut[k], Wi, ev); }
        else       { x = wrap(x - ys, Wi, ev); y = wrap(y + xs, Wi, ev); z = wrap(z - lut[k], Wi, ev); }
    }
    int64_t c = wrap(asr(x, P), W, ev), s = wrap(asr(y, P), W, ev);      /* :218-219 */
    int64_t dc, ds;
    switch (q) {                                     /* :232-246  not(v)+1 at DATA_WIDTH bits */
    case 0: ds = s; dc = c; break;
    case 1: ds = c; dc = wrap(-s, W, ev); break;
    case 2: ds = wrap(-s, W, ev); dc = wrap(-c, W, ev); break;
    default: ds = wrap(-c, W, ev); dc = s; break;
    }
    *oc = dc; *os = ds;
}

/* Variant generators src/cordic_dds48.vhd:94-260 (SIZE = DWPH = 48) and src/cordic_dds_scaled.vhd:98-286
 * (SIZE = SEL_SIZE(DATA_WIDTH-8) :102-107, DWPH = max(SIZE, PHASE_WIDTH) :133-143).  Unlike models A-C the quadrant
 * is folded into the start vector (dds48 :170-216), the x/y update has the opposite sense (:234-242), all DATA_WIDTH
 * stages rotate (:233) and the arctangent ROM is T2 at full DWPH width (:130-138 / scaled :147-155). */
static const unsigned k_sel_size[25] = { 15, 15, 15, 18, 21, 22, 23, 26, 30, 31, 32, 33,
                                         38, 38, 38, 42, 42, 45, 47, 47, 47, 48, 48, 48, 48 };

static void cordic_prerot(unsigned PW, unsigned W, unsigned SIZE, unsigned DWPH, uint64_t theta,
                          int64_t *oc, int64_t *os, uint64_t *ev)
{
    const int64_t gain = g_g46 >> (48 - SIZE);       /* GAIN48(47 downto 48-SIZE): dds48 :113, scaled :112-113 */
    const unsigned q = (unsigned)((theta >> (PW - 2)) & 3u);             /* dds48 :167 */
    const uint64_t low = theta & ((1ull << (PW - 2)) - 1ull);
    uint64_t t;                                      /* init_t: dds48 :169-186 */
    int64_t x, y;                                    /* init_x / init_y: :191-216 */
    switch (q) {
    case 1:  t = low;                          x = 0;    y = wrap(-gain, SIZE, ev); break;   /* "00" & low, y = not(G)+1 */
    case 2:  t = (3ull << (PW - 2)) | low;     x = 0;    y = gain; break;                    /* "11" & low */
    default: t = theta;                        x = gain; y = 0; break;
    }
    /* init_z: phase at the MSBs of the DWPH-bit word (dds48 :163-164, scaled :180-186) */
    int64_t z = wrap((int64_t)(t << (DWPH - PW)), DWPH, NULL);
    for (unsigned ii = 0; ii < W; ++ii) {            /* dds48 :233-243 */
        const int64_t xs = asr(x, ii), ys = asr(y, ii);
        if (z >= 0) { x = wrap(x + ys, SIZE, ev); y = wrap(y - xs, SIZE, ev); }
        else        { x = wrap(x - ys, SIZE, ev); y = wrap(y + xs, SIZE, ev); }
        if (ii + 1 < W) {                            /* :245-251  sigZ has DATA_WIDTH entries */
            const int64_t rom = g_t2[ii] >> (48 - DWPH);
            z = wrap(z < 0 ? z + rom : z - rom, DWPH, ev);
        }
    }
    *os = asr(y, SIZE - W);                          /* :257-258  top DATA_WIDTH bits */
    *oc = asr(x, SIZE - W);
}

static int widths_ok(uint32_t model, uint32_t PW, uint32_t W, uint32_t P)
{
    if (PW < 3 || PW > 32 || W < 4 || W > 32) return 0;
    if (model == BHWO_MODEL_HLS && PW > W + 2) return 0;   /* HLS init_t truncation, SURVEY section 7 */
    if (model == BHWO_MODEL_VHDL && (P < 1 || P > 7)) return 0;
    if (model == BHWO_MODEL_SCALED && W < 8) return 0;      /* SEL_SIZE(DATA_WIDTH-8) */
    return model <= BHWO_MODEL_SCALED;
}

int bhwo_cordic(uint32_t model, uint32_t PW, uint32_t W, uint32_t P, uint64_t theta,
                int32_t *out_cos, int32_t *out_sin, uint64_t *wrap_events)
{
    if (!widths_ok(model, PW, W, P)) return -1;
    build_tables();
    theta &= (1ull << PW) - 1ull;
    int64_t c = 0, s = 0;
    if (model == BHWO_MODEL_HLS) cordic_hls(PW, W, theta, &c, &s, wrap_events);
    else if (model == BHWO_MODEL_CPP) cordic_cpp(PW, W, theta, &c, &s);
    else if (model == BHWO_MODEL_VHDL) cordic_vhdl(PW, W, P, theta, &c, &s, wrap_events);
    else if (model == BHWO_MODEL_DDS48) cordic_prerot(PW, W, 48, 48, theta, &c, &s, wrap_events);
    else {
        const unsigned size = k_sel_size[W - 8];
        cordic_prerot(PW, W, size, size < PW ? PW : size, theta, &c, &s, wrap_events);
    }
    if (out_cos) *out_cos = (int32_t)c;
    if (out_sin) *out_sin = (int32_t)s;
    return 0;
}

/* Vectoring CORDIC src/cordic_atan2.vhd:82-219 (as read; no simulator, no testbench upstream).
 * x, y are INPUT_WIDTH-bit two's-complement inputs; the result is the ANGLE_WIDTH-bit word the entity drives. */
int bhwo_atan2(uint32_t P, uint32_t IW, uint32_t AW, int64_t vx, int64_t vy, int32_t *out_phi)
{
    if (P < 1 || P > 7 || AW < 4 || AW > 32 || IW < AW - 1 || IW > 32) return -1;   /* VEC_DX(ii), ii <= ANGLE_WIDTH-2 :142-145 */
    build_tables();
    const unsigned B = AW + P;
    const uint64_t ux = (uint64_t)vx & ((1ull << IW) - 1ull), uy = (uint64_t)vy & ((1ull << IW) - 1ull);
    const unsigned sx = (unsigned)(ux >> (IW - 1)) & 1u, sy = (unsigned)(uy >> (IW - 1)) & 1u;
    const uint64_t lowm = (1ull << (AW - 1)) - 1ull;
    int64_t x = (int64_t)((sx ? ~ux : ux) & lowm);   /* :142-147  bit-wise xor with the sign, upper bits 0 */
    int64_t y = (int64_t)((sy ? ~uy : uy) & lowm);
    int64_t z = 0;                                   /* :152 */
    for (unsigned ii = 0; ii + 1 < AW; ++ii) {       /* :172-190  ANGLE_WIDTH-1 stages, steered by the sign of y */
        const int64_t rom = g_t4[ii] >> (49 - B);    /* :100-103  top B-1 bits, MSB 0 */
        const int64_t xs = asr(x, ii), ys = asr(y, ii);
        if (y >= 0) { x = wrap(x + ys, B, NULL); y = wrap(y - xs, B, NULL); z = wrap(z - rom, B, NULL); }
        else        { x = wrap(x - ys, B, NULL); y = wrap(y + xs, B, NULL); z = wrap(z + rom, B, NULL); }
    }
    const int64_t phi = wrap(asr(z, P), AW, NULL);   /* :194  sigZ(ANGLE_WIDTH-1)(B-1 downto PRECISION) */
    const int64_t pi_word = (int64_t)1 << (AW - 2);  /* :112  PHI_PI */
    int64_t out;
    switch ((sx << 1) | sy) {                        /* :126-128, :207-213 */
    case 0:  out = phi; break;
    case 1:  out = wrap(phi + pi_word, AW, NULL); break;
    case 2:  out = wrap(-phi, AW, NULL); break;
    default: out = wrap(phi - pi_word, AW, NULL); break;
    }
    if (out_phi) *out_phi = (int32_t)out;
    return 0;
}

/* ------------------------------------------------------------ Taylor feeder */
/*
 * src/taylor_sincos.vhd:91-255 + src/tay1_order.vhd:108-640 (as read; no simulator).
 * ROM entry ii: (S,C) = INTEGER((2^(W-1)-1) * sin/cos(pi*ii / 2^(L+1)))   taylor_sincos.vhd:98-106
 * VHDL INTEGER(real) rounds to nearest, ties away from zero.
 */
static int64_t rom_round(__float128 v)
{
    return (int64_t)(v < 0 ? -floorq(-v + 0.5Q) : floorq(v + 0.5Q));
}

int bhwo_taylor(uint32_t PW, uint32_t W, uint32_t L, uint64_t cnt, int32_t *out_cos, int32_t *out_sin)
{
    if (PW < 3 || PW > 32 || W < 4 || W > 32 || L < 1 || L > 20) return -1;
    cnt &= (1ull << PW) - 1ull;
    const unsigned q = (unsigned)(cnt >> (PW - 2));              /* :141 */
    const __float128 amp = ldexpq(1.0Q, (int)W - 1) - 1.0Q;
    uint64_t addr, f = 0;
    int taylor = 0;
    if ((int)PW - (int)L < 2)       addr = (cnt & ((1ull << (PW - 2)) - 1ull)) << (L - PW + 2);   /* :157-161 */
    else if (PW - L == 2)           addr = cnt & ((1ull << L) - 1ull);                             /* :164-167 */
    else {                                                                                         /* :190-191 */
        addr = (cnt >> (PW - L - 2)) & ((1ull << L) - 1ull);
        f = cnt & ((1ull << (PW - L - 2)) - 1ull);
        taylor = 1;
    }
    __float128 ang = (__float128)addr * M_PIq / ldexpq(1.0Q, (int)L + 1);
    int64_t S = rom_round(amp * sinq(ang)), C = rom_round(amp * cosq(ang));
    int64_t c = C, s = S;
    if (taylor) {
        const int stage = (int)PW - (int)L - 3;                  /* taylor_sincos.vhd:200 */
        const int64_t rpi = (int64_t)floorq(M_PIq * ldexpq(1.0Q, 17 - stage) + 0.5Q);   /* tay1_order.vhd:133 */
        const int64_t m = (rpi * (int64_t)f) & 0xFFFFFF;         /* 24-bit ROM word, :139,146 */
        const unsigned X = 19 + L;                               /* :112 */
        if (W < 19) {                                            /* :180-503  one MACC each, result slice :501-502 */
            __int128 pc = ((__int128)C << X) - (__int128)m * S;
            __int128 ps = ((__int128)S << X) + (__int128)m * C;
            c = wrap((int64_t)(pc >> X), W, NULL);
            s = wrap((int64_t)(ps >> X), W, NULL);
        } else {                                                 /* :506-640 */
            int64_t dc = wrap((int64_t)(((__int128)m * S) >> X), W, NULL);
            int64_t ds = wrap((int64_t)(((__int128)m * C) >> X), W, NULL);
            c = wrap(C - dc, W, NULL);                           /* :595-596 */
            s = wrap(S + ds, W, NULL);
            const int64_t sat = ((int64_t)1 << (W - 1)) - 1;     /* :602-616 */
            if (c < 0) c = sat;
            if (s < 0) s = sat;
        }
    }
    int64_t dc, ds;
    switch (q) {                                                 /* taylor_sincos.vhd:240-253 */
    case 0: ds = s; dc = c; break;
    case 1: ds = c; dc = wrap(-s, W, NULL); break;
    case 2: ds = wrap(-s, W, NULL); dc = wrap(-c, W, NULL); break;
    default: ds = wrap(-c, W, NULL); dc = s; break;
    }
    if (out_cos) *out_cos = (int32_t)dc;
    if (out_sin) *out_sin = (int32_t)ds;
    return 0;
}

/* ------------------------------------------------------- a_k derivation (a9) */
/*
 * hls/windows/win_function.cpp:173-177,191-192,206-212,253-261,306-316,341-355:
 * a_k = round(coe_k * (2^(W-s) - 1)), s = 1 for 2/3/4-term, 2 for 5/7-term; C round().
 */
static const double k_hamming[2] = { 0.5434783, 1 - 0.5434783 };
static const double k_hann[2]    = { 0.5, 0.5 };
static const double k_bh3[3]     = { 0.21, 0.25, 0.04 };
static const double k_bh4[4]     = { 0.35875, 0.48829, 0.14128, 0.01168 };
static const double k_bh5[5]     = { 0.3232153788877343, 0.4714921439576260, 0.1755341299601972,
                                     0.0284969901061499, 0.0012613570882927 };
static const double k_bh7[7]     = { 0.271220360585039, 0.433444612327442, 0.218004122892930,
                                     0.065785343295606, 0.010761867305342, 0.000770012710581,
                                     0.000013680883060 };

static int terms_of(uint32_t win_type)
{
    switch (win_type) {
    case 1: case 2: return 2;
    case 3: return 3; case 4: return 4; case 5: return 5; case 7: return 7;
    default: return 0;
    }
}

int bhwo_coeffs_from_float(uint32_t win_type, uint32_t W, const double *a, int32_t aa[7])
{
    int K = terms_of(win_type);
    if (!K || W < 4 || W > 32) return -1;
    if (!a) {
        switch (win_type) {
        case 1: a = k_hamming; break; case 2: a = k_hann; break; case 3: a = k_bh3; break;
        case 4: a = k_bh4; break; case 5: a = k_bh5; break; default: a = k_bh7; break;
        }
    }
    const unsigned s = (K >= 5) ? 2 : 1;
    const double scale = pow(2.0, (double)(W - s)) - 1.0;
    memset(aa, 0, 7 * sizeof(int32_t));
    for (int k = 0; k < K; ++k) aa[k] = (int32_t)(int64_t)round(a[k] * scale);
    return 0;
}

/* ------------------------------------------------------------------ combine */

static int64_t feeder_cos(const bhwo_params *p, uint64_t theta, unsigned harmonic)
{
    int32_t c = 0, s = 0;
    if (p->sin_type != BHWO_SIN_CORDIC) {
        /* bh_win_3term.vhd:221-226: the 2nd harmonic is a second generator with PHASE_WIDTH-1
         * driven by its own +1 counter, i.e. phase n mod 2^(PW-1) at width PW-1.
         * Extension (TAYLOR_ALL, no reference counterpart): harmonic k = m * 2^v, m odd, reads a generator of width
         * PW - v at phase (m*n) mod 2^(PW-v) = theta >> v; for k <= 2 that is the reference's wiring. */
        unsigned v = 0;
        while (!((harmonic >> v) & 1u)) ++v;
        bhwo_taylor(p->phi_width - v, p->dat_width, p->lut_size, theta >> v, &c, &s);
    } else {
        bhwo_cordic(p->model, p->phi_width, p->dat_width, p->precision, theta, &c, &s, NULL);
    }
    return c;
}

static int32_t window_sample(const bhwo_params *p, uint64_t n)
{
    const unsigned W = p->dat_width, K = p->n_terms;
    const uint64_t mask = (1ull << p->phi_width) - 1ull;
    int64_t acc = p->aa[0];
    if (p->combine == BHWO_COMBINE_HLS) {
        /* hls/windows/win_function.cpp:182,197,222-225,271-275,327-332,368-375 */
        for (unsigned k = 1; k < K; ++k) {
            int64_t c = feeder_cos(p, (k * n) & mask, k);            /* cordic(k*i): wraps into phi_t */
            __int128 prod = (__int128)p->aa[k] * c;
            int64_t m = wrap((int64_t)(prod >> (W - 2)), 2 * W + 1, NULL);   /* dbl_t mlt_k */
            acc += (k & 1) ? -m : m;
        }
        return (int32_t)wrap(acc, W, NULL);                          /* (win_t)(...) */
    }
    /* VHDL rule: src/bh_win_7term.vhd:353-438, bh_win_5term.vhd:289-347, bh_win_4term.vhd:230-280,
     * bh_win_3term.vhd:264-306, hamming_win.vhd:195-231 */
    for (unsigned k = 1; k < K; ++k) {
        int64_t c = feeder_cos(p, (k * n) & mask, k);
        __int128 prod = (__int128)p->aa[k] * c;                      /* int_multNxN_dsp48.vhd:102 */
        int64_t r = wrap((int64_t)(prod >> (W - 2)), W + 1, NULL);   /* slice (2W-2 downto W-2) */
        int64_t b = wrap(asr(r, 1) + (r & 1), W, NULL);              /* round to W bits */
        acc += (k & 1) ? -b : b;
    }
    if (K == 2) {
        int64_t S = wrap(acc, W + 1, NULL);                          /* hamming_win.vhd:214 */
        return (int32_t)wrap(asr(S, 1) + (S & 1), W, NULL);          /* :224-228 */
    }
    int64_t S = wrap(acc, W + 2, NULL);                              /* bh_win_7term.vhd:409-421 */
    return (int32_t)wrap(asr(S, 2) + (asr(S, 1) & 1), W, NULL);      /* :431-435 */
}

static int params_ok(const bhwo_params *p)
{
    if (!p) return 0;
    if (!(p->n_terms == 2 || p->n_terms == 3 || p->n_terms == 4 || p->n_terms == 5 || p->n_terms == 7)) return 0;
    if (p->combine > BHWO_COMBINE_VHDL || p->sin_type > BHWO_SIN_TAYLOR_ALL) return 0;
    if (p->sin_type != BHWO_SIN_CORDIC) {
        if (p->n_terms > 3 && p->sin_type == BHWO_SIN_TAYLOR) return 0;   /* win_selector.vhd:93-135 */
        if (p->phi_width < 4 || p->phi_width > 32 || p->dat_width < 4 || p->dat_width > 32) return 0;
        return p->lut_size >= 1 && p->lut_size <= 20;
    }
    if (p->model > BHWO_MODEL_VHDL) return 0;                        /* dds48 / scaled feed no window upstream */
    return widths_ok(p->model, p->phi_width, p->dat_width, p->precision);
}

int bhwo_generate(const bhwo_params *p, uint64_t n0, uint64_t count, int32_t *out)
{
    if (!params_ok(p) || (count && !out)) return -1;
    build_tables();
    for (uint64_t i = 0; i < count; ++i) out[i] = window_sample(p, n0 + i);
    return 0;
}

int bhwo_sincos(const bhwo_params *p, uint64_t theta0, uint64_t count, int32_t *out_sin, int32_t *out_cos)
{
    if (!p) return -1;
    build_tables();
    for (uint64_t i = 0; i < count; ++i) {
        int32_t c, s;
        int rc = p->sin_type != BHWO_SIN_CORDIC
                     ? bhwo_taylor(p->phi_width, p->dat_width, p->lut_size, theta0 + i, &c, &s)
                     : bhwo_cordic(p->model, p->phi_width, p->dat_width, p->precision, theta0 + i, &c, &s, NULL);
        if (rc) return rc;
        if (out_sin) out_sin[i] = s;
        if (out_cos) out_cos[i] = c;
    }
    return 0;
}

uint64_t bhwo_fnv1a64(const int32_t *v, uint64_t count, uint64_t seed)
{
    uint64_t h = seed ? seed : 0xcbf29ce484222325ull;
    const unsigned char *b = (const unsigned char *)v;
    for (uint64_t i = 0; i < 4 * count; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}
