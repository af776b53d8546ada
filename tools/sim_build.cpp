// sim_build.cpp -- CPU model of the table-build kernel's arithmetic (k_table_build_mirror), used to establish on the host
// that its short cuts are exact before (and after) they are written for the GPU:
//   (1) octant mirror: chain(E - u) == swap(chain(u)) unless some z_k == 0 before the hand-over rotation KS;
//   (2) narrow state: inside a "safe" 64-leaf group every x_k, y_k lies in [0, 2^NITER) from the group's split rotation on, so
//       the rotations run on 32-bit unsigned words with logical shifts;
//   (3) table tail: for the last TAIL rotations  (y_k >> k) == (y_KS >> KS) >> (k - KS)  as long as the low KS bits of x_KS, y_KS
//       keep a margin of 128 to both ends, so  x_end = x_KS - D[p][y_KS >> KS],  y_end = y_KS + D[p][x_KS >> KS]  with
//       D[p][v] = sum_j sigma_j(p) (v >> j)  and p the decision pattern looked up from z_KS;
//   (4) biased narrow state (round 4): a group whose x runs up to 2^NITER (1 + eps) -- the first third of a degree of the octant,
//       where x alone does not fit the 32-bit word -- keeps x - 2^(NITER-1) in the word instead:  (x >> k) = ((x - B) >> k) + (B >> k)
//       exactly, B = 2^(NITER-1) being a multiple of 2^k for every k < NITER, and the tail table gets a 65th row (x >> KS = 64).
// Every entry of the first-quadrant table is compared with the plain chain (the rotation loop of
// hls/windows/win_function.cpp:110-125 | cpp/cordic_sincos.cpp:49-63 | src/cordic_dds.vhd:197-213).
// Test infrastructure / design evidence only: nothing here is linked into the product.
//   usage: sim_build <model 0 hls | 1 cpp | 2 vhdl> <PW> <W> [precision [tail rotations 6 | 7]]     (exit code 0 = every entry identical)
#include <quadmath.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

static int64_t T2[48], T4[48], G46, G47;

static void tables()
{
    const __float128 pi = M_PIq;
    __float128 k = 1.0Q;
    for (int i = 0; i < 48; ++i) {
        const __float128 a = atanq(ldexpq(1.0Q, -i)) / pi;
        T2[i] = (int64_t)floorq(ldexpq(a, 47) + 0.5Q);
        T4[i] = (int64_t)floorq(ldexpq(a, 48) + 0.5Q);
    }
    T4[47] = 0;
    for (int i = 0; i < 120; ++i) k *= sqrtq(1.0Q + ldexpq(1.0Q, -2 * i));
    G46 = (int64_t)floorq(ldexpq(1.0Q, 46) / k + 0.5Q);
    G47 = (int64_t)floorq(ldexpq(1.0Q, 47) / k + 0.5Q);
}

struct Cfg {
    int64_t lut[32];
    int64_t x0;
    int n_iter, z_shl, out_shr;
    uint32_t E;   // first-quadrant entries (z_shr == 0)
};

static void step(int64_t &x, int64_t &y, int64_t &z, int k, int64_t l)
{
    const int64_t xs = x >> k, ys = y >> k;
    if (z < 0) { x += ys; y -= xs; z += l; }
    else       { x -= ys; y += xs; z -= l; }
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: sim_build model PW W [precision]\n"); return 2; }
    tables();
    const int model = atoi(argv[1]), PW = atoi(argv[2]), W = atoi(argv[3]), P = argc > 4 ? atoi(argv[4]) : 1;
    Cfg c{};
    if (model == 0)      { for (int i = 0; i + 1 < W; ++i) c.lut[i] = T4[i] >> (47 - W); c.x0 = G46 >> (46 - W); c.n_iter = W; c.z_shl = W - PW + 2; c.out_shr = 2; }
    else if (model == 1) { for (int i = 0; i + 1 < W; ++i) c.lut[i] = T2[i] >> (47 - W); c.x0 = G46 >> (46 - W); c.n_iter = W; c.z_shl = W - PW + 1; c.out_shr = 2; }
    else { const int Wi = W + P; for (int i = 0; i + 1 < W; ++i) c.lut[i] = T4[i] >> (49 - Wi); c.x0 = G47 >> (49 - Wi); c.n_iter = W - 1; c.z_shl = W - PW + P; c.out_shr = P; }
    if (c.z_shl < 0 || (model == 0 ? PW - 1 >= W : model == 1 ? PW - 1 >= W : PW >= W)) { fprintf(stderr, "z_shr != 0: not a mirror-kernel configuration\n"); return 2; }
    c.E = 1u << (PW - 2);
    const int N = c.n_iter, TAIL = argc > 5 ? atoi(argv[5]) : 6, KS = N - TAIL, PREFIX_MAX = 20;
    if (TAIL != 6 && TAIL != 7) { fprintf(stderr, "tail of 6 or 7 rotations\n"); return 2; }
    const int NV = 1 << TAIL, NP = 1 << TAIL, MARGIN = 2 << TAIL;     // rows / patterns of D; the tail moves x by less than MARGIN
    const int kcap = KS < PREFIX_MAX ? KS : PREFIX_MAX;
    const uint32_t E = c.E;
    const int s = c.z_shl;
    if (2 * c.lut[0] != ((int64_t)E << s)) { fprintf(stderr, "2 lut[0] is not a quarter turn\n"); return 2; }

    // reference: plain chain for every entry
    std::vector<int32_t> rc(E), rs(E);
    for (uint32_t u = 0; u < E; ++u) {
        int64_t x = c.x0, y = 0, z = (int64_t)u << s;
        for (int k = 0; k < N; ++k) step(x, y, z, k, c.lut[k]);
        rc[u] = (int32_t)(x >> c.out_shr);
        rs[u] = (int32_t)(y >> c.out_shr);
    }

    // tail tables: pattern of z_KS (range [-ZB, ZB)), D[p][v]
    const int ZB = 4 << TAIL;                                                // |z_KS| < ZB (checked per lane below)
    std::vector<uint16_t> Pz(2 * ZB);
    for (int zi = 0; zi < 2 * ZB; ++zi) {
        int64_t z = zi - ZB;
        unsigned p = 0;
        for (int j = 0; j < TAIL; ++j) {
            const int k = KS + j;
            if (z < 0) { p |= 1u << j; z += c.lut[k]; } else z -= c.lut[k];
        }
        Pz[zi] = (uint16_t)p;
    }
    std::vector<std::vector<int16_t>> D(NP, std::vector<int16_t>(NV + 1));
    for (unsigned p = 0; p < (unsigned)NP; ++p)
        for (int v = 0; v <= NV; ++v) {
            int d = 0;
            for (int j = 0; j < TAIL; ++j) d += ((p >> j) & 1u) ? -(v >> j) : (v >> j);    // sigma = +1 when z >= 0
            D[p][v] = (int16_t)d;
        }

    std::vector<int32_t> gc(E, INT32_MIN), gs(E, INT32_MIN);
    uint64_t hi_groups = 0;
    const int KA = KS - 10;                                                  // the kernel's ten-rotation block starts here
    const int64_t BX = (int64_t)1 << (N - 1);                                // bias of the biased narrow state
    uint64_t n_groups = E >> 7, wide_groups = 0, unsafe_waves = 0, unsafe_lanes = 0, zero_lanes = 0, zero_prefix = 0;
    const int64_t LIM = (int64_t)1 << N;
    auto full_chain = [&](uint32_t u, int32_t &oc, int32_t &os) {           // the kernel's deferred / fallback chain
        int64_t x = c.x0, y = c.x0, z = ((int64_t)u << s) - c.lut[0];
        for (int k = 1; k < N; ++k) step(x, y, z, k, c.lut[k]);
        oc = (int32_t)(x >> c.out_shr); os = (int32_t)(y >> c.out_shr);
    };
    for (uint64_t g = 0; g < n_groups; ++g) {
        // phase 1: shared prefix
        int64_t X = c.x0, Y = c.x0, zf = ((int64_t)(g << 6) << s) - c.lut[0];
        const int64_t span = (int64_t)63 << s;
        int k0 = 1;
        bool zero0 = false;
        for (int kk = 1; kk < kcap; ++kk) {
            const int64_t zl = zf + span;
            if ((zf < 0) != (zl < 0)) break;
            zero0 |= zf == 0;
            step(X, Y, zf, kk, c.lut[kk]);
            k0 = kk + 1;
        }
        zero_prefix += zero0;
        // narrow-state test (scalar, per group)
        const int64_t Dr = ((int64_t)1 << (N - k0 + 1)) + 2;
        bool narrow = X >= Dr && Y >= Dr && X + Dr < LIM && Y + Dr < LIM && N <= 32;
        // biased: x - B in [Dr, 2^32 - Dr), x below 65 * 2^KS (row 64 of the tail table), y as in the plain narrow state; only groups
        // that enter the ten-rotation block (split rotation >= KA)
        const bool hi = !narrow && N <= 32 && k0 >= KA && Y >= Dr && Y + Dr < LIM && X - BX >= Dr && X - BX + Dr < ((int64_t)1 << 32) &&
                        X + Dr < LIM + ((int64_t)1 << KS);
        const int64_t bias = hi ? BX : 0;
        hi_groups += hi;
        narrow = narrow || hi;
        wide_groups += !narrow;
        if (!narrow && getenv("SIM_VERBOSE")) fprintf(stderr, "wide g=%llu k0=%d X=%lld Y=%lld Dr=%lld\n", (unsigned long long)g, k0, (long long)X, (long long)Y, (long long)Dr);
        bool wave_unsafe = false;
        struct Lane { int64_t x, y, z; bool zero; } L[64];
        for (int lane = 0; lane < 64; ++lane) {
            int64_t x = X, y = Y, z = zf + ((int64_t)lane << s);
            bool zero = false;
            for (int k = k0; k < KS; ++k) {
                zero |= z == 0;
                if (narrow) {
                    if (x - bias < 0 || y < 0 || x - bias >= ((int64_t)1 << 32) || y >= LIM) { fprintf(stderr, "narrow range violated g=%llu lane=%d k=%d\n", (unsigned long long)g, lane, k); return 1; }
                    const uint32_t xu = (uint32_t)(x - bias), yu = (uint32_t)y;
                    const uint32_t a = yu >> k, b = (xu >> k) + (uint32_t)(bias >> k);     // logical shifts on 32-bit words
                    uint32_t xn, yn;
                    if (z < 0) { xn = xu + a; yn = yu - b; z += c.lut[k]; } else { xn = xu - a; yn = yu + b; z -= c.lut[k]; }
                    x = (int64_t)xn + bias; y = yn;
                } else step(x, y, z, k, c.lut[k]);
            }
            L[lane] = Lane{x, y, z, zero};
            if (narrow) {
                if (x - bias < 0 || y < 0 || x - bias >= ((int64_t)1 << 32) || y >= LIM) { fprintf(stderr, "narrow range violated at KS\n"); return 1; }
                const uint32_t lowm = (1u << KS) - 1u;
                const bool ok = (((uint32_t)(x - bias) + (uint32_t)MARGIN) & lowm) >= 2u * (uint32_t)MARGIN && (((uint32_t)y + (uint32_t)MARGIN) & lowm) >= 2u * (uint32_t)MARGIN &&
                                (uint64_t)(z + ZB) < (uint64_t)(2 * ZB);
                if (!ok) { wave_unsafe = true; ++unsafe_lanes; }
            }
        }
        unsafe_waves += wave_unsafe;
        for (int lane = 0; lane < 64; ++lane) {
            const uint32_t u = (uint32_t)(g << 6) + lane;
            int64_t x = L[lane].x, y = L[lane].y, z = L[lane].z;
            int64_t x2 = y, y2 = x, z2 = -z;
            int32_t oc, os, oc2, os2;
            if (narrow && !wave_unsafe) {
                const uint32_t xu = (uint32_t)(x - bias);
                const uint32_t xx = (xu >> KS) + (uint32_t)(bias >> KS), yy = (uint32_t)y >> KS;
                if (xx > (uint32_t)NV || yy >= (uint32_t)NV) { fprintf(stderr, "tail operand out of range\n"); return 1; }
                const unsigned p = Pz[z + ZB], p2 = Pz[-z + ZB];
                const uint32_t bo = (uint32_t)(bias >> c.out_shr);
                oc = (int32_t)(((xu - (uint32_t)(int32_t)D[p][yy]) >> c.out_shr) + bo);
                os = (int32_t)(((uint32_t)y + (uint32_t)(int32_t)D[p][xx]) >> c.out_shr);
                oc2 = (int32_t)(((uint32_t)y - (uint32_t)(int32_t)D[p2][xx]) >> c.out_shr);
                os2 = (int32_t)(((xu + (uint32_t)(int32_t)D[p2][yy]) >> c.out_shr) + bo);
            } else {
                for (int k = KS; k < N; ++k) { step(x, y, z, k, c.lut[k]); step(x2, y2, z2, k, c.lut[k]); }
                oc = (int32_t)(x >> c.out_shr); os = (int32_t)(y >> c.out_shr);
                oc2 = (int32_t)(x2 >> c.out_shr); os2 = (int32_t)(y2 >> c.out_shr);
            }
            gc[u] = oc; gs[u] = os;
            if (u >= 1 && u <= (E >> 1) - 1) {
                if (L[lane].zero || (zero0 && lane == 0)) { ++zero_lanes; full_chain(E - u, oc2, os2); }   // worklist / leaf-0 replay
                gc[E - u] = oc2; gs[E - u] = os2;
            }
        }
    }
    full_chain(E >> 1, gc[E >> 1], gs[E >> 1]);                                 // the middle entry: a deferred chain
    uint64_t bad = 0;
    for (uint32_t u = 0; u < E; ++u)
        if (gc[u] != rc[u] || gs[u] != rs[u]) { if (bad < 5) fprintf(stderr, "mismatch u=%u got (%d,%d) want (%d,%d)\n", u, gc[u], gs[u], rc[u], rs[u]); ++bad; }
    printf("model %d PW %d W %d P %d: entries %u, groups %llu, wide-state groups %llu (%.2f %%), waves with an unsafe tail lane %llu (lanes %llu), "
           "deferred images %llu (prefix zeros %llu), biased-narrow groups %llu, mismatches %llu\n", model, PW, W, P, E, (unsigned long long)n_groups,
           (unsigned long long)wide_groups, 100.0 * wide_groups / n_groups, (unsigned long long)unsafe_waves, (unsigned long long)unsafe_lanes,
           (unsigned long long)zero_lanes, (unsigned long long)zero_prefix, (unsigned long long)hi_groups, (unsigned long long)bad);
    return bad ? 1 : 0;
}
