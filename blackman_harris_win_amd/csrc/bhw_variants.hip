// bhw_variants.hip -- variant generators: cordic_dds48, cordic_dds_scaled, cordic_atan2
//
// Part of the hand-written HIP kernels for gfx950 (MI355X, CDNA4) behind include/bhw.h.  Hot path of the reference: phase
// accumulator -> CORDIC rotation chain (or Taylor LUT) -> weighted N-term cosine sum -> int32 coefficient (SURVEY section 8a
// rows a1-a11).  Integer semantics follow SURVEY App. A; reference lines are cited at each step.
#include "bhw_device.h"

namespace {

// ---------------------------------------------------------------------------------------
// Variant generators.  One lane per sample, 64-bit state wrapped to the entity's vector widths.
// ---------------------------------------------------------------------------------------
// cordic_dds48 (src/cordic_dds48.vhd:160-258) and cordic_dds_scaled (src/cordic_dds_scaled.vhd:176-283)
// The SIZE- and DWPH-bit stores of the entities never wrap (|x|, |y| <= 2^(SIZE-2) * 1.0000..., |z| < 2^(DWPH-2) + atan
// terms; tests/test_oracle.py::test_variant_generator_wraps_never_fire), so the kernel keeps plain 64-bit state.
// Rotation ii in "mad" form once the shifted operand fits 32 bits (SIZE <= 48: ii >= 16), as in rot_step; the first
// rotations use 64-bit select-and-add.  dds48 :233-251: z >= 0 -> x += y>>ii, y -= x>>ii, z -= rom; else the opposite.
template <int II>
__device__ __forceinline__ void prerot_step(int64_t &x, int64_t &y, int64_t &z, int64_t rom, bool last)
{
    const int32_t m = (int32_t)(z >> 63);               // -1 when z < 0
    if constexpr (II >= 16) {
        const int32_t sg = m | 1;                       // -1 when z < 0, +1 otherwise
        const int32_t nsg = -sg;
        int32_t ys = (int32_t)(y >> II), xs = (int32_t)(x >> II);
        asm volatile("" : "+v"(ys), "+v"(xs));          // both shifts read the old state
        x += (int64_t)sg * (int64_t)ys;
        y += (int64_t)nsg * (int64_t)xs;
        if (!last) z += (int64_t)nsg * (int64_t)(int32_t)rom;       // rom < 2^(45-II) here
    } else {
        const int64_t ys = y >> II, xs = x >> II;
        const bool neg = m != 0;
        x += neg ? -ys : ys;
        y += neg ? xs : -xs;
        if (!last) z += neg ? rom : -rom;
    }
}

template <int NITER>
__global__ __launch_bounds__(kBlock) void k_sincos_prerot(BhwPrerotCfg c, uint64_t theta0, uint64_t count,
                                                           int32_t *__restrict__ d_sin, int32_t *__restrict__ d_cos)
{
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t PW = c.phi_width;
    const uint64_t theta = (theta0 + i) & ((1ull << PW) - 1ull);
    const uint32_t q = (uint32_t)(theta >> (PW - 2)) & 3u;              // dds48 :167
    const uint64_t low = theta & ((1ull << (PW - 2)) - 1ull);
    uint64_t t = theta;                                                 // init_t :169-186
    int64_t x = c.gain, y = 0;                                          // init_x / init_y :191-216
    if (q == 1u)      { t = low;                      x = 0; y = -c.gain; }
    else if (q == 2u) { t = (3ull << (PW - 2)) | low; x = 0; y = c.gain; }
    int64_t z = wrap_bits((int64_t)(t << (c.dwph - PW)), c.dwph);       // init_z :163-164 (scaled :180-186): sign-extend the phase
#define BHW_PREROT(II) if constexpr (NITER > II) prerot_step<II>(x, y, z, c.lut[II], II + 1 == NITER);
    BHW_PREROT(0) BHW_PREROT(1) BHW_PREROT(2) BHW_PREROT(3) BHW_PREROT(4) BHW_PREROT(5) BHW_PREROT(6) BHW_PREROT(7)
    BHW_PREROT(8) BHW_PREROT(9) BHW_PREROT(10) BHW_PREROT(11) BHW_PREROT(12) BHW_PREROT(13) BHW_PREROT(14) BHW_PREROT(15)
    BHW_PREROT(16) BHW_PREROT(17) BHW_PREROT(18) BHW_PREROT(19) BHW_PREROT(20) BHW_PREROT(21) BHW_PREROT(22) BHW_PREROT(23)
    BHW_PREROT(24) BHW_PREROT(25) BHW_PREROT(26) BHW_PREROT(27) BHW_PREROT(28) BHW_PREROT(29) BHW_PREROT(30) BHW_PREROT(31)
#undef BHW_PREROT
    if (d_sin) d_sin[i] = (int32_t)(y >> (c.size - NITER));             // :257-258  top DATA_WIDTH bits
    if (d_cos) d_cos[i] = (int32_t)(x >> (c.size - NITER));
}

// The same generators over one whole period (count == 2^PHASE_WIDTH, any start phase): inside a quadrant consecutive phases are
// consecutive angles on one start vector, so a group of 64 phases shares its rotations until the first one whose threshold
// falls inside the group -- exactly the structure of k_table_build_shared, with the quadrant's start vector per group.  Phase 1:
// one lane per group runs the shared prefix (these are the expensive rotations: 64-bit select-and-add below stage 16) and parks
// it in LDS; phase 2: one wave per group, one lane per phase, only the remaining stages.
template <int NITER>
__global__ __launch_bounds__(kBuildThreads) void k_prerot_sweep(BhwPrerotCfg c, uint32_t theta0, int32_t *__restrict__ d_sin, int32_t *__restrict__ d_cos)
{
    __shared__ int64_t gx[kGroupsPerWg], gy[kGroupsPerWg], gz[kGroupsPerWg];
    __shared__ int32_t gk[kGroupsPerWg];
    // The per-leaf stages read their ROM word from shared memory (a broadcast read) once there are more than 26 of them: 32 64-bit
    // words as scalar kernel arguments are 64 SGPRs, and the instances from 27 stages on used to spill 2 .. 12 of them.
    constexpr bool kLutLds = NITER > 26;
    __shared__ int64_t lut_s[kLutLds ? 32 : 1];
    if (kLutLds && threadIdx.x >= 64u && threadIdx.x < 96u) lut_s[threadIdx.x - 64u] = c.lut[threadIdx.x - 64u];
    const uint32_t PW = c.phi_width, zs = c.dwph - PW;
    const uint32_t group0 = blockIdx.x * kGroupsPerWg, n_groups = 1u << (PW - 6);
    constexpr int kmax = NITER < kPrefixMax ? NITER : kPrefixMax;
    if (threadIdx.x < (uint32_t)kGroupsPerWg && group0 + threadIdx.x < n_groups) {
        const uint64_t theta = (uint64_t)(group0 + threadIdx.x) << 6;       // first phase of the group
        const uint32_t q = (uint32_t)(theta >> (PW - 2)) & 3u;              // dds48 :167
        const uint64_t low = theta & ((1ull << (PW - 2)) - 1ull);
        uint64_t t = theta;                                                 // init_t :169-186
        int64_t x = c.gain, y = 0;                                          // init_x / init_y :191-216
        if (q == 1u)      { t = low;                      x = 0; y = -c.gain; }
        else if (q == 2u) { t = (3ull << (PW - 2)) | low; x = 0; y = c.gain; }
        int64_t zf = wrap_bits((int64_t)(t << zs), c.dwph);                 // init_z: sign-extended phase
        const int64_t span = (int64_t)63 << zs;
        int k = 0;
        bool live = true;
#define BHW_PRE(II)                                                                        \
        if constexpr (kmax > II) {                                                          \
            if (live) {                                                                     \
                if ((zf < 0) != (zf + span < 0)) live = false;                              \
                else { prerot_step<II>(x, y, zf, c.lut[II], II + 1 == NITER); k = II + 1; } \
            }                                                                               \
        }
        BHW_PRE(0) BHW_PRE(1) BHW_PRE(2) BHW_PRE(3) BHW_PRE(4) BHW_PRE(5) BHW_PRE(6) BHW_PRE(7) BHW_PRE(8) BHW_PRE(9)
        BHW_PRE(10) BHW_PRE(11) BHW_PRE(12) BHW_PRE(13) BHW_PRE(14) BHW_PRE(15) BHW_PRE(16) BHW_PRE(17) BHW_PRE(18) BHW_PRE(19)
        BHW_PRE(20) BHW_PRE(21) BHW_PRE(22) BHW_PRE(23)
#undef BHW_PRE
        gx[threadIdx.x] = x; gy[threadIdx.x] = y; gz[threadIdx.x] = zf; gk[threadIdx.x] = k;
    }
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t nmask = (1u << PW) - 1u;                                  // PW <= 32 here: a sweep of 2^PW phases in int32 indices
    for (uint32_t gi = wave; gi < (uint32_t)kGroupsPerWg; gi += kBuildThreads / 64) {
        const uint32_t g = group0 + gi;
        if (g >= n_groups) break;
        int64_t x = gx[gi], y = gy[gi];
        int64_t z = gz[gi] + ((int64_t)lane << zs);
        const int k0 = __builtin_amdgcn_readfirstlane(gk[gi]);
#define BHW_POST(II) if constexpr (NITER > II) { if (II >= kmax || II >= k0) prerot_step<II>(x, y, z, kLutLds ? lut_s[II] : c.lut[II], II + 1 == NITER); }
        BHW_POST(0) BHW_POST(1) BHW_POST(2) BHW_POST(3) BHW_POST(4) BHW_POST(5) BHW_POST(6) BHW_POST(7)
        BHW_POST(8) BHW_POST(9) BHW_POST(10) BHW_POST(11) BHW_POST(12) BHW_POST(13) BHW_POST(14) BHW_POST(15)
        BHW_POST(16) BHW_POST(17) BHW_POST(18) BHW_POST(19) BHW_POST(20) BHW_POST(21) BHW_POST(22) BHW_POST(23)
        BHW_POST(24) BHW_POST(25) BHW_POST(26) BHW_POST(27) BHW_POST(28) BHW_POST(29) BHW_POST(30) BHW_POST(31)
#undef BHW_POST
        const uint32_t i = ((g << 6) + lane - theta0) & nmask;
        if (d_sin) d_sin[i] = (int32_t)(y >> (c.size - NITER));             // :257-258  top DATA_WIDTH bits
        if (d_cos) d_cos[i] = (int32_t)(x >> (c.size - NITER));
    }
}

// cordic_atan2 (src/cordic_atan2.vhd:126-213).  The B = ANGLE_WIDTH + PRECISION bit registers do wrap (PRECISION 1 with
// full-scale inputs), so the state is kept shifted left by (word size - B): overflow of the word then *is* the B-bit wrap, and
// the only extra work is clearing the low bits that an arithmetic right shift drags in.  U = uint32_t when B <= 32 (half the
// instructions of the 64-bit form: every shift, add and select is one 32-bit operation), uint64_t otherwise.
template <typename U>
__global__ __launch_bounds__(kBlock) void k_atan2(BhwAtan2Cfg c, uint64_t count, const int32_t *__restrict__ d_x,
                                                   const int32_t *__restrict__ d_y, int32_t *__restrict__ d_phi)
{
    using S = typename std::make_signed<U>::type;
    constexpr uint32_t WB = 8u * sizeof(U);
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const uint32_t IW = c.input_width, AW = c.angle_width, B = AW + c.precision, sh = WB - B;
    const uint64_t im = (1ull << IW) - 1ull;                             // IW <= 32
    const uint64_t ux = (uint64_t)(int64_t)d_x[i] & im, uy = (uint64_t)(int64_t)d_y[i] & im;
    const uint32_t sx = (uint32_t)(ux >> (IW - 1)) & 1u, sy = (uint32_t)(uy >> (IW - 1)) & 1u;
    const uint64_t lowm = (1ull << (AW - 1)) - 1ull;
    const U keep = (U)~(((U)1 << sh) - (U)1);
    U X = (U)((sx ? ~ux : ux) & lowm) << sh;                             // :142-147
    U Y = (U)((sy ? ~uy : uy) & lowm) << sh;
    U Z = 0;                                                            // :152
    for (uint32_t ii = 0; ii + 1 < AW; ++ii) {                          // :172-190
        const U xs = (U)((S)X >> ii) & keep, ys = (U)((S)Y >> ii) & keep;
        const U rom = (U)c.lut[ii] << sh;
        const bool pos = (S)Y >= 0;
        X = pos ? X + ys : X - ys;
        Y = pos ? Y - xs : Y + xs;
        Z = pos ? Z - rom : Z + rom;
    }
    // :194  sigZ(ANGLE_WIDTH-1)(B-1 downto PRECISION): the top ANGLE_WIDTH bits of the B-bit word
    const int64_t phi = (int64_t)((S)Z >> (WB - AW));
    const int64_t pi_word = (int64_t)1 << (AW - 2);                     // PHI_PI :112
    const uint32_t quad = (sx << 1) | sy;                               // :126-128
    const int64_t out = quad == 0u ? phi : quad == 1u ? phi + pi_word : quad == 2u ? -phi : phi - pi_word;   // :207-213
    d_phi[i] = (int32_t)wrap_bits(out, AW);
}

} // namespace

int bhwk_sincos_prerot(const BhwLaunch &l, const BhwPrerotCfg &c, uint64_t theta0, uint64_t count, int32_t *d_sin, int32_t *d_cos)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    hipStream_t st = (hipStream_t)l.stream;
    if (c.phi_width >= 16 && c.phi_width <= 30 && count == (1ull << c.phi_width)) {      // one whole period: shared rotation prefixes
        const uint32_t groups = 1u << (c.phi_width - 6);
        const dim3 grid((groups + kGroupsPerWg - 1) / kGroupsPerWg), block(kBuildThreads);
        const uint32_t th0 = (uint32_t)(theta0 & ((1ull << c.phi_width) - 1ull));
        switch (c.dat_width) {
#define BHW_CASE(N) case N: BHW_LAUNCH(k_prerot_sweep<N>, grid, block, 0, st, c, th0, d_sin, d_cos); break;
            BHW_CASE(8) BHW_CASE(9) BHW_CASE(10) BHW_CASE(11) BHW_CASE(12) BHW_CASE(13) BHW_CASE(14) BHW_CASE(15) BHW_CASE(16)
            BHW_CASE(17) BHW_CASE(18) BHW_CASE(19) BHW_CASE(20) BHW_CASE(21) BHW_CASE(22) BHW_CASE(23) BHW_CASE(24)
            BHW_CASE(25) BHW_CASE(26) BHW_CASE(27) BHW_CASE(28) BHW_CASE(29) BHW_CASE(30) BHW_CASE(31) BHW_CASE(32)
#undef BHW_CASE
        default: return (int)hipErrorInvalidValue;
        }
        return finish(hipSuccess);
    }
    const dim3 grid(grid_for(count)), block(kBlock);
    switch (c.dat_width) {                                              // DATA_WIDTH stages, unrolled
#define BHW_CASE(N) case N: BHW_LAUNCH(k_sincos_prerot<N>, grid, block, 0, st, c, theta0, count, d_sin, d_cos); break;
        BHW_CASE(8) BHW_CASE(9) BHW_CASE(10) BHW_CASE(11) BHW_CASE(12) BHW_CASE(13) BHW_CASE(14) BHW_CASE(15) BHW_CASE(16)
        BHW_CASE(17) BHW_CASE(18) BHW_CASE(19) BHW_CASE(20) BHW_CASE(21) BHW_CASE(22) BHW_CASE(23) BHW_CASE(24)
        BHW_CASE(25) BHW_CASE(26) BHW_CASE(27) BHW_CASE(28) BHW_CASE(29) BHW_CASE(30) BHW_CASE(31) BHW_CASE(32)
#undef BHW_CASE
    default: return (int)hipErrorInvalidValue;
    }
    return finish(hipSuccess);
}

int bhwk_atan2(const BhwLaunch &l, const BhwAtan2Cfg &c, uint64_t count, const int32_t *d_x, const int32_t *d_y, int32_t *d_phi)
{
    if (!count) return 0;
    BHW_SET_DEVICE(l);
    if (c.angle_width + c.precision <= 32u) BHW_LAUNCH(k_atan2<uint32_t>, dim3(grid_for(count)), dim3(kBlock), 0, (hipStream_t)l.stream, c, count, d_x, d_y, d_phi);
    else                                    BHW_LAUNCH(k_atan2<uint64_t>, dim3(grid_for(count)), dim3(kBlock), 0, (hipStream_t)l.stream, c, count, d_x, d_y, d_phi);
    return finish(hipSuccess);
}

