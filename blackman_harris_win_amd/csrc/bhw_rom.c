/* bhw_rom.c -- host-side derivation of the Taylor feeder's quarter-wave ROM.
 *
 * src/taylor_sincos.vhd:91-111 builds the ROM at elaboration time with ieee.math_real:
 *   ROM[ii] = ( INTEGER((2^(W-1)-1) * sin(pi*ii / 2^(L+1))), INTEGER((2^(W-1)-1) * cos(...)) )
 * INTEGER(real) rounds to nearest.  The simulator's libm is unpinned upstream, so the entries are
 * evaluated here in binary128 (correctly rounded for every W <= 32); this is parameter derivation
 * (2^L pairs, once per parameter set), not part of the per-sample path, and is uploaded to the
 * device by bhw_api.cpp.  Compiled with gcc because libquadmath is a GCC runtime library.
 */
#include <quadmath.h>
#include <stdint.h>

static int64_t round_nearest(__float128 v)
{
    return (int64_t)(v < 0 ? -floorq(-v + 0.5Q) : floorq(v + 0.5Q));
}

void bhw_taylor_rom(uint32_t dat_width, uint32_t lut_size, int32_t *rom_sin_cos)
{
    const __float128 amp = ldexpq(1.0Q, (int)dat_width - 1) - 1.0Q;
    const uint32_t depth = 1u << lut_size;
    for (uint32_t ii = 0; ii < depth; ++ii) {
        const __float128 ang = (__float128)ii * M_PIq / ldexpq(1.0Q, (int)lut_size + 1);
        rom_sin_cos[2 * ii + 0] = (int32_t)round_nearest(amp * sinq(ang));
        rom_sin_cos[2 * ii + 1] = (int32_t)round_nearest(amp * cosq(ang));
    }
}

/* round(pi * 2^e) for the 24-bit pi ROM of src/tay1_order.vhd:133 (e = 17 - STAGE, may be negative). */
uint32_t bhw_taylor_pi_word(int e)
{
    return (uint32_t)round_nearest(ldexpq(M_PIq, e));
}
