/* Sweeps the oracle (and the product's host-side ROM derivation) under AddressSanitizer + UBSan on the CPU.
 * upstream models A/B read one entry past their rescaled ROM in the last rotation; the restatement
 * must not.  Built and run by tests/test_sanitizers.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "bhw_oracle.h"

void bhw_taylor_rom(uint32_t dat_width, uint32_t lut_size, int32_t *rom_sin_cos);
uint32_t bhw_taylor_pi_word(int e);

int main(void)
{
    int32_t *out = malloc(8192 * 4), *s = malloc(8192 * 4), *c = malloc(8192 * 4), *rom = malloc((2u << 12) * 4);
    const int wins[] = {1, 2, 3, 4, 5, 7};
    unsigned long long total = 0;
    for (int m = 0; m < 3; m++)
        for (int cb = 0; cb < 2; cb++)
            for (int wi = 0; wi < 6; wi++)
                for (int w = 8; w <= 32; w += 3)
                    for (int pw = 4; pw <= 26; pw += 5) {
                        bhwo_params p = {0};
                        p.model = m; p.combine = cb;
                        p.n_terms = (wins[wi] == 1 || wins[wi] == 2) ? 2 : wins[wi];
                        p.phi_width = pw; p.dat_width = w; p.precision = 1 + (w % 3); p.lut_size = 9;
                        bhwo_coeffs_from_float(wins[wi], w, 0, p.aa);
                        if (bhwo_generate(&p, (1ull << pw) - 100, 300, out) == 0) total += 300;
                        bhwo_sincos(&p, 5, 200, s, c);
                        if (p.n_terms <= 3) {
                            p.sin_type = 1;
                            for (int L = 1; L <= 12; L += 2) {
                                p.lut_size = L;
                                if (bhwo_generate(&p, 3, 200, out) == 0) total += 200;
                            }
                        }
                    }
    for (unsigned w = 8; w <= 32; w += 8)
        for (unsigned L = 1; L <= 12; L += 3) bhw_taylor_rom(w, L, rom);
    for (int e = -10; e <= 17; ++e) total += bhw_taylor_pi_word(e) != 0;
    free(out); free(s); free(c); free(rom);
    printf("ok %llu\n", total);
    return 0;
}
