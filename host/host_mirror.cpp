// host_mirror.cpp -- the reference's host programs, re-hosted on the C ABI (compute on the GPU).
//   coe <PW> <W>        : what main() of cpp/cordic_sincos.cpp writes to coe.dat ("%d %d\n" = s c, :135-138)
//   dout <sel> <PW> <W> : what hls/windows/window_test.cpp writes to dout.dat ("%d \n", :200)
//   golden <sel> <PW> <W>: what it writes to golden_dat.dat ("%d \n", :196,201): the rounded double-precision window -- host
//                         arithmetic of the testbench itself (no generator involved), so it runs without a GPU
//   stream              : win_selector driven like the testbench: RESET, then ENABLE in uneven bursts
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "bhw.hpp"

int main(int argc, char **argv)
{
    try {
        if (argc >= 4 && !strcmp(argv[1], "coe")) {
            const unsigned pw = atoi(argv[2]), w = atoi(argv[3]);
            std::vector<int32_t> s, c;
            bhw::cordic(BHW_MODEL_CPP, pw, w, 0, 1ull << pw, s, c);
            for (size_t i = 0; i < s.size(); ++i) printf("%d %d\n", s[i], c[i]);
            return 0;
        }
        if (argc >= 5 && !strcmp(argv[1], "dout")) {
            const int sel = atoi(argv[2]);
            const unsigned pw = atoi(argv[3]), w = atoi(argv[4]);
            for (int32_t v : bhw::win_function((char)sel, 0, 1ull << pw, pw, w)) printf("%d \n", v);
            return 0;
        }
        if (argc >= 5 && !strcmp(argv[1], "golden")) {
            // window_test.cpp:93-196: calc_dbl = a0 - a1 cos(2 pi i / N) + a2 cos(2 * 2 pi i / N) - ..., then
            // (win_t) round((2^(NWIDTH - shift) - 1) * calc_dbl), shift = 1 for 2/3/4 terms and 2 for 5/7 terms
            static const double coef[8][7] = {{0}, {0.5434783, 1.0 - 0.5434783}, {0.5, 0.5}, {0.21, 0.25, 0.04},
                                              {0.35875, 0.48829, 0.14128, 0.01168},
                                              {0.3232153788877343, 0.4714921439576260, 0.1755341299601972, 0.0284969901061499,
                                               0.0012613570882927}, {0},
                                              {0.271220360585039, 0.433444612327442, 0.218004122892930, 0.065785343295606,
                                               0.010761867305342, 0.000770012710581, 0.000013680883060}};
            static const int terms[8] = {0, 2, 2, 3, 4, 5, 0, 7};
            const int sel = atoi(argv[2]);
            const unsigned pw = atoi(argv[3]), w = atoi(argv[4]);
            if (sel < 1 || sel > 7 || !terms[sel]) return 64;
            const int shift = terms[sel] >= 5 ? 2 : 1;
            const double n = (double)(1ull << pw), scale = std::pow(2.0, (double)(w - shift)) - 1.0;
            for (unsigned long long i = 0; i < (1ull << pw); ++i) {
                double v = 0.0, sign = 1.0;
                for (int k = 0; k < terms[sel]; ++k, sign = -sign) v += sign * coef[sel][k] * std::cos((k * 2.0 * (double)i * M_PI) / n);
                const long long r = (long long)std::round(scale * v);
                const unsigned sh = 64 - w;                            // (win_t) store: wraps to NWIDTH bits
                printf("%d \n", (int)((long long)((unsigned long long)r << sh) >> sh));
            }
            return 0;
        }
        if (argc >= 2 && !strcmp(argv[1], "stream")) {
            bhw::win_selector sel(10, 24, "BH5TERM");
            sel.RESET();
            size_t total = 0;
            for (size_t burst : {1u, 7u, 500u, 516u, 1024u, 333u}) {   // wraps the 10-bit counter twice
                for (int32_t v : sel.ENABLE(burst)) printf("%d \n", v);
                total += burst;
            }
            fprintf(stderr, "%zu\n", total);
            return 0;
        }
        if (argc >= 2 && !strcmp(argv[1], "errors")) {
            int bad = 0;
            try { bhw::win_selector s(10, 16, "KAISER"); } catch (const bhw::error &e) { bad += e.code == BHW_ERR_BADARG; }
            try { bhw::win_selector s(26, 16, "BH4TERM"); } catch (const bhw::error &e) { bad += e.code == BHW_ERR_UNSUPPORTED; }
            {   // the selector hands SIN_TYPE only to the 2-/3-term entities (src/win_selector.vhd:137-199): BH4 stays CORDIC
                bhw::win_selector s(12, 16, "BH4TERM", "TAYLOR");
                bhw_params raw = s.params();
                raw.sin_type = BHW_SIN_TAYLOR;                 // the ABI itself refuses what the reference cannot build
                bad += (s.params().sin_type == BHW_SIN_CORDIC && bhw_params_validate(&raw) == BHW_ERR_UNSUPPORTED);
            }
            {   // ownership parts tile the window (host arithmetic)
                bhw::win_selector s(20, 32, "BH7TERM");
                uint64_t owned = 0;
                for (uint32_t g = 0; g < 4; ++g)
                    for (const bhw_segment &sg : s.SEGMENTS(g, 4)) owned += sg.count;
                bad += (owned == (1ull << 20)) ? 0 : 50;
            }
            auto z = bhw::win_function(6, 0, 8, 10, 16);   // unknown selector -> win_empty
            for (int32_t v : z) bad += (v == 0) ? 0 : 100;
            printf("%d\n", bad);
            return bad == 3 ? 0 : 1;
        }
    } catch (const bhw::error &e) {
        fprintf(stderr, "bhw error %d: %s\n", e.code, e.what());
        return 2;
    }
    fprintf(stderr, "usage: host_mirror coe PW W | dout SEL PW W | golden SEL PW W | stream | errors\n");
    return 64;
}
