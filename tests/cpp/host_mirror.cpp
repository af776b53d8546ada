// host_mirror.cpp -- the reference's host programs, re-hosted on the C ABI (compute on the GPU).
//   coe <PW> <W>        : what main() of cpp/cordic_sincos.cpp writes to coe.dat ("%d %d\n" = s c, :135-138)
//   dout <sel> <PW> <W> : what hls/windows/window_test.cpp writes to dout.dat ("%d \n", :200)
//   stream              : win_selector driven like the testbench: RESET, then ENABLE in uneven bursts
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
            auto z = bhw::win_function(6, 0, 8, 10, 16);   // unknown selector -> win_empty
            for (int32_t v : z) bad += (v == 0) ? 0 : 100;
            printf("%d\n", bad);
            return bad == 3 ? 0 : 1;
        }
    } catch (const bhw::error &e) {
        fprintf(stderr, "bhw error %d: %s\n", e.code, e.what());
        return 2;
    }
    fprintf(stderr, "usage: host_mirror coe PW W | dout SEL PW W | stream | errors\n");
    return 64;
}
