// san_plan.cpp -- the product's host-side planning arithmetic (blackman_harris_win_amd/csrc/bhw_plan.cpp: no HIP in it) swept over
// the whole parameter lattice under AddressSanitizer + UBSan (tests/test_sanitizers.py builds both with -fsanitize=address,undefined;
// GPU sanitizers are not available on the pool).  Besides "no report", every step checks an invariant the launch code relies on.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "bhw_plan.h"

extern "C" {
uint64_t bhw_workspace_bytes_ex(const bhw_params *p, uint64_t n0, uint64_t count, const bhw_exec *ex);
int bhw_dbg_table_format_verdict(const bhw_params *p, uint32_t dlog, int set);
}

static long g_checks = 0;
#define REQUIRE(cond, ...) do { ++g_checks; if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } } while (0)

static const uint32_t kWins[6] = {BHW_WIN_HAMMING, BHW_WIN_HANN, BHW_WIN_BH3, BHW_WIN_BH4, BHW_WIN_BH5, BHW_WIN_BH7};

// parts: 0 = skip the ownership arithmetic (it does not depend on the cosine-sum rule or the weights), 1 = ten part counts, 2 = every count 1..64
static void sweep_config(bhw_params &p, int parts)
{
    const int rc = bhw_params_validate(&p);
    REQUIRE(rc == BHW_OK || rc == BHW_ERR_BADARG || rc == BHW_ERR_UNSUPPORTED, "rc %d", rc);
    if (rc) { REQUIRE(bhw_last_error()[0] != 0, "no error text"); return; }
    const uint64_t N = 1ull << p.phi_width;
    char buf[256], tiny[1], small[16];
    if (p.sin_type != BHW_SIN_CORDIC) {
        REQUIRE(bhw_describe_plan(&p, 0, N, nullptr, buf, sizeof buf) == BHW_OK, "describe (taylor)");
        REQUIRE(bhw_workspace_bytes(&p, 0, N, BHW_ALGO_AUTO) == 0, "taylor needs no scratch");
        return;
    }
    BhwCordicCfg c;
    bhwp_resolve_cordic(&p, c);
    BhwWinCfg w;
    bhwp_resolve_window(&p, w);
    REQUIRE(c.n_iter >= 7 && c.n_iter <= 32 && c.z_shl < 64 && c.z_shr < 32 && c.phi_width - 2 >= c.z_shr, "shifts: n_iter %u z_shl %u z_shr %u", c.n_iter, c.z_shl, c.z_shr);
    // the 32-bit z of the shared-prefix / fused / mirror kernels: whenever the state fits 34 bits the quarter circle fits 2^32
    if (c.dat_width + c.out_shr <= 34)
        REQUIRE(c.z_shl < 32 && (((uint64_t)1 << (c.phi_width - 2 - c.z_shr)) << c.z_shl) <= (1ull << 32) && (uint64_t)c.lut[0] < (1ull << 32), "quarter circle beyond 2^32: z_shl %u", c.z_shl);
    for (uint32_t k = 0; k < 32; ++k) REQUIRE(c.lut[k] >= 0 && (k == 0 || c.lut[k] <= c.lut[k - 1]), "ROM word %u", k);
    REQUIRE(c.x0 > 0, "gain");
    const uint64_t E = bhwp_table_entries(c);
    REQUIRE(E >= 1 && E <= (1ull << 28), "entries %" PRIu64, E);
    const uint64_t shapes[][2] = {{0, N}, {0, N / 2}, {5, N + 7}, {N - 1, 2}, {0, 3 * N}, {N / 8, N / 4}, {3 * (N / 8), N / 2}, {0, 1}};
    for (const auto &sh : shapes)
        for (uint32_t algo = 0; algo <= 3; ++algo) {
            const uint32_t a = bhwp_pick_algo(&p, c, w, sh[0], sh[1], algo);
            REQUIRE(a == BHW_ALGO_DIRECT || a == BHW_ALGO_TABLE || a == BHW_ALGO_FUSED, "algo %u", a);
            if (a == BHW_ALGO_FUSED) REQUIRE(bhwk_fold_direct_applicable(c) && bhwp_has_whole_period(&p, sh[0], sh[1]), "fused without a whole period");
            bhw_exec ex;
            memset(&ex, 0, sizeof ex);
            ex.struct_size = sizeof ex;
            ex.algo = algo;
            for (uint32_t limit = 0; limit <= (algo == BHW_ALGO_TABLE ? (uint32_t)BHW_TABLE_NIBBLE_ESC : 0u); ++limit) {
                ex.table_format = limit;
                const uint64_t tight = bhw_workspace_bytes_ex(&p, sh[0], sh[1], &ex), bound = bhw_workspace_bytes(&p, sh[0], sh[1], algo);
                REQUIRE(tight <= bound && (bound == 0 || bound == E * 8), "scratch %" PRIu64 " > bound %" PRIu64, tight, bound);
                REQUIRE((a == BHW_ALGO_TABLE) == (tight != 0), "scratch of a table call");
            }
            REQUIRE(bhw_describe_plan(&p, sh[0], sh[1], &ex, buf, sizeof buf) == BHW_OK && strlen(buf) < sizeof buf, "describe");
            if (algo == 0) {
                REQUIRE(bhw_describe_plan(&p, sh[0], sh[1], &ex, small, sizeof small) == BHW_OK && strlen(small) < sizeof small, "describe (16 bytes)");
                REQUIRE(bhw_describe_plan(&p, sh[0], sh[1], &ex, tiny, 1) == BHW_OK && tiny[0] == 0, "describe (1 byte)");
            }
            const BhwTableCall t = bhwp_table_call(&p, c, w, sh[0], sh[1], false);
            REQUIRE(!(t.images && t.has_period) && (!t.images || (t.img_mask != 0 && t.img_mask != 0xFFu)), "image subset %x", t.img_mask);
        }
    // table formats: candidates, layouts, scratch under every verdict state
    for (int tiled = 0; tiled < 2; ++tiled) {
        if (tiled && !bhwk_tile_applicable(c, w)) continue;
        BhwCordicCfg ct = c;
        ct.tab_split = (tiled && c.z_shr == 0) ? 1u : 0u;
        for (uint32_t limit = 0; limit <= BHW_TABLE_NIBBLE_ESC; ++limit) {
            uint32_t cand[kMaxFormats];
            const int n = bhwp_table_format_candidates(ct, tiled != 0, limit, cand);
            REQUIRE(n >= 1 && n <= kMaxFormats && cand[n - 1] == 0, "candidates %d", n);
            uint64_t prev = 0;
            for (int i = 0; i < n; ++i) {
                const BhwTableLayout l = bhwp_table_layout(E, cand[i]);
                REQUIRE(l.bytes <= E * 8 || E < 64, "format %u needs %" PRIu64 " > 8E", cand[i], l.bytes);
                REQUIRE(cand[i] == 0 || (l.coarse_off < l.check_off && l.check_off + 8 <= l.bytes && (l.coarse_off & 255) == 0 && (l.check_off & 255) == 0), "layout of %u", cand[i]);
                REQUIRE(i == 0 || l.bytes >= prev, "candidates not narrowest first");
                if (fmt_of(cand[i]) == 5) {                       // escape tables: one per build workgroup, between the records and the check word
                    const uint64_t n_wg = ((E >> 1) + (1ull << l.esc_wg_log) - 1) >> l.esc_wg_log;
                    REQUIRE(l.esc_off >= l.coarse_off + (E >> fmt_cell_log(cand[i])) * 16 && (l.esc_off & 255) == 0 && l.esc_off + n_wg * kEscSlots * 16 <= l.check_off,
                            "escape tables of %u", cand[i]);
                    REQUIRE((1u << (l.esc_wg_log - 6)) == bhwk_build_mirror_threads((uint32_t)E) / 4u, "escape tables per build workgroup");
                } else REQUIRE(l.esc_off == 0, "escape tables in format %u", cand[i]);
                prev = l.bytes;
                if (cand[i]) {
                    BhwCordicCfg cf = ct;
                    cf.tab_dlog = cand[i];
                    REQUIRE(fmt_of(cand[i]) == 1 || bhwk_build_mirror_applies(table_layout(cf), (uint32_t)E), "format %u proposed without a build kernel", cand[i]);
                    char b1[64], b2[64];
                    bhwk_describe_table(cf, w, tiled != 0, false, b1, b2, sizeof b1);
                    bhwk_describe_table(cf, w, tiled != 0, true, b1, b2, sizeof b1);
                }
            }
            for (int state = 0; state < 3; ++state) {             // every packed format unknown / exact / overflowing
                for (int i = 0; i + 1 < n; ++i) bhw_dbg_table_format_verdict(&p, cand[i], state == 0 ? 3 : state);   // 3: any value but 1 / 2 reads as "unknown"
                for (int cap = 0; cap < 2; ++cap) {
                    const uint64_t need = bhwp_table_scratch_bytes(&p, ct, tiled != 0, limit, cap != 0);
                    REQUIRE(need >= bhwp_table_layout(E, state == 1 ? cand[0] : 0).bytes || (state == 0 && !cap), "scratch %" PRIu64 " in state %d", need, state);
                    REQUIRE(need <= E * 8 || E < 64, "scratch beyond the bound");
                }
            }
        }
    }
    if (bhwk_tile_applicable(c, w)) {
        BhwTilePlan tp;
        int nb;
        uint32_t lanes;
        bhwp_tile_plan(c, w, tp, nb, lanes);
        const uint32_t H = 1u << (c.phi_width - 3);
        REQUIRE((nb == 1 || nb == 3 || nb == 15) && (lanes == (uint32_t)kTileLanes || lanes == (uint32_t)kTileThreads) && tp.n_tiles >= 1, "tile plan nb %d", nb);
        for (int i = 0; i < 16; ++i) REQUIRE(tp.offs[i] < H, "run offset %d", i);
        REQUIRE((uint64_t)tp.n_tiles * lanes * (uint64_t)nb >= H, "tiles do not cover the ring");
        (void)bhwp_tile_fast(c, w, nb);
    }
    if (bhwk_fold_direct_applicable(c)) {
        for (uint64_t total : {64ull, 1ull << 13, 1ull << 15, (1ull << 15) + 64, 1ull << 18, 1ull << 20, 1ull << 27}) {
            const int f = bhwp_fold_form(c, w, total);
            REQUIRE(f >= 0 && f <= 3, "fold form %d", f);
        }
        const uint32_t k24 = bhwp_fold_k24(c);
        REQUIRE(k24 >= 1 && k24 <= 32, "k24 %u", k24);
    }
    (void)bhwk_runlength_applicable(c, w, nullptr);
    // ownership parts: the segments of the parts of one window are sorted, inside the window, and cover it
    const uint32_t few[] = {1, 2, 3, 4, 5, 7, 8, 16, 33, 64};
    std::vector<bhw_segment> segs(256), all;
    for (uint32_t gi = 0; gi < (parts == 2 ? 64u : parts == 1 ? 10u : 0u); ++gi) {
        const uint32_t G = parts == 2 ? gi + 1 : few[gi];
        uint32_t n = 0;
        const int r0 = bhw_part_segments(&p, 0, G, segs.data(), 256, &n);
        REQUIRE(r0 == BHW_OK || r0 == BHW_ERR_UNSUPPORTED, "part_segments rc %d", r0);
        if (r0) break;
        all.clear();
        for (uint32_t g = 0; g < G; ++g) {
            REQUIRE(bhw_part_segments(&p, g, G, segs.data(), 256, &n) == BHW_OK && n <= 256, "part %u of %u", g, G);
            for (uint32_t i = 0; i < n; ++i) {
                REQUIRE(segs[i].count > 0 && segs[i].n0 + segs[i].count <= N && (i == 0 || segs[i].n0 > segs[i - 1].n0 + segs[i - 1].count), "segment %u of part %u / %u", i, g, G);
                all.push_back(segs[i]);
            }
            BhwFoldRun runs[32];
            uint32_t t0, tc;
            const int nr = bhwk_part_runs(c, w, g, G, runs, &t0, &tc);
            REQUIRE(nr >= 0 && nr <= 32, "runs %d", nr);
            int prc;
            for (uint32_t algo = 0; algo <= 3; ++algo) (void)bhwp_part_fused(&p, c, runs, nr, tc, algo, &prc);
        }
        // union == [0, N): sort by start, sweep
        for (size_t i = 1; i < all.size(); ++i)
            for (size_t j = i; j > 0 && all[j - 1].n0 > all[j].n0; --j) { bhw_segment t = all[j]; all[j] = all[j - 1]; all[j - 1] = t; }
        uint64_t covered = 0;
        for (const bhw_segment &s : all) {
            REQUIRE(s.n0 <= covered, "gap before %" PRIu64 " (%u parts)", s.n0, G);
            if (s.n0 + s.count > covered) covered = s.n0 + s.count;
        }
        REQUIRE(covered == N, "parts cover %" PRIu64 " of %" PRIu64, covered, N);
    }
    REQUIRE(bhw_part_segments(&p, 1, 1, nullptr, 0, nullptr) != BHW_OK && bhw_part_segments(&p, 0, 65, nullptr, 0, nullptr) != BHW_OK, "bad part numbers accepted");
}

int main()
{
    long configs = 0;
    // the window lattice: every model / rule / source / term count / phase width / data width
    for (uint32_t model = 0; model <= BHW_MODEL_SCALED; ++model)
        for (uint32_t combine = 0; combine <= BHW_COMBINE_VHDL; ++combine)
            for (uint32_t sin_type = 0; sin_type <= BHW_SIN_TAYLOR_ALL; ++sin_type)
                for (uint32_t wi = 0; wi < 6; ++wi)
                    for (uint32_t pw = 3; pw <= 31; ++pw)
                        for (uint32_t W = 7; W <= 33; ++W) {
                            if (sin_type && (model || (W % 4 != 0 && W != 19 && W != 18))) continue;   // Taylor ignores the CORDIC model
                            bhw_params p;
                            const int rc = bhw_params_init(&p, kWins[wi], pw, W);
                            if (rc) { REQUIRE(pw < 4 || pw > 30 || W < 8 || W > 32 || pw > W + 2, "init failed at %u/%u", pw, W); if (pw < 4 || pw > 30 || W < 8 || W > 32) continue; }
                            p.model = model;
                            p.combine = combine;
                            p.sin_type = sin_type;
                            const uint32_t precs[3] = {1, 3, 7}, luts[3] = {1, 9, 16};
                            for (int v = 0; v < (model == BHW_MODEL_VHDL || sin_type ? 3 : 1); ++v) {
                                p.precision = precs[v];
                                p.lut_size = luts[v];
                                const int parts = (combine != 0 || sin_type != 0 || v != 0) ? 0 : ((kWins[wi] == BHW_WIN_BH4 || kWins[wi] == BHW_WIN_BH7) && (W == 16 || W == 32)) ? 2 : (W % 4 == 0) ? 1 : 0;
                                sweep_config(p, parts);
                                ++configs;
                                // caller-scaled weights at the edges of the one-instruction products and of the one-word sums
                                if (sin_type == 0 && (W == 32 || W == 12) && v == 0) {
                                    bhw_params q = p;
                                    for (int k = 0; k < 7; ++k) q.aa[k] = (k & 1) ? INT32_MIN : INT32_MAX;
                                    sweep_config(q, 0);
                                    for (int k = 0; k < 7; ++k) q.aa[k] = (int32_t)((1u << (W - 3)) - (uint32_t)(k & 1));
                                    sweep_config(q, 0);
                                    configs += 2;
                                }
                            }
                        }
    // malformed input
    bhw_params p;
    REQUIRE(bhw_params_validate(nullptr) == BHW_ERR_BADARG && bhw_params_init(nullptr, 7, 26, 32) == BHW_ERR_BADARG && bhw_params_init(&p, 6, 26, 32) == BHW_ERR_BADARG, "NULL / unknown window");
    bhw_params_init(&p, 7, 26, 32);
    p.struct_size = 8;
    REQUIRE(bhw_params_validate(&p) == BHW_ERR_BADARG, "struct_size");
    bhw_params_init(&p, 7, 26, 32);
    for (uint32_t n_terms = 0; n_terms <= 9; ++n_terms) { p.n_terms = n_terms; (void)bhw_params_validate(&p); }
    bhw_params_init(&p, 7, 26, 32);
    bhw_exec ex;
    memset(&ex, 0, sizeof ex);
    for (uint32_t sz : {0u, 8u, 32u, 40u, 48u}) { ex.struct_size = sz; (void)bhwp_check_exec(&ex); (void)bhwp_exec_table_format(&ex); }
    ex.struct_size = sizeof ex;
    ex.table_format = 9;
    REQUIRE(bhwp_check_exec(&ex) == BHW_ERR_BADARG, "table_format 9");
    // weights: float -> integer at every width, values that do not fit
    int32_t aa[7];
    const double big[7] = {1e30, -1e30, 3.0, -3.0, 0.5, 1e300, -1e300}, nan7[7] = {0.0 / 1.0, __builtin_nan(""), 0, 0, 0, 0, 0};
    for (uint32_t W = 0; W <= 40; ++W)
        for (uint32_t wi = 0; wi < 6; ++wi) {
            (void)bhw_coeffs_from_float(kWins[wi], W, nullptr, aa);
            (void)bhw_coeffs_from_float(kWins[wi], W, big, aa);
            (void)bhw_coeffs_from_float(kWins[wi], W, nan7, aa);
            for (uint32_t preset = 0; preset <= 9; ++preset) {
                uint32_t wt;
                double a[7];
                (void)bhw_coeffs_preset(preset, W, &wt, a, aa);
            }
        }
    int64_t tab[48], gains[2];
    REQUIRE(bhw_constant_tables(0, tab, gains) == BHW_OK && bhw_constant_tables(1, tab, nullptr) == BHW_OK && bhw_constant_tables(2, tab, gains) == BHW_ERR_BADARG, "constant tables");
    // the variant generators and atan2
    for (uint32_t model = BHW_MODEL_DDS48; model <= BHW_MODEL_SCALED; ++model)
        for (uint32_t pw = 4; pw <= 30; ++pw)
            for (uint32_t W = 8; W <= 32; ++W) {
                bhw_params q;
                bhw_params_init(&q, 7, pw, W);
                q.model = model;
                if (bhwp_validate(&q, true)) continue;
                BhwPrerotCfg c;
                bhwp_resolve_prerot(&q, c);
                REQUIRE(c.size >= 15 && c.size <= 48 && c.dwph >= c.size && c.dwph <= 48 && c.gain > 0, "prerot %u/%u", pw, W);
            }
    for (uint32_t prec = 0; prec <= 8; ++prec)
        for (uint32_t in = 0; in <= 33; ++in)
            for (uint32_t ang = 3; ang <= 33; ++ang) {
                bhw_atan2_params a{(uint32_t)sizeof(bhw_atan2_params), prec, in, ang};
                if (bhwp_validate_atan2(&a)) continue;
                BhwAtan2Cfg c;
                bhwp_resolve_atan2(&a, c);
            }
    for (int code = -6; code <= 1; ++code) REQUIRE(bhw_strerror(code) != nullptr, "strerror");
    printf("ok %ld configurations, %ld checks\n", configs, g_checks);
    return 0;
}
