// igemm_timeline.hip -- two tuning tools around csrc/igemm.hip built from this one source (never part of the library):
//   (1) -DFZ_IGEMM_TIMING: s_memtime totals per loop segment of waves 0 (group 0) and 4 (group 1) of workgroup 0, per K-loop form:
//       where the cycles of a phase go (counted wait / fragment-read + DMA issue / barrier 1 / LDS latency / MFMA issue / barrier 2);
//   (2) without it: within-process A/B of tile / loop variants, INTERLEAVED rounds (one launch of every variant per round, so that
//       clock drift and thermal state hit all variants alike), random operands, median and best TFLOP/s per variant.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -DFZ_IGEMM_TRIALS [-DFZ_IGEMM_TIMING]
//         -o build_tmp/igemm_{timeline,ab} scripts/igemm_timeline.hip
#include "../fatezero_amd/csrc/igemm.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

static unsigned rng_state = 12345;
static float rnd() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return ((rng_state >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f;
}
static _Float16* dev_random(size_t n, float scale) {
    std::vector<_Float16> h(n);
    for (auto& v : h) v = (_Float16)(rnd() * scale);
    _Float16* d;
    hipMalloc(&d, n * 2);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    return d;
}

struct Problem {
    char name[64];
    bool conv;
    bool geglu, temporal;
    int n, hw, cin, cout;     // conv
    int64_t rows; int k, o;   // gemm
    double flops;
    _Float16 *x, *w, *b, *y;
};

static int launch(const Problem& p0, int cfg, void* ws, int64_t ws_floats) {
    Problem p = p0;
    if (getenv("FZ_TIMELINE_NOBIAS")) p.b = nullptr;   // (what the bias fetch in the epilogue costs: run with and without)
    if (cfg < 0) {  // the library's own choice: -1 = ring tiles only, -2 = with the ping-pong substitution, -N (N >= 4) = also under
        fz_igemm_trial_no_pp = cfg == -1;                      // split-K when a K slice has at least N K-64 steps
        fz_igemm_trial_pp_splitk_min = cfg <= -4 ? -cfg : 0;
        cfg = 0;
    }
    if (p.temporal)
        return fz_temporal_conv3(p.x, p.w, nullptr, nullptr, nullptr, 0, p.y, p.n, p.hw, p.cin, p.cout, 8, ws, ws_floats, nullptr);
    if (p.conv)
        return fz_conv3x3(p.x, p.w, p.b, nullptr, 0, nullptr, p.y, p.n, p.hw, p.hw, p.cin, p.cout, 1, 0, p.n, ws, ws_floats, cfg, cfg ? 1 : 0, nullptr);
    FzGemmDesc d = {};
    d.rows = p.rows; d.in_features = p.k; d.out_features = p.o; d.ldx = p.k; d.ldw = p.k; d.ldy = p.o; d.batch = 1;
    d.tile_cfg = cfg; d.split_k = cfg ? 1 : 0; d.workspace_floats = ws_floats;
    if (p.geglu) {
        d.epilogue = FZ_GEMM_GEGLU;
        d.ldy = p.o / 2;
    }
    return fz_gemm(&d, p.x, p.w, p.b, nullptr, nullptr, p.y, p.geglu ? nullptr : ws, nullptr);
}

int main(int argc, char** argv) {
    std::vector<int> cfgs;
    for (int i = 1; i < argc; ++i)
        if (strcmp(argv[i], "prod") && strcmp(argv[i], "gemmsweep") && strcmp(argv[i], "shortk")) cfgs.push_back(atoi(argv[i]));
    const bool named = argc > 1 && (!strcmp(argv[1], "prod") || !strcmp(argv[1], "gemmsweep") || !strcmp(argv[1], "shortk"));
    if (cfgs.empty() && !named) cfgs = {254222, 254218, 1254218, 3254218, 5254218, 244222, 244218, 1244218};
    std::vector<Problem> probs;
    auto add_conv = [&](const char* name, int n, int hw, int cin, int cout) {
        Problem p = {};
        strncpy(p.name, name, 63); p.conv = true; p.n = n; p.hw = hw; p.cin = cin; p.cout = cout;
        p.flops = 2.0 * n * hw * hw * (double)cout * cin * 9;
        p.x = dev_random((size_t)n * hw * hw * cin, 1.0f);
        p.w = dev_random((size_t)cout * 9 * cin, 0.02f);
        p.b = dev_random(cout, 0.1f);
        hipMalloc(&p.y, (size_t)n * hw * hw * cout * 2);
        probs.push_back(p);
    };
    auto add_gemm = [&](const char* name, int64_t rows, int k, int o) {
        Problem p = {};
        strncpy(p.name, name, 63); p.conv = false; p.rows = rows; p.k = k; p.o = o;
        p.flops = 2.0 * rows * (double)k * o;
        p.x = dev_random((size_t)rows * k, 1.0f);
        p.w = dev_random((size_t)o * k, 0.03f);
        p.b = dev_random(o, 0.1f);
        hipMalloc(&p.y, (size_t)rows * o * 2);
        probs.push_back(p);
    };
    const bool prod = argc > 1 && !strcmp(argv[1], "prod");
    if (prod) {  // every conv / GEMM / temporal-conv shape of the 8-frame inversion and the 16-frame edit forward (SD-1.x at 512^2)
        if (cfgs.empty()) cfgs = {-1, -2};
        char nm[64];
        for (int n : {8, 16}) {
            const int cs[][3] = {{64, 320, 320}, {64, 640, 320}, {64, 960, 320}, {32, 320, 640}, {32, 640, 640}, {32, 960, 640}, {32, 1280, 640},
                                 {32, 1920, 640}, {16, 640, 1280}, {16, 1280, 1280}, {16, 1920, 1280}, {16, 2560, 1280}, {8, 1280, 1280}, {8, 2560, 1280}};
            for (auto& c : cs) {
                snprintf(nm, 64, "conv %2df %2d^2 %4d->%4d      ", n, c[0], c[1], c[2]);
                add_conv(nm, n, c[0], c[1], c[2]);
            }
            const int gs[][4] = {{4096, 320, 320, 0}, {4096, 320, 640, 0}, {4096, 320, 2560, 1}, {4096, 1280, 320, 0}, {1024, 640, 640, 0}, {1024, 640, 1280, 0},
                                 {1024, 640, 5120, 1}, {1024, 2560, 640, 0}, {256, 1280, 1280, 0}, {256, 1280, 2560, 0}, {256, 1280, 10240, 1}, {256, 5120, 1280, 0},
                                 {64, 1280, 1280, 0}, {64, 1280, 10240, 1}, {64, 5120, 1280, 0}};
            for (auto& c : gs) {
                snprintf(nm, 64, "gemm%s %6d x %4d -> %5d", c[3] ? " geglu" : "      ", n * c[0], c[1], c[2]);
                add_gemm(nm, (int64_t)n * c[0], c[1], c[2]);
                probs.back().geglu = c[3] != 0;
            }
            const int ts[][3] = {{4096, 320, 160}, {4096, 160, 320}, {1024, 640, 160}, {1024, 160, 640}, {256, 1280, 160}, {256, 160, 1280}, {64, 1280, 160}, {64, 160, 1280}};
            for (auto& c : ts) {
                snprintf(nm, 64, "tconv %2df %4d tok %4d->%4d  ", n, c[0], c[1], c[2]);
                Problem p = {};
                strncpy(p.name, nm, 63); p.temporal = true; p.n = n; p.hw = c[0]; p.cin = c[1]; p.cout = c[2];
                p.flops = 2.0 * n * c[0] * (double)c[1] * c[2] * 3;
                p.x = dev_random((size_t)n * c[0] * c[1], 1.0f);
                p.w = dev_random((size_t)c[2] * 3 * c[1], 0.03f);
                p.b = nullptr;
                hipMalloc(&p.y, (size_t)n * c[0] * c[2] * 2);
                probs.push_back(p);
            }
        }
    } else if (argc > 1 && !strcmp(argv[1], "shortk")) {  // the epilogue-dominated projections: GEGLU and the K = 320 / 640 GEMMs of the 64^2 / 32^2 levels
        char nm[64];
        for (int n : {8, 16}) {
            const int gs[][4] = {{4096, 320, 2560, 1}, {1024, 640, 5120, 1}, {256, 1280, 10240, 1}, {64, 1280, 10240, 1},
                                 {4096, 320, 320, 0}, {4096, 320, 640, 0}, {4096, 320, 960, 0}, {4096, 1280, 320, 0},
                                 {1024, 640, 640, 0}, {1024, 640, 1280, 0}, {1024, 640, 1920, 0}, {1024, 2560, 640, 0}};
            for (auto& c : gs) {
                snprintf(nm, 64, "gemm%s %6d x %4d -> %5d", c[3] ? " geglu" : "      ", n * c[0], c[1], c[2]);
                add_gemm(nm, (int64_t)n * c[0], c[1], c[2]);
                probs.back().geglu = c[3] != 0;
            }
        }
    } else if (argc > 1 && !strcmp(argv[1], "gemmsweep")) {  // the long-K / GEGLU shapes on which hipBLASLt is ahead (profiles/r03_kbench_gemm.json)
        char nm[64];
        const int gs[][4] = {{8192, 2560, 640, 0}, {16384, 2560, 640, 0}, {2048, 1280, 2560, 0}, {2048, 1280, 3840, 0}, {4096, 1280, 2560, 0},
                             {4096, 1280, 3840, 0}, {2048, 5120, 1280, 0}, {4096, 5120, 1280, 0}, {1024, 5120, 1280, 0}, {8192, 640, 5120, 1},
                             {16384, 640, 5120, 1}, {2048, 1280, 10240, 1}, {4096, 1280, 10240, 1}, {512, 1280, 10240, 1}};
        for (auto& c : gs) {
            snprintf(nm, 64, "gemm%s %6d x %4d -> %5d", c[3] ? " geglu" : "      ", c[0], c[1], c[2]);
            add_gemm(nm, c[0], c[1], c[2]);
            probs.back().geglu = c[3] != 0;
        }
    } else {
    add_conv("conv 16f 64^2 320->320 ", 16, 64, 320, 320);
    add_conv("conv  8f 64^2 320->320 ", 8, 64, 320, 320);
    add_conv("conv 16f 32^2 640->640 ", 16, 32, 640, 640);
    add_gemm("gemm 65536 x 1280 -> 1280", 65536, 1280, 1280);
    add_gemm("gemm  8192 x 2560 ->  640", 8192, 2560, 640);
    add_gemm("gemm  4096 x 1280 -> 3840", 4096, 1280, 3840);
    }
    const int64_t ws_floats = 64ll << 20;
    float* ws;
    hipMalloc(&ws, ws_floats * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
#ifdef FZ_IGEMM_TIMING
    printf("segments (s_memtime ticks per K step of a wave; ring loop: one step = K 64, ping-pong: one phase = K 16 [K 32 for the k32 form])\n");
    printf("  ring:      vmwait | dma-issue | barrier | - | reads+mfma issue\n  ping-pong: vmwait | reads+dma issue | barrier 1 | lds wait | mfma issue | barrier 2\n");
    for (const Problem& p : probs) {
        for (int cfg : cfgs) {
            if (launch(p, cfg, ws, ws_floats) != 0) continue;
            hipDeviceSynchronize();
            long long zero[2][8] = {};
            hipMemcpyToSymbol(HIP_SYMBOL(fz_igemm_timing), zero, sizeof(zero));
            hipEventRecord(e0);
            launch(p, cfg, ws, ws_floats);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long t[2][8], t2[2][6];
            hipMemcpyFromSymbol(t, HIP_SYMBOL(fz_igemm_timing), sizeof(t));
            hipMemcpyFromSymbol(t2, HIP_SYMBOL(fz_igemm_timing2), sizeof(t2));
            printf("%s cfg %8d: wave 0 of workgroup 0: set-up %lld ticks, K loop %lld ticks, whole kernel %lld ticks; launch %.1f us | wall clock: entry -> K loop %.2f us, K loop %.2f us, "
                   "epilogue %.2f us\n", p.name, cfg, t2[0][0], t[0][7], t2[0][1], ms * 1e3, (t2[0][3] - t2[0][2]) / 100.0, (t2[0][4] - t2[0][3]) / 100.0, (t2[0][5] - t2[0][4]) / 100.0);
            if (getenv("FZ_TIMELINE_BRIEF")) continue;
            for (int w = 0; w < 2; ++w) {
                const double n = t[w][6] > 0 ? (double)t[w][6] : 1.0;
                double sum = 0;
                for (int s = 0; s < 6; ++s) sum += t[w][s];
                printf("%s cfg %8d wave %d: steps %4lld | %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f | sum %7.0f ticks/step | %6.1f us launch (%.0f TF/s instrumented)\n",
                       p.name, cfg, w * 4, t[w][6], t[w][0] / n, t[w][1] / n, t[w][2] / n, t[w][3] / n, t[w][4] / n, t[w][5] / n, sum / n,
                       ms * 1e3, p.flops / ms / 1e9);
            }
        }
    }
#else
    const int ROUNDS = prod ? 7 : 15;
    std::vector<double> total_ms(cfgs.size(), 0.0);
    for (const Problem& p : probs) {
        std::vector<std::vector<float>> ms(cfgs.size());
        std::vector<bool> ok(cfgs.size());
        for (size_t v = 0; v < cfgs.size(); ++v) ok[v] = launch(p, cfgs[v], ws, ws_floats) == 0;  // warm-up + support check
        hipDeviceSynchronize();
        for (int r = 0; r < ROUNDS; ++r)
            for (size_t v = 0; v < cfgs.size(); ++v) {
                if (!ok[v]) continue;
                hipEventRecord(e0);
                for (int i = 0; i < 3; ++i) launch(p, cfgs[v], ws, ws_floats);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float t;
                hipEventElapsedTime(&t, e0, e1);
                ms[v].push_back(t / 3);
            }
        printf("%s", p.name);
        for (size_t v = 0; v < cfgs.size(); ++v) {
            if (!ok[v]) { printf(" | %8d     -      ", cfgs[v]); continue; }
            std::sort(ms[v].begin(), ms[v].end());
            total_ms[v] += ms[v][ms[v].size() / 2];
            printf(" | %8d %5.0f/%5.0f", cfgs[v], p.flops / ms[v][ms[v].size() / 2] / 1e9, p.flops / ms[v][0] / 1e9);
        }
        printf("   (median/best TF/s)\n");
    }
    printf("sum of the median launch times over the listed problems:");
    for (size_t v = 0; v < cfgs.size(); ++v) printf(" | %8d %8.3f ms", cfgs[v], total_ms[v]);
    printf("\n");
#endif
    return 0;
}
