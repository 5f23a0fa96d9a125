// flash_ab.hip -- within-process A/B of attn_flash_kernel variants on the 64x64 SD level (d = 40, Lq 4096, Lk 8192,
// 8 and 16 frames x 8 heads), interleaved rounds, TF/s priced at the full 4 Lq (2 Lkf) C per frame whatever the kernel skips, uniform random [-1.5, 1.5) operands, outputs cross-checked.
// Tuning tool, never part of the library (the product's variant choice lives in fz_attn_flash_dispatch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -o build_tmp/flash_ab scripts/flash_ab.hip
#define FZ_FLASH_NO_DISPATCH 1
#include "../fatezero_amd/csrc/attn_flash.hip"
#ifdef FLASH_AB_OLD
#undef FQBLK
#undef FKVBLK
#undef FVSTR
#undef FZ_TICK
#include "flash_old_r02v1.inc"  // the kernel as of commit 13e2b00, renamed *_old
#endif
#include <stdio.h>
#include <algorithm>
#include <vector>

typedef int (*LaunchFn)(const FzAttnSelfDesc&, const void*, const void*, const void*, void*, void*);
struct Variant {
    const char* name;
    LaunchFn fn;
    bool distinct;  // kv slots [-1, 'last'] instead of [-1, 'first']: no frame has coinciding slots (every frame reads 2 x Lkf keys)
    bool k_head_major = false;  // K as a contiguous [frame][head][key][d] tensor (a key tile = 5 KB contiguous) instead of a slice of q | k rows
};

int main(int argc, char** argv) {
    const int only = argc > 1 ? atoi(argv[1]) : -1;  // PMC runs: one variant, 8 frames, a few launches
    const Variant vars[] = {
#ifdef FLASH_AB_OLD
        {"r02 v1 kernel (commit 13e2b00) <40,W2,QB2,bias>     ", launch_flash_old<40, 2, 2>, false},
#endif
        {"ring2 [-1,first]: frames 0,1 single-source <40,W2,QB2>", launch_flash<40, 2, 2, true, 2>, false},
        {"ring2 [-1,last ]: all frames two sources   <40,W2,QB2>", launch_flash<40, 2, 2, true, 2>, true},
        {"QB1 W4 [-1,first]                          <40,W4,QB1>", launch_flash<40, 4, 1, true, 2>, false},
        {"QB1 W4 [-1,last ]                          <40,W4,QB1>", launch_flash<40, 4, 1, true, 2>, true},
#ifdef FLASH_AB_KHM
        {"ring2 [-1,first], K head-major [n][h][key][d]  <40,W2,QB2>", launch_flash<40, 2, 2, true, 2>, false, true},
        {"ring2 [-1,last ], K head-major [n][h][key][d]  <40,W2,QB2>", launch_flash<40, 2, 2, true, 2>, true, true},
#endif
#ifdef FLASH_AB_POLY  // exp split: NPOLY of every 32 exponentials on packed FMAs (fz_exp2_poly2) instead of the transcendental unit
        {"exp split  4 / 32 polynomial [-1,first]    <40,W2,QB2>", launch_flash<40, 2, 2, true, 2, 4>, false},
        {"exp split  8 / 32 polynomial [-1,first]    <40,W2,QB2>", launch_flash<40, 2, 2, true, 2, 8>, false},
        {"exp split 16 / 32 polynomial [-1,first]    <40,W2,QB2>", launch_flash<40, 2, 2, true, 2, 16>, false},
        {"exp split 32 / 32 polynomial [-1,first]    <40,W2,QB2>", launch_flash<40, 2, 2, true, 2, 32>, false},
#endif
#ifdef FLASH_AB_ALL
        {"ring4 (barrier per two tiles) <40,W2,QB2,bias,4>", launch_flash<40, 2, 2, true, 4>, false},
        {"no bias slot ring2            <40,W2,QB2,fma ,2>", launch_flash<40, 2, 2, false, 2>, false},
#endif
    };
    const int NV = sizeof(vars) / sizeof(vars[0]);
    const int H = 8, L = 4096, D = 40, C = H * D;
    for (int F : {8, 16}) {
        if (only >= 0 && F != 8) continue;
        FzAttnSelfDesc d = {};
        d.n_frames = F; d.frame0 = 0; d.clip_len = 8; d.heads = H; d.head_dim = D; d.lq = L; d.lkf = L; d.n_kv = 2;
        d.kv_abs[0] = 0; d.kv_val[0] = -1; d.kv_abs[1] = 1; d.kv_val[1] = 0;
        d.scale = 0.158113883f; d.mode = 0; d.q_log2_scaled = 1;
        d.q_frame_stride = (int64_t)L * 2 * C; d.q_row_stride = 2 * C;
        d.k_frame_stride = (int64_t)L * 2 * C; d.k_row_stride = 2 * C;
        d.vt_frame_stride = (int64_t)C * L; d.vt_chan_stride = L;
        d.o_frame_stride = (int64_t)L * C; d.o_row_stride = C;
        const size_t nqk = (size_t)F * L * 2 * C, nv = (size_t)F * C * L, no = (size_t)F * L * C;
        std::vector<_Float16> hqk(nqk), hv(nv);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; };
        // q is in the log2 domain: a 1.5-scaled operand times scale * log2(e)
        for (size_t i = 0; i < nqk; ++i) {
            const bool is_q = (i % (2 * C)) < (size_t)C;
            hqk[i] = (_Float16)(rnd() * 1.5f * (is_q ? 0.158113883f * 1.44269504f : 1.0f));
        }
        for (auto& x : hv) x = (_Float16)rnd();
        _Float16 *qk, *vt, *o[12], *khm;
        hipMalloc(&qk, nqk * 2); hipMalloc(&vt, nv * 2); hipMalloc(&khm, (size_t)F * L * C * 2);
        {   // the same K values, head-major
            std::vector<_Float16> hk((size_t)F * L * C);
            for (int n = 0; n < F; ++n)
                for (int l = 0; l < L; ++l)
                    for (int c = 0; c < C; ++c)
                        hk[(((size_t)n * H + c / D) * L + l) * D + c % D] = hqk[((size_t)n * L + l) * 2 * C + C + c];
            hipMemcpy(khm, hk.data(), hk.size() * 2, hipMemcpyHostToDevice);
        }
        for (int v = 0; v < NV; ++v) { hipMalloc(&o[v], no * 2); hipMemset(o[v], 0, no * 2); }
        hipMemcpy(qk, hqk.data(), nqk * 2, hipMemcpyHostToDevice);
        hipMemcpy(vt, hv.data(), nv * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int ROUNDS = 7, REP = 5;
        std::vector<std::vector<float>> ms(NV);
        for (int r = 0; r < ROUNDS; ++r)
            for (int v = 0; v < NV; ++v) {
                if (only >= 0 && (v != only || r > 1)) { ms[v].push_back(1.0f); continue; }
                hipEventRecord(e0);
                FzAttnSelfDesc dv = d;
                if (vars[v].distinct) dv.kv_val[1] = d.clip_len - 1;
                const _Float16* kp = qk + C;
                if (vars[v].k_head_major) {
                    dv.k_frame_stride = (int64_t)H * L * D; dv.k_row_stride = D; dv.k_head_stride = (int64_t)L * D;
                    kp = khm;
                }
                for (int i = 0; i < REP; ++i) vars[v].fn(dv, qk, kp, vt, o[v], nullptr);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                float t; hipEventElapsedTime(&t, e0, e1);
                if (r > 0) ms[v].push_back(t / REP);
            }
        const double flops = 4.0 * L * (2.0 * L) * C * F;
        std::vector<_Float16> ref(no), got(no);
        hipMemcpy(ref.data(), o[0], no * 2, hipMemcpyDeviceToHost);
        for (int v = 0; v < NV; ++v) {
            std::sort(ms[v].begin(), ms[v].end());
            hipMemcpy(got.data(), o[v], no * 2, hipMemcpyDeviceToHost);
            double maxd = 0, maxa = 0;
            for (size_t i = 0; i < no; ++i) {
                maxd = std::max(maxd, (double)fabsf((float)got[i] - (float)ref[i]));
                maxa = std::max(maxa, (double)fabsf((float)ref[i]));
            }
            const double med = ms[v][ms[v].size() / 2], mn = ms[v][0];
            printf("F=%2d %-56s median %.4f ms %7.1f TF/s | min %.4f ms %7.1f TF/s | max|o - o[0]| %.2e (max|o| %.3f)\n", F,
                   vars[v].name, med, flops / med / 1e9, mn, flops / mn / 1e9, maxd, maxa);
        }
        hipFree(qk); hipFree(vt);
        for (int v = 0; v < NV; ++v) hipFree(o[v]);
    }
    return 0;
}
