// self_ab.hip -- timing of the capture / inject kernel and of its ablations (no pass 1 / no map stores / no P.V) on the
// 32x32 SD level (d = 80, Lq 1024, Lk 2048, 8 frames x 8 heads: one 268 MB map per launch).  Tuning tool, never part of
// the library.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -o build_tmp/self_ab scripts/self_ab.hip
#define FZ_SELF_NO_ENTRY 1
#include "../fatezero_amd/csrc/attn_self.hip"
#include <stdio.h>
#include <algorithm>
#include <vector>

template <int MODE, int ABL>
static void run(const FzAttnSelfDesc& d, const half_t* q, const half_t* k, const half_t* vt, half_t* o, half_t* p, const float* mask) {
    const int nq = (d.lq + QBLK - 1) / QBLK;
    dim3 grid(nq * d.heads * d.n_frames), block(256);
    hipLaunchKernelGGL((attn_self_kernel<80, MODE, ABL>), grid, block, 0, 0, d, q, k, vt, o, p, mask);
}

int main() {
    const int F = 8, H = 8, L = 1024, D = 80, C = H * D;
    FzAttnSelfDesc d = {};
    d.n_frames = F; d.frame0 = 0; d.clip_len = 8; d.heads = H; d.head_dim = D; d.lq = L; d.lkf = L; d.n_kv = 2;
    d.kv_abs[0] = 0; d.kv_val[0] = -1; d.kv_abs[1] = 1; d.kv_val[1] = 0;
    d.scale = 0.1118034f; d.q_log2_scaled = 0;
    d.q_frame_stride = (int64_t)L * 2 * C; d.q_row_stride = 2 * C;
    d.k_frame_stride = (int64_t)L * 2 * C; d.k_row_stride = 2 * C;
    d.vt_frame_stride = (int64_t)C * L; d.vt_chan_stride = L;
    d.o_frame_stride = (int64_t)L * C; d.o_row_stride = C;
    d.p_row_stride = 2 * L; d.p_head_stride = (int64_t)L * 2 * L; d.p_frame_stride = (int64_t)H * L * 2 * L;
    const size_t nqk = (size_t)F * L * 2 * C, nv = (size_t)F * C * L, no = (size_t)F * L * C, np = (size_t)F * H * L * 2 * L;
    std::vector<_Float16> hqk(nqk), hv(nv);
    std::vector<float> hm((size_t)F * L);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; };
    for (auto& x : hqk) x = (_Float16)(rnd() * 1.5f);
    for (auto& x : hv) x = (_Float16)rnd();
    for (auto& x : hm) x = rnd() > 0.0f ? 1.0f : 0.0f;
    _Float16 *qk, *vt, *o, *p; float* mask;
    hipMalloc(&qk, nqk * 2); hipMalloc(&vt, nv * 2); hipMalloc(&o, no * 2); hipMalloc(&p, np * 2); hipMalloc(&mask, hm.size() * 4);
    hipMemcpy(qk, hqk.data(), nqk * 2, hipMemcpyHostToDevice);
    hipMemcpy(vt, hv.data(), nv * 2, hipMemcpyHostToDevice);
    hipMemcpy(mask, hm.data(), hm.size() * 4, hipMemcpyHostToDevice);
    struct V { const char* name; void (*fn)(const FzAttnSelfDesc&, const half_t*, const half_t*, const half_t*, half_t*, half_t*, const float*); bool masked; };
    const V vars[] = {
        {"capture (shipped)", run<FZ_ATTN_CAPTURE, 0>, false}, {"capture, no pass 1", run<FZ_ATTN_CAPTURE, 1>, false},
        {"capture, no map stores", run<FZ_ATTN_CAPTURE, 2>, false}, {"capture, no pass 1, no stores", run<FZ_ATTN_CAPTURE, 3>, false},
        {"capture, no P.V", run<FZ_ATTN_CAPTURE, 4>, false}, {"capture, pass 1 only-ish (no stores, no P.V)", run<FZ_ATTN_CAPTURE, 6>, false},
        {"inject, no mask (shipped)", run<FZ_ATTN_INJECT, 0>, false}, {"inject, 50 % mask (shipped)", run<FZ_ATTN_INJECT, 0>, true},
        {"inject, no mask, no P.V", run<FZ_ATTN_INJECT, 4>, false},
    };
    const int NV = sizeof(vars) / sizeof(vars[0]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int ROUNDS = 6, REP = 5;
    std::vector<std::vector<float>> ms(NV);
    for (int r = 0; r < ROUNDS; ++r)
        for (int v = 0; v < NV; ++v) {
            hipEventRecord(e0);
            for (int i = 0; i < REP; ++i) vars[v].fn(d, qk, qk + C, vt, o, p, vars[v].masked ? mask : nullptr);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float t; hipEventElapsedTime(&t, e0, e1);
            if (r > 0) ms[v].push_back(t / REP);
        }
    for (int v = 0; v < NV; ++v) {
        std::sort(ms[v].begin(), ms[v].end());
        const double med = ms[v][ms[v].size() / 2];
        printf("%-48s median %.4f ms   map bytes / time = %7.1f GB/s\n", vars[v].name, med, np * 2.0 / med / 1e6);
    }
    return 0;
}
