// flash_timing.hip -- s_memtime totals per segment of the key-tile loop of attn_flash_kernel<40, 2, 2> (wave 0 of workgroup 0), on the
// judged shape (8 frames x 8 heads x 4096 queries x 8192 keys, q in the log2 domain).  Tuning tool, never part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -DFZ_FLASH_TIMING -o build_tmp/flash_timing scripts/flash_timing.hip
// Segments (FZ_TICK slots in csrc/attn_flash.hip): 0 = prefetch issue (global -> registers) of the next tile, 1 = K fragment reads + QK^T
// MFMAs, 2 = running max / rescale branch / exponentials / conversions, 3 = V^T fragment reads + PV MFMAs, 4 = stash (registers -> LDS),
// 5 = barrier.
#define FZ_FLASH_NO_DISPATCH 1
#include "../fatezero_amd/csrc/attn_flash.hip"
#include <stdio.h>
#include <vector>

int main() {
    const int H = 8, L = 4096, D = 40, C = H * D, F = 8;
    FzAttnSelfDesc d = {};
    d.n_frames = F; d.frame0 = 0; d.clip_len = 8; d.heads = H; d.head_dim = D; d.lq = L; d.lkf = L; d.n_kv = 2;
    d.kv_abs[0] = 0; d.kv_val[0] = -1; d.kv_abs[1] = 1; d.kv_val[1] = 7;   // [-1, 'last']: every frame reads two distinct sources
    d.scale = 0.158113883f; d.mode = 0; d.q_log2_scaled = 1;
    d.q_frame_stride = (int64_t)L * 2 * C; d.q_row_stride = 2 * C;
    d.k_frame_stride = (int64_t)L * 2 * C; d.k_row_stride = 2 * C;
    d.vt_frame_stride = (int64_t)C * L; d.vt_chan_stride = L;
    d.o_frame_stride = (int64_t)L * C; d.o_row_stride = C;
    const size_t nqk = (size_t)F * L * 2 * C, nv = (size_t)F * C * L, no = (size_t)F * L * C;
    std::vector<_Float16> hqk(nqk), hv(nv);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; };
    for (size_t i = 0; i < nqk; ++i) hqk[i] = (_Float16)(rnd() * 1.5f * ((i % (2 * C)) < (size_t)C ? 0.158113883f * 1.44269504f : 1.0f));
    for (auto& x : hv) x = (_Float16)rnd();
    _Float16 *qk, *vt, *o;
    hipMalloc(&qk, nqk * 2); hipMalloc(&vt, nv * 2); hipMalloc(&o, no * 2);
    hipMemcpy(qk, hqk.data(), nqk * 2, hipMemcpyHostToDevice);
    hipMemcpy(vt, hv.data(), nv * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        long long zero[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(fz_flash_timing), zero, sizeof(zero));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        launch_flash<40, 2, 2, true, 2>(d, qk, qk + C, vt, o, nullptr);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long t[8];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(fz_flash_timing), sizeof(t));
        const int ntiles = 128;
        long long sum = 0;
        for (int i = 0; i < 7; ++i) sum += t[i];
        printf("launch %.3f ms (instrumented); wave 0 of workgroup 0, %d key tiles, s_memtime ticks per tile: prefetch %lld | K reads + QK MFMA %lld | softmax %lld | V reads + PV MFMA %lld | wait for the prefetched tile %lld | registers -> LDS %lld | barrier %lld | total %lld\n",
               ms, ntiles, t[0] / ntiles, t[1] / ntiles, t[2] / ntiles, t[3] / ntiles, t[6] / ntiles, t[4] / ntiles, t[5] / ntiles, sum / ntiles);
    }
    return 0;
}
