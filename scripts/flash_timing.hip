// flash_timing.hip -- where does one wave of the d=40 flash kernel spend its cycles?  Compiles the product kernel
// source with FZ_FLASH_TIMING (s_memtime probes between the segments of the K/V tile loop, wave 0 of block 0) and
// runs it on the 64x64 SD level (8 frames x 8 heads x 4096 x 8192).  Tuning tool, never part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -w -o build_tmp/flash_timing scripts/flash_timing.hip
#define FZ_FLASH_TIMING 1
#include "../fatezero_amd/csrc/attn_flash.hip"
#include <stdio.h>
#include <vector>

int main() {
    const int F = 8, H = 8, L = 4096, D = 40, C = H * D;
    FzAttnSelfDesc d = {};
    d.n_frames = F; d.frame0 = 0; d.clip_len = F; d.heads = H; d.head_dim = D; d.lq = L; d.lkf = L; d.n_kv = 2;
    d.kv_abs[0] = 0; d.kv_val[0] = -1; d.kv_abs[1] = 1; d.kv_val[1] = 0;
    d.scale = 0.158113883f; d.mode = 0; d.q_log2_scaled = 1;
    d.q_frame_stride = (int64_t)L * 2 * C; d.q_row_stride = 2 * C;
    d.k_frame_stride = (int64_t)L * 2 * C; d.k_row_stride = 2 * C;
    d.vt_frame_stride = (int64_t)C * L; d.vt_chan_stride = L;
    d.o_frame_stride = (int64_t)L * C; d.o_row_stride = C;
    const size_t nqk = (size_t)F * L * 2 * C, nv = (size_t)F * C * L;
    std::vector<_Float16> hqk(nqk), hv(nv);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f * 2.0f - 1.0f; };
    for (auto& x : hqk) x = (_Float16)(rnd() * 1.5f);
    for (auto& x : hv) x = (_Float16)rnd();
    _Float16 *qk, *vt, *o;
    hipMalloc(&qk, nqk * 2); hipMalloc(&vt, nv * 2); hipMalloc(&o, (size_t)F * L * C * 2);
    hipMemcpy(qk, hqk.data(), nqk * 2, hipMemcpyHostToDevice);
    hipMemcpy(vt, hv.data(), nv * 2, hipMemcpyHostToDevice);
    long long zero[8] = {0};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipMemcpyToSymbol(HIP_SYMBOL(fz_flash_timing), zero, sizeof(zero));
        hipEventRecord(e0);
        launch_flash<40, 2, 2>(d, qk, qk + C, vt, o, nullptr);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long t[8];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(fz_flash_timing), sizeof(t));
        const double tiles = 128.0;
        printf("launch %.3f ms | per tile (s_memtime ticks, wave 0 of block 0): fetch-issue %.0f  kfr+QK %.0f  softmax %.0f  PV %.0f  stash %.0f  barrier %.0f  | total %.0f\n",
               ms, t[0] / tiles, t[1] / tiles, t[2] / tiles, t[3] / tiles, t[4] / tiles, t[5] / tiles,
               (t[0] + t[1] + t[2] + t[3] + t[4] + t[5]) / tiles);
    }
    return 0;
}
