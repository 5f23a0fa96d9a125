// ubench_ingest.hip -- how fast does a CU ingest operand tiles from L2 by LDS-DMA (global_load_lds_dwordx4), as the K loops of csrc/*.hip do?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_ingest scripts/ubench_ingest.hip && /tmp/ubench_ingest
// One 512-thread workgroup per CU runs STEPS ring steps; a step fetches `shared_kb` KB that EVERY workgroup reads at the same offsets (a weight
// panel) + `private_kb` KB of its own (activation rows), into an NS-slot LDS ring with NS - 1 steps in flight, and -- optionally -- runs `mfma`
// MFMAs per wave beside it.  Reported: bytes per clock per CU (wall time x the clock the job sustains), for
//   depth (tiles in flight) 1 / 2 / 3, shared-only / private-only / the 40 + 16 KB mix of the 320 x 128 tile, with and without the MFMAs.
// Question behind it: every K loop of the library takes ~1.2 us per 56 KB step = ~22 B/clk/CU whatever its
// structure -- is that the CU's ingest limit, a latency (in-flight depth) limit, or the cost of all CUs streaming the SAME panel at once?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const char* sbase, unsigned lane_off, void* lds) {
    const unsigned l = (unsigned)(unsigned long long)lds;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(lane_off), "s"(sbase) : "memory", "m0");
}

template <int NS, int MFMA>
__global__ void __launch_bounds__(512) ingest(const char* shared_src, const char* private_src, int shared_kb, int private_kb, int steps, long long panel_bytes,
                                              float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int slot_kb = shared_kb + private_kb;
    const char* priv = private_src + (long long)blockIdx.x * private_kb * 1024 * 64;   // 64 steps of private rows per workgroup, then wrap
    f32x16 acc[4];
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    auto issue = [&](int s) {
        unsigned char* dst = lds + (s % NS) * slot_kb * 1024;
        const char* sp = shared_src + ((long long)s * shared_kb * 1024) % panel_bytes;
        const char* pp = priv + (long long)(s % 64) * private_kb * 1024;
        for (int j = wave; j < slot_kb; j += 8) {
            const char* src = j < shared_kb ? sp + j * 1024 : pp + (j - shared_kb) * 1024;
            glds16(src, lane * 16u, dst + j * 1024);
        }
    };
    for (int s = 0; s < NS - 1 && s < steps; ++s) issue(s);
    for (int s = 0; s < steps; ++s) {
        // the ring's wait: everything but the NS - 2 younger steps this wave issued
        const int per = (slot_kb - wave + 7) / 8;
        if (NS == 2 || s + NS - 1 > steps) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {   // (counts are wave-dependent: a generic countdown through the few values that occur)
            const int keep = (NS - 2) * per;
            if (keep >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else if (keep >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (keep >= 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else if (keep >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (s + NS - 1 < steps) issue(s + NS - 1);
#pragma unroll
        for (int m = 0; m < MFMA; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
    }
    if (sink) sink[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + lds[threadIdx.x];
}

template <int NS, int MFMA>
static double run(const char* sh, const char* pr, int skb, int pkb, int steps, long long panel, float* sink) {
    const size_t smem = (size_t)NS * (skb + pkb) * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&ingest<NS, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> t;
    for (int it = 0; it < 7; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ingest<NS, MFMA>), dim3(256), dim3(512), smem, 0, sh, pr, skb, pkb, steps, panel, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2] * 1e-3;
}

int main() {
    const long long panel = 4ll << 20;          // a 4 MB weight panel, cycled (stays in L2 / Infinity Cache)
    const long long priv = 256ll * 64 * 64 * 1024;  // up to 64 KB x 64 steps per workgroup
    char *sh, *pr;
    float* sink;
    hipMalloc(&sh, panel + (1 << 20));
    hipMalloc(&pr, priv);
    hipMalloc(&sink, 256 * 512 * 4);
    hipMemset(sh, 1, panel + (1 << 20));
    hipMemset(pr, 1, priv);
    const int steps = 400;
    const double ghz = 2.1;  // (the clock these launches sustain: bench.py's box block; only scales the B/clk column)
    struct { int skb, pkb; const char* what; } mixes[] = {{40, 16, "40 KB shared + 16 KB private (the 320 x 128 tile, K step 64)"}, {56, 0, "56 KB shared"},
                                                          {0, 56, "56 KB private"}, {20, 8, "20 + 8 KB (K step 32)"}, {40, 32, "40 + 32 KB (320 x 256)"}};
    printf("LDS-DMA ingest per CU, 256 workgroups of 512 threads, %d steps; B/clk at %.1f GHz\n", steps, ghz);
    for (auto& m : mixes) {
        const double bytes = (double)(m.skb + m.pkb) * 1024 * steps;
        auto row = [&](const char* name, double s) { printf("   %-28s %7.2f us/step  %6.1f B/clk/CU  %6.2f TB/s chip\n", name, s / steps * 1e6, bytes / (s * ghz * 1e9), bytes * 256 / s / 1e12); };
        printf("%s\n", m.what);
        row("1 step in flight, no MFMA", run<2, 0>(sh, pr, m.skb, m.pkb, steps, panel, nullptr));
        if ((m.skb + m.pkb) * 3 <= 160) row("2 steps in flight, no MFMA", run<3, 0>(sh, pr, m.skb, m.pkb, steps, panel, nullptr));
        if ((m.skb + m.pkb) * 4 <= 160) row("3 steps in flight, no MFMA", run<4, 0>(sh, pr, m.skb, m.pkb, steps, panel, nullptr));
        row("1 in flight + 20 MFMA/wave", run<2, 20>(sh, pr, m.skb, m.pkb, steps, panel, sink));
        if ((m.skb + m.pkb) * 3 <= 160) row("2 in flight + 20 MFMA/wave", run<3, 20>(sh, pr, m.skb, m.pkb, steps, panel, sink));
    }
    return 0;
}
