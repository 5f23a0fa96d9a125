// ubench_launch_gap.hip -- what a kernel boundary costs on this GPU (never part of the library).
// Back-to-back launches on ONE stream; every workgroup stamps the constant 100 MHz wall clock (s_memrealtime) at entry and exit, the launch's
// min entry / max exit are kept per launch.  Reported: per-launch time by HIP events, the in-kernel span (first entry .. last exit), the gap from
// the last exit of launch i to the first entry of launch i + 1, and the dispatch ramp (first entry .. last entry), for grids of 256 / 1024
// workgroups, with and without a large LDS allocation, plain launches and one hipGraph of the same launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build_tmp/ubench_launch_gap scripts/ubench_launch_gap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

constexpr int NL = 64, MAXG = 1024;
__device__ unsigned long long stamp[NL][MAXG][2];   // entry / exit of every workgroup (own slots: same-address atomics would serialise at ~30 ns each)

extern __shared__ unsigned char dyn[];
__global__ void __launch_bounds__(512) k_span(int id, int busy_ticks, float* sink) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) stamp[id][blockIdx.x][0] = t0;
    if (busy_ticks > 0) {
        while (wall_clock64() - t0 < (unsigned long long)busy_ticks) __builtin_amdgcn_s_sleep(4);
    }
    if (sink != nullptr && threadIdx.x == 9999) sink[0] = dyn[0];
    __syncthreads();
    if (threadIdx.x == 0) stamp[id][blockIdx.x][1] = wall_clock64();
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_span, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("%-46s | event us / launch | in-kernel span us | exit -> next entry us | entry ramp us\n", "launches of 64, one stream");
    for (int graph = 0; graph < 2; ++graph)
        for (int grid : {256, 1024})
            for (int threads : {256, 512})
                for (int lds : {0, 140 * 1024})
                    for (int busy_us : {0, 20}) {
                        auto reset = [&]() {};
                        hipGraphExec_t ge = nullptr;
                        if (graph) {
                            hipGraph_t gr;
                            hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
                            for (int i = 0; i < NL; ++i) hipLaunchKernelGGL(k_span, dim3(grid), dim3(threads), lds, st, i, busy_us * 100, nullptr);
                            hipStreamEndCapture(st, &gr);
                            hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
                        }
                        float ms = 0;
                        for (int rep = 0; rep < 3; ++rep) {
                            reset();
                            hipDeviceSynchronize();
                            hipEventRecord(e0, st);
                            if (graph) hipGraphLaunch(ge, st);
                            else for (int i = 0; i < NL; ++i) hipLaunchKernelGGL(k_span, dim3(grid), dim3(threads), lds, st, i, busy_us * 100, nullptr);
                            hipEventRecord(e1, st);
                            hipEventSynchronize(e1);
                            hipEventElapsedTime(&ms, e0, e1);
                        }
                        static unsigned long long hs[NL][MAXG][2];
                        static unsigned long long h[NL][3];
                        hipMemcpyFromSymbol(hs, HIP_SYMBOL(stamp), sizeof(hs));
                        for (int i = 0; i < NL; ++i) {
                            h[i][0] = ~0ull; h[i][1] = 0; h[i][2] = 0;
                            for (int w = 0; w < grid; ++w) {
                                h[i][0] = std::min(h[i][0], hs[i][w][0]);
                                h[i][2] = std::max(h[i][2], hs[i][w][0]);
                                h[i][1] = std::max(h[i][1], hs[i][w][1]);
                            }
                        }
                        std::vector<double> sp, gap, ramp;
                        for (int i = 8; i < NL; ++i) {
                            sp.push_back((h[i][1] - h[i][0]) / 100.0);
                            ramp.push_back((h[i][2] - h[i][0]) / 100.0);
                            gap.push_back(((double)h[i][0] - (double)h[i - 1][1]) / 100.0);
                        }
                        char nm[96];
                        snprintf(nm, 96, "%s %4d wg x %3d thr, LDS %3d KB, busy %2d us", graph ? "graph" : "plain", grid, threads, lds / 1024, busy_us);
                        printf("%-46s | %8.2f          | %8.2f          | %8.2f              | %6.2f\n", nm, ms * 1e3 / NL, med(sp), med(gap), med(ramp));
                    }
    return 0;
}
