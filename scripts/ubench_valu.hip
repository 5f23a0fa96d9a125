// ubench_valu.hip -- issue cost of the softmax instruction mix on gfx950, alone and beside MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_valu scripts/ubench_valu.hip && /tmp/ubench_valu
// Every kernel runs ITER iterations of a 16-instruction body with independent destination registers; reported is
// shader cycles (s_memtime) per instruction per wave with 1 and 2 waves per SIMD.  Tuning evidence for
// csrc/attn_flash.hip (is v_exp_f32 a quarter-rate op on CDNA4? does it overlap with MFMA / plain VALU?).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define ITER 4096
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int KIND>
__global__ void __launch_bounds__(512) body(float* out, long long* cyc, float seed) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    f32x16 acc[4];
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = seed;
    const float c1 = seed * 0.5f, c2 = seed * 0.25f;
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {  // 16 v_fma_f32
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
            R16(M)
#undef M
        } else if (KIND == 1) {  // 16 v_exp_f32
#define M(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            R16(M)
#undef M
        } else if (KIND == 2) {  // 8 exp + 8 fma interleaved
#define M(i) if ((i) & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
            R16(M)
#undef M
        } else if (KIND == 3) {  // 16 v_max3_f32
#define M(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
            R16(M)
#undef M
        } else if (KIND == 4) {  // 16 v_cvt_pk_f16_f32
#define M(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c1));
            R16(M)
#undef M
        } else if (KIND == 5) {  // 8 v_pk_fma_f32 (16 elements)
#define M(i) if (!((i) & 1)) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&v[i]) : "v"(*(const double*)&v[(i + 2) & 15]));
            R16(M)
#undef M
        } else if (KIND == 6) {  // 4 MFMA, independent accumulators
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
        } else if (KIND == 7) {  // 4 MFMA + 16 exp
#define M(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
            M(0) M(1) M(2) M(3)
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
            M(4) M(5) M(6) M(7)
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[2], 0, 0, 0);
            M(8) M(9) M(10) M(11)
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[3], 0, 0, 0);
            M(12) M(13) M(14) M(15)
#undef M
        } else if (KIND == 8) {  // 4 MFMA + 16 fma
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c1), "v"(c2));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
            M(0) M(1) M(2) M(3)
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
            M(4) M(5) M(6) M(7)
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[2], 0, 0, 0);
            M(8) M(9) M(10) M(11)
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[3], 0, 0, 0);
            M(12) M(13) M(14) M(15)
#undef M
        } else if (KIND == 9) {  // 4 MFMA + 8 exp + 8 fma + 4 cvt (the softmax mix of one 32x32 block quarter)
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[0])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[1]) : "v"(c1), "v"(c2));
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[2])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[3]) : "v"(c1), "v"(c2));
            asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[4]) : "v"(c1));
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[1], 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[5])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[6]) : "v"(c1), "v"(c2));
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[7])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[8]) : "v"(c1), "v"(c2));
            asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[9]) : "v"(c1));
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[2], 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[10])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[11]) : "v"(c1), "v"(c2));
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[12])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[13]) : "v"(c1), "v"(c2));
            asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[14]) : "v"(c1));
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[3], 0, 0, 0);
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[15])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(c1), "v"(c2));
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[2])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[3]) : "v"(c1), "v"(c2));
            asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[4]) : "v"(c1));
        } else if (KIND == 10) {  // 16 v_exp_f16
#define M(i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
            R16(M)
#undef M
        }
    }
    long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
static void run(const char* name, int n_instr, float* out, long long* cyc) {
    for (int wps = 1; wps <= 2; ++wps) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        body<KIND><<<256, 256 * wps>>>(out, cyc, 0.001f);  // warm
        hipEventRecord(e0);
        body<KIND><<<256, 256 * wps>>>(out, cyc, 0.001f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long c;
        hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-34s waves/SIMD %d: %8.3f ms  %7.2f s_memtime ticks / body  (%d instr/body)\n", name, wps, ms,
               (double)c / ITER, n_instr);
    }
}

int main() {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, sizeof(long long));
    run<0>("16 v_fma_f32", 16, out, cyc);
    run<1>("16 v_exp_f32", 16, out, cyc);
    run<10>("16 v_exp_f16", 16, out, cyc);
    run<2>("8 v_exp_f32 + 8 v_fma_f32", 16, out, cyc);
    run<3>("16 v_max3_f32", 16, out, cyc);
    run<4>("16 v_cvt_pk_f16_f32", 16, out, cyc);
    run<5>("8 v_pk_fma_f32", 8, out, cyc);
    run<6>("4 mfma_32x32x16_f16", 4, out, cyc);
    run<7>("4 mfma + 16 v_exp_f32", 20, out, cyc);
    run<8>("4 mfma + 16 v_fma_f32", 20, out, cyc);
    run<9>("4 mfma + 8 exp + 8 fma + 4 cvt", 24, out, cyc);
    return 0;
}
