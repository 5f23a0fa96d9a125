// fz_rt.h -- runtime layer shared by every kernel translation unit of libfatezero_hip.so
//
// Two build modes of the SAME kernel sources:
//   * default (hipcc --offload-arch=gfx950): real HIP; this is the product.
//   * -DFZ_EMU (host clang++): a deterministic CPU emulation of the HIP execution model
//     (blocks -> OS threads, lanes -> fibers, 64-wide waves, LDS, __syncthreads, wave shuffles, MFMA
//     fragment semantics).  TEST INFRASTRUCTURE ONLY: it produces libfatezero_emu.so, which nothing in
//     the product ever loads; tests use it to exercise kernel index math and the host orchestration
//     in the GPU-less authoring container.  It is not a fallback and is never shipped as one.
#pragma once
#include <stdint.h>
#include <stddef.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FZ_OK 0
#define FZ_ERR_BAD_ARG -1
#define FZ_ERR_UNSUPPORTED -2
#define FZ_ERR_LAUNCH -3

#ifndef FZ_EMU
// ============================================================================================
//                                       real HIP (gfx950)
// ============================================================================================
#include <hip/hip_runtime.h>

#define FZ_KERNEL __global__
#define FZ_DEVICE __device__ __forceinline__
#define FZ_HOST_DEVICE __host__ __device__
#define FZ_SHARED __shared__
#define FZ_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

typedef hipStream_t fz_stream_t;

// every kernel of the library is launched through fz_plan::launch (bottom of this file): the launch itself, or -- while a native issue
// plan is being recorded (fz_plan_begin, csrc/plan.hip) -- the launch AND a record of it that fz_plan_replay re-issues later
#define FZ_RAW_LAUNCH(kernel, grid, block, smem, stream, ...) hipLaunchKernelGGL(kernel, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
#define FZ_LAUNCH(kernel, grid, block, smem, stream, ...) fz_plan::launch(kernel, grid, block, smem, (void*)(stream), __VA_ARGS__)

static inline int fz_last_launch_status() { return hipGetLastError() == hipSuccess ? FZ_OK : FZ_ERR_LAUNCH; }

FZ_DEVICE f32x16 fz_mfma_32x32x16_f16(half8_t a, half8_t b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_f16:  A[i][k]: lane = i + 16*(k/8), element k%8;  B[k][n]: lane = n + 16*(k/8), element k%8;
// C/D[row][col]: col = lane&15, row = 4*(lane>>4) + reg
FZ_DEVICE f32x4 fz_mfma_16x16x32_f16(half8_t a, half8_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// sum over each aligned group of 4 lanes, result in all 4 (DPP quad_perm [1,0,3,2] then [2,3,0,1]; a fixed association order)
FZ_DEVICE float fz_sum4(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}
// a value the program knows to be wave-uniform (e.g. threadIdx.x / 64): tell the compiler, so that what depends on it stays in SGPRs
FZ_DEVICE int fz_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// keeps a rarely taken branch a branch (hipcc otherwise if-converts it into per-use v_cndmask on the common path)
#define FZ_COLD_PATH() asm volatile("" ::: "memory")
#define FZ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)  /* the scheduler moves nothing across: keeps unrolled loads from piling up */
// the value is materialised in VGPRs HERE: address arithmetic computed ahead of its use is not re-derived (sunk) next to the use
#define FZ_PIN_V(x) asm volatile("" : "+v"(x))
// A byte address inside the workgroup's LDS as a plain 32-bit value (the low half of the flat address): survives FZ_PIN_V without
// the pointer decaying to a flat one (flat_load instead of ds_read), and `fz_lds_ld_h8(a, constant)` folds the constant into the
// instruction's 16-bit offset field -- a ds_read_b128 with no address arithmetic at the point of use.
typedef uint32_t fz_lds_addr;
FZ_DEVICE fz_lds_addr fz_lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)p; }
FZ_DEVICE half8_t fz_lds_ld_h8(fz_lds_addr a, uint32_t byte_off) {
    return *reinterpret_cast<const __attribute__((address_space(3))) half8_t*>((uintptr_t)(a + byte_off));
}
FZ_DEVICE float fz_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
FZ_DEVICE int fz_shfl_xor_i(int v, int mask) { return __shfl_xor(v, mask, 64); }
FZ_DEVICE float fz_shfl(float v, int lane) { return __shfl(v, lane, 64); }
FZ_DEVICE unsigned long long fz_ballot(int pred) { return __ballot(pred); }
// sum over each aligned group of 8 lanes, result in all 8, by DPP (VALU only -- no ds_bpermute round trip): quad_perm
// [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror; a fixed association order
FZ_DEVICE float fz_sum8(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}
// max over the lane pair (l, l^32) without touching LDS: v_permlane32_swap (gfx950) instead of ds_bpermute
FZ_DEVICE float fz_pair_max32(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
FZ_DEVICE float fz_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
FZ_DEVICE float fz_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
FZ_DEVICE float fz_rsqrt(float x) { return rsqrtf(x); }
// LDS-DMA (global_load_lds_dwordx4): every lane fetches 16 bytes from its OWN global address; the wave's 64 chunks land
// lane-linear at lds_wave_base + 16 * lane (M0 base + lane * 16, cdna_hip_programming.md section 5).  Asynchronous: tracked
// by vmcnt only -- fz_wait_vm0() + a barrier must separate it from the ds_reads of the data.
FZ_DEVICE void fz_glds16(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// the same with a cache policy (AUX: 1 = sc0, 2 = nt, 16 = sc1; sc1 / nt loads are served by L2 and leave the CU's vector L1 alone)
template <int AUX>
FZ_DEVICE void fz_glds16_aux(const void* gsrc_lane, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}
// the same with a wave-uniform 64-bit base (kept in scalar registers) plus a per-lane 32-bit byte offset: the SGPR-base addressing
// form of the instruction -- advancing the base is scalar work, no VALU per issue
FZ_DEVICE void fz_glds16_so(const char* sbase_uniform, uint32_t lane_byte_off, void* lds_wave_base) {
    // (hipcc selects only the VGPR-address form for the builtin -- a 64-bit VALU add per issue -- so the instruction is written
    // out: M0 = LDS base of the wave's 1 KB, then `global_load_lds_dwordx4 voffset, s[base:base+1]`.  M0 is a set-before-use
    // register for the compiler as well (it rewrites it in front of every LDS-DMA builtin), so clobbering it here is safe.)
    const uint32_t lds = (uint32_t)(uintptr_t)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane_byte_off), "s"(sbase_uniform)
                 : "memory", "m0");
}
#ifndef FZ_A_CPOL_ASM
#define FZ_A_CPOL_ASM ""   /* trial builds: " nt" / " sc1" / " sc1 nt" on the weight operand's LDS-DMA (scalar-base form) */
#endif
FZ_DEVICE void fz_glds16_so_a(const char* sbase_uniform, uint32_t lane_byte_off, void* lds_wave_base) {
    const uint32_t lds = (uint32_t)(uintptr_t)lds_wave_base;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" FZ_A_CPOL_ASM ::"s"(lds), "v"(lane_byte_off), "s"(sbase_uniform)
                 : "memory", "m0");
}
FZ_DEVICE void fz_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// counted wait: at most N of this wave's vector-memory operations (LDS-DMA included) still outstanding; they retire in order
template <int N>
FZ_DEVICE void fz_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// workgroup barrier that does NOT drain vmcnt (hipcc's __syncthreads() waits vmcnt(0) while an LDS-DMA is in flight):
// the caller orders its own LDS traffic -- fz_wait_vm<N>() before it for DMA'd data, and every ds_read of the buffer being
// recycled already consumed (an MFMA cannot issue before its LDS operands arrived)
FZ_DEVICE void fz_barrier_nodrain() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the bare s_barrier: neither vmcnt nor lgkmcnt is drained (gfx950 backs off a barrier with memory operations outstanding), so
// ds_reads issued before it are still in flight after it -- the compiler's own lgkmcnt wait in front of their first use orders them
FZ_DEVICE void fz_barrier_raw() { __builtin_amdgcn_s_barrier(); }
// this wave's LDS writes (and reads) are complete: in front of a bare barrier behind which OTHER waves read what this one wrote
FZ_DEVICE void fz_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// LDS hand-over INSIDE one wave (lane a's ds_write read by lane b of the same wave): the LDS unit executes a wave's
// instructions in order, so no s_barrier is needed -- only the compiler has to keep the order
FZ_DEVICE void fz_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#define FZ_DEVICE_GLOBAL __device__

#else
// ============================================================================================
//                                CPU emulation (tests only)
// ============================================================================================
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* fz_stream_t;

namespace fz_emu {
struct BlockCtx;
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local unsigned char* t_dyn_smem;
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
void sync_block();                                  // __syncthreads (hipcc: waits vmcnt(0) first -> retires the lane's LDS-DMA)
void sync_block_nodrain();                          // s_barrier alone
void dma_issue(const void* src16, void* dst16);     // one lane's 16 bytes of an LDS-DMA instruction
void dma_wait(int max_outstanding);                 // s_waitcnt vmcnt(N) for the calling lane
void wave_exchange(const void* mine, void* all, size_t bytes);  // gather `bytes` from each of the wave's 64 lanes
int lane_id();
f32x16 mfma_32x32x16_f16(half8_t a, half8_t b, f32x16 c);        // one wave-wide MFMA (all 64 lane fibers call it)
f32x4 mfma_16x16x32_f16(half8_t a, half8_t b, f32x4 c);
}  // namespace fz_emu

#define threadIdx (fz_emu::t_threadIdx)
#define blockIdx (fz_emu::t_blockIdx)
#define blockDim (fz_emu::t_blockDim)
#define gridDim (fz_emu::t_gridDim)

#define FZ_KERNEL
#define FZ_DEVICE static inline
#define FZ_HOST_DEVICE static
#define FZ_SHARED static thread_local
#define FZ_DYN_SMEM(name) unsigned char* name = fz_emu::t_dyn_smem
#define __launch_bounds__(...)
#define __restrict__

#define FZ_RAW_LAUNCH(kernel, grid, block, smem, stream, ...) fz_emu::launch(grid, block, smem, [=]() { kernel(__VA_ARGS__); })
#define FZ_LAUNCH(kernel, grid, block, smem, stream, ...) fz_plan::launch(kernel, grid, block, smem, (void*)(stream), __VA_ARGS__)

static inline int fz_last_launch_status() { return FZ_OK; }
static inline void __syncthreads() { fz_emu::sync_block(); }

// v_mfma_f32_32x32x16_f16 fragment semantics (cdna_hip_programming.md §3):
//   A[i][k]: lane = i + 32*(k/8), element k%8;  B[k][n]: lane = n + 32*(k/8), element k%8
//   C/D[row][col]: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
static inline f32x16 fz_mfma_32x32x16_f16(half8_t a, half8_t b, f32x16 c) { return fz_emu::mfma_32x32x16_f16(a, b, c); }
static inline f32x4 fz_mfma_16x16x32_f16(half8_t a, half8_t b, f32x4 c) { return fz_emu::mfma_16x16x32_f16(a, b, c); }
static inline int fz_uniform(int v) { return v; }
#define FZ_COLD_PATH() ((void)0)
#define FZ_SCHED_FENCE() ((void)0)
#define FZ_PIN_V(x) ((void)0)
typedef uint32_t fz_lds_addr;  // emulation: byte offset from the block's dynamic LDS
static inline fz_lds_addr fz_lds_addr_of(const void* p) { return (uint32_t)((const unsigned char*)p - fz_emu::t_dyn_smem); }
static inline float fz_shfl_xor(float v, int mask) {
    float all[64];
    fz_emu::wave_exchange(&v, all, sizeof(float));
    return all[fz_emu::lane_id() ^ mask];
}
static inline int fz_shfl_xor_i(int v, int mask) {
    int all[64];
    fz_emu::wave_exchange(&v, all, sizeof(int));
    return all[fz_emu::lane_id() ^ mask];
}
static inline float fz_shfl(float v, int lane) {
    float all[64];
    fz_emu::wave_exchange(&v, all, sizeof(float));
    return all[lane & 63];
}
static inline unsigned long long fz_ballot(int pred) {
    int all[64];
    fz_emu::wave_exchange(&pred, all, sizeof(int));
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) m |= (unsigned long long)(all[i] != 0) << i;
    return m;
}
static inline float fz_pair_max32(float v) { return fmaxf(v, fz_shfl_xor(v, 32)); }
static inline float fz_sum4(float v) {  // same association order as the DPP form: xor 1, xor 2
    v += fz_shfl_xor(v, 1);
    v += fz_shfl_xor(v, 2);
    return v;
}
static inline float fz_sum8(float v) {  // same association order as the DPP form: xor 1, xor 2, mirror within 8
    v += fz_shfl_xor(v, 1);
    v += fz_shfl_xor(v, 2);
    float all[64];
    fz_emu::wave_exchange(&v, all, sizeof(float));
    const int l = fz_emu::lane_id();
    return v + all[(l & ~7) | (7 - (l & 7))];
}
static inline float fz_exp2(float x) { return exp2f(x); }
static inline float fz_rcp(float x) { return 1.0f / x; }
static inline float fz_rsqrt(float x) { return 1.0f / sqrtf(x); }
// LDS-DMA is ASYNCHRONOUS on the emulator too: the 16 bytes are queued per lane and land when the lane's vmcnt wait retires them
// (the latest moment the hardware allows -- a missing or too-loose fz_wait_vm<N>() reads stale LDS and fails the test); with
// FZ_EMU_DMA=early in the environment they land at issue (the earliest moment -- a DMA issued before every reader of the recycled
// buffer passed the barrier corrupts their tile).  Tests run the GEMM / conv cases both ways.
static inline void fz_glds16(const void* gsrc_lane, void* lds_wave_base) {
    fz_emu::dma_issue(gsrc_lane, (unsigned char*)lds_wave_base + 16 * fz_emu::lane_id());
}
static inline void fz_glds16_so(const char* sbase_uniform, uint32_t lane_byte_off, void* lds_wave_base) {
    fz_glds16(sbase_uniform + lane_byte_off, lds_wave_base);
}
template <int AUX>
static inline void fz_glds16_aux(const void* gsrc_lane, void* lds_wave_base) { fz_glds16(gsrc_lane, lds_wave_base); }
static inline void fz_glds16_so_a(const char* sbase_uniform, uint32_t lane_byte_off, void* lds_wave_base) { fz_glds16_so(sbase_uniform, lane_byte_off, lds_wave_base); }
static inline void fz_wait_vm0() { fz_emu::dma_wait(0); }
template <int N>
static inline void fz_wait_vm() { fz_emu::dma_wait(N); }
static inline void fz_barrier_nodrain() { fz_emu::sync_block_nodrain(); }
static inline void fz_barrier_raw() { fz_emu::sync_block_nodrain(); }
static inline void fz_lds_fence() {}
static inline void fz_wave_lds_sync() {  // all 64 lane fibers of the wave meet
    int mine = 0, all[64];
    fz_emu::wave_exchange(&mine, all, sizeof(int));
}
#define FZ_DEVICE_GLOBAL static
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
#endif

// --------------------------------------------------------------------------------------------
// small helpers common to both modes
// --------------------------------------------------------------------------------------------
FZ_DEVICE half8_t fz_zero_h8() {
    half8_t z;
    for (int i = 0; i < 8; ++i) z[i] = (half_t)0.0f;
    return z;
}
FZ_DEVICE f32x16 fz_zero_f16v() {
    f32x16 z;
    for (int i = 0; i < 16; ++i) z[i] = 0.0f;
    return z;
}
FZ_DEVICE half8_t fz_ld_h8(const half_t* p) { return *reinterpret_cast<const half8_t*>(p); }
#ifdef FZ_EMU
static inline half8_t fz_lds_ld_h8(fz_lds_addr a, uint32_t byte_off) {
    return *reinterpret_cast<const half8_t*>(fz_emu::t_dyn_smem + a + byte_off);
}
#endif
// 16-byte load from a wave-uniform base plus a 32-bit per-lane byte offset (the SGPR-base + VGPR-offset global_load form)
FZ_DEVICE half8_t fz_ld_h8_off(const char* base, uint32_t byte_off) {
    return *reinterpret_cast<const half8_t*>(base + byte_off);
}
// wave issue priority hint (0..3); no-op on the emulator
FZ_DEVICE void fz_setprio_hi() {
#ifndef FZ_EMU
    __builtin_amdgcn_s_setprio(1);
#endif
}
FZ_DEVICE void fz_setprio_lo() {
#ifndef FZ_EMU
    __builtin_amdgcn_s_setprio(0);
#endif
}
// a * b + c with a, b < 2^24: one full-rate v_mad_u32_u24 instead of a quarter-rate 32-bit multiply
FZ_DEVICE uint32_t fz_mad24(uint32_t a, uint32_t b, uint32_t c) {
#ifdef FZ_EMU
    return a * b + c;
#else
    return __umul24(a, b) + c;
#endif
}
FZ_DEVICE void fz_st_h8(half_t* p, half8_t v) { *reinterpret_cast<half8_t*>(p) = v; }
// streaming 16-byte store (nt: the line is not kept for re-use) for data this GPU job writes once and reads much later
FZ_DEVICE void fz_st_h8_nt(half_t* p, half8_t v) {
#ifdef FZ_EMU
    *reinterpret_cast<half8_t*>(p) = v;
#else
    __builtin_nontemporal_store(v, reinterpret_cast<half8_t*>(p));
#endif
}

// gelu(x) = x Phi(x) with the exact (erf) definition of torch's F.gelu, branch-free: Phi(-|x|) = erfc(|x| / sqrt 2) / 2 from
// Abramowitz & Stegun 7.1.26 (|erf error| <= 1.5e-7; measured against fp64: |gelu error| <= 4.3e-7 everywhere, relative
// error <= 1.7e-4 wherever |gelu| > 1e-3, i.e. below half an fp16 ulp of the stored value).  libm's erff is a two-branch
// ~45-instruction sequence per call, which made the GEGLU epilogue of the K = 320 projection longer than its K loop.
FZ_DEVICE float fz_gelu_erf(float x) {
#ifdef FZ_GELU_TRIAL_IDENTITY  // (trial builds only, scripts/build_variant.sh: an upper bound on what a cheaper GELU could buy)
    return x;
#endif
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = fz_rcp(fmaf(0.3275911f, z, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float q = 0.5f * poly * t * fz_exp2(-1.4426950408889634f * z * z);  // Phi(-|x|)
    return x * (x < 0.0f ? q : 1.0f - q);
}

// 2^x for a PAIR of arguments WITHOUT the transcendental unit: fract / floor split, a cubic minimax polynomial of 2^f on [0, 1) in
// packed fp32 FMAs (max relative error 7.5e-5: a sixth of an fp16 ulp), v_ldexp.  4.5-5 full-rate VALU instructions per element
// against one v_exp_f32 -- which, unlike FMAs, does not co-issue with MFMAs (profiles/r01_ubench_valu.txt).  Trial use only:
// csrc/attn_flash.hip NPOLY, scripts/flash_ab.hip.
FZ_DEVICE f32x2 fz_exp2_poly2(f32x2 x) {
#ifdef FZ_EMU
    const f32x2 fl = {floorf(x[0]), floorf(x[1])};
#else
    const f32x2 fl = {__builtin_floorf(x[0]), __builtin_floorf(x[1])};
#endif
    const f32x2 fr = x - fl;
    const f32x2 c0 = {0.9999249577522278f, 0.9999249577522278f}, c1 = {0.6958348155021667f, 0.6958348155021667f};
    const f32x2 c2 = {0.22606661915779114f, 0.22606661915779114f}, c3 = {0.07802354544401169f, 0.07802354544401169f};
    f32x2 p = c3 * fr + c2;
    p = p * fr + c1;
    p = p * fr + c0;
    f32x2 r;
#ifdef FZ_EMU
    r[0] = ldexpf(p[0], (int)(fl[0] < -200.0f ? -200.0f : fl[0]));
    r[1] = ldexpf(p[1], (int)(fl[1] < -200.0f ? -200.0f : fl[1]));
#else
    // (v_cvt_i32_f32 saturates and v_ldexp_f32 flushes an exponent below the fp32 range to 0: no clamp needed)
    r[0] = __builtin_amdgcn_ldexpf(p[0], (int)fl[0]);
    r[1] = __builtin_amdgcn_ldexpf(p[1], (int)fl[1]);
#endif
    return r;
}

// LayerNorm of ONE 320-channel row held by 8 consecutive lanes (lane l8 of the 8 holds chunks l8 + 8 i, i = 0..4, of 8 channels; gm / bt: the
// same chunks of gamma / beta): exact two-sweep statistics in fp32 on the fp16 values, sums in a FIXED order (even / odd elements apart,
// chunk-major, then fz_sum8's DPP tree).  ONE body shared by every launch that writes LN(y) beside y (igemm.hip GS == -1, ff_chain.hip,
// xattn_chain.hip), with re-association switched OFF inside it: under -ffast-math the sums of an inlined copy are re-associated by the context
// it is inlined into, and the launches that replace each other must agree bit for bit (found on MI355X: two textually identical epilogues
// that differed by an fp16 ulp).  (Out of line it costs the caller its live registers around the call: gemm_lnout +40 %.)
struct FzRow5 {
    half8_t c[5];
};
FZ_DEVICE FzRow5 fz_ln_row320(const FzRow5& v, const FzRow5& gm, const FzRow5& bt, float eps) {
#ifndef FZ_EMU
#pragma clang fp reassociate(off)
#endif
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            s0 += (float)v.c[i][e];
            s1 += (float)v.c[i][e + 1];
        }
    const float mean = fz_sum8(s0 + s1) * (1.0f / 320.0f);
    float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float d0 = (float)v.c[i][e] - mean, d1 = (float)v.c[i][e + 1] - mean;
            q0 += d0 * d0;
            q1 += d1 * d1;
        }
    const float rstd = 1.0f / sqrtf(fz_sum8(q0 + q1) * (1.0f / 320.0f) + eps);
    FzRow5 o;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) o.c[i][e] = (half_t)(((float)v.c[i][e] - mean) * rstd * (float)gm.c[i][e] + (float)bt.c[i][e]);
    return o;
}

static inline int fz_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int fz_round_up(int a, int b) { return fz_ceil_div(a, b) * b; }

// ============================================================================================
//                     native issue plans: record a launch list once, re-issue it later
// ============================================================================================
// One UNet forward is ~520-700 launches whose grid, arguments and order are a pure function of (clip geometry, controller plan); Python
// decides them again at every DDIM step.  A plan is that decision taken once: while fz_plan_begin ... fz_plan_end brackets a forward, every
// launch of the library is issued as usual AND appended to the plan -- kernel, grid, block, LDS bytes and a byte copy of the kernel
// arguments -- and fz_plan_replay re-issues a range of the records with one call.  What changes from step to step is data behind pointers
// (latents, the timestep embedding, the step's map slabs, masks): fz_plan_relocate rewrites every pointer-sized argument word of a record
// range that points into [old, old + bytes) to the same offset from `new_base`.  (csrc/plan.hip; host side: fatezero_amd/issue.py)
#include <string.h>
#include <vector>

namespace fz_plan {

struct Record {
    void (*run)(const Record&, const unsigned char* args, void* stream);
    int (*argptrs)(unsigned char* packed, void** out);   // addresses of the packed arguments, in kernel-parameter order (graph nodes)
    const void* kernel;
    dim3 grid, block;
    size_t smem;
    uint32_t arg_off, arg_len;   // this record's argument bytes in Plan::args (arg_off is a multiple of 16)
};

struct Plan {
    std::vector<Record> recs;
    std::vector<uint64_t> args;   // (uint64_t: the relocation pass walks pointer-sized, pointer-aligned words)
};

inline thread_local Plan* g_recording = nullptr;   // the plan THIS THREAD is recording (nullptr: none).  Thread-local: the library keeps no
                                                   // process-wide state -- a recording belongs to the thread that called fz_plan_begin, launches
                                                   // other threads make meanwhile are issued and not recorded.  (a C++17 inline variable: one
                                                   // object per thread for all translation units of the library, and the stand-alone trial
                                                   // programs under scripts/ that include a kernel file link without csrc/plan.hip)

// the arguments of one launch as an aggregate of the kernel's own parameter types: trivially copyable, pointers at their natural alignment
template <typename... P> struct Args;
template <> struct Args<> {};
template <typename H, typename... T> struct Args<H, T...> {
    H head;
    Args<T...> tail;
};
template <typename... T> struct Packer;
template <> struct Packer<> {
    static Args<> make() { return Args<>{}; }
};
template <typename H, typename... T> struct Packer<H, T...> {
    template <typename A0, typename... AR> static Args<H, T...> make(A0&& a0, AR&&... ar) {
        return Args<H, T...>{(H)a0, Packer<T...>::make(ar...)};
    }
};
template <typename F, typename... Done> inline void unpack(F&& f, const Args<>&, const Done&... d) { f(d...); }
template <typename F, typename H, typename... T, typename... Done> inline void unpack(F&& f, const Args<H, T...>& a, const Done&... d) {
    unpack(f, a.tail, d..., a.head);
}

inline int collect(Args<>&, void**, int i) { return i; }
template <typename H, typename... T> inline int collect(Args<H, T...>& a, void** out, int i) {
    out[i] = (void*)&a.head;
    return collect(a.tail, out, i + 1);
}

template <typename... P> struct Thunk {
    static int argptrs(unsigned char* packed, void** out) { return collect(*(Args<P...>*)packed, out, 0); }
    static void run(const Record& r, const unsigned char* args, void* stream) {
        Args<P...> a;
        memcpy(&a, args + r.arg_off, sizeof(a));
        void (*k)(P...) = (void (*)(P...))r.kernel;
        const dim3 grid = r.grid, block = r.block;
        const size_t smem = r.smem;
        unpack([&](const P&... p) { FZ_RAW_LAUNCH(k, grid, block, smem, stream, p...); }, a);
    }
};

template <typename... P, typename... A>
inline void launch(void (*kernel)(P...), dim3 grid, dim3 block, size_t smem, void* stream, A&&... a) {
    if (g_recording != nullptr) {
        Plan& pl = *g_recording;
        const Args<P...> packed = Packer<P...>::make(a...);
        Record r;
        r.run = &Thunk<P...>::run;
        r.argptrs = &Thunk<P...>::argptrs;
        static_assert(sizeof...(P) <= 32 && alignof(Args<P...>) <= 16, "kernel argument list beyond what a plan record holds");
        r.kernel = (const void*)kernel;
        r.grid = grid; r.block = block; r.smem = smem;
        r.arg_off = (uint32_t)(pl.args.size() * 8);
        r.arg_len = (uint32_t)sizeof(packed);
        pl.args.resize(pl.args.size() + (sizeof(packed) + 15) / 16 * 2, 0);
        memcpy((unsigned char*)pl.args.data() + r.arg_off, &packed, sizeof(packed));
        pl.recs.push_back(r);
    }
    FZ_RAW_LAUNCH(kernel, grid, block, smem, stream, a...);
}

}  // namespace fz_plan
