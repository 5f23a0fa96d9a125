// ff_chain.hip -- the FeedForward CHAIN of the 64x64-level transformer block in ONE launch:
//     hidden_states = ff(norm3(hidden_states)) + hidden_states ; norm_temporal(hidden_states)          (attention.py:312-321, 327-331)
//   with ff = diffusers FeedForward [GEGLU(dim, 4 dim) -> Linear(4 dim, dim)] [3P]:  val | gate = x W1^T + b1 ;  h = val * gelu(gate) ;
//   y = h W2^T + b2.   Per-op this was fz_gemm(GEGLU) + fz_gemm_lnout: two launches with a rows x 1280 fp16 round trip through HBM between
//   them (168 MB at 16 frames), both short-K / epilogue-dominated (0.25 of the MFMA roof, profiles/r05_job_breakdown_by_op_and_shape.txt).
//
// One workgroup = 128 rows = four PAIRS of waves, each pair 32 rows; wave w (UP) and wave w + 4 (DOWN) share a SIMD.  Nothing of the
// activations touches LDS but the 2 KB of h a pair hands over per step:
//   x        the pair's 32 rows x 320 channels of LN(x) live in the UP wave's REGISTERS for the whole launch as the 20 B fragments of the up
//            projection (lane = row + 32 * k half, 8 halves: 80 VGPRs);
//   hidden   walked in chunks of 32 units: the UP wave computes the val / gate tiles U[32 units][32 rows] = W1 chunk . x^T (2 x 20 MFMAs
//            32x32x16), + b1, gates them in registers.  The accumulator layout of the MFMA (lane = row, 16 registers = 16 units) IS the
//            B-fragment layout of the down projection when W1's rows are packed in the order the registers want (unit 16 s + 8 hi + j at A
//            row 8 (2 s + j / 4) + 4 hi + j % 4): h never crosses lanes; it goes to LDS as two lane-linear 1 KB fragments and the DOWN wave
//            contracts it in natural k order one step later;
//   y        the pair's 32 rows x 320 outputs accumulate in fp32 in the DOWN wave's registers across all 40 chunks (10 tiles, 160 registers);
//   weights  the only thing streamed: pre-packed once (fz_ff_chain_pack) into the exact 1 KB MFMA A FRAGMENTS the loops consume, in
//            consumption order -- LDS-DMA (global_load_lds_dwordx4) copies them lane-linear into a 2-stage ring of 64 KB stages and a
//            fragment read is one conflict-free ds_read_b128 at lane * 16 (no swizzle, no address arithmetic).  One read per MFMA, 125 of
//            the LDS's 256 B/clk at full matrix rate; the stream (2.6 MB) is the same for every workgroup: it lives in L2;
//   pipeline stage t holds W1 and b1 of chunk t and W2 of chunk t - 1: in iteration t the UP waves run up(t) + gate(t) and the DOWN waves
//            down(t - 1); one barrier per iteration.  Why two roles: the state of a 32-row strip (x 80 + y 160 registers + operands) needs a
//            whole SIMD's register file, so a wave that does everything is ALONE on its SIMD and its LDS-DMA issue (~60 cycles each), its
//            fragment reads and the erf GELU (2 transcendentals + 12 VALU per element) serialise with its MFMAs -- measured: 105 us per
//            workgroup round, 36 us of it MFMA (profiles/r06_ff_chain_ablation.txt).  Split by role each wave fits 256 registers, two share
//            a SIMD, and one's VALU / DMA / LDS traffic runs under the other's MFMAs;
//   epilogue y + b2 -> fp16 -> LDS tile -> (+ res) -> full-row 16-byte stores of y AND of LayerNorm(y), all eight waves (the arithmetic of
//            fz_gemm_lnout's epilogue, igemm.hip GS == -1: exact two-sweep statistics on the stored fp16 values).
// Arithmetic = the two launches': fp32 accumulation over k ascending with the same MFMA, bias added in fp32 after the K loop, h rounded
// to fp16, y rounded to fp16 before the residual -- the results are BIT-IDENTICAL to fz_gemm(GEGLU) + fz_gemm_lnout (tests/kernel_cases.py
// case_ff_chain).
#include "fz_rt.h"
#include <atomic>
#include <type_traits>
#include "../../include/fatezero_hip.h"

namespace {
constexpr int FC_C = 320;                       // channels (the 64x64 level of SD-1.x)
constexpr int FC_KS = FC_C / 16;                // 20 k steps of the up projection
constexpr int FC_CT = FC_C / 32;                // 10 output tiles of the down projection
constexpr int FC_ROWS = 128;                    // rows per workgroup
constexpr int FC_FRAG = 1024;                   // bytes of one A fragment (64 lanes x 16 B)
constexpr int FC_W1F = 2 * FC_KS;               // 40 fragments: (k step, val | gate)
constexpr int FC_W2F = 2 * FC_CT;               // 20 fragments: (k step of the chunk's 32 units, output tile)
constexpr int FC_BIASF = FC_W1F + FC_W2F;       // fragment slot 60: b1 of the chunk as fp32 [val | gate][32 A rows]
constexpr int FC_STAGE_FRAGS = 64;              // 61 used; 64 = 16 DMA instructions per wave
constexpr int FC_STAGE = FC_STAGE_FRAGS * FC_FRAG;
#ifndef FC_GATE_UP
#define FC_GATE_UP 2   /* register quads (of 4) of a chunk the UP wave gates itself: 0, 2 (k step 0 of the down projection) */
#endif
static_assert(FC_GATE_UP == 0 || FC_GATE_UP == 2, "whole k steps");
constexpr int FC_OSTR = FC_C + 8;               // staging row stride of the epilogue (halves)
constexpr int FC_HBYTES = 4 * 8 * FC_FRAG;       // U hand-over from the UP to the DOWN wave of a pair: [pair][val | gate][4 register quads] x 1 KB (fp32)
constexpr size_t FC_LDS_BYTES = 2 * (size_t)FC_STAGE + FC_HBYTES;
static_assert(4 * 32 * FC_OSTR * 2 <= 2 * FC_STAGE, "epilogue staging fits the ring");
static_assert(FC_LDS_BYTES <= 160 * 1024, "LDS");

// hidden unit (within a chunk of 32) that A row i of the W1 fragments carries: see the header -- unit u sits at accumulator register
// r = 8 (u / 16) + u % 8 of lane half hi = (u % 16) / 8, i.e. A row 8 (r / 4) + 4 hi + r % 4
FZ_HOST_DEVICE int fc_unit_of_arow(int i) {
    const int gq = i >> 3, hi = (i >> 2) & 1, e = i & 3;
    return 16 * (gq >> 1) + 8 * hi + 4 * (gq & 1) + e;
}
}  // namespace

#ifdef FC_TIMING  // trial build (scripts/ff_chain_variants.sh): cycle totals per loop segment of the UP and the DOWN wave of pair 0, workgroup 0
__device__ long long fc_timing[2][8];
#define FC_TK(i) __builtin_amdgcn_sched_barrier(0); const long long tk##i = clock64(); __builtin_amdgcn_sched_barrier(0)
#define FC_TK_ADD(slot, a, b) tacc[slot] += (b) - (a)
#else
#define FC_TK(i) ((void)0)
#define FC_TK_ADD(slot, a, b) ((void)0)
#endif

struct FcArgs {
    const half_t* xn;      // [rows][320]  LayerNorm'ed input of the feed-forward
    const char* packed;    // fz_ff_chain_pack's stream: (inner / 32 + 1) stages of 64 KB
    const half_t* b2;      // [320] or null
    const half_t* res;     // [rows][320] or null
    half_t* y;             // [rows][320]
    half_t* yln;           // [rows][320] or null
    const half_t* gamma;   // LayerNorm of y (with yln)
    const half_t* beta;
    int64_t rows;
    int nchunk;            // inner / 32
    float eps;
};

FZ_KERNEL void __launch_bounds__(512, 2) ff_chain_kernel(FcArgs g) {
    FZ_DYN_SMEM(raw);
    const int tid = threadIdx.x, wave = fz_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    // Wave w and wave w + 4 sit on the same SIMD and own the same 32 rows: w projects UP and gates, w + 4 projects DOWN.
    const bool is_up = wave < 4;
    const int pair = wave & 3;
    // XCD-aware order: blocks b, b + 8, ... share an XCD; give each XCD a contiguous run of row blocks (neighbouring rows, one weight stream)
    const int nt = gridDim.x, bid = blockIdx.x;
    const int q8 = nt >> 3, r8 = nt & 7, xcd = bid & 7;
    const int blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int64_t row0 = (int64_t)blk * FC_ROWS + pair * 32;
    unsigned char* const hbase = raw + 2 * FC_STAGE;  // h hand-over: [slot 2][pair 4][k step 2] fragments of 1 KB

    // ---- weight stream: stage t -> ring buffer t & 1; wave w copies fragments [8 w, 8 w + 8) of the stage ---------------------------------
    // (an LDS-DMA instruction costs the issuing wave ~60 cycles: with two waves per SIMD it runs under the partner's MFMAs)
    const uint32_t lane_off = (uint32_t)lane * 16u;
    auto issue = [&](int t) {
        const char* src = g.packed + (int64_t)t * FC_STAGE + wave * 8 * FC_FRAG;
        unsigned char* dst = raw + (t & 1) * FC_STAGE + wave * 8 * FC_FRAG;
#pragma unroll
        for (int i = 0; i < 8; ++i) fz_glds16_so(src + i * FC_FRAG, lane_off, dst + i * FC_FRAG);
    };
    const int NC = g.nchunk;
    f32x16 yacc[FC_CT];  // (DOWN waves; the UP waves never touch them)

    if (is_up) {
        // ================================================ UP waves ==========================================================================
        // the wave's rows as B fragments of the up projection: lane (row l31, k half hi) holds xn[row][16 s + 8 hi .. + 8)
        half8_t xb[FC_KS];
        {
            int64_t row = row0 + l31;
            row = row < g.rows ? row : g.rows - 1;  // clamped: the tail rows of the last workgroup are computed and never stored
            const half_t* src = g.xn + row * FC_C + hi * 8;
#pragma unroll
            for (int s = 0; s < FC_KS; ++s) xb[s] = fz_ld_h8(src + s * 16);
        }
        // (the compiler's scoreboard must see the row loads retired HERE: with them formally pending it guards every MFMA of the loop with
        //  a counted vmcnt wait that, the DMA of the next stage being younger, drains that DMA early in each iteration)
#pragma unroll
        for (int s = 0; s < FC_KS; ++s) asm volatile("" : "+v"(xb[s]));
        issue(0);
#ifdef FC_TIMING
        long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const long long tk_start = clock64();
#endif
        for (int t = 0; t <= NC; ++t) {
            FC_TK(0);
            fz_wait_vm0();          // this wave's part of stage t has landed ...
            FC_TK(1);
            fz_barrier_nodrain();   // ... and everybody's; stage t - 1 and the h slot of chunk t - 2 are free
            FC_TK(2);
            FC_TK_ADD(0, tk0, tk1);
            FC_TK_ADD(1, tk1, tk2);
            if (t == NC) break;
#ifndef FC_TRIAL_NODMA
            issue(t + 1);
#else
            if (t == 0) issue(t + 1);
#endif
            FC_TK(3);
            FC_TK_ADD(2, tk2, tk3);
            const fz_lds_addr base = fz_lds_addr_of(raw + (t & 1) * FC_STAGE) + lane_off;
            // up(t): U = W1 chunk . x^T -- 40 fragments in 8 batches of 5 (k steps 5 q / 2 ..), the reads of batch q + 1 issued before the
            // MFMAs of batch q
            half8_t fa[10], fb[10];
            auto load = [&](half8_t* f, int q) {
#pragma unroll
                for (int i = 0; i < 10; ++i) f[i] = fz_lds_ld_h8(base, (q * 10 + i) * FC_FRAG);
            };
            f32x16 uv = fz_zero_f16v(), ug = fz_zero_f16v();
            auto up5 = [&](const half8_t* f, int q) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    uv = fz_mfma_32x32x16_f16(f[2 * i], xb[5 * q + i], uv);
                    ug = fz_mfma_32x32x16_f16(f[2 * i + 1], xb[5 * q + i], ug);
                }
            };
            load(fa, 0);
            FZ_SCHED_FENCE();
            load(fb, 1);
            up5(fa, 0);
            FZ_SCHED_FENCE();
            load(fa, 2);
            up5(fb, 1);
            FZ_SCHED_FENCE();
            load(fb, 3);
            up5(fa, 2);
            FZ_SCHED_FENCE();
            up5(fb, 3);
            FC_TK(4);
            FC_TK_ADD(3, tk3, tk4);
            // + b1 (fp32, after the K loop as fz_gemm's epilogue does: register 4 gq + e <-> A row 8 gq + 4 hi + e); the val / gate tiles go to
            // the DOWN wave in fp32, lane-linear (8 x 1 KB): it gates them (the erf GELU runs beside THIS wave's next 40 MFMAs)
            const float* bl = reinterpret_cast<const float*>(raw + (t & 1) * FC_STAGE + FC_BIASF * FC_FRAG);
            // The erf GELU is ~100 VALU cycles per element, as much VALU time as the chunk has MFMA time: the pair shares it.  This wave gates
            // register quads [0, FC_GATE_UP) -- k step 0 of the down projection -- and hands them over as an fp16 B fragment; quads
            // [FC_GATE_UP, 4) travel as fp32 val / gate and the DOWN wave gates them beside this wave's next 40 MFMAs.
            f32x4 ov[4], og[4];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + 8 * gq + 4 * hi);
                const f32x4 bg = *reinterpret_cast<const f32x4*>(bl + 32 + 8 * gq + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ov[gq][e] = uv[4 * gq + e] + bv[e];
                    og[gq][e] = ug[4 * gq + e] + bg[e];
                }
            }
            half8_t hb0;
#pragma unroll
            for (int gq = 0; gq < FC_GATE_UP; ++gq)
#pragma unroll
                for (int e = 0; e < 4; ++e) hb0[4 * gq + e] = (half_t)(ov[gq][e] * fz_gelu_erf(og[gq][e]));
            FC_TK(5);
            FC_TK_ADD(4, tk4, tk5);
            fz_barrier_nodrain();   // (B) the DOWN wave has taken chunk t - 1 out of the hand-over slot
            FC_TK(6);
            FC_TK_ADD(5, tk5, tk6);
            unsigned char* us = hbase + pair * 8 * FC_FRAG + lane_off;
            if (FC_GATE_UP == 2) *reinterpret_cast<half8_t*>(us) = hb0;
#pragma unroll
            for (int gq = FC_GATE_UP; gq < 4; ++gq) {
                *reinterpret_cast<f32x4*>(us + gq * FC_FRAG) = ov[gq];
                *reinterpret_cast<f32x4*>(us + (4 + gq) * FC_FRAG) = og[gq];
            }
            FC_TK(7);
            FC_TK_ADD(6, tk6, tk7);
        }
#ifdef FC_TIMING
        tacc[7] = clock64() - tk_start;
        if (blockIdx.x == 0 && tid == 0)
            for (int i = 0; i < 8; ++i) fc_timing[0][i] = tacc[i];
#endif
    } else {
        // ================================================ DOWN waves ========================================================================
#ifndef FC_TRIAL_NOPRIO
        fz_setprio_hi();  // this wave's VALU (the gate) and LDS / DMA issue go first: they fit between the partner's MFMAs, not the other way round
#endif
#pragma unroll
        for (int c = 0; c < FC_CT; ++c) yacc[c] = fz_zero_f16v();
        issue(0);
#ifdef FC_TIMING
        long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const long long tk_start = clock64();
#endif
        for (int t = 0; t <= NC; ++t) {
            FC_TK(0);
            fz_wait_vm0();
            FC_TK(1);
            fz_barrier_nodrain();
            FC_TK(2);
            FC_TK_ADD(0, tk0, tk1);
            FC_TK_ADD(1, tk1, tk2);
#ifndef FC_TRIAL_NODMA
            if (t < NC) issue(t + 1);
#else
            if (t == 0) issue(t + 1);
#endif
            FC_TK(3);
            FC_TK_ADD(2, tk2, tk3);
            if (t == 0) {
                fz_barrier_nodrain();  // (B)
                continue;
            }
            // gate(t - 1) + down(t - 1): U(t - 1) was written before this iteration's barrier; W2 of chunk t - 1 travels in stage t
            const fz_lds_addr base = fz_lds_addr_of(raw + (t & 1) * FC_STAGE) + lane_off;
            const unsigned char* us = hbase + pair * 8 * FC_FRAG + lane_off;
            half8_t hb[2];  // h (fp16) as the two B fragments of the down projection: k step s <- registers 8 s .. 8 s + 7
            f32x4 pv[4], pg[4];
            if (FC_GATE_UP == 2) hb[0] = *reinterpret_cast<const half8_t*>(us);
#pragma unroll
            for (int gq = FC_GATE_UP; gq < 4; ++gq) {
                pv[gq] = *reinterpret_cast<const f32x4*>(us + gq * FC_FRAG);
                pg[gq] = *reinterpret_cast<const f32x4*>(us + (4 + gq) * FC_FRAG);
            }
            half8_t fa[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) fa[i] = fz_lds_ld_h8(base, (FC_W1F + i) * FC_FRAG);
#pragma unroll
            for (int gq = FC_GATE_UP; gq < 4; ++gq)
#pragma unroll
                for (int e = 0; e < 4; ++e) hb[gq >> 1][4 * (gq & 1) + e] = (half_t)(pv[gq][e] * fz_gelu_erf(pg[gq][e]));
#ifdef FC_TIMING
            asm volatile("" : "+v"(hb[0]), "+v"(hb[1]));
#endif
            FC_TK(4);
            FC_TK_ADD(3, tk3, tk4);
#pragma unroll
            for (int c = 0; c < FC_CT; ++c) yacc[c] = fz_mfma_32x32x16_f16(fa[c], hb[0], yacc[c]);
            FZ_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < 10; ++i) fa[i] = fz_lds_ld_h8(base, (FC_W1F + 10 + i) * FC_FRAG);
#pragma unroll
            for (int c = 0; c < FC_CT; ++c) yacc[c] = fz_mfma_32x32x16_f16(fa[c], hb[1], yacc[c]);
            FC_TK(5);
            FC_TK_ADD(4, tk4, tk5);
            if (t < NC) fz_barrier_nodrain();  // (B) U(t - 1) is in registers (long since): the UP wave may overwrite the slot
            FC_TK(6);
            FC_TK_ADD(5, tk5, tk6);
        }
#ifdef FC_TIMING
        tacc[7] = clock64() - tk_start;
        if (blockIdx.x == 0 && tid == 256)
            for (int i = 0; i < 8; ++i) fc_timing[1][i] = tacc[i];
#endif
    }

    // ---- epilogue: + b2 -> fp16 -> LDS tile [128 rows][320 + 8] (the DOWN waves own the accumulators) -> (+ res) -> y and LayerNorm(y), full
    //      rows, ALL eight waves: wave w takes rows [16 w, 16 w + 16) of the tile
    __syncthreads();  // every wave is done with the ring (no DMA in flight: the last stage was waited for)
    half_t* Call = reinterpret_cast<half_t*>(raw);
    if (!is_up) {
        half_t* Cs = Call + pair * 32 * FC_OSTR;
#pragma unroll
        for (int c = 0; c < FC_CT; ++c)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int co = c * 32 + 8 * gq + 4 * hi;
                half4_t bv;
                if (g.b2 != nullptr) {
                    bv = *reinterpret_cast<const half4_t*>(g.b2 + co);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = (half_t)0.0f;
                }
                half4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (half_t)(yacc[c][4 * gq + e] + (float)bv[e]);
                *reinterpret_cast<half4_t*>(Cs + l31 * FC_OSTR + co) = v;
            }
    }
    __syncthreads();
    const int l8 = lane & 7;
    FzRow5 gmv, btv;
    if (g.yln != nullptr) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            gmv.c[i] = fz_ld_h8(g.gamma + (l8 + 8 * i) * 8);
            btv.c[i] = fz_ld_h8(g.beta + (l8 + 8 * i) * 8);
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {  // 8 rows per pass: 8 lanes per row, 5 chunks of 8 channels per lane
        const int rl = wave * 16 + it * 8 + (lane >> 3);
        const int64_t px = (int64_t)blk * FC_ROWS + rl;
        const bool ok = px < g.rows;
        const int64_t pxc = ok ? px : g.rows - 1;
        FzRow5 v;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const half8_t a = fz_ld_h8(Call + rl * FC_OSTR + (l8 + 8 * i) * 8);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)a[e];
            if (g.res != nullptr) {
                const half8_t r = fz_ld_h8(g.res + pxc * FC_C + (l8 + 8 * i) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v.c[i][e] = (half_t)f[e];
            if (ok) fz_st_h8(g.y + px * FC_C + (l8 + 8 * i) * 8, v.c[i]);
        }
        if (g.yln == nullptr) continue;
        // LayerNorm of the stored row: the out-of-line body igemm.hip's GS == -1 epilogue calls (fz_rt.h): the same bits
        const FzRow5 o = fz_ln_row320(v, gmv, btv, g.eps);
        if (ok) {
#pragma unroll
            for (int i = 0; i < 5; ++i) fz_st_h8(g.yln + px * FC_C + (l8 + 8 * i) * 8, o.c[i]);
        }
    }
}

// ---- packing: one thread per 16 bytes of the stream ----------------------------------------------------------------------------------------
struct FcPackArgs {
    const half_t* w1;   // [2 * inner][320]: rows [0, inner) = val, [inner, 2 inner) = gate (diffusers GEGLU.proj, chunk(2, dim=-1))
    const half_t* b1;   // [2 * inner] or null
    const half_t* w2;   // [320][inner]
    char* out;
    int inner, nchunk;
};

FZ_KERNEL void __launch_bounds__(256) ff_chain_pack_kernel(FcPackArgs g) {
    const int64_t total = (int64_t)(g.nchunk + 1) * FC_STAGE / 16;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int t = (int)(id / (FC_STAGE / 16));
        const int in = (int)(id - (int64_t)t * (FC_STAGE / 16));
        const int f = in >> 6, ln = in & 63, l31 = ln & 31, hi = ln >> 5;
        half8_t v = fz_zero_h8();
        if (f < FC_W1F) {
            if (t < g.nchunk) {  // W1 of chunk t: fragment (k step s, val | gate), A row l31 <- hidden unit fc_unit_of_arow(l31)
                const int s = f >> 1, gate = f & 1;
                const int row = (gate ? g.inner : 0) + t * 32 + fc_unit_of_arow(l31);
                v = fz_ld_h8(g.w1 + (int64_t)row * FC_C + 16 * s + 8 * hi);
            }
        } else if (f < FC_BIASF) {
            if (t > 0) {  // W2 of chunk t - 1: fragment (k step s of the chunk, output tile c), natural k order
                const int q = f - FC_W1F, s = q / FC_CT, c = q - s * FC_CT;
                v = fz_ld_h8(g.w2 + (int64_t)(c * 32 + l31) * g.inner + (t - 1) * 32 + 16 * s + 8 * hi);
            }
        } else if (f == FC_BIASF) {
            if (t < g.nchunk && g.b1 != nullptr && ln < 16) {  // 64 floats: [val | gate][A row]; this thread writes floats [4 ln, 4 ln + 4)
                f32x4 b;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int idx = 4 * ln + e, gate = idx >> 5, i = idx & 31;
                    b[e] = (float)g.b1[(gate ? g.inner : 0) + t * 32 + fc_unit_of_arow(i)];
                }
                v = __builtin_bit_cast(half8_t, b);
            }
        }
        *reinterpret_cast<half8_t*>(g.out + id * 16) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
//                                                   host side
// ---------------------------------------------------------------------------------------------------------------
#ifdef FC_TIMING
extern "C" int fz_ff_chain_timing(long long* out) {  // trial builds only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fc_timing), sizeof(long long) * 16) == hipSuccess ? FZ_OK : FZ_ERR_LAUNCH;
}
#endif

extern "C" int fz_ff_chain_ok(int64_t rows, int channels, int inner) {
    return rows > 0 && rows < (1ll << 38) && channels == FC_C && inner > 0 && inner % 32 == 0 && inner <= 32 * 4096;
}

// Where is the one launch the faster form on MI355X?  Measured against fz_gemm(GEGLU) + fz_gemm_lnout (profiles/r06_ff_chain_ab.txt): see
// DESIGN.md section 3.  A workgroup streams the whole 2.6 MB weight set for its 128 rows, so the launch wants the chip full.
extern "C" int fz_ff_chain_preferred(int64_t rows, int channels, int inner) {
    return fz_ff_chain_ok(rows, channels, inner) && inner == 4 * FC_C && rows >= 128 * 192;
}

extern "C" int64_t fz_ff_chain_pack_bytes(int channels, int inner) {
    if (!fz_ff_chain_ok(1, channels, inner)) return 0;
    return (int64_t)(inner / 32 + 1) * FC_STAGE;
}

extern "C" int fz_ff_chain_pack(const void* w1, const void* b1, const void* w2, void* packed, int channels, int inner, void* stream) {
    if (!w1 || !w2 || !packed) return FZ_ERR_BAD_ARG;
    if (!fz_ff_chain_ok(1, channels, inner)) return FZ_ERR_UNSUPPORTED;
    FcPackArgs g = {(const half_t*)w1, (const half_t*)b1, (const half_t*)w2, (char*)packed, inner, inner / 32};
    const int64_t total = (int64_t)(g.nchunk + 1) * FC_STAGE / 16;
    FZ_LAUNCH(ff_chain_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, g);
    return fz_last_launch_status();
}

extern "C" int fz_ff_chain(const void* xn, const void* packed, const void* b2, const void* res, void* y, const void* ln_gamma,
                           const void* ln_beta, float ln_eps, void* y_ln, int64_t rows, int channels, int inner, void* stream) {
    if (!xn || !packed || !y) return FZ_ERR_BAD_ARG;
    if (!fz_ff_chain_ok(rows, channels, inner)) return FZ_ERR_UNSUPPORTED;
    if (y_ln != nullptr && (!ln_gamma || !ln_beta)) return FZ_ERR_BAD_ARG;
    FcArgs g = {};
    g.xn = (const half_t*)xn;
    g.packed = (const char*)packed;
    g.b2 = (const half_t*)b2;
    g.res = (const half_t*)res;
    g.y = (half_t*)y;
    g.yln = (half_t*)y_ln;
    g.gamma = (const half_t*)ln_gamma;
    g.beta = (const half_t*)ln_beta;
    g.rows = rows;
    g.nchunk = inner / 32;
    g.eps = ln_eps;
    const int64_t nwg = (rows + FC_ROWS - 1) / FC_ROWS;
#ifndef FZ_EMU
    static std::atomic<uint64_t> attr_set_mask{0};  // LDS above 64 KB is an opt-in function attribute, per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FZ_ERR_LAUNCH;
    if (dev >= 64 || !(attr_set_mask.load(std::memory_order_relaxed) >> dev & 1)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)FC_LDS_BYTES) != hipSuccess)
            return FZ_ERR_LAUNCH;
        if (dev < 64) attr_set_mask.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
#endif
    FZ_LAUNCH(ff_chain_kernel, dim3((unsigned)nwg), dim3(512), FC_LDS_BYTES, stream, g);
    return fz_last_launch_status();
}
