// ff_chain.hip -- the FeedForward CHAIN of the 64x64-level transformer block in ONE launch:
//     hidden_states = ff(norm3(hidden_states)) + hidden_states ; norm_temporal(hidden_states)          (attention.py:312-321, 327-331)
//   with ff = diffusers FeedForward [GEGLU(dim, 4 dim) -> Linear(4 dim, dim)] [3P]:  val | gate = x W1^T + b1 ;  h = val * gelu(gate) ;
//   y = h W2^T + b2.   Per-op this was fz_gemm(GEGLU) + fz_gemm_lnout: two launches with a rows x 1280 fp16 round trip through HBM between
//   them (168 MB at 16 frames), both short-K / epilogue-dominated (0.25 of the MFMA roof, profiles/r05_job_breakdown_by_op_and_shape.txt).
//
// One workgroup = 128 rows, four waves of 32 rows, ONE wave per SIMD (512 registers each).  Nothing of the activations ever touches LDS:
//   x        a wave's 32 rows x 320 channels of LN(x) live in REGISTERS for the whole launch as the 20 B fragments of the up projection
//            (lane = row + 32 * k half, 8 halves: 80 VGPRs);
//   hidden   walked in chunks of 32 units: val / gate tiles U[32 units][32 rows] = W1 chunk . x^T (2 x 20 MFMAs 32x32x16), + b1, gated in
//            registers.  The accumulator layout of the MFMA (lane = row, 16 registers = 16 units) IS the B-fragment layout of the down
//            projection when W1's rows are packed in the order the registers want (unit 16 s + 8 hi + j at A row 8 (2 s + j / 4) + 4 hi
//            + j % 4): h never crosses lanes, never leaves the wave, and the down projection contracts it in natural k order;
//   y        the 32 rows x 320 outputs accumulate in fp32 REGISTERS across all 40 chunks (10 tiles, 160 registers);
//   weights  the only thing streamed: pre-packed once (fz_ff_chain_pack) into the exact 1 KB MFMA A FRAGMENTS the loop consumes, in
//            consumption order -- LDS-DMA (global_load_lds_dwordx4) copies them lane-linear into a 2-stage ring of 64 KB stages and a
//            fragment read is one conflict-free ds_read_b128 at lane * 16 (no swizzle, no address arithmetic, no im2row).  One read per
//            MFMA, 125 of the LDS's 256 B/clk at full matrix rate; the stream (2.6 MB) is the same for every workgroup: it lives in L2.
//   pipeline stage t holds W1 of chunk t, b1 of chunk t and W2 of chunk t - 1: iteration t runs up(t) while the VALU gates chunk t - 1
//            (erf GELU: ~25 instructions per element in the shadow of the MFMAs) and then down(t - 1).  One barrier per iteration
//            (60 MFMAs per wave); the DMA of stage t + 1 is in flight for the whole of iteration t.
//   epilogue y + b2 -> fp16 -> wave-private LDS tile -> (+ res) -> full-row 16-byte stores of y AND of LayerNorm(y) (the arithmetic of
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
constexpr int FC_OSTR = FC_C + 8;               // staging row stride of the epilogue (halves)
constexpr size_t FC_LDS_BYTES = 2 * (size_t)FC_STAGE;
static_assert(4 * 32 * FC_OSTR * 2 <= 2 * FC_STAGE, "epilogue staging fits the ring");

// hidden unit (within a chunk of 32) that A row i of the W1 fragments carries: see the header -- unit u sits at accumulator register
// r = 8 (u / 16) + u % 8 of lane half hi = (u % 16) / 8, i.e. A row 8 (r / 4) + 4 hi + r % 4
FZ_HOST_DEVICE int fc_unit_of_arow(int i) {
    const int gq = i >> 3, hi = (i >> 2) & 1, e = i & 3;
    return 16 * (gq >> 1) + 8 * hi + 4 * (gq & 1) + e;
}
}  // namespace

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

FZ_KERNEL void __launch_bounds__(256, 1) ff_chain_kernel(FcArgs g) {
    FZ_DYN_SMEM(raw);
    const int tid = threadIdx.x, wave = fz_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order: blocks b, b + 8, ... share an XCD; give each XCD a contiguous run of row blocks (neighbouring rows, one weight stream)
    const int nt = gridDim.x, bid = blockIdx.x;
    const int q8 = nt >> 3, r8 = nt & 7, xcd = bid & 7;
    const int blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int64_t row0 = (int64_t)blk * FC_ROWS + wave * 32;

    // ---- the wave's rows as B fragments of the up projection: lane (row l31, k half hi) holds xn[row][16 s + 8 hi .. + 8) ----------------
    half8_t xb[FC_KS];
    {
        int64_t row = row0 + l31;
        row = row < g.rows ? row : g.rows - 1;  // clamped: the tail rows of the last workgroup are computed and never stored
        const half_t* src = g.xn + row * FC_C + hi * 8;
#pragma unroll
        for (int s = 0; s < FC_KS; ++s) xb[s] = fz_ld_h8(src + s * 16);
    }

    // ---- weight stream: stage t -> ring buffer t & 1; wave w copies fragments [16 w, 16 w + 16) of the stage ------------------------------
    const uint32_t lane_off = (uint32_t)lane * 16u;
    // (issued in quarters: an LDS-DMA instruction costs the issuing wave ~60 cycles, two MFMA slots -- spread between the MFMA batches of an
    //  iteration they fill the matrix pipe's queue time instead of stopping it for 1 000 cycles at the top of the iteration)
    auto issue4 = [&](int t, int part) {
        const char* src = g.packed + (int64_t)t * FC_STAGE + (wave * 16 + part * 4) * FC_FRAG;
        unsigned char* dst = raw + (t & 1) * FC_STAGE + (wave * 16 + part * 4) * FC_FRAG;
#pragma unroll
        for (int i = 0; i < 4; ++i) fz_glds16_so(src + i * FC_FRAG, lane_off, dst + i * FC_FRAG);
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int part = 0; part < 4; ++part) issue4(t, part);
    };

    f32x16 yacc[FC_CT];
#pragma unroll
    for (int c = 0; c < FC_CT; ++c) yacc[c] = fz_zero_f16v();
    f32x16 pv = fz_zero_f16v(), pg = fz_zero_f16v();  // val / gate tile (+ b1) of the previous chunk

    // (the compiler's scoreboard must see the row loads retired HERE: with them formally pending it guards every MFMA of the loop with a
    //  counted vmcnt wait that, the DMA of the next stage being younger, drains that DMA in the first quarter of each iteration)
#pragma unroll
    for (int s = 0; s < FC_KS; ++s) asm volatile("" : "+v"(xb[s]));
    issue(0);
    const int NC = g.nchunk;
    // One iteration = 60 fragment reads + 60 MFMAs in 6 batches of 10: W1 (k steps 5 q .. 5 q + 4, val | gate) for q = 0..3, W2 (k step
    // s, the 10 output tiles) for s = 0, 1.  The reads of batch q + 1 are issued BEFORE the MFMAs of batch q (two register sets), so an
    // MFMA never waits for its own ds_read; the gating VALU of chunk t - 1 (4 registers per batch) rides behind the MFMAs of up(t).
    // The first iteration (nothing to gate yet) and the last one (nothing to project up any more) are instantiations of their own: the
    // steady-state loop body is ONE basic block without a branch.
    half8_t fa[10], fb[10];
    auto iteration = [&](int t, auto UP, auto DOWN) {
        constexpr bool up = decltype(UP)::value, down = decltype(DOWN)::value;
        fz_wait_vm0();          // this wave's part of stage t has landed ...
        fz_barrier_nodrain();   // ... and everybody's; every wave is done reading stage t - 1 (its ds_reads fed MFMAs already issued)
#ifdef FC_TRIAL_NODMA
        const bool dma = up && t == 0;
#else
        constexpr bool dma = up;
#endif
#ifdef FC_TRIAL_DMA_BURST
        if (dma) issue(t + 1);
#endif
        const fz_lds_addr base = fz_lds_addr_of(raw + (t & 1) * FC_STAGE) + lane_off;
        auto load = [&](half8_t* f, int q) {
#ifdef FC_TRIAL_NOLDS
            if (t > 1) return;
#endif
#pragma unroll
            for (int i = 0; i < 10; ++i) f[i] = fz_lds_ld_h8(base, (q * 10 + i) * FC_FRAG);
        };
        f32x16 uv = fz_zero_f16v(), ug = fz_zero_f16v();
#ifdef FC_TRIAL_UP4
        f32x16 uv2 = fz_zero_f16v(), ug2 = fz_zero_f16v();
#endif
        half8_t hb[2];
        auto gate4 = [&](int p) {  // registers 4 p .. 4 p + 3 of the previous chunk's val / gate tiles -> h (fp16), the down projection's B operand
#pragma unroll
            for (int e = 0; e < 4; ++e) hb[p >> 1][4 * (p & 1) + e] = (half_t)(pv[4 * p + e] * fz_gelu_erf(pg[4 * p + e]));
        };
        auto up5 = [&](const half8_t* f, int q) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
#ifdef FC_TRIAL_UP4
                if (i & 1) {
                    uv2 = fz_mfma_32x32x16_f16(f[2 * i], xb[5 * q + i], uv2);
                    ug2 = fz_mfma_32x32x16_f16(f[2 * i + 1], xb[5 * q + i], ug2);
                    continue;
                }
#endif
                uv = fz_mfma_32x32x16_f16(f[2 * i], xb[5 * q + i], uv);
                ug = fz_mfma_32x32x16_f16(f[2 * i + 1], xb[5 * q + i], ug);
            }
        };
        auto down10 = [&](const half8_t* f, int s) {
#pragma unroll
            for (int c = 0; c < FC_CT; ++c) yacc[c] = fz_mfma_32x32x16_f16(f[c], hb[s], yacc[c]);
        };
        if (up) {
            load(fa, 0);
            FZ_SCHED_FENCE();
            load(fb, 1);
            up5(fa, 0);
#ifndef FC_TRIAL_DMA_BURST
            if (dma) issue4(t + 1, 0);
#endif
            if (down) gate4(0);
            FZ_SCHED_FENCE();
            load(fa, 2);
            up5(fb, 1);
#ifndef FC_TRIAL_DMA_BURST
            if (dma) issue4(t + 1, 1);
#endif
            if (down) gate4(1);
            FZ_SCHED_FENCE();
            load(fb, 3);
            up5(fa, 2);
#ifndef FC_TRIAL_DMA_BURST
            if (dma) issue4(t + 1, 2);
#endif
            if (down) gate4(2);
            FZ_SCHED_FENCE();
            if (down) load(fa, 4);
            up5(fb, 3);
#ifndef FC_TRIAL_DMA_BURST
            if (dma) issue4(t + 1, 3);
#endif
            if (down) gate4(3);
            FZ_SCHED_FENCE();
        } else {  // the last iteration: only the previous chunk's gate and down projection are left
            load(fa, 4);
#pragma unroll
            for (int p = 0; p < 4; ++p) gate4(p);
        }
        if (down) {
            load(fb, 5);
            down10(fa, 0);
            FZ_SCHED_FENCE();
            down10(fb, 1);
        }
        if (up) {  // + b1 (fp32, after the K loop as fz_gemm's epilogue does): register 4 gq + e <-> A row 8 gq + 4 hi + e
#ifdef FC_TRIAL_UP4
            uv += uv2;
            ug += ug2;
#endif
            const float* bl = reinterpret_cast<const float*>(raw + (t & 1) * FC_STAGE + FC_BIASF * FC_FRAG);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bl + 8 * gq + 4 * hi);
                const f32x4 bg = *reinterpret_cast<const f32x4*>(bl + 32 + 8 * gq + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pv[4 * gq + e] = uv[4 * gq + e] + bv[e];
                    pg[4 * gq + e] = ug[4 * gq + e] + bg[e];
                }
            }
        }
    };
    iteration(0, std::true_type(), std::false_type());
    for (int t = 1; t < NC; ++t) iteration(t, std::true_type(), std::true_type());
    iteration(NC, std::false_type(), std::true_type());

    // ---- epilogue: + b2 -> fp16 -> wave-private LDS tile [32 rows][320 + 8] -> (+ res) -> y and LayerNorm(y), full rows --------------------
    __syncthreads();  // every wave is done with the ring (no DMA in flight: the last stage was waited for)
    half_t* Cs = reinterpret_cast<half_t*>(raw) + wave * 32 * FC_OSTR;
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
    fz_wave_lds_sync();
    const int l8 = lane & 7;
    half8_t gmv[5], btv[5];
    if (g.yln != nullptr) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            gmv[i] = fz_ld_h8(g.gamma + (l8 + 8 * i) * 8);
            btv[i] = fz_ld_h8(g.beta + (l8 + 8 * i) * 8);
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {  // 8 rows per pass: 8 lanes per row, 5 chunks of 8 channels per lane
        const int rl = it * 8 + (lane >> 3);
        const int64_t px = row0 + rl;
        const bool ok = px < g.rows;
        const int64_t pxc = ok ? px : g.rows - 1;
        half8_t v[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const half8_t a = fz_ld_h8(Cs + rl * FC_OSTR + (l8 + 8 * i) * 8);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)a[e];
            if (g.res != nullptr) {
                const half8_t r = fz_ld_h8(g.res + pxc * FC_C + (l8 + 8 * i) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[i][e] = (half_t)f[e];
            if (ok) fz_st_h8(g.y + px * FC_C + (l8 + 8 * i) * 8, v[i]);
        }
        if (g.yln == nullptr) continue;
        // LayerNorm of the stored row: the arithmetic (and summation order) of igemm.hip's GS == -1 epilogue
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                s0 += (float)v[i][e];
                s1 += (float)v[i][e + 1];
            }
        const float mean = fz_sum8(s0 + s1) * (1.0f / 320.0f);
        float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const float d0 = (float)v[i][e] - mean, d1 = (float)v[i][e + 1] - mean;
                q0 += d0 * d0;
                q1 += d1 * d1;
            }
        const float rstd = 1.0f / sqrtf(fz_sum8(q0 + q1) * (1.0f / 320.0f) + g.eps);
        if (ok) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)v[i][e] - mean) * rstd * (float)gmv[i][e] + (float)btv[i][e]);
                fz_st_h8(g.yln + px * FC_C + (l8 + 8 * i) * 8, o);
            }
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
    FZ_LAUNCH(ff_chain_kernel, dim3((unsigned)nwg), dim3(256), FC_LDS_BYTES, stream, g);
    return fz_last_launch_status();
}
