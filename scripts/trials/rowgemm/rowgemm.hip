// rowgemm.hip -- the row-streaming projection of the 64x64 level: LayerNorm + Linear (+ bias, residuals, V^T) in ONE launch
//   * `x = attn(norm(x)) + x` of SpatioTemporalTransformerBlock.forward (attention.py:295-337): norm1 + to_q | to_k | to_v (:340-372),
//     norm2 + attn2.to_q (attention_register.py:71-80), norm_temporal + attn_temporal.to_q | to_k | to_v (attention.py:327-337);
//   * the out-projections + residual (`to_out[0]`, attention.py:216 [3P diffusers CrossAttention]) and proj_in (attention.py:64-66).
// These launches move 2 (rows K + rows N (1 + residuals)) bytes for 2 rows K N FLOP with K = 320: <= 160 FLOP per byte, far below the
// chip's ridge -- their roof is HBM.  The implicit-GEMM kernel (igemm.hip) runs them at 0.22-0.25 of it: a 128 / 256-row tile per CU,
// every CU loading, multiplying and storing in lockstep, a LayerNorm launch in front that reads and writes the same rows once more.
//
// Here a workgroup owns 128 ROWS and ALL of their K = 320 channels:
//   weights  a 320-column slice of W stays in REGISTERS for the whole launch: wave w holds output channels [80 w, 80 w + 80) x K = 320
//            as 5 x 10 A fragments of v_mfma_f32_16x16x32_f16 (200 VGPRs of the 512 a one-wave-per-SIMD kernel owns) -- the matrix pipe
//            is fed from registers on one side, and the LDS reads per FLOP are those of the row operand alone;
//   rows     the 128 x 320 block lands in LDS by LDS-DMA (global_load_lds_dwordx4) ONCE, as [k step][16 tokens][4 chunks] blocks of 1 KB
//            -- one DMA instruction each -- whose chunk position is XOR-swizzled so that both the B-fragment reads (lane = token + 16 k
//            chunk) and the LayerNorm pass (4 lanes per token) are conflict-free ds_read_b128;
//   LN       whole rows are in the workgroup, so the LayerNorm of the consuming projection is a pass over the LDS block (two-sweep
//            statistics in fp32, 4 lanes per token, DPP sums), written back in place as the fp16 the stand-alone kernel would have
//            stored: the LayerNorm launch and its round trip through HBM disappear;
//   N > 320  further 320-column slices of W (q | k | v: three) are passes over the SAME resident rows; the slices at or beyond
//            `vt_split` leave transposed as V^T[frame][channel][token], the attention kernels' value operand (fz_gemm_qkvt's contract);
//   out      accumulators (+ bias) -> fp16 -> LDS staging -> (+ res) (+ res2) -> full-row 16-byte stores.
// fp32 accumulation over k ascending; one rounding to fp16 at the end, as fz_gemm.
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"

struct RgArgs {
    const half_t* x;      // [rows][ldx], 320 channels read
    const half_t* w;      // [n_out][ldw]
    const half_t* bias;   // [n_out] or null
    const half_t* res;    // [rows][ldres] or null (plain columns only)
    const half_t* res2;
    half_t* y;            // [rows][ldy]: columns [0, vt_split) (all of them without V^T)
    half_t* yt;           // [frames][n_out - vt_split][ldyt] or null
    const half_t* gamma;  // LayerNorm weight / bias [320] or null
    const half_t* beta;
    int64_t ldx, ldw, ldy, ldres, yt_bs, ldyt, rows;
    int n_out, vt_split, vt_rows;
    float eps;
};

namespace {
constexpr int RG_K = 320, RG_KS = RG_K / 32;   // 10 k steps of 32
constexpr int RG_ROWS = 128, RG_SUB = 32;      // rows per workgroup / per sub-tile (one per wave for the DMA and the LayerNorm pass)
constexpr int RG_SLOT = RG_SUB * RG_K;         // halves per sub-tile slot (20 KB)
constexpr int RG_OSTR = RG_K + 8;              // staging row stride (plain): 328 halves
constexpr int RG_TSTR = RG_SUB + 8;            // staging row stride (transposed): 32 tokens + 8
constexpr int RG_STAGE = RG_K * RG_TSTR;       // 12 800 halves >= 32 * 328
constexpr int RG_MAX_N = 1920;
constexpr size_t RG_LDS_BYTES = (size_t)(4 * RG_SLOT + RG_STAGE) * 2;

// chunk swizzle: physical 16-byte position of logical chunk c of token t inside its 64-byte row = c ^ g(t >> 2), g = (0, 2, 3, 1):
// conflict-free for ds_read_b128's lane groups {0-3, 12-15, 20-27} ... both when lane = token + 16 * chunk (MFMA B fragments) and when
// lane = 4 * token + chunk (LayerNorm pass, LDS-DMA destination order)
FZ_DEVICE int rg_g(int tq) { return (0x78 >> (2 * tq)) & 3; }
// halves offset of (k step ks, token tt of the sub-tile, logical chunk c) inside a sub-tile slot
FZ_DEVICE int rg_off(int ks, int tt, int c) { return ((ks * 2 + (tt >> 4)) * 16 + (tt & 15)) * 32 + ((c ^ rg_g((tt & 15) >> 2)) * 8); }
}  // namespace

template <bool LN, bool VT>
FZ_KERNEL void __launch_bounds__(256, 1) rowgemm320_kernel(RgArgs g) {
    FZ_DYN_SMEM(raw);
    half_t* XS = reinterpret_cast<half_t*>(raw);
    half_t* OS = XS + 4 * RG_SLOT;
    const int tid = threadIdx.x, wave = fz_uniform(tid >> 6), lane = tid & 63;
    const int64_t row0 = (int64_t)blockIdx.x * RG_ROWS;

    // ---- 1. this wave's 32 rows -> LDS: 20 LDS-DMA instructions of 1 KB = (k step, 16-token half) blocks; lane = 4 * token + position
    {
        const int t = lane >> 2, c = (lane & 3) ^ rg_g(t >> 2);
        half_t* dst = XS + wave * RG_SLOT;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int64_t row = row0 + wave * RG_SUB + h * 16 + t;
            row = row < g.rows ? row : g.rows - 1;  // clamped: the tail rows of the last workgroup are computed and never stored
            const half_t* src = g.x + row * g.ldx + c * 8;
#pragma unroll
            for (int ks = 0; ks < RG_KS; ++ks) fz_glds16(src + ks * 32, dst + (ks * 2 + h) * 512);
        }
    }
    half8_t gm[LN ? RG_KS : 1], bt[LN ? RG_KS : 1];
    if constexpr (LN) {  // LayerNorm pass: lane = 4 * token + j owns chunks (k step i, j), i = 0..9
#pragma unroll
        for (int i = 0; i < RG_KS; ++i) {
            gm[i] = fz_ld_h8(g.gamma + i * 32 + (lane & 3) * 8);
            bt[i] = fz_ld_h8(g.beta + i * 32 + (lane & 3) * 8);
        }
    }
    fz_wait_vm0();  // this wave's rows have landed (its own LDS-DMA only: nothing else is outstanding yet)
    fz_wave_lds_sync();  // (a lane reads chunks other lanes of its wave fetched: a wave-wide wait on the hardware, a meeting point on the emulator)

    // ---- 2. weights of the first slice -> registers; they arrive while the LayerNorm pass runs
    half8_t wf[5][RG_KS];
    auto load_w = [&](int pass) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const half_t* wr = g.w + (int64_t)(pass * 320 + wave * 80 + i * 16 + (lane & 15)) * g.ldw + (lane >> 4) * 8;
#pragma unroll
            for (int ks = 0; ks < RG_KS; ++ks) wf[i][ks] = fz_ld_h8(wr + ks * 32);
        }
    };
    load_w(0);

    if constexpr (LN) {
        half_t* xs = XS + wave * RG_SLOT;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tt = h * 16 + (lane >> 2), j = lane & 3;
            half8_t v[RG_KS];
#pragma unroll
            for (int i = 0; i < RG_KS; ++i) v[i] = fz_ld_h8(xs + rg_off(i, tt, j));
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int i = 0; i < RG_KS; ++i)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    s0 += (float)v[i][e];
                    s1 += (float)v[i][e + 1];
                }
            const float mean = fz_sum4(s0 + s1) * (1.0f / RG_K);
            float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
            for (int i = 0; i < RG_KS; ++i)
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const float d0 = (float)v[i][e] - mean, d1 = (float)v[i][e + 1] - mean;
                    q0 += d0 * d0;
                    q1 += d1 * d1;
                }
            const float rstd = 1.0f / sqrtf(fz_sum4(q0 + q1) * (1.0f / RG_K) + g.eps);
#pragma unroll
            for (int i = 0; i < RG_KS; ++i) {
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)v[i][e] - mean) * rstd * (float)gm[i][e] + (float)bt[i][e]);
                fz_st_h8(xs + rg_off(i, tt, j), o);
            }
        }
    }
    __syncthreads();  // every wave's rows (normalised) are in LDS

    // ---- 3. passes over 320-column slices of W; per sub-tile: 2 x (10 B-fragment reads, 50 MFMAs) -> staging -> stores
    const int npass = g.n_out / 320;
    const int tok = lane & 15, kq = lane >> 4;
    const int boff = tok * 32 + ((kq ^ rg_g(tok >> 2)) * 8);  // B fragment of k step ks, half h: slot + (ks * 2 + h) * 512 + boff
    const bool vec_res = g.res != nullptr;
    for (int pass = 0; pass < npass; ++pass) {
        const bool tpass = VT && pass * 320 >= g.vt_split;  // this slice leaves transposed (V^T)
        half4_t bv[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (g.bias != nullptr) {
                bv[i] = *reinterpret_cast<const half4_t*>(g.bias + pass * 320 + wave * 80 + i * 16 + kq * 4);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[i][e] = (half_t)0.0f;
            }
        }
        for (int s = 0; s < 4; ++s) {
            const half_t* xs = XS + s * RG_SLOT;
            f32x4 acc[2][5];
            // all 20 B fragments of the sub-tile are requested up front (80 VGPRs): the MFMAs of k step ks wait for ITS fragment only,
            // instead of a read -> wait -> 5 MFMAs chain per k step with the LDS latency exposed every time
            half8_t bf[2][RG_KS];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < RG_KS; ++ks) bf[h][ks] = fz_ld_h8(xs + (ks * 2 + h) * 512 + boff);
            FZ_SCHED_FENCE();  // (left alone the scheduler sinks every read back in front of its first use to save registers)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[h][i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int ks = 0; ks < RG_KS; ++ks)
#pragma unroll
                    for (int i = 0; i < 5; ++i) acc[h][i] = fz_mfma_16x16x32_f16(wf[i][ks], bf[h][ks], acc[h][i]);
            }
            // the weights of the next slice start travelling as soon as the last MFMA of this one has issued: they ride under the epilogue
            if (s == 3 && pass + 1 < npass) load_w(pass + 1);
            // (barriers that do NOT drain vmcnt: __syncthreads() would wait for the weights just requested and for the stores in flight)
            fz_barrier_nodrain();  // the previous sub-tile's staging tile has been read by everyone
            if (!tpass) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        half4_t o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (half_t)(acc[h][i][e] + (float)bv[i][e]);
                        *reinterpret_cast<half4_t*>(OS + (h * 16 + tok) * RG_OSTR + wave * 80 + i * 16 + kq * 4) = o;
                    }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 5; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            OS[(wave * 80 + i * 16 + kq * 4 + e) * RG_TSTR + h * 16 + tok] = (half_t)(acc[h][i][e] + (float)bv[i][e]);
            }
            fz_barrier_nodrain();  // (its s_waitcnt lgkmcnt(0): this wave's staging writes have completed)
            const int64_t srow = row0 + s * RG_SUB;
            if (!tpass) {
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    const int id = tid + 256 * it, tk = id / 40, ch = (id - tk * 40) * 8;
                    const int64_t row = srow + tk;
                    if (row >= g.rows) continue;
                    half8_t v = fz_ld_h8(OS + tk * RG_OSTR + ch);
                    const int col = pass * 320 + ch;
                    if (vec_res) {
                        float f[8];
                        const half8_t r = fz_ld_h8(g.res + row * g.ldres + col);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = (float)v[e] + (float)r[e];
                        if (g.res2 != nullptr) {
                            const half8_t r2 = fz_ld_h8(g.res2 + row * g.ldres + col);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += (float)r2[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (half_t)f[e];
                    }
                    fz_st_h8(g.y + row * g.ldy + col, v);
                }
            } else {
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    const int id = tid + 256 * it, ch = id >> 2, tc = (id & 3) * 8;
                    const int64_t row = srow + tc;
                    if (row >= g.rows) continue;  // rows % 8 == 0 (host): a piece is inside the rows or outside
                    const int64_t fr = row / g.vt_rows, tk = row - fr * g.vt_rows;
                    fz_st_h8(g.yt + fr * g.yt_bs + (int64_t)(pass * 320 + ch - g.vt_split) * g.ldyt + tk, fz_ld_h8(OS + ch * RG_TSTR + tc));
                }
            }
        }
    }
}

static int rg_run(const RgArgs& g, void* stream) {
    const int64_t nblk = (g.rows + RG_ROWS - 1) / RG_ROWS;
    if (nblk <= 0 || nblk >= (1ll << 31)) return FZ_ERR_BAD_ARG;
    dim3 grid((unsigned)nblk), block(256);
    const bool ln = g.gamma != nullptr, vt = g.yt != nullptr;
    if (ln && vt) {
        FZ_LAUNCH((rowgemm320_kernel<true, true>), grid, block, RG_LDS_BYTES, stream, g);
    } else if (ln) {
        FZ_LAUNCH((rowgemm320_kernel<true, false>), grid, block, RG_LDS_BYTES, stream, g);
    } else if (vt) {
        FZ_LAUNCH((rowgemm320_kernel<false, true>), grid, block, RG_LDS_BYTES, stream, g);
    } else {
        FZ_LAUNCH((rowgemm320_kernel<false, false>), grid, block, RG_LDS_BYTES, stream, g);
    }
    return fz_last_launch_status();
}

extern "C" int fz_ln_gemm_ok(int64_t rows, int in_features, int out_features) {
    return rows > 0 && rows < (1ll << 40) && in_features == RG_K && out_features > 0 && out_features % 320 == 0 && out_features <= RG_MAX_N;
}

// Where is this launch the FASTER form on MI355X?  Nowhere by a margin worth a second code path (profiles/r05_rowgemm_negative_result.txt):
// kernel-level A/B against fz_layernorm + fz_gemm / fz_gemm_qkvt on the 64x64-level shapes reads 1.22x / 1.14x at 8 frames with the LayerNorm
// fused, 0.95-1.05x at 16 frames, 0.76-0.99x without a LayerNorm to fuse -- every 128-row workgroup re-ingests its 205 KB weight slice
// (4.5 us per CU) behind a 4.1 us launch floor, and with the weights in registers (one wave per SIMD) nothing hides a latency.  The model
// therefore does not call it; the entry points stay (tested on the emulator and on MI355X) for callers with >= 512 rows per workgroup slice.
extern "C" int fz_ln_gemm_preferred(int64_t rows, int in_features, int out_features) {
    (void)rows; (void)in_features; (void)out_features;
    return 0;
}

static int rg_common(RgArgs& g, const FzGemmDesc* d, const void* x, const void* gamma, const void* beta, float eps, const void* w) {
    if (!d || !x || !w || d->rows <= 0) return FZ_ERR_BAD_ARG;
    if (!fz_ln_gemm_ok(d->rows, d->in_features, d->out_features)) return FZ_ERR_UNSUPPORTED;
    if (d->epilogue != FZ_GEMM_PLAIN || d->transpose_out || d->batch > 1 || d->w_batch_stride) return FZ_ERR_UNSUPPORTED;
    if ((gamma == nullptr) != (beta == nullptr)) return FZ_ERR_BAD_ARG;
    if (d->ldx < d->in_features || d->ldw < d->in_features || (d->ldx % 8) || (d->ldw % 8)) return FZ_ERR_BAD_ARG;
    g.x = (const half_t*)x;
    g.w = (const half_t*)w;
    g.gamma = (const half_t*)gamma;
    g.beta = (const half_t*)beta;
    g.eps = eps;
    g.ldx = d->ldx;
    g.ldw = d->ldw;
    g.rows = d->rows;
    g.n_out = d->out_features;
    g.vt_split = d->out_features;
    g.vt_rows = 1;
    return FZ_OK;
}

extern "C" int fz_ln_gemm(const FzGemmDesc* d, const void* x, const void* gamma, const void* beta, float eps, const void* w, const void* bias,
                          const void* res, const void* res2, void* y, void* stream) {
    RgArgs g = {};
    const int rc = rg_common(g, d, x, gamma, beta, eps, w);
    if (rc != FZ_OK) return rc;
    if (!y || d->ldy < d->out_features || (d->ldy % 8)) return FZ_ERR_BAD_ARG;
    g.bias = (const half_t*)bias;
    g.res = (const half_t*)res;
    g.res2 = res ? (const half_t*)res2 : nullptr;
    if (!res && res2) g.res = (const half_t*)res2;
    g.ldres = d->ldres ? d->ldres : d->ldy;
    if ((res || res2) && (g.ldres % 8)) return FZ_ERR_BAD_ARG;
    g.y = (half_t*)y;
    g.ldy = d->ldy;
    return rg_run(g, stream);
}

extern "C" int fz_ln_gemm_qkvt(const FzGemmDesc* d, const void* x, const void* gamma, const void* beta, float eps, const void* w, void* y,
                               void* yt, int split_col, int64_t rows_per_frame, int64_t yt_frame_stride, int64_t ldyt, void* stream) {
    RgArgs g = {};
    const int rc = rg_common(g, d, x, gamma, beta, eps, w);
    if (rc != FZ_OK) return rc;
    if (!y || !yt || split_col <= 0 || split_col >= d->out_features || split_col % 320 || rows_per_frame <= 0 || rows_per_frame % 32 ||
        d->rows % rows_per_frame || rows_per_frame >= (1ll << 31) || ldyt < rows_per_frame || (ldyt % 8) || (yt_frame_stride % 8) ||
        d->ldy < split_col || (d->ldy % 8))
        return FZ_ERR_BAD_ARG;
    g.y = (half_t*)y;
    g.ldy = d->ldy;
    g.yt = (half_t*)yt;
    g.yt_bs = yt_frame_stride;
    g.ldyt = ldyt;
    g.vt_split = split_col;
    g.vt_rows = (int)rows_per_frame;
    return rg_run(g, stream);
}
