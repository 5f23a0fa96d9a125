// norms.hip -- GroupNorm(+SiLU) over token-major activations and LayerNorm, for gfx950.
//
// GroupNorm replaces torch.nn.GroupNorm + F.silu of the reference: on 5-D tensors [b,c,f,h,w] inside
// ResnetBlockPseudo3D (resnet.py:338-339, :369, :384) and conv_norm_out (unet_3d_condition.py:439-440) the
// statistics of one (batch, group) span ALL frames ("span" = F); inside SpatioTemporalTransformerModel
// (attention.py:110) the input is 4-D [(b f),c,h,w] so they are per frame (span = 1).
// Layout is x[n][token][C] fp16.  Where a (stat set, group) fits the registers of one workgroup: ONE launch (gn_fused_kernel below).
// Otherwise three small HBM-bound kernels:
//   gn_stats    : per (frame, token chunk) Welford partials (n, mean, M2) for each group      (reads x once)
//   gn_finalize : deterministic Chan merge of the partials of a span -> (mean, rstd) per (span, group)
//   gn_apply    : y = silu?((x - mean) * rstd * gamma + beta)                          (reads x once, writes y)
// (Both ways of saving the finalize launch were measured and lost: the last-arriving statistics workgroup doing the merge --
// cross-XCD hand-off latency, 2.4x slower -- and every apply workgroup doing it redundantly -- 1.7x slower even with the
// group-major partial layout.)
// Split stats/apply is also the form frame-sharded multi-GPU needs: the finalize step is where the per-rank
// partials are all-reduced.
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"
#include <stdlib.h>

// Tokens per block ("chunk").  A block has V = C/8 vector lanes x R rows of threads; the chunk is sized so that a thread
// owns <= 16 rows (all of them in flight as loads at once) and so that the small pyramid levels still launch >= 32 blocks
// per frame -- at 16x16 tokens x 1280 channels a fixed 64-token chunk left 32 blocks to fill 256 CUs.
static int gn_tb(int tokens, int channels) {
    const int V = channels / 8;
    const int R = V >= 256 ? 1 : 256 / (V > 0 ? V : 1);
    int tb = 64;
    while (tb > 8 && tb > 16 * R) tb >>= 1;
    while (tb > 8 && tokens / tb < 32) tb >>= 1;
    return tb;
}

extern "C" int fz_groupnorm_chunks(int tokens, int channels) {
    const int tb = gn_tb(tokens, channels);
    return (tokens + tb - 1) / tb;
}

struct GnArgs {
    const half_t* x;
    const half_t* x2;  // channels [C1, C) come from a SECOND tensor [n][tokens][C - C1] (the skip connection of an up block:
    int C1;            // GroupNorm of torch.cat([x, skip], channel) without the concatenated copy); x2 == nullptr: C1 = C
    half_t* y;
    const half_t *gamma, *beta;
    float* partial;  // [n_frames][G][chunks][3]: the partials of one (frame, group) are contiguous
    float* stats;    // [n_frames/span][G][2]
    int n_frames, span, tokens, C, G, chunks, V, R;
    int tb;          // tokens per chunk
    int fin_span;    // frames merged per stat set by gn_finalize (= span unless the partials were gathered from other ranks)
    int pchunks;     // partial records per (frame, group) that gn_finalize reads (= chunks unless a producer's epilogue wrote them)
    float eps;
    int silu;
};

// ROWS > 0: the thread's (<= ROWS) rows of the chunk are loaded ONCE, all loads in flight together, and kept in registers
// for the second sweep; ROWS == 0: rows are re-read (second sweep hits L2) -- wide channel counts that do not fit.
// (register-resident forms: <= 512 threads, so that 32 rows x 4 VGPRs fit without scratch -- at the default 1024-thread bound the
// compiler has 128 VGPRs and spilled 56 of them)
template <int ROWS>
FZ_KERNEL void __launch_bounds__(ROWS > 0 ? 512 : 1024) gn_stats_kernel(GnArgs a) {
    // two sweeps over the chunk: exact chunk mean first, then sum (x-mean)^2, so the partial variance never suffers the
    // E[x^2]-E[x]^2 cancellation
    FZ_DYN_SMEM(raw);
    float* red = reinterpret_cast<float*>(raw);  // [R][C]
    float* gmean = red + a.R * a.C;              // [G]
    const int tid = threadIdx.x, v = tid % a.V, r = tid / a.V;
    const int chunk = blockIdx.x, n = blockIdx.y;
    const int t0 = chunk * a.tb;
    const int t1 = min(t0 + a.tb, a.tokens);
    const int cg = a.C / a.G;
    const float cnt = (float)((t1 - t0) * cg);
    // (a 16-byte chunk never straddles the two sources: C1 % 8 == 0)
    const bool second = v * 8 >= a.C1;
    const int64_t rs = second ? a.C - a.C1 : a.C1;  // row stride of this thread's source
    const half_t* base = (second ? a.x2 + ((int64_t)n * a.tokens) * rs + (v * 8 - a.C1) : a.x + ((int64_t)n * a.tokens) * rs + v * 8);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.0f;
    half8_t xr[ROWS > 0 ? ROWS : 1];
    if (ROWS > 0) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {  // unconditional, clamped: every load is issued before the first use
            int t = t0 + r + i * a.R;
            t = t < t1 ? t : t1 - 1;
            xr[i] = fz_ld_h8(base + (int64_t)t * rs);
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            if (t0 + r + i * a.R < t1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)xr[i][e];
            }
        }
    } else {
        for (int t = t0 + r; t < t1; t += a.R) {
            const half8_t xv = fz_ld_h8(base + (int64_t)t * rs);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)xv[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[r * a.C + v * 8 + e] = s[e];
    __syncthreads();
    if (tid < a.G) {
        float sum = 0.0f;
        for (int rr = 0; rr < a.R; ++rr)
            for (int c = tid * cg; c < (tid + 1) * cg; ++c) sum += red[rr * a.C + c];
        gmean[tid] = sum / cnt;
    }
    __syncthreads();
    float mu[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mu[e] = gmean[(v * 8 + e) / cg];
        s[e] = 0.0f;
    }
    if (ROWS > 0) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            if (t0 + r + i * a.R < t1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dlt = (float)xr[i][e] - mu[e];
                    s[e] += dlt * dlt;
                }
            }
        }
    } else {
        for (int t = t0 + r; t < t1; t += a.R) {
            const half8_t xv = fz_ld_h8(base + (int64_t)t * rs);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dlt = (float)xv[e] - mu[e];
                s[e] += dlt * dlt;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red[r * a.C + v * 8 + e] = s[e];
    __syncthreads();
    if (tid < a.G) {
        float m2 = 0.0f;
        for (int rr = 0; rr < a.R; ++rr)
            for (int c = tid * cg; c < (tid + 1) * cg; ++c) m2 += red[rr * a.C + c];
        float* out = a.partial + (((int64_t)n * a.G + tid) * a.chunks + chunk) * 3;
        out[0] = cnt;
        out[1] = gmean[tid];
        out[2] = m2;
    }
}

FZ_DEVICE void chan_merge(float& cnt, float& mean, float& m2, float nb, float mb, float m2b) {
    const float tot = cnt + nb;
    if (tot > 0.0f) {
        const float delta = mb - mean;
        mean += delta * (nb / tot);
        m2 += m2b + delta * delta * (cnt * nb / tot);
        cnt = tot;
    }
}

// Chan-merge of the records [e0, e1) of (span sp, group g) by one full wave: lane l takes e0 + l, e0 + l + 64, ... in that order (eight, then
// four records per lane in flight before the first merge: the merge chain is serial, the loads must not be), then the fixed butterfly.
FZ_DEVICE void gn_merge_range(const GnArgs& a, int sp, int g, int lane, int e0, int e1, float& cnt, float& mean, float& m2) {
    cnt = 0.0f; mean = 0.0f; m2 = 0.0f;
    // partial records of (frame f, group g) are contiguous: consecutive lanes read consecutive 12-byte records
    auto rec = [&](int e) -> const float* {
        const int f = e / a.pchunks, c = e - f * a.pchunks;
        return a.partial + (((int64_t)(sp * a.fin_span + f) * a.G + g) * a.pchunks + c) * 3;
    };
    int e = e0 + lane;
    for (; e + 448 < e1; e += 512) {
        float r[8][3];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* pp = rec(e + 64 * u);
            r[u][0] = pp[0]; r[u][1] = pp[1]; r[u][2] = pp[2];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) chan_merge(cnt, mean, m2, r[u][0], r[u][1], r[u][2]);
    }
    for (; e + 192 < e1; e += 256) {
        float r[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* pp = rec(e + 64 * u);
            r[u][0] = pp[0]; r[u][1] = pp[1]; r[u][2] = pp[2];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) chan_merge(cnt, mean, m2, r[u][0], r[u][1], r[u][2]);
    }
    for (; e < e1; e += 64) {
        const float* pp = rec(e);
        chan_merge(cnt, mean, m2, pp[0], pp[1], pp[2]);
    }
#pragma unroll
    for (int msk = 1; msk < 64; msk <<= 1) {
        const float nb = fz_shfl_xor(cnt, msk), mb = fz_shfl_xor(mean, msk), m2b = fz_shfl_xor(m2, msk);
        // both partners must compute the same merged value: merge (lower lane, upper lane) in that order
        float c0 = cnt, me0 = mean, q0 = m2, c1 = nb, me1 = mb, q1 = m2b;
        if (lane & msk) { c0 = nb; me0 = mb; q0 = m2b; c1 = cnt; me1 = mean; q1 = m2; }
        chan_merge(c0, me0, q0, c1, me1, q1);
        cnt = c0; mean = me0; m2 = q0;
    }
}

// (span sp, group g) -> (mean, rstd) by ONE full wave over all of the span's partials (deterministic order -> bitwise reproducible
// statistics, whoever runs it)
FZ_DEVICE void gn_finalize_group(const GnArgs& a, int sp, int g, int lane, float* mean_out, float* rstd_out) {
    float cnt, mean, m2;
    gn_merge_range(a, sp, g, lane, 0, a.fin_span * a.pchunks, cnt, mean, m2);
    const float var = m2 / cnt;  // biased, as torch.nn.GroupNorm
    *mean_out = mean;
    *rstd_out = 1.0f / sqrtf(var + a.eps);
}

FZ_KERNEL void __launch_bounds__(64) gn_finalize_kernel(GnArgs a) {  // one wave per (span, group)
    const int sp = (int)blockIdx.x / a.G, g = (int)blockIdx.x % a.G;
    float mean, rstd;
    gn_finalize_group(a, sp, g, (int)threadIdx.x, &mean, &rstd);
    if (threadIdx.x == 0) {
        a.stats[(sp * a.G + g) * 2 + 0] = mean;
        a.stats[(sp * a.G + g) * 2 + 1] = rstd;
    }
}

// The same for spans with thousands of records (the partials of fz_lora_pair_gn: one record per (frame, group, 128 / clip_len tokens)): four
// waves per (span, group), each over a contiguous quarter of the records, merged in wave order.
#define GN_WIDE_RECORDS 1024
FZ_KERNEL void __launch_bounds__(256) gn_finalize_wide_kernel(GnArgs a) {
    FZ_SHARED float part[4][3];
    const int sp = (int)blockIdx.x / a.G, g = (int)blockIdx.x % a.G;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int total = a.fin_span * a.pchunks, per = (total + 3) / 4;
    float cnt, mean, m2;
    gn_merge_range(a, sp, g, lane, min(wave * per, total), min((wave + 1) * per, total), cnt, mean, m2);
    if (lane == 0) { part[wave][0] = cnt; part[wave][1] = mean; part[wave][2] = m2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt = part[0][0]; mean = part[0][1]; m2 = part[0][2];
        for (int w = 1; w < 4; ++w) chan_merge(cnt, mean, m2, part[w][0], part[w][1], part[w][2]);
        a.stats[(sp * a.G + g) * 2 + 0] = mean;
        a.stats[(sp * a.G + g) * 2 + 1] = 1.0f / sqrtf(m2 / cnt + a.eps);
    }
}

FZ_KERNEL void gn_apply_kernel(GnArgs a) {
    const int tid = threadIdx.x, v = tid % a.V, r = tid / a.V;
    const int chunk = blockIdx.x, n = blockIdx.y;
    const int t0 = chunk * a.tb;
    const int t1 = min(t0 + a.tb, a.tokens);
    const int cg = a.C / a.G;
    const int sp = n / a.span;
    float sc[8], sh[8];
    const half8_t gv = fz_ld_h8(a.gamma + v * 8), bv = fz_ld_h8(a.beta + v * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (v * 8 + e) / cg;
        const float mean = a.stats[(sp * a.G + g) * 2 + 0], rstd = a.stats[(sp * a.G + g) * 2 + 1];
        sc[e] = rstd * (float)gv[e];
        sh[e] = (float)bv[e] - mean * sc[e];
    }
    const bool second = v * 8 >= a.C1;
    const int64_t rs = second ? a.C - a.C1 : a.C1;
    const half_t* xb = (second ? a.x2 + ((int64_t)n * a.tokens) * rs + (v * 8 - a.C1) : a.x + ((int64_t)n * a.tokens) * rs + v * 8);
    half_t* yb = a.y + ((int64_t)n * a.tokens) * a.C + v * 8;
#pragma unroll 4
    for (int t = t0 + r; t < t1; t += a.R) {
        const half8_t xv = fz_ld_h8(xb + (int64_t)t * rs);
        half8_t yv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = (float)xv[e] * sc[e] + sh[e];
            if (a.silu) f = f / (1.0f + __builtin_expf(-f));
            yv[e] = (half_t)f;
        }
        fz_st_h8(yb + (int64_t)t * a.C, yv);
    }
}

// ----------------------------------------------------------------------------------------------------------
// One-launch form for the small launches.  One workgroup per (stat set, group): the group's span x tokens x C/G elements
// (<= GN_FUSED_MAX_PAIRS channel pairs) are loaded ONCE into registers -- NE channel pairs per thread, every load in flight before the
// first use --, then exact mean, exact sum of squared deviations, scale / shift per channel through LDS, normalise (+SiLU), store.
// Below ~1 MB per tensor the three-kernel form is three launch latencies (stats 5-9 us + finalize 4.7 us + apply 5-10 us); this is one.
// Reductions are in a fixed order (thread-serial, xor butterfly, wave results summed in wave order): bitwise reproducible.
// ----------------------------------------------------------------------------------------------------------
#define GN_FUSED_THREADS 1024
#define GN_FUSED_MAX_NE 20
#define GN_FUSED_MAX_PAIRS (GN_FUSED_THREADS * GN_FUSED_MAX_NE)

FZ_DEVICE float gn_block_sum(float v, float* red, int tid) {  // red: GN_FUSED_THREADS / 64 floats of LDS
#pragma unroll
    for (int msk = 1; msk < 64; msk <<= 1) v += fz_shfl_xor(v, msk);
    __syncthreads();  // the previous user of `red` is done
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < GN_FUSED_THREADS / 64; ++w) s += red[w];
    return s;
}

template <int NE>
FZ_KERNEL void __launch_bounds__(GN_FUSED_THREADS) gn_fused_kernel(GnArgs a) {
    FZ_SHARED float red[GN_FUSED_THREADS / 64];
    FZ_SHARED float gsc[128], gsh[128];  // per channel of the group: rstd * gamma, beta - mean * rstd * gamma  (C/G <= 128)
    const int tid = threadIdx.x;
    const int sp = (int)blockIdx.x / a.G, g = (int)blockIdx.x % a.G;
    const int cg = a.C / a.G, P = cg >> 1;       // channel pairs per row of the group
    const int total = a.span * a.tokens * P;     // channel pairs of the (set, group)
    const int c0 = g * cg;
    const int64_t row0 = (int64_t)sp * a.span * a.tokens;  // the set's frames are consecutive: its rows too
    // two wave-uniform bases (x | the skip tensor of a lazy concatenation) + 32-bit element offsets; a group may straddle the seam
    const uint32_t rs1 = (uint32_t)a.C1, rs2 = (uint32_t)(a.C - a.C1);
    const half_t* const src1 = a.x + row0 * a.C1;
    const half_t* const src2 = a.x2 != nullptr ? a.x2 + row0 * (a.C - a.C1) : src1;
    half_t* const dst = a.y + row0 * a.C + c0;
    // thread t owns pairs e = t + 1024 i: (row, pair) advance incrementally
    const int drow = GN_FUSED_THREADS / P, dpr = GN_FUSED_THREADS - drow * P;
    const int row = tid / P, pr = tid - row * P;
    const int last_row = a.span * a.tokens - 1;
    half2_t v[NE];
    {
        int r = row, q = pr;
#pragma unroll
        for (int i = 0; i < NE; ++i) {  // unconditional, clamped: all NE loads are in flight before the first use; issued in order, so
            const int rr = r < last_row ? r : last_row;  // that one address register serves them all (hoisted, the 2 x NE address
            const uint32_t ch = (uint32_t)(c0 + 2 * q);                                          // registers spilled)
            const half_t* p = ch >= rs1 ? src2 + ((uint32_t)rr * rs2 + (ch - rs1)) : src1 + ((uint32_t)rr * rs1 + ch);
            v[i] = *reinterpret_cast<const half2_t*>(p);
            FZ_SCHED_FENCE();
            r += drow;
            q += dpr;
            if (q >= P) { q -= P; ++r; }
        }
    }
    const float cnt = (float)total * 2.0f;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i)
        if (tid + GN_FUSED_THREADS * i < total) s += (float)v[i][0] + (float)v[i][1];
    const float mean = gn_block_sum(s, red, tid) / cnt;
    // (the packed halves stay THE copy of the data: without the pins the fp32 conversions of the first sweep are kept alive for the
    // other two -- 2 x NE more registers, spilled)
#pragma unroll
    for (int i = 0; i < NE; ++i) FZ_PIN_V(v[i]);
    s = 0.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i)
        if (tid + GN_FUSED_THREADS * i < total) {
            const float d0 = (float)v[i][0] - mean, d1 = (float)v[i][1] - mean;
            s += d0 * d0 + d1 * d1;
        }
    const float var = gn_block_sum(s, red, tid) / cnt;  // biased, as torch.nn.GroupNorm
    const float rstd = 1.0f / sqrtf(var + a.eps);
#pragma unroll
    for (int i = 0; i < NE; ++i) FZ_PIN_V(v[i]);
    if (tid < cg) {
        const float sc = rstd * (float)a.gamma[c0 + tid];
        gsc[tid] = sc;
        gsh[tid] = (float)a.beta[c0 + tid] - mean * sc;
    }
    __syncthreads();
    {
        int r = row, q = pr;
        FZ_PIN_V(r);  // a fresh chain: otherwise the (row, pair) of every element is kept from the load loop -- 2 x NE registers
        FZ_PIN_V(q);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            if (tid + GN_FUSED_THREADS * i < total) {
                float f0 = (float)v[i][0] * gsc[2 * q] + gsh[2 * q];
                float f1 = (float)v[i][1] * gsc[2 * q + 1] + gsh[2 * q + 1];
                if (a.silu) {
                    f0 = f0 / (1.0f + __builtin_expf(-f0));
                    f1 = f1 / (1.0f + __builtin_expf(-f1));
                }
                half2_t o;
                o[0] = (half_t)f0;
                o[1] = (half_t)f1;
                *reinterpret_cast<half2_t*>(dst + ((uint32_t)r * (uint32_t)a.C + 2u * (uint32_t)q)) = o;
            }
            FZ_SCHED_FENCE();
            r += drow;
            q += dpr;
            if (q >= P) { q -= P; ++r; }
        }
    }
}

#ifdef FZ_GN_TRIALS  // build_tmp variant (scripts/build_variant.sh ... -DFZ_GN_TRIALS): a second thread layout of the one-launch form
// MEASURED (profiles/r03_gn_one_launch_layout2_trial.txt): 10-20 % faster than the shipped layout on the same shapes (7.6 vs 8.3 us, 11.3 vs
// 13.2 us), same crossover against the three kernels (19 vs 16 us at 40 rows per thread): the one-launch form is bound by what ONE
// workgroup can stream with 4-byte strided accesses (~10 GB/s), not by its instruction count.  Kept as a trial, not shipped.
// Thread t owns ONE channel pair (q = t % P) of rows r0 + i R (r0 = t / P, R = 1024 / P rows per sweep): no per-element index
// arithmetic (one 32-bit add per load / store), the pair's source tensor, scale and shift live in registers.
template <int NE>
FZ_KERNEL void __launch_bounds__(GN_FUSED_THREADS) gn_fused2_kernel(GnArgs a) {
    FZ_SHARED float red[GN_FUSED_THREADS / 64];
    const int tid = threadIdx.x;
    const int sp = (int)blockIdx.x / a.G, g = (int)blockIdx.x % a.G;
    const int cg = a.C / a.G, P = cg >> 1;
    const int R = GN_FUSED_THREADS / P;          // rows per sweep
    const int rows = a.span * a.tokens;
    const int r0 = tid / P, q = tid - r0 * P;
    const bool active = r0 < R;                  // threads beyond R * P idle (they still meet the barriers)
    const int ch = g * cg + 2 * q;               // this thread's channel pair, for all its rows
    const int64_t row0 = (int64_t)sp * rows;
    const bool second = ch >= a.C1;
    const uint32_t rs = (uint32_t)(second ? a.C - a.C1 : a.C1);
    const half_t* const src = second ? a.x2 + row0 * (a.C - a.C1) + (ch - a.C1) : a.x + row0 * a.C1 + ch;
    half_t* const dst = a.y + row0 * a.C + ch;
    const uint32_t step_in = (uint32_t)R * rs, step_out = (uint32_t)R * (uint32_t)a.C;
    const int r_first = active ? r0 : 0;
    half2_t v[NE];
    {
        uint32_t off = (uint32_t)r_first * rs;
        const uint32_t off_max = (uint32_t)(rows - 1) * rs;
#pragma unroll
        for (int i = 0; i < NE; ++i) {  // all loads in flight before the first use, issued in order (one address register)
            v[i] = *reinterpret_cast<const half2_t*>(src + (off < off_max ? off : off_max));
            FZ_SCHED_FENCE();
            off += step_in;
        }
    }
    const int r_lim = active ? rows - r0 : 0;  // row i of this thread is inside the set iff i * R < r_lim
    const float cnt = (float)rows * (float)cg;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i)
        if (i * R < r_lim) s += (float)v[i][0] + (float)v[i][1];
    const float mean = gn_block_sum(s, red, tid) / cnt;
#pragma unroll
    for (int i = 0; i < NE; ++i) FZ_PIN_V(v[i]);
    s = 0.0f;
#pragma unroll
    for (int i = 0; i < NE; ++i)
        if (i * R < r_lim) {
            const float d0 = (float)v[i][0] - mean, d1 = (float)v[i][1] - mean;
            s += d0 * d0 + d1 * d1;
        }
    const float var = gn_block_sum(s, red, tid) / cnt;
    const float rstd = 1.0f / sqrtf(var + a.eps);
#pragma unroll
    for (int i = 0; i < NE; ++i) FZ_PIN_V(v[i]);
    const half2_t gm = *reinterpret_cast<const half2_t*>(a.gamma + ch), bt = *reinterpret_cast<const half2_t*>(a.beta + ch);
    const float sc0 = rstd * (float)gm[0], sc1 = rstd * (float)gm[1];
    const float sh0 = (float)bt[0] - mean * sc0, sh1 = (float)bt[1] - mean * sc1;
    uint32_t off = (uint32_t)r_first * (uint32_t)a.C;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        if (i * R < r_lim) {
            float f0 = (float)v[i][0] * sc0 + sh0, f1 = (float)v[i][1] * sc1 + sh1;
            if (a.silu) {
                f0 = f0 / (1.0f + __builtin_expf(-f0));
                f1 = f1 / (1.0f + __builtin_expf(-f1));
            }
            half2_t o;
            o[0] = (half_t)f0;
            o[1] = (half_t)f1;
            *reinterpret_cast<half2_t*>(dst + off) = o;
        }
        FZ_SCHED_FENCE();
        off += step_out;
    }
}

template <int NE>
static void gn_launch_fused2(const GnArgs& a, int ne, dim3 grid, void* stream) {
    if constexpr (NE > 4) {
        if (ne < NE) return gn_launch_fused2<NE - 4>(a, ne, grid, stream);
    }
    FZ_LAUNCH(gn_fused2_kernel<NE>, grid, dim3(GN_FUSED_THREADS), 0, stream, a);
}
#endif

// The one-launch form where it measured faster than the three kernels on MI355X (profiles/r03_gn_one_launch_vs_three.txt, GPU time from
// the kernel trace): a workgroup's serial time grows with the pairs per thread (~0.55 us each: 2 + 5.5 us at 10, 25 us at 40, where the
// three-kernel form takes 17-28 us whatever the shape), and many workgroups of it are VALU-bound where the three kernels are
// bandwidth-bound.  Wins 1.3-2.7x: <= 20 pairs per thread and <= 2560 pairs-per-thread x workgroups -- the 8^2 level (both widths),
// the per-frame transformer norms of the 16^2 level and of the 32^2 level at 8 frames.  false: the caller runs the three-kernel form.
static bool gn_try_fused(const GnArgs& a, void* stream) {
    const int cg = a.C / a.G;
    if ((cg & 1) || cg > 128 || (a.C1 & 1)) return false;
    const int64_t pairs = (int64_t)a.span * a.tokens * (cg / 2);
#ifndef FZ_GN_TRIALS
    if (pairs > GN_FUSED_MAX_PAIRS) return false;
#endif
    if ((int64_t)a.span * a.tokens * a.C >= (1ll << 31)) return false;  // 32-bit element offsets
    const int ne = (int)((pairs + GN_FUSED_THREADS - 1) / GN_FUSED_THREADS);
    const int wgs = (a.n_frames / a.span) * a.G;
#ifdef FZ_GN_TRIALS  // the second layout on EVERY group that fits (no rule): scripts/gn_ab.py against the three-kernel form
    {
        const int P = cg / 2, R = GN_FUSED_THREADS / P, rows = a.span * a.tokens;
        const int ne2 = (rows + R - 1) / R;  // rows per thread
        if (ne2 > 84) return false;
        gn_launch_fused2<84>(a, ne2 < 4 ? 4 : (ne2 + 3) / 4 * 4, dim3(wgs), stream);
        return true;
    }
#endif
    if ((int64_t)ne * wgs > 2560) return false;
    const dim3 grid(wgs), block(GN_FUSED_THREADS);
    // (a masked iteration costs what a live one does: tight buckets)
    if (ne <= 2) {
        FZ_LAUNCH(gn_fused_kernel<2>, grid, block, 0, stream, a);
    } else if (ne <= 4) {
        FZ_LAUNCH(gn_fused_kernel<4>, grid, block, 0, stream, a);
    } else if (ne <= 5) {
        FZ_LAUNCH(gn_fused_kernel<5>, grid, block, 0, stream, a);
    } else if (ne <= 8) {
        FZ_LAUNCH(gn_fused_kernel<8>, grid, block, 0, stream, a);
    } else if (ne <= 10) {
        FZ_LAUNCH(gn_fused_kernel<10>, grid, block, 0, stream, a);
    } else if (ne <= 16) {
        FZ_LAUNCH(gn_fused_kernel<16>, grid, block, 0, stream, a);
    } else {
        FZ_LAUNCH(gn_fused_kernel<GN_FUSED_MAX_NE>, grid, block, 0, stream, a);
    }
    return true;
}

static void gn_launch_stats(const GnArgs& a, dim3 grid, dim3 block, size_t smem, void* stream) {
    const int rows = (a.tb + a.R - 1) / a.R;  // rows of a chunk per thread
    if (rows > 32 || block.x > 512) {
        FZ_LAUNCH(gn_stats_kernel<0>, grid, block, smem, stream, a);
    } else if (rows <= 12) {
        FZ_LAUNCH(gn_stats_kernel<12>, grid, block, smem, stream, a);
    } else if (rows <= 24) {
        FZ_LAUNCH(gn_stats_kernel<24>, grid, block, smem, stream, a);
    } else {
        FZ_LAUNCH(gn_stats_kernel<32>, grid, block, smem, stream, a);
    }
}

static int gn_setup(GnArgs& a, int n_frames, int span, int tokens, int channels, int groups, int& threads, size_t& smem) {
    if (n_frames <= 0 || span <= 0 || n_frames % span || channels % 8 || channels % groups || groups > 64)
        return FZ_ERR_BAD_ARG;
    a.n_frames = n_frames; a.span = span; a.fin_span = span; a.tokens = tokens; a.C = channels; a.G = groups;
    a.tb = gn_tb(tokens, channels);
    a.chunks = fz_groupnorm_chunks(tokens, channels);
    a.pchunks = a.chunks;
    a.V = channels / 8;
    if (a.V > 1024) return FZ_ERR_UNSUPPORTED;
    a.R = a.V >= 256 ? 1 : 256 / a.V;
    // >= G threads are needed for the per-group reduction; threads beyond V*R would alias rows, so V*R >= 64 is required
    if (a.V * a.R < 64) return FZ_ERR_UNSUPPORTED;
    threads = a.V * a.R;
    smem = ((size_t)a.R * channels + groups) * sizeof(float);
    return FZ_OK;
}

extern "C" int fz_groupnorm(const void* x, void* y, const void* gamma, const void* beta, int n_frames, int span,
                            int tokens, int channels, int groups, float eps, int silu, float* partial, void* stream) {
    if (!x || !y || !gamma || !beta || !partial) return FZ_ERR_BAD_ARG;
    GnArgs a;
    int threads;
    size_t smem;
    const int rc = gn_setup(a, n_frames, span, tokens, channels, groups, threads, smem);
    if (rc != FZ_OK) return rc;
    a.x = (const half_t*)x; a.x2 = nullptr; a.C1 = channels;
    a.y = (half_t*)y; a.gamma = (const half_t*)gamma; a.beta = (const half_t*)beta;
    a.eps = eps; a.silu = silu;
    a.partial = partial;
    a.stats = partial + (int64_t)n_frames * a.chunks * groups * 3;
    if (gn_try_fused(a, stream)) return fz_last_launch_status();
    dim3 grid(a.chunks, n_frames), block(threads);
    gn_launch_stats(a, grid, block, smem, stream);
    const int nst = (n_frames / span) * groups;
    FZ_LAUNCH(gn_finalize_kernel, dim3(nst), dim3(64), 0, stream, a);
    FZ_LAUNCH(gn_apply_kernel, grid, block, 0, stream, a);
    return fz_last_launch_status();
}

extern "C" int fz_groupnorm_cat(const void* x1, int channels1, const void* x2, int channels2, void* y, const void* gamma,
                                const void* beta, int n_frames, int span, int tokens, int groups, float eps, int silu, float* partial,
                                void* stream) {
    if (!x1 || !x2 || !y || !gamma || !beta || !partial || channels1 <= 0 || channels2 <= 0 || (channels1 % 8) || (channels2 % 8))
        return FZ_ERR_BAD_ARG;
    GnArgs a;
    int threads;
    size_t smem;
    const int rc = gn_setup(a, n_frames, span, tokens, channels1 + channels2, groups, threads, smem);
    if (rc != FZ_OK) return rc;
    a.x = (const half_t*)x1; a.x2 = (const half_t*)x2; a.C1 = channels1;
    a.y = (half_t*)y; a.gamma = (const half_t*)gamma; a.beta = (const half_t*)beta;
    a.eps = eps; a.silu = silu;
    a.partial = partial;
    a.stats = partial + (int64_t)n_frames * a.chunks * groups * 3;
    if (gn_try_fused(a, stream)) return fz_last_launch_status();
    dim3 grid(a.chunks, n_frames), block(threads);
    gn_launch_stats(a, grid, block, smem, stream);
    FZ_LAUNCH(gn_finalize_kernel, dim3((n_frames / span) * groups), dim3(64), 0, stream, a);
    FZ_LAUNCH(gn_apply_kernel, grid, block, 0, stream, a);
    return fz_last_launch_status();
}

extern "C" int fz_groupnorm_stats(const void* x, int n_frames, int tokens, int channels, int groups, float* partial,
                                  void* stream) {
    if (!x || !partial) return FZ_ERR_BAD_ARG;
    GnArgs a;
    int threads;
    size_t smem;
    const int rc = gn_setup(a, n_frames, 1, tokens, channels, groups, threads, smem);
    if (rc != FZ_OK) return rc;
    a.x = (const half_t*)x; a.x2 = nullptr; a.C1 = channels;
    a.y = nullptr; a.gamma = nullptr; a.beta = nullptr; a.eps = 0.0f; a.silu = 0;
    a.partial = partial; a.stats = nullptr;
    gn_launch_stats(a, dim3(a.chunks, n_frames), dim3(threads), smem, stream);
    return fz_last_launch_status();
}

extern "C" int fz_groupnorm_apply(const void* x, void* y, const void* gamma, const void* beta, int n_frames, int span,
                                  int tokens, int channels, int groups, float eps, int silu, const float* partial_all,
                                  int stat_sets, int frames_per_set, float* stats, void* stream) {
    if (!x || !y || !gamma || !beta || !partial_all || !stats || stat_sets <= 0 || frames_per_set <= 0) return FZ_ERR_BAD_ARG;
    GnArgs a;
    int threads;
    size_t smem;
    const int rc = gn_setup(a, n_frames, span, tokens, channels, groups, threads, smem);
    if (rc != FZ_OK) return rc;
    if (n_frames / span != stat_sets) return FZ_ERR_BAD_ARG;
    a.x = (const half_t*)x; a.x2 = nullptr; a.C1 = channels;
    a.y = (half_t*)y; a.gamma = (const half_t*)gamma; a.beta = (const half_t*)beta;
    a.eps = eps; a.silu = silu;
    a.partial = const_cast<float*>(partial_all);
    a.stats = stats;
    a.fin_span = frames_per_set;
    FZ_LAUNCH(gn_finalize_kernel, dim3(stat_sets * groups), dim3(64), 0, stream, a);
    FZ_LAUNCH(gn_apply_kernel, dim3(a.chunks, n_frames), dim3(threads), 0, stream, a);
    return fz_last_launch_status();
}

extern "C" int fz_groupnorm_from_partials(const void* x, void* y, const void* gamma, const void* beta, int n_frames, int span, int tokens,
                                          int channels, int groups, float eps, int silu, const float* partial, int partial_chunks,
                                          float* stats, void* stream) {
    if (!x || !y || !gamma || !beta || !partial || !stats || partial_chunks <= 0) return FZ_ERR_BAD_ARG;
    GnArgs a;
    int threads;
    size_t smem;
    const int rc = gn_setup(a, n_frames, span, tokens, channels, groups, threads, smem);
    if (rc != FZ_OK) return rc;
    a.x = (const half_t*)x; a.x2 = nullptr; a.C1 = channels;
    a.y = (half_t*)y; a.gamma = (const half_t*)gamma; a.beta = (const half_t*)beta;
    a.eps = eps; a.silu = silu;
    a.partial = const_cast<float*>(partial);
    a.stats = stats;
    a.pchunks = partial_chunks;
    if ((int64_t)span * partial_chunks > GN_WIDE_RECORDS) {
        FZ_LAUNCH(gn_finalize_wide_kernel, dim3((n_frames / span) * groups), dim3(256), 0, stream, a);
    } else {
        FZ_LAUNCH(gn_finalize_kernel, dim3((n_frames / span) * groups), dim3(64), 0, stream, a);
    }
    FZ_LAUNCH(gn_apply_kernel, dim3(a.chunks, n_frames), dim3(threads), 0, stream, a);
    return fz_last_launch_status();
}

// ----------------------------------------------------------------------------------------------------------
// LayerNorm over channels (attention.py:193-233: norm1/2/3, norm_temporal): one wave per token row, the row is
// held in registers (<= 5 x 8 channels per lane), two-pass mean / variance like torch (sum, then sum of squared deviations).
// ----------------------------------------------------------------------------------------------------------
#define LN_MAXV 5

// LN_ROWS rows per wave, all of their loads issued before the first use: with one row (one 16-byte load per lane) in flight per
// wave the kernel ran at 2.3 TB/s on the 64^2 level.
#define LN_ROWS 4
template <int MAXV>
FZ_KERNEL void __launch_bounds__(256)
layernorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, const half_t* __restrict__ gamma,
                 const half_t* __restrict__ beta, int64_t rows, int C, float eps) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * LN_ROWS;
    const int V = C >> 3;
    half8_t xv[LN_ROWS][MAXV];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        int64_t row = row0 + r;
        row = row < rows ? row : rows - 1;  // clamped: every lane stays alive for the shuffles, stores are guarded
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int v = lane + 64 * i;
            xv[r][i] = v < V ? fz_ld_h8(x + row * C + v * 8) : fz_zero_h8();
        }
    }
    float mean[LN_ROWS], rstd[LN_ROWS];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)xv[r][i][e];
        mean[r] = sum;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int r = 0; r < LN_ROWS; ++r) mean[r] += fz_shfl_xor(mean[r], m);
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        mean[r] /= (float)C;
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            if (lane + 64 * i < V) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dlt = (float)xv[r][i][e] - mean[r];
                    sq += dlt * dlt;
                }
            }
        }
        rstd[r] = sq;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int r = 0; r < LN_ROWS; ++r) rstd[r] += fz_shfl_xor(rstd[r], m);
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) rstd[r] = 1.0f / sqrtf(rstd[r] / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int v = lane + 64 * i;
        if (v < V) {
            const half8_t gv = fz_ld_h8(gamma + v * 8), bv = fz_ld_h8(beta + v * 8);
#pragma unroll
            for (int r = 0; r < LN_ROWS; ++r) {
                if (row0 + r < rows) {
                    half8_t yv;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        yv[e] = (half_t)(((float)xv[r][i][e] - mean[r]) * rstd[r] * (float)gv[e] + (float)bv[e]);
                    fz_st_h8(y + (row0 + r) * C + v * 8, yv);
                }
            }
        }
    }
}

extern "C" int fz_layernorm(const void* x, void* y, const void* gamma, const void* beta, int64_t rows, int channels,
                            float eps, void* stream) {
    if (!x || !y || !gamma || !beta || rows <= 0) return FZ_ERR_BAD_ARG;
    if (channels % 8 || channels / 8 > 64 * LN_MAXV) return FZ_ERR_UNSUPPORTED;
    dim3 grid((unsigned)((rows + 4 * LN_ROWS - 1) / (4 * LN_ROWS))), block(256);
    const int v = channels / 8;
    if (v <= 64) {
        FZ_LAUNCH(layernorm_kernel<1>, grid, block, 0, stream, (const half_t*)x, (half_t*)y, (const half_t*)gamma,
                  (const half_t*)beta, rows, channels, eps);
    } else if (v <= 128) {
        FZ_LAUNCH(layernorm_kernel<2>, grid, block, 0, stream, (const half_t*)x, (half_t*)y, (const half_t*)gamma,
                  (const half_t*)beta, rows, channels, eps);
    } else {
        FZ_LAUNCH(layernorm_kernel<LN_MAXV>, grid, block, 0, stream, (const half_t*)x, (half_t*)y, (const half_t*)gamma,
                  (const half_t*)beta, rows, channels, eps);
    }
    return fz_last_launch_status();
}
