// lora_pair.hip -- the temporal LoRA pair of a PseudoConv3d in ONE launch (reference: video_diffusion/models/lora.py:31-54,
// LoRALinearLayer.forward on '(b h w) c f': `up(down(x)) + x`, two bias-free Conv1d(k=3, pad=1) over the frame axis; called from
// resnet.py:57-80 behind every 3x3 convolution of the UNet).
//
//   d[f]  = sum_t  Wd[t] x[f + t - 1]            rank-160 intermediate, zero padded at the clip ends
//   y[f]  = sum_t  Wu[t] d[f + t - 1] + x[f] (+ temb[clip]) (+ res2[f])
//
// As two launches (fz_temporal_conv3 twice) the pair costs 23 + 31 us at the 64^2 level with 8 frames and 35 + 53 us with 16: d makes a
// round trip through L2 / Infinity Cache, the rank-160 down projection has 128 tiles of 160 x 256 (half the chip), and the K = 480 up
// projection is prologue / epilogue-shaped.  Here a workgroup owns a TOKEN BLOCK x ALL FRAMES of a clip -- 128 rows = (128 / F) tokens x F
// frames, so both temporal neighbours of every row live in the same workgroup -- and runs both GEMMs back to back:
//   phase 1  D[160][128] = Wd (160 x 3C) . X^T: K loop over (tap, 64-channel chunk), both operands HBM / L2 -> LDS by LDS-DMA, 3-stage
//            ring (the loop of csrc/igemm.hip); 8 waves = 2 x 4: the A side splits 3 + 2 MFMA tiles, the B side 4 x 32 rows;
//   --       D rounded to fp16 into LDS as Dl[row][160 + 8] (+ one row of zeros: the clip-end padding of the up convolution);
//   phase 2  Y[320][128] per 320-column slice of C: A = Wu rows streamed by LDS-DMA in K steps of 32 (4-stage ring), B fragments read
//            STRAIGHT from Dl -- tap t of row (f, tok) is row (f + t - 1, tok) of the same tile or the zero row --; 8 waves = 2 x 4 of
//            5 x 1 tiles; epilogue + x (+ temb) (+ res2) through LDS, 16-byte stores.
// 256 workgroups at 8 frames x 64^2, 512 at 16 frames.  Same arithmetic as the two launches: fp32 accumulation, d rounded to fp16 once,
// the same K order inside each GEMM (the result is bit-identical to fz_temporal_conv3 run twice; tests/kernel_cases.py: case_lora_pair).
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"

#define LP_RANK 160
#define LP_ROWS 128
#define LP_DSTR (LP_RANK + 8)
#define LP_CSTR (320 + 8)

FZ_DEVICE_GLOBAL __attribute__((aligned(16))) half_t lp_zero_page[512];  // 1 KB of zeros: lanes of padded rows / missing frames point here

struct LoraPairArgs {
    const half_t* x;     // [N][tokens][C]
    const half_t* wd;    // [160][3][C]
    const half_t* wu;    // [C][3][160]
    const half_t* temb;  // [N / F] rows of C values, temb_stride apart, or null
    const half_t* res2;  // [N][tokens][C] or null
    half_t* y;           // [N][tokens][C]
    float* gs_out;       // GS instantiations: Welford partials [N][gs_groups][gs_chunks][3] of y's GroupNorm statistics
    int64_t temb_stride;
    int N, tokens, C, F, tok_blk, blocks_per_clip, gs_groups, gs_chunks;
};

// LDS map (halves).  Phase 1 ring and {Dl, phase 2 ring} alias: D is written after the last phase-1 tile was consumed.
#define LP_A1 (24 * 512)                    /* 160 weight rows x 64 halves = 20 DMA instructions, padded to 3 per wave */
#define LP_B1 (16 * 512)                    /* 128 x rows x 64 halves = 16 DMA instructions, 2 per wave */
#define LP_STAGE1 (LP_A1 + LP_B1)
#define LP_DL 0
#define LP_DL_HALVES ((LP_ROWS + 1) * LP_DSTR)
#define LP_RING2 21760                      /* >= LP_DL_HALVES (21672), 128-byte aligned */
#define LP_STAGE2 (24 * 512)                /* 320 weight rows x 32 halves = 20 DMA instructions, padded to 3 per wave */
#define LP_NS1 3
#define LP_NS2 4
#define LP_LDS_HALVES (LP_RING2 + LP_NS2 * LP_STAGE2)
static_assert(LP_NS1 * LP_STAGE1 <= LP_LDS_HALVES && LP_DL_HALVES <= LP_RING2 && 64 * LP_CSTR + 512 * 3 * 2 <= LP_NS2 * LP_STAGE2, "LDS map");
static_assert(LP_LDS_HALVES * 2 <= 160 * 1024, "LDS");

// GS = channels per GroupNorm group (0: no statistics; 10 / 20: the launch also writes the Welford partials of what it stores, as the GS
// instantiations of csrc/igemm.hip do -- one record per (frame, group, min(64, 128 / F)-row piece of the workgroup's tile)).
template <int GS>
FZ_KERNEL void __launch_bounds__(512, 2) lora_pair_kernel(LoraPairArgs g) {
    FZ_DYN_SMEM(raw);
    half_t* smem = reinterpret_cast<half_t*>(raw);
    const int tid = threadIdx.x, wave = fz_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int clip = (int)blockIdx.x / g.blocks_per_clip, tblk = (int)blockIdx.x % g.blocks_per_clip;
    const int t0 = tblk * g.tok_blk, TOK = g.tok_blk, F = g.F, C = g.C;
    const char* zero = reinterpret_cast<const char*>(lp_zero_page);
    auto row_frame = [&](int r) { return r / TOK; };
    auto row_px = [&](int r, int f) { return ((int64_t)(clip * F + f) * g.tokens + t0 + (r - (r / TOK) * TOK)); };  // pixel row of tile row r in frame f

    // =========================================== phase 1: D = Wd . X^T ===========================================
    {
        const int wa = wave >> 2, wb = wave & 3;
        const int na = wa == 0 ? 3 : 2, atile0 = wa == 0 ? 0 : 3;  // A tiles (of 32 rank rows) of this wave
        // per-lane DMA sources: instruction i of wave `wave` covers tile rows 8 * (i * 8 + wave) .. + 8, lane -> (row = lane / 8, chunk lane % 8)
        const int pos = lane & 7;
        const char* aptr[3];
        for (int i = 0; i < 3; ++i) {
            const int row = (i * 8 + wave) * 8 + (lane >> 3);
            const int sc = pos ^ ((row >> 1) & 7);
            aptr[i] = row < LP_RANK ? reinterpret_cast<const char*>(g.wd + (int64_t)row * 3 * C) + sc * 16 : nullptr;
        }
        int brow[2], bsc[2];
        for (int i = 0; i < 2; ++i) {
            brow[i] = (i * 8 + wave) * 8 + (lane >> 3);
            bsc[i] = pos ^ ((brow[i] >> 1) & 7);
        }
        const char* bptr[2];
        auto retarget = [&](int tap) {
            for (int i = 0; i < 2; ++i) {
                const int f = row_frame(brow[i]), fs = f + tap - 1;
                bptr[i] = (fs >= 0 && fs < F) ? reinterpret_cast<const char*>(g.x + row_px(brow[i], fs) * C) + bsc[i] * 16 : nullptr;
            }
        };
        const int kchunks = C / 64, nkt = 3 * kchunks;
        int itap = 0, ikc = 0;
        retarget(0);
        auto issue = [&](int buf) {
            char* Ab = reinterpret_cast<char*>(smem + buf * LP_STAGE1);
            char* Bb = Ab + LP_A1 * 2;
            const int64_t ka = (int64_t)(itap * C + ikc * 64) * 2, kb = (int64_t)ikc * 128;
            for (int i = 0; i < 3; ++i) fz_glds16(aptr[i] != nullptr ? aptr[i] + ka : zero, Ab + (i * 8 + wave) * 1024);
            for (int i = 0; i < 2; ++i) fz_glds16(bptr[i] != nullptr ? bptr[i] + kb : zero, Bb + (i * 8 + wave) * 1024);
            if (++ikc == kchunks) {
                ikc = 0;
                ++itap;
                if (itap < 3) retarget(itap);
            }
        };
        f32x16 acc[3];
        for (int i = 0; i < 3; ++i) acc[i] = fz_zero_f16v();
        const int fsw = (l31 >> 1) & 7;
        issue(0);
        issue(1);  // (nkt >= 15)
        int buf = 0;
        for (int it = 0; it < nkt; ++it) {
            if (it + 2 <= nkt) {
                fz_wait_vm<5>();  // 5 DMA instructions per wave and stage: this tile has landed, the next may stay in flight
            } else {
                fz_wait_vm0();
            }
            fz_barrier_nodrain();
            if (it + 2 < nkt) {
                int nb = buf + 2;
                nb = nb >= LP_NS1 ? nb - LP_NS1 : nb;
                issue(nb);
            }
            const half_t* As = smem + buf * LP_STAGE1;
            const half_t* Bs = As + LP_A1;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int co = ((2 * kk + hi) ^ fsw) * 8;
                const half8_t bf = fz_ld_h8(Bs + (wb * 32 + l31) * 64 + co);
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (i < na) acc[i] = fz_mfma_32x32x16_f16(fz_ld_h8(As + ((atile0 + i) * 32 + l31) * 64 + co), bf, acc[i]);
            }
            buf = buf + 1 == LP_NS1 ? 0 : buf + 1;
        }
        __syncthreads();  // every wave is done with the ring: Dl may overwrite it
        half_t* Dl = smem + LP_DL;
        const int r = wb * 32 + l31;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < na) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    half4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (half_t)acc[i][4 * gq + e];
                    *reinterpret_cast<half4_t*>(Dl + r * LP_DSTR + (atile0 + i) * 32 + 8 * gq + 4 * hi) = v;
                }
            }
        if (tid < LP_DSTR / 8) fz_st_h8(Dl + LP_ROWS * LP_DSTR + tid * 8, fz_zero_h8());  // the zero row
        __syncthreads();
    }

    // =========================================== phase 2: Y = Wu . D^T (+ x + temb + res2) ========================
    const int wa = wave >> 2, wb = wave & 3;
    const half_t* Dl = smem + LP_DL;
    // rows of Dl that hold d[f + t - 1] for this lane's B row (tile row wb * 32 + l31), or the zero row beyond the clip ends
    int drow[3];
    {
        const int r = wb * 32 + l31, f = row_frame(r);
        for (int t = 0; t < 3; ++t) {
            const int fs = f + t - 1;
            drow[t] = (fs >= 0 && fs < F) ? r + (t - 1) * TOK : LP_ROWS;
        }
    }
    const int pos4 = lane & 3;
    half_t* ring2 = smem + LP_RING2;
    for (int at = 0; at < C / 320; ++at) {
        // DMA sources of this 320-row slice of Wu: instruction i of wave `wave` covers rows 16 * (i * 8 + wave) .. + 16 (64 B per row)
        const char* aptr[3];
        for (int i = 0; i < 3; ++i) {
            const int row = (i * 8 + wave) * 16 + (lane >> 2);
            const int sc = pos4 ^ ((row >> 2) & 3);
            aptr[i] = row < 320 ? reinterpret_cast<const char*>(g.wu + (int64_t)(at * 320 + row) * 3 * LP_RANK) + sc * 16 : nullptr;
        }
        int is = 0;  // K-32 step to issue next: (tap, chunk) = (is / 5, is % 5), K offset is * 32 halves
        auto issue2 = [&](int buf) {
            char* Ab = reinterpret_cast<char*>(ring2 + buf * LP_STAGE2);
            for (int i = 0; i < 3; ++i) fz_glds16(aptr[i] != nullptr ? aptr[i] + (int64_t)is * 64 : zero, Ab + (i * 8 + wave) * 1024);
            ++is;
        };
        f32x16 acc[5];
        for (int i = 0; i < 5; ++i) acc[i] = fz_zero_f16v();
        const int fsw = (l31 >> 2) & 3;
        constexpr int NKT = 15;
        issue2(0);
        issue2(1);
        issue2(2);
        int buf = 0;
        for (int it = 0; it < NKT; ++it) {
            if (it + 3 <= NKT) {
                fz_wait_vm<6>();  // 3 DMA instructions per wave and stage: two later tiles may stay in flight
            } else if (it + 2 <= NKT) {
                fz_wait_vm<3>();
            } else {
                fz_wait_vm0();
            }
            fz_barrier_nodrain();
            if (it + 3 < NKT) {
                int nb = buf + 3;
                nb = nb >= LP_NS2 ? nb - LP_NS2 : nb;
                issue2(nb);
            }
            const half_t* As = ring2 + buf * LP_STAGE2;
            const int t = it / 5, kc = it - 5 * t;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const half8_t bf = fz_ld_h8(Dl + drow[t] * LP_DSTR + kc * 32 + (2 * kk + hi) * 8);
                const int co = ((2 * kk + hi) ^ fsw) * 8;
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = fz_mfma_32x32x16_f16(fz_ld_h8(As + ((wa * 5 + i) * 32 + l31) * 32 + co), bf, acc[i]);
            }
            buf = buf + 1 == LP_NS2 ? 0 : buf + 1;
        }
        // ---- epilogue of the slice: two passes of 64 rows through the (now idle) ring
        half_t* Cs = ring2;
        for (int ps = 0; ps < 2; ++ps) {
            __syncthreads();
            if ((wb >> 1) == ps) {
                half_t* crow = Cs + ((wb & 1) * 32 + l31) * LP_CSTR;
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        half4_t v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = (half_t)acc[i][4 * gq + e];
                        *reinterpret_cast<half4_t*>(crow + (wa * 5 + i) * 32 + 8 * gq + 4 * hi) = v;
                    }
            }
            __syncthreads();
            for (int id = tid; id < 64 * 40; id += 512) {
                const int pl = id / 40, ch = id - pl * 40;
                const int r = ps * 64 + pl, f = row_frame(r);
                const int64_t px = row_px(r, f);
                const int co = at * 320 + ch * 8;
                const half8_t v = fz_ld_h8(Cs + pl * LP_CSTR + ch * 8);
                const half8_t xv = fz_ld_h8(g.x + px * C + co);
                float fv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) fv[e] = (float)v[e];
                // (the order of fz_temporal_conv3's epilogue: + temb, + res (= x), + res2)
                if (g.temb != nullptr) {
                    const half8_t tv = fz_ld_h8(g.temb + (int64_t)clip * g.temb_stride + co);
#pragma unroll
                    for (int e = 0; e < 8; ++e) fv[e] += (float)tv[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) fv[e] += (float)xv[e];
                if (g.res2 != nullptr) {
                    const half8_t rv = fz_ld_h8(g.res2 + px * C + co);
#pragma unroll
                    for (int e = 0; e < 8; ++e) fv[e] += (float)rv[e];
                }
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)fv[e];
                fz_st_h8(g.y + px * C + co, o);
                if constexpr (GS > 0) fz_st_h8(Cs + pl * LP_CSTR + ch * 8, o);  // the staging tile now holds what was STORED
            }
            if constexpr (GS > 0) {
                // GroupNorm statistics of the 64 rows x 320 columns this pass stored.  Thread (group gi, row slice rs) sums RPT consecutive
                // rows of the group's GS channels, shifted by the first value it sees (all LDS loads in flight before the first use); one
                // thread per (frame piece, group) Chan-merges the piece's slices in slice order.  Deterministic.
                constexpr int NGRP = 320 / GS, RPT = NGRP / 8, SL = 64 / RPT;
                static_assert(NGRP * SL == 512 && GS % 2 == 0, "one (group, slice) item per thread");
                float* red = reinterpret_cast<float*>(Cs + 64 * LP_CSTR);
                __syncthreads();
                {
                    const int gi = tid % NGRP, rs = tid / NGRP;
                    const half_t* base = Cs + gi * GS + rs * RPT * LP_CSTR;
                    half2_t v[RPT][GS / 2];
#pragma unroll
                    for (int r = 0; r < RPT; ++r)
#pragma unroll
                        for (int c = 0; c < GS / 2; ++c) v[r][c] = *reinterpret_cast<const half2_t*>(base + r * LP_CSTR + 2 * c);
                    const float p0 = (float)v[0][0][0];
                    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                    for (int r = 0; r < RPT; ++r)
#pragma unroll
                        for (int c = 0; c < GS / 2; ++c) {
                            const float d0 = (float)v[r][c][0] - p0, d1 = (float)v[r][c][1] - p0;
                            s1 += d0 + d1;
                            s2 += d0 * d0 + d1 * d1;
                        }
                    const float n = (float)(RPT * GS);
                    red[(gi * SL + rs) * 3 + 0] = n;
                    red[(gi * SL + rs) * 3 + 1] = p0 + s1 / n;
                    red[(gi * SL + rs) * 3 + 2] = s2 - s1 * s1 / n;
                }
                __syncthreads();
                const int rows_rec = TOK < 64 ? TOK : 64;       // rows of one frame inside this pass
                const int spr = rows_rec / RPT;                 // slices per record
                if (tid < NGRP * (64 / rows_rec)) {
                    const int gi = tid % NGRP, piece = tid / NGRP;
                    const float* q0 = red + (gi * SL + piece * spr) * 3;
                    float cnt = q0[0], mean = q0[1], m2 = q0[2];
                    for (int q = 1; q < spr; ++q) {  // Chan et al., the order gn_finalize uses
                        const float nb = q0[q * 3], mb = q0[q * 3 + 1], m2b = q0[q * 3 + 2];
                        const float tot = cnt + nb, delta = mb - mean;
                        mean += delta * (nb / tot);
                        m2 += m2b + delta * delta * (cnt * nb / tot);
                        cnt = tot;
                    }
                    const int r0 = ps * 64 + piece * rows_rec, f = r0 / TOK;
                    const int chunk = tblk * (TOK > 64 ? TOK / 64 : 1) + (r0 - f * TOK) / 64;
                    float* out = g.gs_out + (((int64_t)(clip * F + f) * g.gs_groups + at * NGRP + gi) * g.gs_chunks + chunk) * 3;
                    out[0] = cnt;
                    out[1] = mean;
                    out[2] = m2;
                }
            }
        }
        __syncthreads();  // the staging tile is read: the next slice's prologue may refill the ring
    }
}

extern "C" int fz_lora_pair_ok(int n, int tokens, int channels, int rank, int clip_len) {
    if (n <= 0 || tokens <= 0 || clip_len <= 0 || n % clip_len || rank != LP_RANK) return 0;
    if (channels % 320 || channels > 1280 || LP_ROWS % clip_len) return 0;
    return tokens % (LP_ROWS / clip_len) == 0 ? 1 : 0;
}

// Where the one launch is FASTER than fz_temporal_conv3 twice (profiles/r04_lora_pair_ab.txt): every workgroup streams both weight
// matrices through its LDS for 128 rows, which only pays when the launch fills the chip -- >= 256 workgroups (the 64^2 level of the UNet).
extern "C" int fz_lora_pair_preferred(int n, int tokens, int channels, int rank, int clip_len) {
    if (!fz_lora_pair_ok(n, tokens, channels, rank, clip_len)) return 0;
    return (int64_t)n * tokens / LP_ROWS >= 256 ? 1 : 0;
}

// records per (frame, group) of the statistics form, or 0 where it does not exist: group width 10 / 20 (320 / 640 channels in 32 groups),
// >= 4 rows of a frame per workgroup (clip_len <= 32)
extern "C" int fz_lora_pair_gn_chunks(int n, int tokens, int channels, int rank, int clip_len, int gn_groups) {
    if (gn_groups <= 0 || channels % gn_groups || !fz_lora_pair_ok(n, tokens, channels, rank, clip_len)) return 0;
    const int cpg = channels / gn_groups, tok = LP_ROWS / clip_len;
    if ((cpg != 10 && cpg != 20) || tok < 4) return 0;
    return tokens / tok * (tok > 64 ? tok / 64 : 1);
}

static int lora_pair_launch(const void* x, const void* w_down, const void* w_up, const void* temb, int64_t temb_stride, const void* res2,
                            void* y, int n, int tokens, int channels, int rank, int clip_len, float* gn_partial, int gn_groups,
                            void* stream) {
    if (!x || !w_down || !w_up || !y) return FZ_ERR_BAD_ARG;
    if (!fz_lora_pair_ok(n, tokens, channels, rank, clip_len)) return FZ_ERR_UNSUPPORTED;
    if (temb != nullptr && (temb_stride % 8)) return FZ_ERR_UNSUPPORTED;
    LoraPairArgs g = {};
    g.x = (const half_t*)x; g.wd = (const half_t*)w_down; g.wu = (const half_t*)w_up;
    g.temb = (const half_t*)temb; g.temb_stride = temb_stride ? temb_stride : channels;
    g.res2 = (const half_t*)res2; g.y = (half_t*)y;
    g.N = n; g.tokens = tokens; g.C = channels; g.F = clip_len;
    g.tok_blk = LP_ROWS / clip_len;
    g.blocks_per_clip = tokens / g.tok_blk;
    int gs = 0;
    if (gn_partial != nullptr) {
        g.gs_chunks = fz_lora_pair_gn_chunks(n, tokens, channels, rank, clip_len, gn_groups);
        if (g.gs_chunks == 0) return FZ_ERR_UNSUPPORTED;
        g.gs_out = gn_partial; g.gs_groups = gn_groups;
        gs = channels / gn_groups;
    }
    const int64_t blocks = (int64_t)(n / clip_len) * g.blocks_per_clip;
    if (blocks <= 0 || blocks >= (1ll << 31)) return FZ_ERR_BAD_ARG;
    const size_t lds = (size_t)LP_LDS_HALVES * sizeof(half_t);
    void (*kern)(LoraPairArgs) = gs == 10 ? &lora_pair_kernel<10> : (gs == 20 ? &lora_pair_kernel<20> : &lora_pair_kernel<0>);
#ifndef FZ_EMU
    static unsigned long long attr_mask[3] = {0, 0, 0};
    const int ki = gs == 10 ? 1 : (gs == 20 ? 2 : 0);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FZ_ERR_LAUNCH;
    if (dev >= 64 || !((attr_mask[ki] >> dev) & 1ull)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return FZ_ERR_LAUNCH;
        if (dev < 64) attr_mask[ki] |= 1ull << dev;
    }
#endif
    if (gs == 10) {
        FZ_LAUNCH(lora_pair_kernel<10>, dim3((unsigned)blocks), dim3(512), lds, stream, g);
    } else if (gs == 20) {
        FZ_LAUNCH(lora_pair_kernel<20>, dim3((unsigned)blocks), dim3(512), lds, stream, g);
    } else {
        FZ_LAUNCH(lora_pair_kernel<0>, dim3((unsigned)blocks), dim3(512), lds, stream, g);
    }
    (void)kern;
    return fz_last_launch_status();
}

extern "C" int fz_lora_pair(const void* x, const void* w_down, const void* w_up, const void* temb, int64_t temb_stride, const void* res2,
                            void* y, int n, int tokens, int channels, int rank, int clip_len, void* stream) {
    return lora_pair_launch(x, w_down, w_up, temb, temb_stride, res2, y, n, tokens, channels, rank, clip_len, nullptr, 0, stream);
}

extern "C" int fz_lora_pair_gn(const void* x, const void* w_down, const void* w_up, const void* temb, int64_t temb_stride, const void* res2,
                               void* y, int n, int tokens, int channels, int rank, int clip_len, float* gn_partial, int gn_groups,
                               void* stream) {
    if (!gn_partial) return FZ_ERR_BAD_ARG;
    return lora_pair_launch(x, w_down, w_up, temb, temb_stride, res2, y, n, tokens, channels, rank, clip_len, gn_partial, gn_groups, stream);
}
