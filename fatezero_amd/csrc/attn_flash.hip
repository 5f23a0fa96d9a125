// attn_flash.hip -- the MFMA-bound member of the attention family: plain (no map) sparse-causal self-attention,
// i.e. the 64x64-token level of SD-1.x (Lq 4096, Lk 8192, d 40: 86 % of all attention FLOPs) and the
// unconditional CFG half of every other level.  Same transposed formulation as attn_self.hip
// (S^T = K Q^T, O^T = V^T P^T, lane <-> query row, registers of S^T are the B operand of the PV MFMA), plus:
//   * software pipelining: the next K/V tile is fetched global->registers before the MFMAs of the current tile
//     and written to the OTHER LDS buffer after them: one __syncthreads per 64-key tile, HBM/L2 latency hidden
//     under the 14 (d=40) MFMAs + softmax of the current tile;
//   * deferred rescale: O/l are only rescaled when some row's running max grew by more than 2^6
//     (wave-uniform branch); P stays < 2^6 in fp16, the accumulators are fp32;
//   * the softmax denominator costs no VALU when d % 32 != 0: one padding row of the V^T tile is set to 1.0, so
//     row D of O^T accumulates sum_k P[q,k] inside the PV MFMA (and is rescaled with O for free);
//   * key masking only on the (wave-uniform) tail tile of a kv slot;
//   * when d % 16 != 0 (d = 40) and q arrives in the log2 domain (desc.q_log2_scaled: scale*log2e folded into Wq by
//     the host) the running max costs no VALU either: the free contraction slot D holds 1.0 on the K side and -m
//     (kept fp16-representable, so the product is exact) on the Q side, and the QK^T MFMA delivers s - m directly:
//     per score the VALU work is exp2 + 1/2 max3 + 1/2 cvt_pk.
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"
#include <stdlib.h>

#define FQBLK 128
#define FKVBLK 64
#define FVSTR 72

template <int D, int QB, int NSTG = 2>
struct FlashCfg {
    static constexpr int QROWS = 128 * QB;  // query rows per workgroup: 4 waves x QB blocks of 32
    static constexpr int DP16 = (D + 15) / 16 * 16;
    static constexpr int NC = DP16 / 16;
    static constexpr int NT = (D + 31) / 32;
    static constexpr int KSTR = DP16 + 8;
    static constexpr int KCH = DP16 / 8;
    static constexpr int VROWS = NT * 32;
    static constexpr int OSTR = NT * 32 + 8;
    static constexpr bool ONES_ROW = (D % 32) != 0;  // a free padding row of V^T carries the softmax denominator
    static constexpr bool BIAS_SLOT = (D % 16) != 0 && (D % 8) == 0;  // a free contraction slot can carry -max
    static constexpr int BC = D / 16, BHI = (D % 16) / 8, BE = D % 8;  // Q fragment chunk / lane half / element of slot D
    static constexpr int KS = FKVBLK * KSTR;
    static constexpr int VS = VROWS * FVSTR;
    static constexpr int STAGE = KS + VS;
    static constexpr int OS = QROWS * OSTR;
    static constexpr int LDS_HALVES = (NSTG * STAGE > OS) ? NSTG * STAGE : OS;
    static constexpr int DCH = D / 8;                       // 16-byte DATA chunks per K row; chunks [DCH, KCH) are padding
    static constexpr int KLD = (FKVBLK * DCH + 255) / 256;  // 16-byte K chunks per thread per tile (data chunks only)
    static constexpr int VLD = (D * 8 + 255) / 256;         // 16-byte V^T chunks per thread per tile (rows < D only)
    static_assert(D % 8 == 0, "head dim in 16-byte chunks");
};

#ifdef FZ_FLASH_TIMING  // scripts/flash_timing.hip: per-segment s_memtime totals of wave 0 of block 0 (never in the product)
__device__ long long fz_flash_timing[8];
#define FZ_TICK(slot)                                         \
    do {                                                      \
        const long long now_ = clock64();                     \
        if (blockIdx.x == 0 && tid == 0) fz_flash_timing[slot] += now_ - tick_; \
        tick_ = now_;                                         \
    } while (0)
#else
#define FZ_TICK(slot)
#endif

FZ_DEVICE int fl_pi(int i) {
    const int a = i >> 3, hp = (i >> 2) & 1, t = i & 3;
    return ((a & 2) << 3) + 8 * hp + 4 * (a & 1) + t;
}

// Source frame of kv slot j for clip frame f of batch element b (attention.py:383-386: relative slots clamp into the clip).
FZ_HOST_DEVICE inline int fl_kv_source(const FzAttnSelfDesc& d, int b, int f, int j) {
    const int kvl = d.kv_clip_len ? d.kv_clip_len : d.clip_len;  // frames per batch element of k / vt
    int s = d.kv_abs[j] ? d.kv_val[j] : f + (d.kv_clip_len ? d.kv_frame_off : 0) + d.kv_val[j];
    s = s < 0 ? 0 : (s > kvl - 1 ? kvl - 1 : s);
    return b * kvl + s;
}

// Dispatch order of the frames of a launch.  A frame whose kv slots ALL resolve to one source frame after clamping (frames 0 and 1
// of a clip with [-1, 'first']: both slots are frame 0) reads that source once -- the softmax over a key set listed n times, against
// values listed n times, is the softmax over the set listed once (not so when only SOME slots coincide: [0, 1, 1] weighs frame 1
// twice) -- so its workgroups run 1/n as long; the launcher lists the full-length frames first so that the short ones fill the
// last round instead of opening a new one.
struct FlashOrder {
    int heads_per_xcd;  // > 0: every XCD owns whole heads and walks the frames in `fl` order; 0: plain group order
    unsigned char fl[64];
};

// NPOLY (trial, scripts/flash_ab.hip): of the 32 exponentials a lane evaluates per query block and key tile, NPOLY go through the
// polynomial fz_exp2_poly2 (packed FMAs, which co-issue with MFMAs) instead of v_exp_f32 (which does not); the library ships 0 --
// profiles/r04_flash_exp_split_ab.txt has the A/B.
template <int D, int W, int QB, bool USE_BIAS, int NSTG, int NPOLY = 0>
FZ_KERNEL void __launch_bounds__(256, W)
attn_flash_kernel(FzAttnSelfDesc d, const half_t* __restrict__ q, const half_t* __restrict__ k,
                  const half_t* __restrict__ vt, half_t* __restrict__ o, FlashOrder ord) {
    typedef FlashCfg<D, QB, NSTG> C;
    static_assert(NSTG == 2 || NSTG == 4, "K/V ring of 2 stages (barrier per tile) or 4 (barrier per pair of tiles)");
    static_assert(C::LDS_HALVES * 2 <= 160 * 1024 / W, "LDS per workgroup at W workgroups per CU");
    FZ_SHARED __attribute__((aligned(16))) half_t smem[C::LDS_HALVES];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq_ = lane & 31, hi = lane >> 5;
    const int nq = (d.lq + C::QROWS - 1) / C::QROWS;
    const int groups = d.heads * d.n_frames;
    int h, fl, qt;
    {
        const int bid = blockIdx.x;
        if ((groups & 7) == 0) {
            const int xcd = bid & 7, idx = bid >> 3, lg = idx / nq;  // lg: this XCD's lg-th (head, frame) group
            qt = idx % nq;
            if (ord.heads_per_xcd > 0) {
                h = xcd * ord.heads_per_xcd + lg % ord.heads_per_xcd;
                fl = ord.fl[lg / ord.heads_per_xcd];
            } else {
                const int group = xcd * (groups >> 3) + lg;
                h = group / d.n_frames;
                fl = group % d.n_frames;
            }
        } else {
            const int group = bid / nq;
            qt = bid % nq;
            h = group / d.n_frames;
            fl = group % d.n_frames;
        }
    }
    const int n = d.frame0 + fl, b = n / d.clip_len, f = n % d.clip_len;
    constexpr bool BIAS = C::BIAS_SLOT && USE_BIAS;
    // Padded keys (last tile of a kv slot whose length is not a multiple of 64).  With the bias slot AND the ones row the
    // tile can be neutralised where it is written to LDS -- K row all zeros (slot D included: score - m = 0, P = 1) against
    // V^T / ones-row columns of zeros (no contribution to O^T or to the denominator) -- so the softmax of the hot variant
    // carries no mask arithmetic at all (hipcc speculates the 32 compares per tile into the common path otherwise).  The
    // launcher keeps lkf < 64 away from this variant (a first tile made of padding alone would pin the running max at 0).
    constexpr bool STASH_MASKS = BIAS && C::ONES_ROW;
    const float cs = d.q_log2_scaled ? 1.0f : d.scale * 1.4426950408889634f;
    // each wave owns QB blocks of 32 query rows: every K / V^T fragment read from LDS feeds QB MFMAs, and the
    // independent blocks let the MFMAs of one overlap the softmax VALU of the other inside the wave
    half8_t qf[QB][C::NC];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const int qrow = qt * C::QROWS + (wave * QB + u) * 32 + lq_;
#pragma unroll
        for (int c = 0; c < C::NC; ++c) {
            const int dd = 16 * c + 8 * hi;
            qf[u][c] = (qrow < d.lq && dd < D)
                           ? fz_ld_h8(q + (int64_t)n * d.q_frame_stride + (int64_t)qrow * d.q_row_stride + h * D + dd)
                           : fz_zero_h8();
        }
    }
    // source frames of the kv slots (wave-uniform scalars); when ALL slots resolve to one frame the keys are read once
    int src[FZ_MAX_KV_SLOTS];
    bool one_source = true;
#pragma unroll
    for (int j = 0; j < FZ_MAX_KV_SLOTS; ++j) {
        src[j] = fl_kv_source(d, b, f, j);
        one_source = one_source && (j >= d.n_kv || src[j] == src[0]);
    }
    const int nkv = one_source ? 1 : d.n_kv;
    const int lkfp = (d.lkf + FKVBLK - 1) / FKVBLK * FKVBLK;
    const int tps = lkfp / FKVBLK;
    const int ntiles = nkv * tps;
    const int64_t khs = d.k_head_stride ? d.k_head_stride : (int64_t)D;

    // ---- constant parts of the K / V^T stages, written once: the padding chunks of every K row (zeros; slot D = 1.0 when
    // it carries the running max), the padding rows of V^T (zeros; row D = 1.0 when it carries the softmax denominator).
    // fix_pads(stage, valid) rewrites the parts that depend on the number of valid keys of the tile in that stage.
    half8_t kpad = fz_zero_h8();  // what the first padding chunk of a K row holds
    if (BIAS) kpad[C::BE] = (half_t)1.0f;
    auto fix_pads = [&](int st, int valid) {
        half_t* Ks = smem + st * C::STAGE;
        half_t* Vs = Ks + C::KS;
        constexpr int NPAD = C::KCH - C::DCH;
        for (int id = tid; id < FKVBLK * NPAD; id += 256) {
            const int key = id / (NPAD > 0 ? NPAD : 1), ch = C::DCH + id % (NPAD > 0 ? NPAD : 1);
            fz_st_h8(Ks + key * C::KSTR + ch * 8, (ch == C::DCH && key < valid) ? kpad : fz_zero_h8());
        }
        if (C::ONES_ROW && tid < 8) {
            half8_t ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (tid * 8 + e < valid) ? (half_t)1.0f : (half_t)0.0f;
            fz_st_h8(Vs + D * FVSTR + tid * 8, ov);
        }
    };
    for (int st = 0; st < NSTG; ++st) {
        constexpr int PADCH = (C::VROWS > D + (C::ONES_ROW ? 1 : 0) ? C::VROWS - D - (C::ONES_ROW ? 1 : 0) : 0) * 8;
        for (int id = tid; id < PADCH; id += 256)
            fz_st_h8(smem + st * C::STAGE + C::KS + (D + (C::ONES_ROW ? 1 : 0) + id / 8) * FVSTR + (id % 8) * 8, fz_zero_h8());
        fix_pads(st, FKVBLK);
    }

    // Prefetch registers.  The loads are UNCONDITIONAL (addresses clamped into the tensor, invalid data replaced when it
    // is written to LDS): a predicated load makes hipcc wrap it in control flow and then wait vmcnt(0) at the join,
    // i.e. right before the first MFMA of the tile -- which would expose the whole global-load latency every tile.
    // Addresses are strength-reduced: a wave-uniform 64-bit base that walks along the keys of the current kv slot, plus a
    // 32-bit per-lane byte offset whose only per-tile work is the clamp of the key row on a slot's ragged last tile
    // (one v_min + one v_mad_u32_u24 per K chunk; the V^T offsets never change).  A thread whose chunk index runs past the
    // tile repeats the tile's last chunk (same data to the same LDS address as its owner): no predication in the stash either.
    half8_t kreg[C::KLD], vreg[C::VLD];
    uint32_t kkey[C::KLD], kdd[C::KLD], voff[C::VLD];
    int klds[C::KLD], vlds[C::VLD];  // LDS offsets (halves) inside a stage
    const uint32_t krs_bytes = (uint32_t)d.k_row_stride * 2u;
#pragma unroll
    for (int i = 0; i < C::KLD; ++i) {
        int id = tid + 256 * i;
        id = id < FKVBLK * C::DCH ? id : FKVBLK * C::DCH - 1;
        const int ch = id % C::DCH;
        kkey[i] = id / C::DCH;
        kdd[i] = 2u * (ch * 8);
        klds[i] = (id / C::DCH) * C::KSTR + ch * 8;
    }
#pragma unroll
    for (int i = 0; i < C::VLD; ++i) {
        int id = tid + 256 * i;
        id = id < D * 8 ? id : D * 8 - 1;
        voff[i] = (uint32_t)(id >> 3) * (uint32_t)d.vt_chan_stride * 2u + (id & 7) * 16u;
        vlds[i] = C::KS + (id >> 3) * FVSTR + (id & 7) * 8;
    }
    int fj = 0, fr0 = 0;  // kv slot and first key row of the next tile to fetch
    const char *kcur = nullptr, *vcur = nullptr;
    auto fetch = [&]() {
        if (fr0 == 0) {  // wave-uniform: first tile of a kv slot
            kcur = reinterpret_cast<const char*>(k + (int64_t)src[fj] * d.k_frame_stride + (int64_t)h * khs);
            vcur = reinterpret_cast<const char*>(vt + (int64_t)src[fj] * d.vt_frame_stride + (int64_t)(h * D) * d.vt_chan_stride);
        }
        const uint32_t kmax = (uint32_t)(d.lkf - 1 - fr0);  // padded keys are masked / neutralised; any finite data will do
#pragma unroll
        for (int i = 0; i < C::KLD; ++i) {
            const uint32_t key = kkey[i] < kmax ? kkey[i] : kmax;
            kreg[i] = fz_ld_h8_off(kcur, fz_mad24(key, krs_bytes, kdd[i]));
        }
#pragma unroll
        for (int i = 0; i < C::VLD; ++i) vreg[i] = fz_ld_h8_off(vcur, voff[i]);
        kcur += (int64_t)FKVBLK * krs_bytes;
        vcur += FKVBLK * 2;
        fr0 += FKVBLK;
        if (fr0 >= lkfp) {
            fr0 = 0;
            ++fj;
        }
    };
    int sr0 = 0;        // first key row (within its kv slot) of the tile in the prefetch registers
    int dirty = 0;      // bit st: stage st holds the pads of a ragged tile
    auto stash = [&](int st) {
        half_t* Sb = smem + st * C::STAGE;
        const int valid = d.lkf - sr0;  // keys [valid, 64) of this tile are padding
        if (STASH_MASKS && (valid < FKVBLK || ((dirty >> st) & 1))) {  // wave-uniform, rare: ragged tile, or the tile after one
            const int vc = valid < FKVBLK ? valid : FKVBLK;
            fix_pads(st, vc);
            dirty = valid < FKVBLK ? (dirty | (1 << st)) : (dirty & ~(1 << st));
#pragma unroll
            for (int i = 0; i < C::KLD; ++i)
                fz_st_h8(Sb + klds[i], (int)kkey[i] < vc ? kreg[i] : fz_zero_h8());
#pragma unroll
            for (int i = 0; i < C::VLD; ++i) {
                half8_t vv = vreg[i];
                const int col = (vlds[i] - C::KS) % FVSTR;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (col + e >= vc) vv[e] = (half_t)0.0f;
                fz_st_h8(Sb + vlds[i], vv);
            }
        } else {
#pragma unroll
            for (int i = 0; i < C::KLD; ++i) fz_st_h8(Sb + klds[i], kreg[i]);
#pragma unroll
            for (int i = 0; i < C::VLD; ++i) fz_st_h8(Sb + vlds[i], vreg[i]);
        }
        sr0 += FKVBLK;
        if (sr0 >= lkfp) sr0 = 0;
    };

    float m[QB], l[QB];
    f32x16 oacc[QB][C::NT];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        m[u] = BIAS ? 0.0f : -1e30f;
        l[u] = 0.0f;
#pragma unroll
        for (int t = 0; t < C::NT; ++t) oacc[u][t] = fz_zero_f16v();
    }

#ifdef FZ_FLASH_TIMING
    long long tick_ = 0;
#endif
    // One 64-key tile: S^T = K Q^T (MFMA) -> softmax (VALU) -> O^T += V^T P^T (MFMA).
    auto tile = [&](const int kt, const int r0, const half_t* Ks) {
        const half_t* Vs = Ks + C::KS;
        f32x16 acc[QB][2];
        half8_t kfr[2][C::NC];  // every K fragment of the tile is requested before the first MFMA
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const half_t* row = Ks + (32 * sub + fl_pi(lq_)) * C::KSTR + 8 * hi;
#pragma unroll
            for (int c = 0; c < C::NC; ++c) kfr[sub][c] = fz_ld_h8(row + 16 * c);
        }
        // query block outermost: block 0's scores are complete while block 1's MFMAs still run
#pragma unroll
        for (int u = 0; u < QB; ++u)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                acc[u][sub] = fz_zero_f16v();
#pragma unroll
                for (int c = 0; c < C::NC; ++c) acc[u][sub] = fz_mfma_32x32x16_f16(kfr[sub][c], qf[u][c], acc[u][sub]);
            }
        FZ_TICK(1);
        float tmax[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            if (!STASH_MASKS && r0 + FKVBLK > d.lkf) {  // wave-uniform: last tile of a kv slot has padded keys
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 32 * sub + (r < 8 ? 8 * hi + r : 8 + 8 * hi + r);
                        if (r0 + key >= d.lkf) acc[u][sub][r] = -INFINITY;
                    }
            }
            float t = acc[u][0][0];
#pragma unroll
            for (int i = 1; i < 32; ++i) t = fmaxf(t, acc[u][i >> 4][i & 15]);
            tmax[u] = fz_pair_max32(t);  // row max of this tile over both lane halves
        }
        half8_t pf[QB][2][2];
        if (BIAS) {
            // acc already is (log2-domain score) - m[u]; m[u] only moves when a row grew by more than 2^6, and on the
            // first tile (where it starts at 0 and must come DOWN to the row max as well).  ONE wave-uniform branch for all
            // query blocks keeps the exp / PV part below a single scheduling region.
            const bool first = (kt == 0);
            bool grew = false;
#pragma unroll
            for (int u = 0; u < QB; ++u) grew = grew || tmax[u] > 6.0f;
            if (first || fz_ballot(grew) != 0ull) {
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    float mn = first ? tmax[u] : m[u] + fmaxf(tmax[u], 0.0f);
                    mn = fminf(fmaxf(mn, -60000.0f), 60000.0f);
                    mn = (float)(half_t)mn;  // fp16-representable: (1.0 * -mn) is exact inside the MFMA
                    const float delta = mn - m[u];
                    const float alpha = first ? 1.0f : fz_exp2(-delta);
                    m[u] = mn;
                    l[u] *= alpha;
#pragma unroll
                    for (int t = 0; t < C::NT; ++t) oacc[u][t] *= alpha;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[u][sub][r] -= delta;
                    if (hi == C::BHI) qf[u][C::BC][C::BE] = (half_t)(-mn);
                }
            }
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 32; i += 2) {
                    constexpr int PSTEP = NPOLY > 0 ? 32 / NPOLY : 0;  // every PSTEP-th pair takes the polynomial
                    f32x2 pe;
                    if (NPOLY > 0 && ((i >> 1) % (PSTEP > 0 ? PSTEP : 1)) == 0) {
                        pe = fz_exp2_poly2(f32x2{acc[u][i >> 4][i & 15], acc[u][i >> 4][(i & 15) + 1]});
                    } else {
                        pe = f32x2{fz_exp2(acc[u][i >> 4][i & 15]), fz_exp2(acc[u][i >> 4][(i & 15) + 1])};
                    }
                    if (!C::ONES_ROW) sum += pe[0] + pe[1];
                    pf[u][i >> 4][(i >> 3) & 1][i & 7] = (half_t)pe[0];
                    pf[u][i >> 4][(i >> 3) & 1][(i & 7) + 1] = (half_t)pe[1];
                }
                if (!C::ONES_ROW) l[u] += sum;
            }
        } else {
            bool grew = false;
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                tmax[u] *= cs;  // log2 domain (cs > 0)
                grew = grew || tmax[u] > m[u] + 6.0f;
            }
            if (fz_ballot(grew) != 0ull) {  // some row's max grew by more than 2^6: rescale
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const float mn = fmaxf(m[u], tmax[u]);
                    const float alpha = fz_exp2(m[u] - mn);
                    m[u] = mn;
                    l[u] *= alpha;
#pragma unroll
                    for (int t = 0; t < C::NT; ++t) oacc[u][t] *= alpha;
                }
            }
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const float pe = fz_exp2(acc[u][i >> 4][i & 15] * cs - m[u]);
                    if (!C::ONES_ROW) sum += pe;
                    pf[u][i >> 4][(i >> 3) & 1][i & 7] = (half_t)pe;
                }
                if (!C::ONES_ROW) l[u] += sum;
            }
        }
        FZ_TICK(2);
        // (key group, d tile, query block) order: the first MFMAs only need the first eighth of the exponentials, so the
        // exp of key group g+1 runs in the shadow of the MFMAs of group g
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int t = 0; t < C::NT; ++t) {
                    const half8_t vf = fz_ld_h8(Vs + (32 * t + lq_) * FVSTR + 8 * hi + 32 * sub + 16 * mm);
#pragma unroll
                    for (int u = 0; u < QB; ++u) oacc[u][t] = fz_mfma_32x32x16_f16(vf, pf[u][sub][mm], oacc[u][t]);
                }
        FZ_TICK(3);
    };

    // K/V ring.  NSTG == 2: tile t+1 is fetched (global -> registers) before the MFMAs of tile t and written to the other
    // stage after them, one barrier per tile.  NSTG == 4: the fetch runs two tiles ahead and the workgroup only meets
    // every second tile -- the stage written during tile t (tile t+2) was last read during tile t-2, which every wave
    // finished before the barrier that ended tile t-1 (t-1 odd) or tile t-2 (t-1 even is never the case for even t).
    constexpr int AHEAD = NSTG / 2;
    fetch();
    stash(0);
    if (AHEAD == 2 && ntiles > 1) {
        fetch();
        stash(1);
    }
    __syncthreads();
#ifdef FZ_FLASH_TIMING
    tick_ = clock64();
#endif
    int r0 = 0;  // first key row (within its kv slot) of the tile being consumed
    for (int kt = 0; kt < ntiles; ++kt) {
        const bool more = kt + AHEAD < ntiles;
        if (more) fetch();  // in flight during the MFMAs below
        FZ_TICK(0);
        const half_t* Ks = smem + (kt & (NSTG - 1)) * C::STAGE;
        tile(kt, r0, Ks);
#ifdef FZ_FLASH_TIMING  // the stash split into "wait for the prefetched tile" (slot 6) and "registers -> LDS" (slot 4)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FZ_TICK(6);
#endif
        if (more) stash((kt + AHEAD) & (NSTG - 1));
#ifdef FZ_FLASH_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        FZ_TICK(4);
        if (NSTG == 2 || (kt & 1) || kt + 1 == ntiles) __syncthreads();
        FZ_TICK(5);
        r0 += FKVBLK;
        if (r0 >= lkfp) r0 = 0;
    }

    // ---- epilogue ---------------------------------------------------------------------------------------
    half_t* Os = smem;
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        float denom;
        if (C::ONES_ROW) {
            // row D of O^T: tile D/32, row i = D%32 = (r&3) + 8(r>>2) + 4hi'
            constexpr int i = D % 32, t = D / 32;
            constexpr int hp = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
            const float mine = oacc[u][t][r];
            const float other = fz_shfl_xor(mine, 32);
            denom = (hi == hp) ? mine : other;
        } else {
            denom = l[u] + fz_shfl_xor(l[u], 32);
        }
        const float fin = 1.0f / denom;
#pragma unroll
        for (int t = 0; t < C::NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (half_t)(oacc[u][t][4 * g + e] * fin);
                *reinterpret_cast<half4_t*>(Os + ((wave * QB + u) * 32 + lq_) * C::OSTR + 32 * t + 8 * g + 4 * hi) = v;
            }
    }
    __syncthreads();
    constexpr int OCH = D / 8;
    for (int id = lane; id < 32 * QB * OCH; id += 64) {
        const int row = id / OCH, ch = id % OCH;
        const int qg = qt * C::QROWS + wave * 32 * QB + row;
        if (qg < d.lq)
            fz_st_h8(o + (int64_t)n * d.o_frame_stride + (int64_t)qg * d.o_row_stride + h * D + ch * 8,
                     fz_ld_h8(Os + (wave * 32 * QB + row) * C::OSTR + ch * 8));
    }
}

template <int D, int W, int QB, bool USE_BIAS = true, int NSTG = 2, int NPOLY = 0>
static int launch_flash(const FzAttnSelfDesc& d, const void* q, const void* k, const void* vt, void* o, void* stream) {
    // 32-bit per-lane byte offsets inside one (frame, head) K / V^T panel
    if (d.k_row_stride >= (1 << 22) || (int64_t)D * d.vt_chan_stride >= (1ll << 30)) return FZ_ERR_UNSUPPORTED;
    const int nq = (d.lq + 128 * QB - 1) / (128 * QB);
    FlashOrder ord = {};
    const int groups = d.heads * d.n_frames;
    if ((groups & 7) == 0 && (groups >> 3) % d.n_frames == 0 && d.n_frames <= 64) {
        ord.heads_per_xcd = (groups >> 3) / d.n_frames;
        int pos = 0;
        for (int pass = 0; pass < 2; ++pass)  // full-length frames first, the single-source ones after them
            for (int fl = 0; fl < d.n_frames; ++fl) {
                const int n = d.frame0 + fl, b = n / d.clip_len, f = n % d.clip_len;
                bool one_source = d.n_kv > 1;
                for (int j = 1; j < d.n_kv; ++j) one_source = one_source && fl_kv_source(d, b, f, j) == fl_kv_source(d, b, f, 0);
                if (one_source == (pass == 1)) ord.fl[pos++] = (unsigned char)fl;
            }
    }
    dim3 grid(nq * d.heads * d.n_frames), block(256);
    FZ_LAUNCH((attn_flash_kernel<D, W, QB, USE_BIAS, NSTG, NPOLY>), grid, block, 0, stream, d, (const half_t*)q, (const half_t*)k,
              (const half_t*)vt, (half_t*)o, ord);
    return fz_last_launch_status();
}

#ifndef FZ_FLASH_NO_DISPATCH
// called by fz_attn_self (attn_self.hip) for mode == FZ_ATTN_FLASH
int fz_attn_flash_dispatch(const FzAttnSelfDesc& d, const void* q, const void* k, const void* vt, void* o, void* stream) {
    // Variant choice is fixed by measurements (profiles/r01_kbench_v0.json, profiles/r02_flash_ab.txt; the
    // 4-stage ring with a barrier every second tile measured 0.1-0.7 % SLOWER than the 2-stage one and is not instantiated):
    // two query blocks per wave at two waves per SIMD once there are >= 512 query rows, four waves per SIMD below.
    // (s_setprio 1 around the PV -- or the QK and PV -- MFMA clusters: 3 % SLOWER, 764 -> 741 TFLOP/s at 16 frames, same box;
    // profiles/r02_flash_ab.txt has the numbers.)
    const bool big = d.lq >= 512;  // two query blocks per wave only pay when there are enough rows to fill the chip
    switch (d.head_dim) {
        case 16: return launch_flash<16, 2, 1>(d, q, k, vt, o, stream);
        case 32: return launch_flash<32, 2, 1>(d, q, k, vt, o, stream);
        case 40:
            if (!d.q_log2_scaled || d.lkf < FKVBLK)  // q as to_q produces it (or a first tile with padded keys, see STASH_MASKS)
                return big ? launch_flash<40, 2, 2, false>(d, q, k, vt, o, stream)
                           : launch_flash<40, 4, 1, false>(d, q, k, vt, o, stream);
            return big ? launch_flash<40, 2, 2>(d, q, k, vt, o, stream) : launch_flash<40, 4, 1>(d, q, k, vt, o, stream);
        case 64: return launch_flash<64, 2, 1>(d, q, k, vt, o, stream);
        case 80: return launch_flash<80, 2, 1>(d, q, k, vt, o, stream);
        case 128: return launch_flash<128, 1, 1>(d, q, k, vt, o, stream);
        case 160: return launch_flash<160, 1, 1>(d, q, k, vt, o, stream);
        default: return FZ_ERR_UNSUPPORTED;
    }
}
#endif
