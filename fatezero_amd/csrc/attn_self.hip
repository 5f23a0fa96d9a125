// attn_self.hip -- the HBM-bound members of the self-attention family for gfx950: K2 capture (exact two-pass softmax,
// probability map written to the HBM arena) and K3 inject (rows take the stored map, or the live attention where the blend
// mask says so).  Replaces attention_register.py:131-218 + :23-59 of the reference (see include/fatezero_hip.h); the plain
// (no map) case is csrc/attn_flash.hip.
//
// Formulation ("everything transposed", so every softmax quantity is lane-local):
//   S^T[k,q] = K[k,:] . Q[q,:]        v_mfma_f32_32x32x16_f16, A = K tile (LDS), B = Q (registers)
//   O^T[d,q] = sum_k V^T[d,k] P^T[k,q]                          A = V^T tile (LDS), B = P (registers)
// In the 32x32 accumulator layout a lane owns one query column q = lane&31 and 16 rows of each 32-key sub-tile; the K tile
// is laid out in LDS (fz_krow_of_key) so that over both sub-tiles those are 32 CONSECUTIVE keys -- the B-operand k-slot order
// of the second MFMA and 64 contiguous bytes of the stored map at once: P moves between the softmax registers, the P.V MFMA
// and HBM without an LDS round trip, permlane or shuffle; the only cross-lane op per key tile is one lane-pair max.
// V arrives transposed from the projection GEMM ([channel][token]) so both LDS tiles are read with 16-byte
// ds_read_b128 at conflict-free strides (stride/16B odd).
//
// Work decomposition: 256 threads = 4 waves x 32 query rows; key tiles of 64, fetched one tile ahead (global -> registers
// before the MFMAs of the current tile, registers -> the other LDS stage after them, one barrier per tile); 1-D grid with an
// XCD-aware mapping (all query tiles of a (head, frame) pair land on one XCD so its K/V stay in that XCD's L2).
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"
#include <stdlib.h>

#define QBLK 128
#define KVBLK 64
#define VSTR 72

#ifndef FZ_MAP_STORE
#define FZ_MAP_STORE fz_st_h8_nt
#endif
#define PSTR 72 /* halves; 144 B = 9 x 16 B (odd) */

template <int D, int MODE>
struct SelfCfg {
    static_assert(D % 8 == 0, "head dim in 16-byte chunks");
    static constexpr int DP16 = (D + 15) / 16 * 16;
    static constexpr int NC = DP16 / 16;      // QK^T k-steps
    static constexpr int NT = (D + 31) / 32;  // O^T row tiles
    static constexpr int KSTR = DP16 + 8;     // halves; (DP16+8)/8 is odd
    static constexpr int DCH = D / 8;         // 16-byte data chunks per K row
    static constexpr int KCH = DP16 / 8;
    static constexpr int VROWS = NT * 32;
    static constexpr int OSTR = NT * 32 + 8;
    static constexpr int KS = KVBLK * KSTR;
    static constexpr int VS = VROWS * VSTR;
    static constexpr int STAGE = KS + VS;
    static constexpr int OS = QBLK * OSTR;
    // CAPTURE: a wave-private 32 x 64 staging tile turns the lanes' 16-byte pieces of 32 different rows into full 128-byte
    // row segments before they leave for HBM (a 16-byte write per lane is a 16-byte L2 request: 17 M requests per 268 MB map)
    static constexpr int PS = (MODE == FZ_ATTN_CAPTURE) ? 4 * 32 * PSTR : 0;
    static constexpr int MAIN = 2 * STAGE + PS;
    static constexpr int LDS_HALVES = MAIN > OS ? MAIN : OS;
    static constexpr int KLD = (KVBLK * DCH + 255) / 256;  // 16-byte K chunks per thread per tile
    static constexpr int VLD = (D * 8 + 255) / 256;        // 16-byte V^T chunks per thread per tile
};

// Key order inside a 64-key tile.  In the 32x32 accumulator of S^T = K Q^T a lane (query column lane & 31, half hi = lane >> 5)
// owns rows (r & 3) + 8 (r >> 2) + 4 hi, r = 0..15, of each 32-row sub-tile.  The K tile is laid out in LDS so that those are
// the keys 32 hi + 16 sub + r: over both sub-tiles a lane holds the 32 CONSECUTIVE keys [32 hi, 32 hi + 32) of its query row
// in register order.  That is at once (a) the B-operand k-slot order of the P.V MFMA (P goes from the softmax registers into
// the MFMA, the V^T fragment of step (sub, mm) is read at column 32 hi + 16 sub + 8 mm), and (b) 64 contiguous bytes of the
// stored probability map per lane: captured rows are written, and injected rows read, straight between registers and HBM
// with 16-byte accesses -- no LDS staging of P, no barrier for it.
FZ_DEVICE int fz_krow_of_key(int kk) {
    const int hi = kk >> 5, sub = (kk >> 4) & 1, r = kk & 15;
    return 32 * sub + (r & 3) + 8 * (r >> 2) + 4 * hi;
}

// ABL: timing ablations of scripts/self_ab.hip only (1: no pass 1, 2: no map stores, 4: no P.V); the library instantiates 0
template <int D, int MODE, int ABL = 0>
FZ_KERNEL void __launch_bounds__(256, (D <= 80 ? 2 : 1))  // two workgroups per SIMD set wherever the registers allow it
attn_self_kernel(FzAttnSelfDesc d, const half_t* __restrict__ q, const half_t* __restrict__ k,
                 const half_t* __restrict__ vt, half_t* __restrict__ o, half_t* __restrict__ p,
                 const float* __restrict__ row_mask) {
    typedef SelfCfg<D, MODE> C;
    FZ_SHARED __attribute__((aligned(16))) half_t smem[C::LDS_HALVES];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq_ = lane & 31, hi = lane >> 5;

    // ---- block -> (head, frame, query tile), XCD aware ------------------------------------------------
    const int nq = (d.lq + QBLK - 1) / QBLK;
    const int groups = d.heads * d.n_frames;
    int group, qt;
    {
        const int bid = blockIdx.x;
        if ((groups & 7) == 0) {
            const int xcd = bid & 7, idx = bid >> 3;
            group = xcd * (groups >> 3) + idx / nq;
            qt = idx % nq;
        } else {
            group = bid / nq;
            qt = bid % nq;
        }
    }
    const int h = group / d.n_frames, fl = group % d.n_frames;
    const int n = d.frame0 + fl, b = n / d.clip_len, f = n % d.clip_len;

    const int qrow = qt * QBLK + wave * 32 + lq_;
    const bool qvalid = qrow < d.lq;

    // ---- Q fragments (B operand of S^T = K Q^T): lane holds d = 16c + 8hi .. +7 of its own row --------
    half8_t qf[C::NC];
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
        const int dd = 16 * c + 8 * hi;
        qf[c] = (qvalid && dd < D)
                    ? fz_ld_h8(q + (int64_t)n * d.q_frame_stride + (int64_t)qrow * d.q_row_stride + h * D + dd)
                    : fz_zero_h8();
    }

    int src[FZ_MAX_KV_SLOTS];
#pragma unroll
    for (int j = 0; j < FZ_MAX_KV_SLOTS; ++j) {
        const int kvl = d.kv_clip_len ? d.kv_clip_len : d.clip_len;  // frames per batch element of k / vt
        int s = d.kv_abs[j] ? d.kv_val[j] : f + (d.kv_clip_len ? d.kv_frame_off : 0) + d.kv_val[j];
        s = s < 0 ? 0 : (s > kvl - 1 ? kvl - 1 : s);
        src[j] = b * kvl + s;
    }

    const int lkfp = (d.lkf + KVBLK - 1) / KVBLK * KVBLK;
    const int tps = lkfp / KVBLK;
    const int ntiles = d.n_kv * tps;
    const int64_t khs = d.k_head_stride ? d.k_head_stride : (int64_t)D;
    const float cs = d.q_log2_scaled ? 1.0f : d.scale * 1.4426950408889634f;  // softmax in the log2 domain

    bool use_cur = true;  // INJECT: does this lane's row keep the live attention?
    bool any_cur = true;  // kernel-uniform: is QK^T needed at all?
    if (MODE == FZ_ATTN_INJECT) {
        use_cur = false;
        if (row_mask != nullptr && qvalid)
            use_cur = row_mask[(int64_t)(fl + d.mask_frame_off) * d.lq + qrow] != 0.0f;
        any_cur = (row_mask != nullptr);
    }

    // ---- the stored map: this lane's 64-byte segment of its row inside the current 64-key tile ----------------------------
    // rows that do not take part (beyond lq; INJECT rows that keep the live attention) are pointed at the tile's first row,
    // so that the loads stay unconditional (a predicated load costs a vmcnt(0) at the join) and hit in L1
    const bool p_vec = ((d.lkf | d.p_row_stride | d.p_head_stride | d.p_frame_stride) & 7) == 0 &&
                       (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    const bool p_mine = qvalid && !(MODE == FZ_ATTN_INJECT && use_cur);
    half_t* prow = p + (int64_t)(fl + d.p_frame_off) * d.p_frame_stride + (int64_t)h * d.p_head_stride +
                   (int64_t)(p_mine ? qrow : qt * QBLK) * d.p_row_stride + 32 * hi;

    // ---- K / V^T tiles: global -> registers (prefetch, one tile ahead) -> LDS (2-stage ring, one barrier per tile) ---------
    half8_t kreg[C::KLD], vreg[C::VLD];
    uint32_t kkey[C::KLD], kdd[C::KLD], voff[C::VLD];
    int klds[C::KLD], vlds[C::VLD];
    const uint32_t krs_bytes = (uint32_t)d.k_row_stride * 2u;
#pragma unroll
    for (int i = 0; i < C::KLD; ++i) {
        int id = tid + 256 * i;
        id = id < KVBLK * C::DCH ? id : KVBLK * C::DCH - 1;  // surplus threads repeat the last chunk
        const int key = id / C::DCH, ch = id % C::DCH;
        kkey[i] = key;
        kdd[i] = 16u * ch;
        klds[i] = fz_krow_of_key(key) * C::KSTR + ch * 8;
    }
#pragma unroll
    for (int i = 0; i < C::VLD; ++i) {
        int id = tid + 256 * i;
        id = id < D * 8 ? id : D * 8 - 1;
        voff[i] = (uint32_t)(id >> 3) * (uint32_t)d.vt_chan_stride * 2u + (id & 7) * 16u;
        vlds[i] = C::KS + (id >> 3) * VSTR + (id & 7) * 8;
    }
    // constant parts of both stages: padding chunks of the K rows, padding rows of V^T
    for (int st = 0; st < 2; ++st) {
        constexpr int NPAD = C::KCH - C::DCH;
        for (int id = tid; id < KVBLK * NPAD; id += 256)
            fz_st_h8(smem + st * C::STAGE + (id / (NPAD > 0 ? NPAD : 1)) * C::KSTR + (C::DCH + id % (NPAD > 0 ? NPAD : 1)) * 8,
                     fz_zero_h8());
        for (int id = tid; id < (C::VROWS - D) * 8; id += 256)
            fz_st_h8(smem + st * C::STAGE + C::KS + (D + id / 8) * VSTR + (id % 8) * 8, fz_zero_h8());
    }
    auto fetch = [&](int kt, bool with_k, bool with_v) {
        const int j = kt / tps, r0 = (kt % tps) * KVBLK;
        if (with_k) {
            const char* kb = reinterpret_cast<const char*>(k + (int64_t)src[j] * d.k_frame_stride + (int64_t)h * khs) +
                             (int64_t)r0 * krs_bytes;
            const uint32_t kmax = (uint32_t)(d.lkf - 1 - r0);  // padded keys are masked in the softmax; any finite data will do
#pragma unroll
            for (int i = 0; i < C::KLD; ++i) {
                const uint32_t key = kkey[i] < kmax ? kkey[i] : kmax;
                kreg[i] = fz_ld_h8_off(kb, fz_mad24(key, krs_bytes, kdd[i]));
            }
        }
        if (with_v) {
            const char* vb = reinterpret_cast<const char*>(vt + (int64_t)src[j] * d.vt_frame_stride +
                                                           (int64_t)(h * D) * d.vt_chan_stride + r0);
#pragma unroll
            for (int i = 0; i < C::VLD; ++i) vreg[i] = fz_ld_h8_off(vb, voff[i]);
        }
    };
    auto stash = [&](int st, bool with_k, bool with_v) {
        half_t* Sb = smem + st * C::STAGE;
        if (with_k) {
#pragma unroll
            for (int i = 0; i < C::KLD; ++i) fz_st_h8(Sb + klds[i], kreg[i]);
        }
        if (with_v) {
#pragma unroll
            for (int i = 0; i < C::VLD; ++i) fz_st_h8(Sb + vlds[i], vreg[i]);
        }
    };
    // S^T of the tile in stage `st` (raw dot products; cs takes them to the log2 domain), padded keys at -inf:
    // s[16 sub + r] = score of key 32 hi + 16 sub + r
    auto scores = [&](int st, int r0, f32x2* s2) {
        float* s = reinterpret_cast<float*>(s2);
        const half_t* Ks = smem + st * C::STAGE;
        half8_t kfr[2][C::NC];  // every K fragment of the tile is requested before the first MFMA (LDS latency paid once)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int c = 0; c < C::NC; ++c) kfr[sub][c] = fz_ld_h8(Ks + (32 * sub + lq_) * C::KSTR + 8 * hi + 16 * c);
        f32x16 acc[2] = {fz_zero_f16v(), fz_zero_f16v()};
#pragma unroll
        for (int c = 0; c < C::NC; ++c)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) acc[sub] = fz_mfma_32x32x16_f16(kfr[sub][c], qf[c], acc[sub]);  // two independent chains
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[16 * sub + r] = acc[sub][r];
        if (r0 + KVBLK > d.lkf) {  // wave-uniform: ragged last tile of a kv slot
#pragma unroll
            for (int i = 0; i < 32; ++i)
                if (r0 + 32 * hi + i >= d.lkf) s[i] = -INFINITY;
        }
    };

    float m = -1e30f, l = 0.0f;
    f32x16 oacc[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) oacc[t] = fz_zero_f16v();

    // ---- CAPTURE pass 1: exact row max and sum (K tiles only) ------------------------------------------------
    if (MODE == FZ_ATTN_CAPTURE && !(ABL & 1)) {
        fetch(0, true, false);
        stash(0, true, false);
        __syncthreads();
        for (int kt = 0; kt < ntiles; ++kt) {
            if (kt + 1 < ntiles) fetch(kt + 1, true, false);
            f32x2 s2[16];  // score pairs: the softmax arithmetic runs on v_pk_* (two scores per VALU instruction)
            scores(kt & 1, (kt % tps) * KVBLK, s2);
            f32x2 mx = s2[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = __builtin_elementwise_max(mx, s2[i]);
            const float tmax = fz_pair_max32(fmaxf(mx[0], mx[1])) * cs;  // cs > 0
            const float mn = fmaxf(m, tmax);
            const f32x2 cs2 = {cs, cs}, nm2 = {-mn, -mn};
            f32x2 sum2 = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x2 t = s2[i] * cs2 + nm2;  // one packed fma + two exps per pair
                const f32x2 e = {fz_exp2(t[0]), fz_exp2(t[1])};
                sum2 += e;
            }
            l = l * fz_exp2(m - mn) + sum2[0] + sum2[1];
            m = mn;
            if (kt + 1 < ntiles) stash((kt + 1) & 1, true, false);
            __syncthreads();
        }
        l += fz_shfl_xor(l, 32);
        m += log2f(l);  // P = 2^(s - m) / l = 2^(s - (m + log2 l)): the normalisation rides in the exponent
    }

    // ---- main pass --------------------------------------------------------------------------------------
    const bool need_k = (MODE != FZ_ATTN_INJECT) || any_cur;
    half8_t pst[4];  // INJECT: the stored segment of the NEXT tile, in flight during the current one
    auto fetch_p = [&](int kt) {
        const int j = kt / tps, r0 = (kt % tps) * KVBLK;
        const half_t* sp = prow + (int64_t)j * d.lkf + r0;
        if (p_vec) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                // a chunk is either entirely inside the row or entirely padding (lkf % 8 == 0): clamp, zero below
                const int rr = r0 + 32 * hi + 8 * c;
                pst[c] = fz_ld_h8(rr < d.lkf ? sp + 8 * c : sp - (r0 + 32 * hi));
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int rr = r0 + 32 * hi + 8 * c + e;
                    pst[c][e] = rr < d.lkf ? sp[8 * c + e] : (half_t)0.0f;
                }
        }
    };
    fetch(0, need_k, true);
    if (MODE == FZ_ATTN_INJECT) fetch_p(0);
    stash(0, need_k, true);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int j = kt / tps, r0 = (kt % tps) * KVBLK;
        const int st = kt & 1;
        half8_t pf[4];  // P^T B operands of the tile: chunk c = 2 sub + mm holds keys 32 hi + 8 c .. + 7
        if (MODE == FZ_ATTN_INJECT) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                pf[c] = pst[c];
                if (p_vec && r0 + KVBLK > d.lkf && r0 + 32 * hi + 8 * c >= d.lkf) pf[c] = fz_zero_h8();
            }
        }
        if (kt + 1 < ntiles) {
            fetch(kt + 1, need_k, true);
            if (MODE == FZ_ATTN_INJECT) fetch_p(kt + 1);
        }
        float alpha = 1.0f;
        if (need_k) {
            f32x2 s2[16];
            scores(st, r0, s2);
            const f32x2 cs2 = {cs, cs};
            if (MODE == FZ_ATTN_CAPTURE) {
                const f32x2 nm2 = {-m, -m};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x2 t = s2[i] * cs2 + nm2;
                    s2[i] = f32x2{fz_exp2(t[0]), fz_exp2(t[1])};  // final, normalised P
                }
            } else {
                f32x2 mx = s2[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) mx = __builtin_elementwise_max(mx, s2[i]);
                const float tmax = fz_pair_max32(fmaxf(mx[0], mx[1])) * cs;
                const float mn = fmaxf(m, tmax);
                alpha = use_cur ? fz_exp2(m - mn) : 1.0f;
                m = mn;
                const f32x2 nm2 = {-mn, -mn};
                f32x2 sum2 = {0.0f, 0.0f};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x2 t = s2[i] * cs2 + nm2;
                    s2[i] = f32x2{fz_exp2(t[0]), fz_exp2(t[1])};
                    sum2 += s2[i];
                }
                l = l * alpha + sum2[0] + sum2[1];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                half8_t pc;
#pragma unroll
                for (int e = 0; e < 8; ++e) pc[e] = (half_t)s2[4 * c + e / 2][e & 1];
                if (MODE == FZ_ATTN_CAPTURE || use_cur) pf[c] = pc;  // per-lane select: lane <-> query row
            }
        }
        if (MODE == FZ_ATTN_CAPTURE && !(ABL & 2)) {
            if (p_vec) {
                // registers -> the wave's own staging tile now; staging tile -> HBM at the END of the step (see below)
                half_t* Pw = smem + 2 * C::STAGE + wave * 32 * PSTR;
#pragma unroll
                for (int c = 0; c < 4; ++c) fz_st_h8(Pw + lq_ * PSTR + 32 * hi + 8 * c, pf[c]);
            } else if (qvalid) {
                half_t* dp = prow + (int64_t)j * d.lkf + r0;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (r0 + 32 * hi + 8 * c + e < d.lkf) dp[8 * c + e] = pf[c][e];
            }
        }
        if (MODE == FZ_ATTN_INJECT && any_cur) {
#pragma unroll
            for (int t = 0; t < C::NT; ++t) oacc[t] *= alpha;
        }
        const half_t* Vs = smem + st * C::STAGE + C::KS;
#pragma unroll
        for (int cp = 0; cp < 4; cp += 2) {  // V^T fragments of two key chunks at a time, requested before their MFMAs
            half8_t vfr[2][C::NT];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < C::NT; ++t) vfr[c][t] = fz_ld_h8(Vs + (32 * t + lq_) * VSTR + 32 * hi + 8 * (cp + c));
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int t = 0; t < C::NT; ++t) {
                    if (ABL & 4) oacc[t][0] += (float)pf[cp + c][t & 7] + (float)vfr[c][t][0];
                    else oacc[t] = fz_mfma_32x32x16_f16(vfr[c][t], pf[cp + c], oacc[t]);
                }
        }
        if (kt + 1 < ntiles) stash(st ^ 1, need_k, true);
        if (MODE == FZ_ATTN_CAPTURE && !(ABL & 2) && p_vec) {
            // The map stores go out AFTER the stash: stores count in vmcnt like loads, so the stash's wait for the prefetched
            // tile would otherwise also wait for this tile's stores to be acknowledged (measured: the stores were fully
            // exposed, 40 us of a 136 us launch).  Issued here they have the whole next step to drain.  Full 128-byte row
            // segments: 8 lanes per row.
            const half_t* Pw = smem + 2 * C::STAGE + wave * 32 * PSTR;
            fz_wave_lds_sync();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + (lane >> 3), ch = lane & 7;
                const int qg = qt * QBLK + wave * 32 + row;
                if (qg < d.lq && r0 + 8 * ch < d.lkf)
                    FZ_MAP_STORE(p + (int64_t)(fl + d.p_frame_off) * d.p_frame_stride + (int64_t)h * d.p_head_stride +
                                 (int64_t)qg * d.p_row_stride + (int64_t)j * d.lkf + r0 + 8 * ch,
                             fz_ld_h8(Pw + row * PSTR + 8 * ch));
            }
            fz_wave_lds_sync();  // the staging tile is free again for the next step's writes (same wave)
        }
        __syncthreads();
    }

    // ---- epilogue: normalise, stage O^T through LDS, store whole head-rows -----------------------------
    float fin = 1.0f;
    if (MODE == FZ_ATTN_INJECT && any_cur) {
        l += fz_shfl_xor(l, 32);
        if (use_cur) fin = 1.0f / l;
    }
    half_t* Os = smem;
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (half_t)(oacc[t][4 * g + e] * fin);
            *reinterpret_cast<half4_t*>(Os + (wave * 32 + lq_) * C::OSTR + 32 * t + 8 * g + 4 * hi) = v;
        }
    __syncthreads();
    constexpr int OCH = D / 8;
    for (int id = lane; id < 32 * OCH; id += 64) {
        const int row = id / OCH, ch = id % OCH;
        const int qg = qt * QBLK + wave * 32 + row;
        if (qg < d.lq)
            fz_st_h8(o + (int64_t)n * d.o_frame_stride + (int64_t)qg * d.o_row_stride + h * D + ch * 8,
                     fz_ld_h8(Os + (wave * 32 + row) * C::OSTR + ch * 8));
    }
}

template <int D>
static int launch_self(const FzAttnSelfDesc& d, const void* q, const void* k, const void* vt, void* o, void* p,
                       const float* row_mask, void* stream) {
    const int nq = (d.lq + QBLK - 1) / QBLK;
    dim3 grid(nq * d.heads * d.n_frames), block(256);
    const half_t* q_ = (const half_t*)q;
    const half_t* k_ = (const half_t*)k;
    const half_t* vt_ = (const half_t*)vt;
    half_t* o_ = (half_t*)o;
    half_t* p_ = (half_t*)p;
    if (d.k_row_stride >= (1 << 22) || (int64_t)D * d.vt_chan_stride >= (1ll << 30)) return FZ_ERR_UNSUPPORTED;  // 32-bit lane offsets
    switch (d.mode) {
        case FZ_ATTN_CAPTURE:
            FZ_LAUNCH((attn_self_kernel<D, FZ_ATTN_CAPTURE>), grid, block, 0, stream, d, q_, k_, vt_, o_, p_, row_mask);
            break;
        case FZ_ATTN_INJECT:
            FZ_LAUNCH((attn_self_kernel<D, FZ_ATTN_INJECT>), grid, block, 0, stream, d, q_, k_, vt_, o_, p_, row_mask);
            break;
        default:
            return FZ_ERR_BAD_ARG;
    }
    return fz_last_launch_status();
}

#ifndef FZ_SELF_NO_ENTRY
int fz_attn_flash_dispatch(const FzAttnSelfDesc& d, const void* q, const void* k, const void* vt, void* o, void* stream);

extern "C" int fz_attn_self(const FzAttnSelfDesc* desc, const void* q, const void* k, const void* vt, void* o,
                            void* p, const float* row_mask, void* stream) {
    if (!desc || !q || !vt || !o) return FZ_ERR_BAD_ARG;
    const FzAttnSelfDesc& d = *desc;
    if (d.n_frames <= 0 || d.lq <= 0 || d.lkf <= 0 || d.n_kv < 1 || d.n_kv > FZ_MAX_KV_SLOTS) return FZ_ERR_BAD_ARG;
    if (d.mode != FZ_ATTN_FLASH && !p) return FZ_ERR_BAD_ARG;
    if (!(d.mode == FZ_ATTN_INJECT && row_mask == nullptr) && !k) return FZ_ERR_BAD_ARG;
    if ((d.q_row_stride | d.k_row_stride | d.vt_chan_stride | d.o_row_stride | d.q_frame_stride | d.k_frame_stride |
         d.vt_frame_stride | d.o_frame_stride) & 7)
        return FZ_ERR_BAD_ARG;  // 16-byte vector access
    if (d.mode == FZ_ATTN_FLASH) return fz_attn_flash_dispatch(d, q, k, vt, o, stream);
    switch (d.head_dim) {
        case 16: return launch_self<16>(d, q, k, vt, o, p, row_mask, stream);
        case 32: return launch_self<32>(d, q, k, vt, o, p, row_mask, stream);
        case 40: return launch_self<40>(d, q, k, vt, o, p, row_mask, stream);
        case 64: return launch_self<64>(d, q, k, vt, o, p, row_mask, stream);
        case 80: return launch_self<80>(d, q, k, vt, o, p, row_mask, stream);
        case 128: return launch_self<128>(d, q, k, vt, o, p, row_mask, stream);
        case 160: return launch_self<160>(d, q, k, vt, o, p, row_mask, stream);
        default: return FZ_ERR_UNSUPPORTED;
    }
}
#endif
