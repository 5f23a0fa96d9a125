// attn_self.hip -- fused sparse-causal spatio-temporal self-attention for gfx950 (K1 flash / K2 capture /
// K3 inject).  Replaces attention_register.py:131-218 + :23-59 of the reference (see include/fatezero_hip.h).
//
// Formulation ("everything transposed", so every softmax quantity is lane-local):
//   S^T[k,q] = K[k,:] . Q[q,:]        v_mfma_f32_32x32x16_f16, A = K tile (LDS), B = Q (registers)
//   O^T[d,q] = sum_k V^T[d,k] P^T[k,q]                          A = V^T tile (LDS), B = P (registers)
// In the 32x32 accumulator layout a lane owns one query column q = lane&31 and 16 rows; with the K-tile rows
// permuted by pi() those 16 rows are two runs of 8 *consecutive* keys, which is exactly the B-operand k-slot
// layout of the second MFMA -- so P goes from the softmax registers into the PV MFMA with no LDS round
// trip, no permlane and no shuffles; the only cross-lane op per key tile is one xor-32 shuffle of the row max.
// V arrives transposed from the projection GEMM ([channel][token]) so both LDS tiles are read with 16-byte
// ds_read_b128 at conflict-free strides (stride/16B odd).
//
// Work decomposition: 256 threads = 4 waves x 32 query rows; key tiles of 64; 1-D grid with an XCD-aware
// mapping (all query tiles of a (head, frame) pair land on one XCD so its K/V stay in that XCD's L2).
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"
#include <stdlib.h>

#define QBLK 128
#define KVBLK 64
#define PSTR 72 /* halves; 144 B = 9 x 16 B (odd) */
#define VSTR 72

template <int D, int MODE>
struct SelfCfg {
    static constexpr int DP16 = (D + 15) / 16 * 16;
    static constexpr int NC = DP16 / 16;      // QK^T k-steps
    static constexpr int NT = (D + 31) / 32;  // O^T row tiles
    static constexpr int KSTR = DP16 + 8;     // halves; (DP16+8)/8 is odd
    static constexpr int KCH = DP16 / 8;      // 16-byte chunks per K row
    static constexpr int VROWS = NT * 32;
    static constexpr int OSTR = NT * 32 + 8;
    static constexpr int KS_HALVES = KVBLK * KSTR;
    static constexpr int VS_HALVES = VROWS * VSTR;
    static constexpr int PS_HALVES = (MODE == FZ_ATTN_FLASH) ? 0 : QBLK * PSTR;  // P staging tile
    static constexpr int OS_HALVES = QBLK * OSTR;
    static constexpr int MAIN_HALVES = KS_HALVES + VS_HALVES + PS_HALVES;
    static constexpr int LDS_HALVES = MAIN_HALVES > OS_HALVES ? MAIN_HALVES : OS_HALVES;
};

// K-tile row permutation: MFMA A-row i (0..31) holds key pi(i) of the 32-key sub-tile.
FZ_DEVICE int fz_pi(int i) {
    const int a = i >> 3, hp = (i >> 2) & 1, t = i & 3;
    return ((a & 2) << 3) + 8 * hp + 4 * (a & 1) + t;
}

template <int D, int MODE>
FZ_KERNEL void __launch_bounds__(256)
attn_self_kernel(FzAttnSelfDesc d, const half_t* __restrict__ q, const half_t* __restrict__ k,
                 const half_t* __restrict__ vt, half_t* __restrict__ o, half_t* __restrict__ p,
                 const float* __restrict__ row_mask) {
    typedef SelfCfg<D, MODE> C;
    FZ_SHARED __attribute__((aligned(16))) half_t smem[C::LDS_HALVES];
    half_t* Ks = smem;
    half_t* Vs = smem + C::KS_HALVES;
    half_t* Ps = smem + C::KS_HALVES + C::VS_HALVES;

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
    bool any_cur = true;  // wave-uniform: is QK^T needed at all?
    if (MODE == FZ_ATTN_INJECT) {
        use_cur = false;
        if (row_mask != nullptr && qvalid)
            use_cur = row_mask[(int64_t)(fl + d.mask_frame_off) * d.lq + qrow] != 0.0f;
        any_cur = (row_mask != nullptr);
    }

    // ---- tile loaders ----------------------------------------------------------------------------------
    auto load_k = [&](int kt) {
        const int j = kt / tps, r0 = (kt % tps) * KVBLK;
        const half_t* base = k + (int64_t)src[j] * d.k_frame_stride + (int64_t)h * khs;
        for (int id = tid; id < KVBLK * C::KCH; id += 256) {
            const int key = id / C::KCH, ch = id % C::KCH;
            const int r = r0 + key, dd = ch * 8;
            half8_t v = (r < d.lkf && dd < D) ? fz_ld_h8(base + (int64_t)r * d.k_row_stride + dd) : fz_zero_h8();
            fz_st_h8(Ks + key * C::KSTR + dd, v);
        }
    };
    auto load_v = [&](int kt) {
        const int j = kt / tps, r0 = (kt % tps) * KVBLK;
        const half_t* base = vt + (int64_t)src[j] * d.vt_frame_stride + (int64_t)(h * D) * d.vt_chan_stride + r0;
        for (int id = tid; id < C::VROWS * 8; id += 256) {
            const int row = id >> 3, ch = id & 7;
            half8_t v = (row < D) ? fz_ld_h8(base + (int64_t)row * d.vt_chan_stride + ch * 8) : fz_zero_h8();
            fz_st_h8(Vs + row * VSTR + ch * 8, v);
        }
    };
    // S^T sub-tile: acc[r] = S[q = lane&31][key = 32*sub + (r<8 ? 8hi+r : 16+8hi+r-8)]
    auto qk_sub = [&](int sub) -> f32x16 {
        f32x16 acc = fz_zero_f16v();
        const half_t* row = Ks + (32 * sub + fz_pi(lq_)) * C::KSTR + 8 * hi;
#pragma unroll
        for (int c = 0; c < C::NC; ++c) acc = fz_mfma_32x32x16_f16(fz_ld_h8(row + 16 * c), qf[c], acc);
        return acc;
    };
    auto key_of = [&](int sub, int r) -> int { return 32 * sub + (r < 8 ? 8 * hi + r : 8 + 8 * hi + r); };

    float m = -1e30f, l = 0.0f, inv_l = 1.0f;
    f32x16 oacc[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) oacc[t] = fz_zero_f16v();

    // ---- CAPTURE pass 1: exact row max and sum ---------------------------------------------------------
    if (MODE == FZ_ATTN_CAPTURE) {
        for (int kt = 0; kt < ntiles; ++kt) {
            const int r0 = (kt % tps) * KVBLK;
            __syncthreads();
            load_k(kt);
            __syncthreads();
            float s[32];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 acc = qk_sub(sub);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s[16 * sub + r] = (r0 + key_of(sub, r) < d.lkf) ? acc[r] * cs : -INFINITY;
            }
            float tmax = s[0];
#pragma unroll
            for (int i = 1; i < 32; ++i) tmax = fmaxf(tmax, s[i]);
            tmax = fmaxf(tmax, fz_shfl_xor(tmax, 32));
            const float mn = fmaxf(m, tmax);
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 32; ++i) sum += fz_exp2(s[i] - mn);
            l = l * fz_exp2(m - mn) + sum;
            m = mn;
        }
        l += fz_shfl_xor(l, 32);
        inv_l = 1.0f / l;
    }

    // ---- main pass --------------------------------------------------------------------------------------
    for (int kt = 0; kt < ntiles; ++kt) {
        const int j = kt / tps, r0 = (kt % tps) * KVBLK;
        __syncthreads();
        if (MODE != FZ_ATTN_INJECT || any_cur) load_k(kt);
        load_v(kt);
        if (MODE == FZ_ATTN_INJECT) {
            // stored map tile -> LDS, coalesced 128-byte row segments; rows that keep the live attention are skipped
            for (int id = tid; id < QBLK * 8; id += 256) {
                const int row = id >> 3, ch = id & 7;
                const int qg = qt * QBLK + row;
                bool need = qg < d.lq;
                if (need && row_mask != nullptr)
                    need = row_mask[(int64_t)(fl + d.mask_frame_off) * d.lq + qg] == 0.0f;
                half8_t v = fz_zero_h8();
                if (need) {
                    const int rr = r0 + ch * 8;
                    const half_t* src_p = p + (int64_t)(fl + d.p_frame_off) * d.p_frame_stride +
                                          (int64_t)h * d.p_head_stride + (int64_t)qg * d.p_row_stride +
                                          (int64_t)j * d.lkf + rr;
                    if (rr + 8 <= d.lkf && ((d.lkf | d.p_row_stride) & 7) == 0) {
                        v = fz_ld_h8(src_p);
                    } else {
                        for (int e = 0; e < 8; ++e) v[e] = (rr + e < d.lkf) ? src_p[e] : (half_t)0.0f;
                    }
                }
                fz_st_h8(Ps + row * PSTR + ch * 8, v);
            }
        }
        __syncthreads();

        half8_t pf[2][2];
        float alpha = 1.0f;
        if (MODE != FZ_ATTN_INJECT || any_cur) {
            float s[32];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 acc = qk_sub(sub);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s[16 * sub + r] = (r0 + key_of(sub, r) < d.lkf) ? acc[r] * cs : -INFINITY;
            }
            if (MODE == FZ_ATTN_CAPTURE) {
#pragma unroll
                for (int i = 0; i < 32; ++i) s[i] = fz_exp2(s[i] - m) * inv_l;  // final, normalised P
            } else {
                float tmax = s[0];
#pragma unroll
                for (int i = 1; i < 32; ++i) tmax = fmaxf(tmax, s[i]);
                tmax = fmaxf(tmax, fz_shfl_xor(tmax, 32));
                const float mn = fmaxf(m, tmax);
                alpha = fz_exp2(m - mn);
                m = mn;
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    s[i] = fz_exp2(s[i] - mn);
                    sum += s[i];
                }
                l = l * alpha + sum;
            }
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[sub][mm][e] = (half_t)s[16 * sub + 8 * mm + e];
        }
        if (MODE == FZ_ATTN_INJECT && !use_cur) {
            alpha = 1.0f;
            const half_t* row = Ps + (wave * 32 + lq_) * PSTR + 8 * hi;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) pf[sub][mm] = fz_ld_h8(row + 32 * sub + 16 * mm);
        }
        if (MODE == FZ_ATTN_CAPTURE) {
            // stage the wave's 32x64 P tile in LDS, then write it out as full 128-byte row segments
            half_t* row = Ps + (wave * 32 + lq_) * PSTR + 8 * hi;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) fz_st_h8(row + 32 * sub + 16 * mm, pf[sub][mm]);
        }

        if (MODE != FZ_ATTN_CAPTURE) {
#pragma unroll
            for (int t = 0; t < C::NT; ++t) oacc[t] *= alpha;
        }
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const half_t* vrow = Vs + (32 * t + lq_) * VSTR + 8 * hi;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
                    oacc[t] = fz_mfma_32x32x16_f16(fz_ld_h8(vrow + 32 * sub + 16 * mm), pf[sub][mm], oacc[t]);
        }

        if (MODE == FZ_ATTN_CAPTURE) {
            __syncthreads();
            for (int id = lane; id < 32 * 8; id += 64) {
                const int row = id >> 3, ch = id & 7;
                const int qg = qt * QBLK + wave * 32 + row;
                const int rr = r0 + ch * 8;
                if (qg < d.lq && rr < d.lkf) {
                    half8_t v = fz_ld_h8(Ps + (wave * 32 + row) * PSTR + ch * 8);
                    half_t* dst = p + (int64_t)(fl + d.p_frame_off) * d.p_frame_stride + (int64_t)h * d.p_head_stride +
                                  (int64_t)qg * d.p_row_stride + (int64_t)j * d.lkf + rr;
                    if (rr + 8 <= d.lkf && ((d.lkf | d.p_row_stride) & 7) == 0) {
                        fz_st_h8(dst, v);
                    } else {
                        for (int e = 0; e < 8; ++e)
                            if (rr + e < d.lkf) dst[e] = v[e];
                    }
                }
            }
        }
    }

    // ---- epilogue: normalise, stage O^T through LDS, store whole head-rows -----------------------------
    float fin = 1.0f;
    if (MODE == FZ_ATTN_FLASH || (MODE == FZ_ATTN_INJECT && any_cur)) {
        l += fz_shfl_xor(l, 32);
        if (MODE == FZ_ATTN_FLASH || use_cur) fin = 1.0f / l;
    }
    __syncthreads();
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
    switch (d.mode) {
        case FZ_ATTN_FLASH:
            FZ_LAUNCH((attn_self_kernel<D, FZ_ATTN_FLASH>), grid, block, 0, stream, d, q_, k_, vt_, o_, p_, row_mask);
            break;
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
