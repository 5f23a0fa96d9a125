// attn_cross.hip -- cross-attention (Lk <= 96, i.e. the 77 CLIP tokens) with per-step map capture and the fused
// prompt-to-prompt edit.  Replaces attention_register.py:71-128 + :23-59 and, for the conditional half of the
// edit pass, AttentionControlEdit.forward / replace_cross_attention (attention_util.py:129-132, :213-223,
// :243-253, :282-286) of the reference.
//
// Same transposed MFMA formulation as attn_self.hip (lane <-> query row, registers <-> keys), but the whole key
// axis (3 sub-tiles of 32) lives in registers, so the softmax is exact in one pass and the edit
//     new = (base @ M) * A + cur * B
// is a register-wise FMA: (base @ M)^T = M^T base^T is one more MFMA chain whose A operand is the (padded,
// transposed) 96x96 mapper in LDS and whose B operand is the lane's own row of the stored inversion map, read
// straight from the HBM arena with 16-byte loads (arena rows are padded to 80 halves for that purpose).
// LDS is one region reused for K -> mapper -> V^T -> O staging.
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"

#define XQBLK 128
#define XKEYS 96
#define XVSTR 104 /* halves: 208 B = 13 x 16 B (odd) */
#define XMSTR 104

template <int D>
struct CrossCfg {
    static constexpr int DP16 = (D + 15) / 16 * 16;
    static constexpr int NC = DP16 / 16;
    static constexpr int NT = (D + 31) / 32;
    static constexpr int KSTR = DP16 + 8;
    static constexpr int KCH = DP16 / 8;
    static constexpr int VROWS = NT * 32;
    static constexpr int OSTR = NT * 32 + 8;
    static constexpr int KS = XKEYS * KSTR;
    static constexpr int VS = VROWS * XVSTR;
    static constexpr int MS = XKEYS * XMSTR;
    static constexpr int OS = XQBLK * OSTR;
    static constexpr int M1 = KS > VS ? KS : VS;
    static constexpr int M2 = MS > OS ? MS : OS;
    static constexpr int LDS_HALVES = M1 > M2 ? M1 : M2;
};

FZ_DEVICE int fz_pi_x(int i) {
    const int a = i >> 3, hp = (i >> 2) & 1, t = i & 3;
    return ((a & 2) << 3) + 8 * hp + 4 * (a & 1) + t;
}

template <int D, int MODE>
FZ_KERNEL void __launch_bounds__(256)
attn_cross_kernel(FzAttnCrossDesc d, const half_t* __restrict__ q, const half_t* __restrict__ k,
                  const half_t* __restrict__ vt, half_t* __restrict__ o, half_t* __restrict__ p,
                  const half_t* __restrict__ mapper_t, const float* __restrict__ coef, half_t* __restrict__ cur_out) {
    typedef CrossCfg<D> C;
    FZ_SHARED __attribute__((aligned(16))) half_t smem[C::LDS_HALVES];
    FZ_SHARED float coefs[2 * XKEYS];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lq_ = lane & 31, hi = lane >> 5;
    const int nq = (d.lq + XQBLK - 1) / XQBLK;
    const int bid = blockIdx.x;
    const int qt = bid % nq, group = bid / nq;
    const int h = group % d.heads, fl = group / d.heads;
    const int n = d.frame0 + fl, b = n / d.clip_len;
    const int qrow = qt * XQBLK + wave * 32 + lq_;
    const bool qvalid = qrow < d.lq;
    const float cs = d.scale * 1.4426950408889634f;

    half8_t qf[C::NC];
#pragma unroll
    for (int c = 0; c < C::NC; ++c) {
        const int dd = 16 * c + 8 * hi;
        qf[c] = (qvalid && dd < D)
                    ? fz_ld_h8(q + (int64_t)n * d.q_frame_stride + (int64_t)qrow * d.q_row_stride + h * D + dd)
                    : fz_zero_h8();
    }
    auto key_of = [&](int sub, int r) -> int { return 32 * sub + (r < 8 ? 8 * hi + r : 8 + 8 * hi + r); };

    // ---- stage K, S^T = K Q^T ---------------------------------------------------------------------------
    {
        const half_t* base = k + (int64_t)b * d.k_batch_stride + h * D;
        for (int id = tid; id < XKEYS * C::KCH; id += 256) {
            const int key = id / C::KCH, ch = id % C::KCH, dd = ch * 8;
            half8_t v = (key < d.lk && dd < D) ? fz_ld_h8(base + (int64_t)key * d.k_row_stride + dd) : fz_zero_h8();
            fz_st_h8(smem + key * C::KSTR + dd, v);
        }
        if (MODE == FZ_ATTN_INJECT && tid < 2 * XKEYS) coefs[tid] = coef[tid];
    }
    __syncthreads();
    float s[48];
#pragma unroll
    for (int sub = 0; sub < 3; ++sub) {
        f32x16 acc = fz_zero_f16v();
        const half_t* row = smem + (32 * sub + fz_pi_x(lq_)) * C::KSTR + 8 * hi;
#pragma unroll
        for (int c = 0; c < C::NC; ++c) acc = fz_mfma_32x32x16_f16(fz_ld_h8(row + 16 * c), qf[c], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) s[16 * sub + r] = (key_of(sub, r) < d.lk) ? acc[r] * cs : -INFINITY;
    }
    // exact softmax over the lane pair (lane, lane^32)
    float mx = s[0];
#pragma unroll
    for (int i = 1; i < 48; ++i) mx = fmaxf(mx, s[i]);
    mx = fmaxf(mx, fz_shfl_xor(mx, 32));
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 48; ++i) {
        s[i] = fz_exp2(s[i] - mx);
        sum += s[i];
    }
    sum += fz_shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < 48; ++i) s[i] *= inv;

    // map store helper: lane's row, 16-byte chunks at columns 32*sub + 16*mm + 8*hi (< 80)
    auto store_map = [&](half_t* dst_base, const float* vals) {
        if (!qvalid) return;
        half_t* dst = dst_base + (int64_t)(fl + d.p_frame_off) * d.p_frame_stride + (int64_t)h * d.p_head_stride +
                      (int64_t)qrow * d.p_row_stride;
#pragma unroll
        for (int sub = 0; sub < 3; ++sub)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int col = 32 * sub + 16 * mm + 8 * hi;
                if (col < FZ_CROSS_P_STRIDE) {
                    half8_t v;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (half_t)vals[16 * sub + 8 * mm + e];
                    fz_st_h8(dst + col, v);
                }
            }
    };
    if (MODE == FZ_ATTN_CAPTURE) store_map(p, s);
    if (MODE == FZ_ATTN_INJECT && d.store_cur && cur_out != nullptr) store_map(cur_out, s);

    // ---- INJECT: new = (base @ M) * A + cur * B ----------------------------------------------------------
    if (MODE == FZ_ATTN_INJECT) {
        __syncthreads();
        for (int id = tid; id < XKEYS * (XKEYS / 8); id += 256) {
            const int row = id / (XKEYS / 8), ch = id % (XKEYS / 8);
            fz_st_h8(smem + row * XMSTR + ch * 8, fz_ld_h8(mapper_t + row * XKEYS + ch * 8));
        }
        __syncthreads();
        half8_t bf[5];
        const half_t* brow = p + (int64_t)(fl + d.p_frame_off) * d.p_frame_stride + (int64_t)h * d.p_head_stride +
                             (int64_t)qrow * d.p_row_stride + 8 * hi;
#pragma unroll
        for (int c = 0; c < 5; ++c) bf[c] = qvalid ? fz_ld_h8(brow + 16 * c) : fz_zero_h8();
#pragma unroll
        for (int sub = 0; sub < 3; ++sub) {
            f32x16 acc = fz_zero_f16v();
            const half_t* mrow = smem + (32 * sub + fz_pi_x(lq_)) * XMSTR + 8 * hi;
#pragma unroll
            for (int c = 0; c < 5; ++c) acc = fz_mfma_32x32x16_f16(fz_ld_h8(mrow + 16 * c), bf[c], acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nkey = key_of(sub, r);
                s[16 * sub + r] = acc[r] * coefs[nkey] + s[16 * sub + r] * coefs[XKEYS + nkey];
            }
        }
    }

    half8_t pf[3][2];
#pragma unroll
    for (int sub = 0; sub < 3; ++sub)
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[sub][mm][e] = (half_t)s[16 * sub + 8 * mm + e];

    // ---- stage V^T, O^T = V^T P^T ------------------------------------------------------------------------
    __syncthreads();
    {
        const half_t* base = vt + (int64_t)b * d.vt_batch_stride + (int64_t)(h * D) * d.vt_chan_stride;
        for (int id = tid; id < C::VROWS * (XKEYS / 8); id += 256) {
            const int row = id / (XKEYS / 8), ch = id % (XKEYS / 8);
            half8_t v = (row < D) ? fz_ld_h8(base + (int64_t)row * d.vt_chan_stride + ch * 8) : fz_zero_h8();
            fz_st_h8(smem + row * XVSTR + ch * 8, v);
        }
    }
    __syncthreads();
    f32x16 oacc[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) {
        oacc[t] = fz_zero_f16v();
        const half_t* vrow = smem + (32 * t + lq_) * XVSTR + 8 * hi;
#pragma unroll
        for (int sub = 0; sub < 3; ++sub)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
                oacc[t] = fz_mfma_32x32x16_f16(fz_ld_h8(vrow + 32 * sub + 16 * mm), pf[sub][mm], oacc[t]);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (half_t)oacc[t][4 * g + e];
            *reinterpret_cast<half4_t*>(smem + (wave * 32 + lq_) * C::OSTR + 32 * t + 8 * g + 4 * hi) = v;
        }
    __syncthreads();
    constexpr int OCH = D / 8;
    for (int id = lane; id < 32 * OCH; id += 64) {
        const int row = id / OCH, ch = id % OCH;
        const int qg = qt * XQBLK + wave * 32 + row;
        if (qg < d.lq)
            fz_st_h8(o + (int64_t)n * d.o_frame_stride + (int64_t)qg * d.o_row_stride + h * D + ch * 8,
                     fz_ld_h8(smem + (wave * 32 + row) * C::OSTR + ch * 8));
    }
}

template <int D>
static int launch_cross(const FzAttnCrossDesc& d, const void* q, const void* k, const void* vt, void* o, void* p,
                        const void* mapper_t, const float* coef, void* cur_out, void* stream) {
    const int nq = (d.lq + XQBLK - 1) / XQBLK;
    dim3 grid(nq * d.heads * d.n_frames), block(256);
    const half_t* q_ = (const half_t*)q;
    const half_t* k_ = (const half_t*)k;
    const half_t* vt_ = (const half_t*)vt;
    half_t* o_ = (half_t*)o;
    half_t* p_ = (half_t*)p;
    const half_t* m_ = (const half_t*)mapper_t;
    half_t* c_ = (half_t*)cur_out;
    switch (d.mode) {
        case FZ_ATTN_FLASH:
            FZ_LAUNCH((attn_cross_kernel<D, FZ_ATTN_FLASH>), grid, block, 0, stream, d, q_, k_, vt_, o_, p_, m_, coef, c_);
            break;
        case FZ_ATTN_CAPTURE:
            FZ_LAUNCH((attn_cross_kernel<D, FZ_ATTN_CAPTURE>), grid, block, 0, stream, d, q_, k_, vt_, o_, p_, m_, coef, c_);
            break;
        case FZ_ATTN_INJECT:
            FZ_LAUNCH((attn_cross_kernel<D, FZ_ATTN_INJECT>), grid, block, 0, stream, d, q_, k_, vt_, o_, p_, m_, coef, c_);
            break;
        default:
            return FZ_ERR_BAD_ARG;
    }
    return fz_last_launch_status();
}

extern "C" int fz_attn_cross(const FzAttnCrossDesc* desc, const void* q, const void* k, const void* vt, void* o,
                             void* p, const void* mapper_t, const float* coef, void* cur_out, void* stream) {
    if (!desc || !q || !k || !vt || !o) return FZ_ERR_BAD_ARG;
    const FzAttnCrossDesc& d = *desc;
    if (d.n_frames <= 0 || d.lq <= 0 || d.lk <= 0 || d.lk > FZ_CROSS_MAX_KEYS) return FZ_ERR_BAD_ARG;
    if (d.mode != FZ_ATTN_FLASH && (!p || d.p_row_stride < FZ_CROSS_P_STRIDE || (d.p_row_stride & 7))) return FZ_ERR_BAD_ARG;
    if (d.mode == FZ_ATTN_INJECT && (!mapper_t || !coef)) return FZ_ERR_BAD_ARG;
    if ((d.q_row_stride | d.k_row_stride | d.vt_chan_stride | d.o_row_stride | d.q_frame_stride | d.k_batch_stride |
         d.vt_batch_stride | d.o_frame_stride) & 7)
        return FZ_ERR_BAD_ARG;
    switch (d.head_dim) {
        case 16: return launch_cross<16>(d, q, k, vt, o, p, mapper_t, coef, cur_out, stream);
        case 32: return launch_cross<32>(d, q, k, vt, o, p, mapper_t, coef, cur_out, stream);
        case 40: return launch_cross<40>(d, q, k, vt, o, p, mapper_t, coef, cur_out, stream);
        case 64: return launch_cross<64>(d, q, k, vt, o, p, mapper_t, coef, cur_out, stream);
        case 80: return launch_cross<80>(d, q, k, vt, o, p, mapper_t, coef, cur_out, stream);
        case 128: return launch_cross<128>(d, q, k, vt, o, p, mapper_t, coef, cur_out, stream);
        case 160: return launch_cross<160>(d, q, k, vt, o, p, mapper_t, coef, cur_out, stream);
        default: return FZ_ERR_UNSUPPORTED;
    }
}
