// elementwise.hip -- the small HBM-bound kernels of the loop: GEGLU gate, V transpose (+pad), the fused
// CFG + DDIM latent update (+ latent blend), and the fp32 running sum of cross maps.
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"

// ---- GEGLU (diffusers FeedForward/GEGLU [3P]): y = h * gelu_erf(gate), [h | gate] = x ----------
FZ_KERNEL void __launch_bounds__(256) geglu_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int64_t rows, int inner) {
    const int vper = inner >> 3;
    const int64_t total = rows * vper;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / vper;
        const int v = (int)(i % vper);
        const half8_t hv = fz_ld_h8(x + r * 2 * inner + v * 8);
        const half8_t gv = fz_ld_h8(x + r * 2 * inner + inner + v * 8);
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = (float)gv[e];
            const float ge = fz_gelu_erf(g);
            o[e] = (half_t)((float)hv[e] * ge);
        }
        fz_st_h8(y + r * inner + v * 8, o);
    }
}

extern "C" int fz_geglu(const void* x, void* y, int64_t rows, int inner, void* stream) {
    if (!x || !y || rows <= 0 || inner <= 0 || (inner & 7)) return FZ_ERR_BAD_ARG;
    const int64_t total = rows * (inner >> 3);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    FZ_LAUNCH(geglu_kernel, dim3(blocks), dim3(256), 0, stream, (const half_t*)x, (half_t*)y, rows, inner);
    return fz_last_launch_status();
}

// ---- row softmax y = softmax(scale * x) of an fp16 score matrix, fp32 arithmetic (the single-head 512-wide attention of the
//      VAE mid block, diffusers AttentionBlock [3P]: scores too wide for the head-dim-specialised attention kernels, outside
//      the timed loop).  One wave per row; the row is read once and kept in registers when it fits (cols <= 64 * 8 * SM_MAXV).
#define SM_MAXV 16
FZ_DEVICE float sm_wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, fz_shfl_xor(v, m));
    return v;
}
FZ_DEVICE float sm_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += fz_shfl_xor(v, m);
    return v;
}
FZ_KERNEL void __launch_bounds__(256)
softmax_rows_kernel(const half_t* __restrict__ x, half_t* __restrict__ y, int64_t rows, int cols, int64_t ldx, int64_t ldy,
                    float scale_log2e) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    const bool rvalid = row < rows;  // every lane stays alive for the shuffles
    const int V = cols >> 3;
    const half_t* xr = x + (rvalid ? row : 0) * ldx;
    if (V <= 64 * SM_MAXV) {
        half8_t xv[SM_MAXV];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < SM_MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < V) {
                xv[i] = fz_ld_h8(xr + v * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)xv[i][e]);
            }
        }
        mx = sm_wave_max(mx) * scale_log2e;  // scale > 0
        float sum = 0.0f;
        float ev[SM_MAXV][8];
#pragma unroll
        for (int i = 0; i < SM_MAXV; ++i) {
            const int v = lane + 64 * i;
            if (v < V) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ev[i][e] = fz_exp2(fmaf((float)xv[i][e], scale_log2e, -mx));
                    sum += ev[i][e];
                }
            }
        }
        const float inv = 1.0f / sm_wave_sum(sum);
#pragma unroll
        for (int i = 0; i < SM_MAXV; ++i) {
            const int v = lane + 64 * i;
            if (rvalid && v < V) {
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)(ev[i][e] * inv);
                fz_st_h8(y + row * ldy + v * 8, o);
            }
        }
    } else {  // long rows: three sweeps, the re-reads hit L2
        float mx = -INFINITY;
        for (int v = lane; v < V; v += 64) {
            const half8_t t = fz_ld_h8(xr + v * 8);
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)t[e]);
        }
        mx = sm_wave_max(mx) * scale_log2e;
        float sum = 0.0f;
        for (int v = lane; v < V; v += 64) {
            const half8_t t = fz_ld_h8(xr + v * 8);
            for (int e = 0; e < 8; ++e) sum += fz_exp2(fmaf((float)t[e], scale_log2e, -mx));
        }
        const float inv = 1.0f / sm_wave_sum(sum);
        for (int v = lane; v < V; v += 64) {
            const half8_t t = fz_ld_h8(xr + v * 8);
            half8_t o;
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(fz_exp2(fmaf((float)t[e], scale_log2e, -mx)) * inv);
            if (rvalid) fz_st_h8(y + row * ldy + v * 8, o);
        }
    }
}

extern "C" int fz_softmax_rows(const void* x, void* y, int64_t rows, int cols, int64_t ldx, int64_t ldy, float scale,
                               void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || !(scale > 0.0f)) return FZ_ERR_BAD_ARG;
    if ((cols & 7) || (ldx & 7) || (ldy & 7) || ldx < cols || ldy < cols) return FZ_ERR_UNSUPPORTED;
    FZ_LAUNCH(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const half_t*)x, (half_t*)y, rows,
              cols, ldx, ldy, scale * 1.4426950408889634f);
    return fz_last_launch_status();
}

// ---- out[n][c][lp] = in[n][l][c]^T, zero padded (the V^T operand of the attention kernels) ------------------
FZ_KERNEL void __launch_bounds__(256)
transpose_pad_kernel(const half_t* __restrict__ in, half_t* __restrict__ out, int l, int c, int64_t in_frame_stride,
                     int64_t in_row_stride, int lp) {
    FZ_SHARED half_t tile[64][66];
    const int n = blockIdx.z, l0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int li = l0 + r, ci = c0 + tx;
        tile[r][tx] = (li < l && ci < c) ? in[(int64_t)n * in_frame_stride + (int64_t)li * in_row_stride + ci] : (half_t)0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int ci = c0 + r, li = l0 + tx;
        if (ci < c && li < lp) out[((int64_t)n * c + ci) * lp + li] = tile[tx][r];
    }
}

extern "C" int fz_transpose_pad(const void* in, void* out, int n, int l, int c, int64_t in_frame_stride,
                                int64_t in_row_stride, int lp, void* stream) {
    if (!in || !out || n <= 0 || l <= 0 || c <= 0 || lp < l) return FZ_ERR_BAD_ARG;
    dim3 grid((lp + 63) / 64, (c + 63) / 64, n), block(256);
    FZ_LAUNCH(transpose_pad_kernel, grid, block, 0, stream, (const half_t*)in, (half_t*)out, l, c, in_frame_stride,
              in_row_stride, lp);
    return fz_last_launch_status();
}

// ---- fused latent update --------------------------------------------------------------------------------------
// eps = eps_u + g (eps_c - eps_u) (p2p_ddim_spatial_temporal.py:400-404); z <- cz z + ce eps, which is
// DDIMScheduler.step(eta=0) [3P] or next_clean2noise_step (p2p_ddim:150-161) with the two scalar coefficients
// folded on the host; optional latent blend z <- inv + mask (z - inv) (spatial_blend.py:121).
FZ_KERNEL void __launch_bounds__(256)
latent_update_kernel(float* __restrict__ z, const half_t* __restrict__ eps_u, const half_t* __restrict__ eps_c,
                     float guidance, float cz, float ce, const float* __restrict__ inv, const float* __restrict__ mask,
                     half_t* __restrict__ next_in, int frames, int hw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)frames * hw) return;
    const int f = (int)(i / hw), pix = (int)(i % hw);
    const half4_t ec = *reinterpret_cast<const half4_t*>(eps_c + i * 4);
    half4_t eu = ec;
    if (eps_u != nullptr) eu = *reinterpret_cast<const half4_t*>(eps_u + i * 4);
    const float mk = mask ? mask[i] : 1.0f;
    half4_t nx;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t zi = ((int64_t)c * frames + f) * hw + pix;
        float e = (float)ec[c];
        if (eps_u != nullptr) e = (float)eu[c] + guidance * ((float)ec[c] - (float)eu[c]);
        float zn = cz * z[zi] + ce * e;
        if (mask != nullptr) zn = inv[zi] + mk * (zn - inv[zi]);
        z[zi] = zn;
        nx[c] = (half_t)zn;
    }
    if (next_in != nullptr) *reinterpret_cast<half4_t*>(next_in + i * 4) = nx;
}

extern "C" int fz_latent_update(float* z, const void* eps_u, const void* eps_c, float guidance, float cz, float ce,
                                const float* inv, const float* mask, void* next_in, int frames, int hw, void* stream) {
    if (!z || !eps_c || frames <= 0 || hw <= 0) return FZ_ERR_BAD_ARG;
    if ((mask != nullptr) != (inv != nullptr)) return FZ_ERR_BAD_ARG;
    const int64_t total = (int64_t)frames * hw;
    FZ_LAUNCH(latent_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, z, (const half_t*)eps_u,
              (const half_t*)eps_c, guidance, cz, ce, inv, mask, (half_t*)next_in, frames, hw);
    return fz_last_launch_status();
}

// ---- acc (fp32) += x (fp16): running sum of the edit pass' own cross maps (attention_store.py:95-101) ---------
FZ_KERNEL void __launch_bounds__(256) accumulate_kernel(float* __restrict__ acc, const half_t* __restrict__ x, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const half8_t v = fz_ld_h8(x + i * 8);
        float* a = acc + i * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += (float)v[e];
    }
}

extern "C" int fz_accumulate(float* acc, const void* x, int64_t n, void* stream) {
    if (!acc || !x || n <= 0 || (n & 7)) return FZ_ERR_BAD_ARG;
    const int64_t n8 = n >> 3;
    int blocks = (int)((n8 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    FZ_LAUNCH(accumulate_kernel, dim3(blocks), dim3(256), 0, stream, acc, (const half_t*)x, n8);
    return fz_last_launch_status();
}

extern "C" const char* fz_version(void) {
#ifdef FZ_EMU
    return "fatezero_amd 0.1 (CPU emulation build -- tests only)";
#else
    return "fatezero_amd 0.1 (hip gfx950)";
#endif
}
