// blend_mask.hip -- the cross-attention blend mask of the reference's SpatialBlender.get_mask
// (spatial_blend.py:24-56, called from :58-124): for every (prompt, frame)
//   m = mean_{layer,head} sum_w maps[..., w] * alpha[w]          (fp32, fixed order: layer-major, head, word)
//   m = max_pool2d(m, 3, stride 1, pad 1)  ->  nearest resize to (h, w)  ->  m / max(m)  ->  m > th
// One workgroup per (prompt, frame); the r x r map lives in LDS.  The result is a 0/1 float mask: the
// index-level parity target of the project (tests require 0 differing elements against the oracle).
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"

#define BM_MAX_MAPS 8
#define BM_MAX_PIX 1600 /* r <= 40 */

struct BlendArgs {
    const half_t* maps[BM_MAX_MAPS];
    int n_maps, n_prompts, frames, heads, res, out_h, out_w, or_first;
    int64_t prompt_stride, row_stride;
    const float* alpha;  // [P][80]
    float th;
    float* out;  // [P][F][h][w]
};

FZ_KERNEL void __launch_bounds__(256) blend_mask_kernel(BlendArgs a) {
    FZ_SHARED float m[BM_MAX_PIX];
    FZ_SHARED float pooled[BM_MAX_PIX];
    FZ_SHARED float red[256];
    FZ_SHARED float al[80];
    const int tid = threadIdx.x;
    const int pr = blockIdx.x / a.frames, f = blockIdx.x % a.frames;
    const int r = a.res, npix = r * r;
    if (tid < 80) al[tid] = a.alpha[pr * 80 + tid];
    __syncthreads();
    const float cnt = (float)(a.n_maps * a.heads);
    for (int pix = tid; pix < npix; pix += 256) {
        float acc = 0.0f;
        for (int mi = 0; mi < a.n_maps; ++mi)
            for (int h = 0; h < a.heads; ++h) {
                const half_t* row = a.maps[mi] + (int64_t)pr * a.prompt_stride +
                                    (((int64_t)f * a.heads + h) * npix + pix) * a.row_stride;
                float s = 0.0f;
                for (int c = 0; c < 10; ++c) {
                    const half8_t v = fz_ld_h8(row + 8 * c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += (float)v[e] * al[8 * c + e];
                }
                acc += s;
            }
        m[pix] = acc / cnt;
    }
    __syncthreads();
    for (int pix = tid; pix < npix; pix += 256) {
        const int y = pix / r, x = pix % r;
        float mx = -INFINITY;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < r && xx >= 0 && xx < r) mx = fmaxf(mx, m[yy * r + xx]);
            }
        pooled[pix] = mx;
    }
    __syncthreads();
    // nearest resize (torch legacy 'nearest': src = min(floor(dst * in/out), in-1)) and the max over the output
    const float sy = (float)r / (float)a.out_h, sx = (float)r / (float)a.out_w;
    const int nout = a.out_h * a.out_w;
    float lmax = -INFINITY;
    for (int i = tid; i < nout; i += 256) {
        const int oy = i / a.out_w, ox = i % a.out_w;
        const int iy = min((int)floorf(oy * sy), r - 1), ix = min((int)floorf(ox * sx), r - 1);
        lmax = fmaxf(lmax, pooled[iy * r + ix]);
    }
    red[tid] = lmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    const float gmax = red[0];
    float* out = a.out + ((int64_t)pr * a.frames + f) * nout;
    for (int i = tid; i < nout; i += 256) {
        const int oy = i / a.out_w, ox = i % a.out_w;
        const int iy = min((int)floorf(oy * sy), r - 1), ix = min((int)floorf(ox * sx), r - 1);
        out[i] = (pooled[iy * r + ix] / gmax > a.th) ? 1.0f : 0.0f;
    }
}

FZ_KERNEL void __launch_bounds__(256) blend_or_first_kernel(float* out, int n_prompts, int64_t per_prompt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_prompt) return;
    const float first = out[i];
    for (int p = 1; p < n_prompts; ++p)
        if (first != 0.0f) out[(int64_t)p * per_prompt + i] = 1.0f;
}

extern "C" int fz_blend_mask(const void* const* maps, int n_maps, int n_prompts, int64_t prompt_stride, int frames,
                             int heads, int res, int64_t p_row_stride, const float* alpha, float th, int out_h,
                             int out_w, int or_with_first, float* out, float* scratch, void* stream) {
    (void)scratch;
    if (!maps || !alpha || !out || n_maps <= 0 || n_maps > BM_MAX_MAPS || res * res > BM_MAX_PIX) return FZ_ERR_BAD_ARG;
    if (p_row_stride < 80 || (p_row_stride & 7)) return FZ_ERR_BAD_ARG;
    BlendArgs a;
    for (int i = 0; i < BM_MAX_MAPS; ++i) a.maps[i] = i < n_maps ? (const half_t*)maps[i] : nullptr;
    a.n_maps = n_maps; a.n_prompts = n_prompts; a.frames = frames; a.heads = heads; a.res = res;
    a.out_h = out_h; a.out_w = out_w; a.or_first = or_with_first;
    a.prompt_stride = prompt_stride; a.row_stride = p_row_stride; a.alpha = alpha; a.th = th; a.out = out;
    FZ_LAUNCH(blend_mask_kernel, dim3(n_prompts * frames), dim3(256), 0, stream, a);
    if (or_with_first && n_prompts > 1) {
        const int64_t per = (int64_t)frames * out_h * out_w;
        FZ_LAUNCH(blend_or_first_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, stream, out, n_prompts, per);
    }
    return fz_last_launch_status();
}
