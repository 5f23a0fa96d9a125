// conv3x3.hip -- 3x3 convolution of the pseudo-3D ResNet blocks as an MFMA implicit GEMM on token-major (NHWC)
// fp16 activations.  Replaces the spatial part of PseudoConv3d.forward (resnet.py:57-64) inside
// ResnetBlockPseudo3D / Up- / DownsamplePseudo3D (resnet.py:123-236, :335-394) with the elementwise tail fused:
//   y[n][oy][ox][co] = sum_{ky,kx,ci} x[n][iy][ix][ci] * w[co][ky][kx][ci]  + bias[co] (+ temb[b][co]) (+ res[n][oy][ox][co])
//   iy = oy*stride + ky - 1 (zero padded); with `upsample` the input is read through a nearest-2x upsampling
//   (resnet.py:145) that is never materialised.
// GEMM view: M = cout (A operand = packed weights wt[co][tap][ci], k contiguous), N = pixels (B operand = x rows,
// k contiguous), K = 9*Cin; so both LDS tiles are [row][k] and are read with ds_read_b128, and the accumulator
// layout (lane <-> pixel, 4 consecutive couts per register group) stages through LDS into full-row 16-byte stores.
// Work decomposition: wave tile 64 couts x 64 pixels (2x2 MFMA 32x32x16 tiles), WM x WN waves per workgroup,
// K step = one tap x BK channels; register prefetch + double-buffered LDS, one barrier per K step.
#include "fz_rt.h"
#include "../../include/fatezero_hip.h"

struct ConvArgs {
    const half_t* x;
    const half_t* wt;    // [Cout][taps][Cin]
    const half_t* bias;  // [Cout] or null
    const half_t* temb;  // [B][Cout] or null (added per batch element; frames_per_batch frames share a row)
    const half_t* res;   // [N][Ho][Wo][Cout] or null
    const half_t* res2;  // second residual, same layout, or null
    half_t* y;
    int64_t temb_stride; // elements between the temb rows of consecutive batch elements
    int N, Hi, Wi, Cin, Ho, Wo, Cout, stride, upsample, frames_per_batch;
    int temporal;  // != 0: 3 taps along the FRAME axis (k=3, zero padded inside each clip of frames_per_batch frames)
};

// SWZ: LDS staging layout.  false: rows padded to BK + 8 halves (fragment reads conflict-free, but the staging writes
// -- four 16-byte chunks of a row per four lanes -- collide 2-way: profiles/r01_pmc_conv64.json).  true: unpadded rows with
// the 16-byte chunk index XOR-ed by (row / rows-per-256-bytes) % chunks-per-row, conflict-free for both the staging writes
// and the 16-row fragment reads.  Selected with FZ_CONV_SWZ=1 until it has been timed on hardware.
template <int WM, int WN, int BK, bool SWZ = false>
struct ConvCfg {
    static constexpr int T = 64 * WM * WN;
    static constexpr int BCO = 64 * WM;  // couts per workgroup
    static constexpr int BPX = 64 * WN;  // pixels per workgroup
    static constexpr int KSTR = SWZ ? BK : BK + 8;  // halves; (BK+8)/8 odd for BK = 32, 64
    static constexpr int NCH = BK / 8;              // 16-byte chunks per row
    static constexpr int RPB = 128 / BK;            // rows per 256 bytes of LDS (one pass over all 64 banks)
    // halves offset of chunk `ch` of row `row` inside a staged operand tile
    static FZ_DEVICE int at(int row, int ch) { return row * KSTR + (SWZ ? (ch ^ ((row / RPB) % NCH)) : ch) * 8; }
    static constexpr int WCH = BCO * BK / 8;  // 16-byte chunks per weight tile
    static constexpr int XCH = BPX * BK / 8;
    static constexpr int WLD = (WCH + T - 1) / T;
    static constexpr int XLD = (XCH + T - 1) / T;
    static constexpr int STAGE = (BCO + BPX) * KSTR;
    static constexpr int CSTR = BCO + 8;
    static constexpr int CS = BPX * CSTR;
    static constexpr int LDS_HALVES = (2 * STAGE > CS) ? 2 * STAGE : CS;
};

template <int WM, int WN, int BK, bool SWZ = false>
FZ_KERNEL void __launch_bounds__(64 * WM * WN) conv3x3_kernel(ConvArgs a) {
    typedef ConvCfg<WM, WN, BK, SWZ> C;
    FZ_SHARED __attribute__((aligned(16))) half_t smem[C::LDS_HALVES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int co0 = blockIdx.y * C::BCO;
    const int64_t px0 = (int64_t)blockIdx.x * C::BPX;
    const int64_t npix = (int64_t)a.N * a.Ho * a.Wo;
    const int Hu = a.upsample ? a.Hi * 2 : a.Hi, Wu = a.upsample ? a.Wi * 2 : a.Wi;  // logical input extent

    // per-thread pixel coordinates of the activation chunks it fetches (fixed over the K loop)
    int xn[C::XLD], xoy[C::XLD], xox[C::XLD], xch[C::XLD];
    bool xok[C::XLD];
#pragma unroll
    for (int i = 0; i < C::XLD; ++i) {
        int id = tid + C::T * i;
        xok[i] = id < C::XCH;
        id = xok[i] ? id : C::XCH - 1;
        const int64_t p = px0 + id / (BK / 8);
        xch[i] = id % (BK / 8);
        const int64_t pc = p < npix ? p : npix - 1;
        xok[i] = xok[i] && p < npix;
        xn[i] = (int)(pc / ((int64_t)a.Ho * a.Wo));
        const int rem = (int)(pc % ((int64_t)a.Ho * a.Wo));
        xoy[i] = rem / a.Wo;
        xox[i] = rem % a.Wo;
    }
    const int kchunks = a.Cin / BK;
    const int ntaps = a.temporal ? 3 : 9;
    const int nk = ntaps * kchunks;

    half8_t wreg[C::WLD], xreg[C::XLD];
    bool xz[C::XLD];
    // Addresses are strength-reduced (the im2col gather used to cost ~14 VALU per MFMA, profiles/r01_pmc_conv64.json):
    //   weights  [cout][tap][cin]: the K index of slab ks is ks*BK, so a wave-uniform pointer advances by one slab per
    //            step and each lane keeps one constant 32-bit byte offset per chunk;
    //   activations: the (tap-dependent) source pixel of a lane's chunk -- bounds test, clamp, upsample shift, frame shift
    //            of the temporal form -- is recomputed only when the tap changes (every Cin/BK steps); inside a tap the
    //            same wave-uniform pointer trick walks along the channels.
    uint32_t woff[C::WLD], xoff[C::XLD];
#pragma unroll
    for (int i = 0; i < C::WLD; ++i) {
        int id = tid + C::T * i;
        id = id < C::WCH ? id : C::WCH - 1;
        int co = co0 + id / (BK / 8);
        co = co < a.Cout ? co : a.Cout - 1;
        woff[i] = 2u * ((uint32_t)co * (uint32_t)(ntaps * a.Cin) + (uint32_t)(id % (BK / 8)) * 8u);
    }
    const char* wcur = reinterpret_cast<const char*>(a.wt);
    const char* xcur = reinterpret_cast<const char*>(a.x);
    int ftap = 0, fkc = 0;  // tap / channel slab of the next fetch
    auto retarget = [&](int tap) {  // per-lane source pixels of tap `tap`
        const int ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int i = 0; i < C::XLD; ++i) {
            if (a.temporal) {
                const int f = xn[i] % a.frames_per_batch;
                int fs = f + tap - 1;
                xz[i] = !(fs >= 0 && fs < a.frames_per_batch && xok[i]);
                fs = fs < 0 ? 0 : (fs >= a.frames_per_batch ? a.frames_per_batch - 1 : fs);
                const int ns = xn[i] - f + fs;
                xoff[i] = 2u * ((((uint32_t)ns * a.Hi + xoy[i]) * a.Wi + xox[i]) * (uint32_t)a.Cin + xch[i] * 8u);
            } else {
                int iy = xoy[i] * a.stride + ky - 1, ix = xox[i] * a.stride + kx - 1;
                const bool inb = iy >= 0 && iy < Hu && ix >= 0 && ix < Wu;
                xz[i] = !(inb && xok[i]);
                iy = iy < 0 ? 0 : (iy >= Hu ? Hu - 1 : iy);
                ix = ix < 0 ? 0 : (ix >= Wu ? Wu - 1 : ix);
                if (a.upsample) {
                    iy >>= 1;
                    ix >>= 1;
                }
                xoff[i] = 2u * ((((uint32_t)xn[i] * a.Hi + iy) * a.Wi + ix) * (uint32_t)a.Cin + xch[i] * 8u);
            }
        }
    };
    auto fetch = [&]() {
        if (fkc == 0) {  // wave-uniform: a new tap starts (no loads inside the branch)
            retarget(ftap);
            xcur = reinterpret_cast<const char*>(a.x);
        }
#pragma unroll
        for (int i = 0; i < C::WLD; ++i) wreg[i] = fz_ld_h8_off(wcur, woff[i]);
#pragma unroll
        for (int i = 0; i < C::XLD; ++i) xreg[i] = fz_ld_h8_off(xcur, xoff[i]);
        wcur += BK * 2;
        xcur += BK * 2;
        if (++fkc == kchunks) {
            fkc = 0;
            ++ftap;
        }
    };
    auto stash = [&](int st) {
        half_t* Ws = smem + st * C::STAGE;
        half_t* Xs = Ws + C::BCO * C::KSTR;
#pragma unroll
        for (int i = 0; i < C::WLD; ++i) {
            const int id = tid + C::T * i;
            if (id < C::WCH) {
                const bool okc = co0 + id / (BK / 8) < a.Cout;
                fz_st_h8(Ws + C::at(id / (BK / 8), id % (BK / 8)), okc ? wreg[i] : fz_zero_h8());
            }
        }
#pragma unroll
        for (int i = 0; i < C::XLD; ++i) {
            const int id = tid + C::T * i;
            if (id < C::XCH)
                fz_st_h8(Xs + C::at(id / (BK / 8), id % (BK / 8)), xz[i] ? fz_zero_h8() : xreg[i]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = fz_zero_f16v();

    fetch();
    stash(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) fetch();
        const half_t* Ws = smem + cur * C::STAGE;
        const half_t* Xs = Ws + C::BCO * C::KSTR;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            half8_t wf[2], xf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wf[i] = fz_ld_h8(Ws + C::at(wm * 64 + i * 32 + l31, 2 * kk + hi));
                xf[i] = fz_ld_h8(Xs + C::at(wn * 64 + i * 32 + l31, 2 * kk + hi));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = fz_mfma_32x32x16_f16(wf[i], xf[j], acc[i][j]);
        }
        if (ks + 1 < nk) stash(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C^T tile -> LDS [pixel][cout] -> bias / temb / residual -> 16-byte row stores ----------------
    half_t* Cs = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)      // cout tile
#pragma unroll
        for (int j = 0; j < 2; ++j)  // pixel tile
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (half_t)acc[i][j][4 * g + e];
                *reinterpret_cast<half4_t*>(Cs + (wn * 64 + j * 32 + l31) * C::CSTR + wm * 64 + i * 32 + 8 * g + 4 * hi) = v;
            }
    __syncthreads();
    constexpr int OCH = C::BCO / 8;
    for (int id = tid; id < C::BPX * OCH; id += C::T) {
        const int pl = id / OCH, ch = id % OCH;
        const int64_t p = px0 + pl;
        const int co = co0 + ch * 8;
        if (p < npix && co < a.Cout) {
            half8_t v = fz_ld_h8(Cs + pl * C::CSTR + ch * 8);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
            if (a.bias != nullptr) {
                const half8_t b = fz_ld_h8(a.bias + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += (float)b[e];
            }
            if (a.temb != nullptr) {
                const int n = (int)(p / ((int64_t)a.Ho * a.Wo));
                const half8_t t = fz_ld_h8(a.temb + (int64_t)(n / a.frames_per_batch) * a.temb_stride + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += (float)t[e];
            }
            if (a.res != nullptr) {
                const half8_t r = fz_ld_h8(a.res + p * a.Cout + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
            }
            if (a.res2 != nullptr) {
                const half8_t r = fz_ld_h8(a.res2 + p * a.Cout + co);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)f[e];
            fz_st_h8(a.y + p * a.Cout + co, v);
        }
    }
}

template <int WM, int WN, int BK, bool SWZ = false>
static int launch_conv(const ConvArgs& a, void* stream) {
    typedef ConvCfg<WM, WN, BK, SWZ> C;
    const int64_t npix = (int64_t)a.N * a.Ho * a.Wo;
    dim3 grid((unsigned)((npix + C::BPX - 1) / C::BPX), (a.Cout + C::BCO - 1) / C::BCO), block(C::T);
    FZ_LAUNCH((conv3x3_kernel<WM, WN, BK, SWZ>), grid, block, 0, stream, a);
    return fz_last_launch_status();
}

#include <stdlib.h>
#include <stdio.h>
static int conv_cfg_override() {  // tuning knob FZ_CONV_CFG=<wm><wn><bk/32> e.g. 122 = (1,2,64); 0 = heuristic
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FZ_CONV_CFG");
        v = e ? atoi(e) : 0;
    }
    return v;
}

static bool conv_swizzled() {  // tuning knob FZ_CONV_SWZ=1: conflict-free LDS staging layout (see ConvCfg)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FZ_CONV_SWZ");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

static int dispatch_conv(const ConvArgs& a, void* stream) {
    // the kernel addresses x and wt with 32-bit byte offsets from a 64-bit base
    if ((int64_t)a.N * a.Hi * a.Wi * a.Cin >= (1ll << 31) || (int64_t)a.Cout * (a.temporal ? 3 : 9) * a.Cin >= (1ll << 31))
        return FZ_ERR_UNSUPPORTED;
    const bool k64 = (a.Cin % 64) == 0;
    const int64_t npix = (int64_t)a.N * a.Ho * a.Wo;
    int cfg = conv_cfg_override();
    if (cfg == 0) {
        cfg = npix >= 16384 ? 221 : 222;  // measured on MI355X (scripts/kbench.py --conv)
    }
    if (!k64 && (cfg % 10) == 2) cfg -= 1;
    if (conv_swizzled()) {
        switch (cfg) {
            case 121: return launch_conv<1, 2, 32, true>(a, stream);
            case 221: return launch_conv<2, 2, 32, true>(a, stream);
            case 222: return launch_conv<2, 2, 64, true>(a, stream);
            default: break;  // other tile shapes: padded layout
        }
    }
    switch (cfg) {
        case 111: return launch_conv<1, 1, 32>(a, stream);  // 64 x 64 tiles, one wave: candidates for the small levels
        case 112: return launch_conv<1, 1, 64>(a, stream);
        case 211: return launch_conv<2, 1, 32>(a, stream);
        case 212: return launch_conv<2, 1, 64>(a, stream);
        case 121: return launch_conv<1, 2, 32>(a, stream);
        case 122: return launch_conv<1, 2, 64>(a, stream);
        case 141: return launch_conv<1, 4, 32>(a, stream);
        case 142: return launch_conv<1, 4, 64>(a, stream);
        case 221: return launch_conv<2, 2, 32>(a, stream);
        case 222: return launch_conv<2, 2, 64>(a, stream);
        case 241: return launch_conv<2, 4, 32>(a, stream);
        default: return FZ_ERR_BAD_ARG;
    }
}

extern "C" int fz_temporal_conv3(const void* x, const void* wt, const void* res, const void* res2, const void* temb,
                                int64_t temb_stride, void* y, int n, int tokens, int cin, int cout, int clip_len,
                                void* stream) {
    if (!x || !wt || !y || n <= 0 || tokens <= 0 || clip_len <= 0 || n % clip_len) return FZ_ERR_BAD_ARG;
    if (cin % 32 || cout % 8) return FZ_ERR_UNSUPPORTED;
    ConvArgs a;
    a.x = (const half_t*)x; a.wt = (const half_t*)wt; a.bias = nullptr; a.temb = (const half_t*)temb;
    a.temb_stride = temb_stride ? temb_stride : cout;
    a.res = (const half_t*)res; a.res2 = (const half_t*)res2; a.y = (half_t*)y;
    a.N = n; a.Hi = 1; a.Wi = tokens; a.Cin = cin; a.Cout = cout; a.stride = 1; a.upsample = 0;
    a.frames_per_batch = clip_len; a.temporal = 1; a.Ho = 1; a.Wo = tokens;
    return dispatch_conv(a, stream);
}

extern "C" int fz_conv3x3(const void* x, const void* wt, const void* bias, const void* temb, int64_t temb_stride,
                          const void* res, void* y, int n, int hi, int wi, int cin, int cout, int stride, int upsample,
                          int frames_per_batch, void* stream) {
    if (!x || !wt || !y || n <= 0 || hi <= 0 || wi <= 0) return FZ_ERR_BAD_ARG;
    if (cin % 32 || cout % 8 || (stride != 1 && stride != 2) || (upsample && stride != 1)) return FZ_ERR_UNSUPPORTED;
    ConvArgs a;
    a.x = (const half_t*)x; a.wt = (const half_t*)wt; a.bias = (const half_t*)bias; a.temb = (const half_t*)temb;
    a.res = (const half_t*)res; a.res2 = nullptr; a.y = (half_t*)y;
    a.temb_stride = temb_stride ? temb_stride : cout;
    a.N = n; a.Hi = hi; a.Wi = wi; a.Cin = cin; a.Cout = cout; a.stride = stride; a.upsample = upsample;
    a.frames_per_batch = frames_per_batch > 0 ? frames_per_batch : 1;
    a.temporal = 0;
    const int hu = upsample ? 2 * hi : hi, wu = upsample ? 2 * wi : wi;
    a.Ho = (hu + 2 - 3) / stride + 1;
    a.Wo = (wu + 2 - 3) / stride + 1;
    return dispatch_conv(a, stream);
}
