// conv_halo.hip -- the stride-1 3x3 convolution of PseudoConv3d's spatial part (resnet.py:57-64) with the PIXEL operand resident in LDS across
// the nine taps.  Same arithmetic as fz_conv3x3's implicit GEMM (csrc/igemm.hip MODE 3: K order = Cin chunk outer, taps inner, fp32
// accumulation with v_mfma_f32_32x32x16_f16, bias in fp32, one fp16 rounding before the time-embedding row / residual): a different data path.
//
// Why: with the chip full, every K loop of igemm.hip runs at what the L2s deliver to 256 CUs -- ~12 TB/s, 56 KB
// per K-64 step per CU for the 320 x 128 tile -- whatever the loop's structure (ring, K groups, ping-pong, loader / consumer: within 3 % of each
// other).  What is left is BYTES per MAC.  In the implicit GEMM a workgroup re-fetches its pixel rows once per tap: 9 x.  Here a workgroup owns
// 256 output pixels = 256 / W whole image rows of one frame and 160 output channels; per 64-channel chunk it stages the pixel rows WITH THEIR
// HALO -- (256 / W + 2) x (W + 2) pixels x 128 bytes, zeros outside the image -- ONCE, and the nine taps read the same LDS tile at a constant
// pixel offset per tap ((ky - 1)(W + 2) + kx - 1).  Per chunk and workgroup: 9 x 20 KB of weights + 51 KB of pixels = 231 KB for 160 x 256 x 64 x 9
// MACs, against 9 x 56 KB = 504 KB for the same MACs on the 320 x 128 tile: 2.2 x fewer bytes per MAC.
//
// One workgroup = 8 waves in two roles (the loader / consumer split measured in igemm.hip, tile 252214):
//   waves 0-3  CONSUMERS, one per SIMD: wave w owns pixels [64 w, 64 w + 64) x all 160 channels as 5 x 2 MFMA tiles (160 accumulator registers);
//              per K-32 step it reads 10 weight fragments from the ring and 4 pixel fragments from the halo tile -- one cluster AHEAD of
//              their use -- and runs 20 MFMAs; nothing else;
//   waves 4-5  WEIGHT loaders: the 160 x 32 weight tile of (chunk, tap, half) = 10 LDS-DMA pieces per step, 4-slot ring, three tiles ahead;
//   waves 6-7  PIXEL loaders: the NEXT chunk's halo tile (<= 50 pieces of 8 pixels x 128 B), spread over the first 16 of the chunk's 18 steps,
//              into the other of two halo buffers.
// One barrier per K-32 step.  Barrier B(j) has weight tiles <= j + 1 landed and, in front of a chunk's last step, the next chunk's halo tile.
#include "fz_rt.h"
#include <atomic>
#include "../../include/fatezero_hip.h"

FZ_DEVICE_GLOBAL __attribute__((aligned(16))) half_t ch_zero_page[512];  // 1 KB of zeros: the source of every lane outside the image

namespace {
constexpr int CH_BA = 160;                    // output channels per workgroup (5 MFMA tiles)
constexpr int CH_BB = 256;                    // output pixels per workgroup (4 consumer waves x 2 MFMA tiles)
constexpr int CH_TA = 5, CH_TB = 2;
constexpr int CH_ASLOT = CH_BA * 32 * 2;      // bytes of one weight tile (K step 32): 10 KB
constexpr int CH_NAS = 4;                     // weight ring slots
constexpr int CH_AP = CH_BA / 16;             // LDS-DMA pieces per weight tile (16 rows x 64 B each): 10
constexpr int CH_OSTR = CH_BA + 8;            // staging row stride of the epilogue (halves)
static_assert(CH_BB * CH_OSTR * 2 <= 160 * 1024, "epilogue staging");
}  // namespace

#ifdef CH_TIMING  // trial build: cycle totals of consumer wave 0, weight loader 4 and pixel loader 6 of workgroup 0
__device__ long long ch_timing[3][4];
#define CH_T0() const long long tt0 = clock64()
#define CH_T1(slot) tacc[slot] += clock64() - tt0
#else
#define CH_T0() ((void)0)
#define CH_T1(slot) ((void)0)
#endif

struct ChArgs {
    const half_t* x;       // [N][H][W][Cin]
    const half_t* wt;      // [Cout][9][Cin]
    const half_t* bias;    // [Cout] or null
    const half_t* temb;    // rows of Cout values, one per temb_group consecutive pixels, or null
    const half_t* res;     // [N H W][Cout] or null
    half_t* y;             // [N H W][Cout]
    int64_t temb_stride, temb_group;
    int N, H, W, Cin, Cout;
    int tiles_a;           // Cout / 160
    int hb_bytes;          // bytes of one halo buffer (pieces of 1 KB)
    int np;                // LDS-DMA pieces of a halo tile
};

FZ_KERNEL void __launch_bounds__(512, 2) conv_halo_kernel(ChArgs g) {
    FZ_DYN_SMEM(raw);
    const int tid = threadIdx.x, wave = fz_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order (igemm.hip): blocks b, b + 8, .. share an XCD and get consecutive tiles; a-tile fastest: the two (or more) channel tiles of
    // one pixel tile sit next to each other and share its halo rows in L2
    const int nt = gridDim.x, bid = blockIdx.x;
    const int q8 = nt >> 3, r8 = nt & 7, xcd = bid & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int ta = lid % g.tiles_a, tb = lid / g.tiles_a;
    const int a0 = ta * CH_BA;
    const int W = g.W, W2 = W + 2, R = CH_BB / W;
    const int64_t px0 = (int64_t)tb * CH_BB;                 // first output pixel of the tile (flattened n, y, x)
    const int hw = g.H * W;
    const int fn = (int)(px0 / hw), y0 = (int)((px0 - (int64_t)fn * hw) / W);
    unsigned char* const Aring = raw;
    unsigned char* const Hbuf = raw + CH_NAS * CH_ASLOT;
    const int nchunk = g.Cin >> 6, nstep = nchunk * 18;     // a step = (chunk, tap, K half)
    const char* zero = reinterpret_cast<const char*>(ch_zero_page);

#ifdef CH_TIMING
    long long tacc[4] = {0, 0, 0, 0};
    const long long tstart = clock64();
#endif
    f32x16 acc[CH_TA][CH_TB];
#pragma unroll
    for (int i = 0; i < CH_TA; ++i)
#pragma unroll
        for (int q = 0; q < CH_TB; ++q) acc[i][q] = fz_zero_f16v();

    if (wave >= 6) {
        // ================================================ pixel loaders ===================================================================
        // piece p covers halo pixels [8 p, 8 p + 8): lane = (pixel 8 p + lane / 8, physical 16-byte chunk lane % 8) fetches logical chunk
        // (lane % 8) ^ ((pixel >> 1) & 7) of that pixel -- the swizzle the consumers' ds_read_b128 undo -- or zeros outside the image / the tile
        const int bl = wave - 6;
        const int npl = (g.np - bl + 1) / 2;   // pieces of this loader: bl, bl + 2, ...
        // per-lane source of every piece of this loader for chunk 0 (null: the zero page); a chunk adds 128 bytes
        const char* src[25];
#pragma unroll
        for (int i = 0; i < 25; ++i) {
            const int p = bl + 2 * i;
            const int hp = 8 * p + (lane >> 3), pc = lane & 7;
            const int yy = hp / W2, xx = hp - yy * W2;
            const int iy = y0 + yy - 1, ix = xx - 1;
            const bool ok = i < npl && yy < R + 2 && iy >= 0 && iy < g.H && ix >= 0 && ix < W;
            const int lc = pc ^ ((hp >> 1) & 7);
            src[i] = ok ? reinterpret_cast<const char*>(g.x + (((int64_t)fn * g.H + iy) * W + ix) * g.Cin + lc * 8) : nullptr;
        }
        auto fire = [&](int i, int chunk, int buf) {
            const char* s = src[i] != nullptr ? src[i] + chunk * 128 : zero + lane * 16;
            fz_glds16(s, Hbuf + buf * g.hb_bytes + (bl + 2 * i) * 1024);
        };
#pragma unroll
        for (int i = 0; i < 25; ++i)
            if (i < npl) fire(i, 0, 0);
        fz_wait_vm0();
        fz_barrier_raw();                                   // B(0)
        for (int c = 0; c < nchunk; ++c) {
            const bool more = c + 1 < nchunk;
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                if (more && s < 13) {                       // two pieces per step: 26 slots for the <= 25 pieces
                    if (2 * s < npl) fire(2 * s, c + 1, (c + 1) & 1);
                    if (2 * s + 1 < npl) fire(2 * s + 1, c + 1, (c + 1) & 1);
                }
                if (s == 16) {
                    CH_T0();
                    fz_wait_vm0();                          // the next chunk's tile is complete in front of B(18 c + 17)
                    CH_T1(0);
                }
                {
                    CH_T0();
                    fz_barrier_raw();                       // B(18 c + s + 1)
                    CH_T1(1);
                }
            }
        }
    } else if (wave >= 4) {
        // ================================================ weight loaders ==================================================================
        // tile (chunk c, tap t, half h) = rows a0 .. a0 + 160 of wt, 32 halves at k = t Cin + 64 c + 32 h; piece p = rows [16 p, 16 p + 16) x 4
        // chunks of 16 B; lane = (row 16 p + lane / 4, physical chunk lane % 4) fetches logical chunk (lane % 4) ^ ((row >> 2) & 3)
        const int al = wave - 4;
        uint32_t aoff[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int row = 16 * (al + 2 * i) + (lane >> 2);
            const int lc = (lane & 3) ^ ((row >> 2) & 3);
            aoff[i] = (uint32_t)(((int64_t)row * 9 * g.Cin + lc * 8) * 2);
        }
        const char* const a_tile = reinterpret_cast<const char*>(g.wt + (int64_t)a0 * 9 * g.Cin);
        auto issue = [&](int j) {                           // step j = 18 c + 2 t + h
            const int c = j / 18, r = j - 18 * c, t = r >> 1, h = r & 1;
            const char* base = a_tile + ((int64_t)t * g.Cin + 64 * c + 32 * h) * 2;
#pragma unroll
            for (int i = 0; i < 5; ++i) fz_glds16_so(base, aoff[i], Aring + (j & 3) * CH_ASLOT + (al + 2 * i) * 1024);
        };
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nstep) issue(j);
        if (nstep > 2) {
            fz_wait_vm<5>();                                // tiles 0, 1 landed; tile 2 in flight
        } else {
            fz_wait_vm0();
        }
        fz_barrier_raw();                                   // B(0)
        for (int j = 0; j < nstep; ++j) {
            {
                CH_T0();
                if (j + 3 < nstep) {
                    issue(j + 3);                           // into the slot tile j - 1 left (all its fragments were read before B(j))
                    CH_T1(2);
                    fz_wait_vm<5>();                        // tile j + 2 landed
                } else {
                    fz_wait_vm0();
                }
                CH_T1(0);
            }
            {
                CH_T0();
                fz_barrier_raw();                           // B(j + 1)
                CH_T1(1);
            }
        }
    } else {
        // ================================================ consumers ========================================================================
        // weight fragment (tile i, k sub-step kk): row 32 i + l31 of the slot, chunk (2 kk + hi) ^ ((row >> 2) & 3); a row = 64 bytes
        const int arow = l31 * 64, asw = (l31 >> 2) & 3;
        // pixel fragment (tile q, tap, chunk lc): halo pixel hp0[q] + tap shift, chunk lc ^ ((hp >> 1) & 7); a pixel = 128 bytes
        int hp0[CH_TB];
#pragma unroll
        for (int q = 0; q < CH_TB; ++q) {
            const int pl = 64 * wave + 32 * q + l31, yl = pl / W, xl = pl - yl * W;
            hp0[q] = (yl + 1) * W2 + xl + 1;
        }
        half8_t af0[CH_TA], bf0[CH_TB], af1[CH_TA], bf1[CH_TB];
        // LDS addresses of the fragments the READ cursor points at -- it runs one k sub-step ahead of the MFMAs: ONE address for the weight tile
        // (the five tiles sit at constant offsets), one per pixel tile.  Incremental: per tap 6 VALU per pixel tile (halo pixel, its byte
        // offset, its swizzle term), per sub-step 2 -- a wave alone on its SIMD hides ~5 instructions per MFMA and nothing between clusters.
        struct Fa {
            fz_lds_addr a, b[CH_TB];
        };
        const fz_lds_addr a_kk[2] = {fz_lds_addr_of(Aring) + arow + ((hi ^ asw) << 4), fz_lds_addr_of(Aring) + arow + (((2 + hi) ^ asw) << 4)};
        const int hi4 = hi << 4;
        // (branch-free: a branch in here is a basic-block boundary between two clusters -- nothing of it can sink into the MFMAs' shadow)
        int rs = 0, rt = 0, rc = 0;                          // read cursor: k sub-step counter (4 per tap), tap, chunk
        auto next_addr = [&]() -> Fa {                      // the cursor's addresses; then one sub-step on
            const int rkk = rs & 1, rh = (rs >> 1) & 1, rj = rs >> 1;
            const int ky = rt / 3, kx = rt - 3 * ky, sh = (ky - 1) * W2 + (kx - 1);
            const fz_lds_addr hs = fz_lds_addr_of(Hbuf) + (rc & 1) * g.hb_bytes;
            const int S = (4 * rh + 2 * rkk) << 4;
            Fa f;
            f.a = (rkk ? a_kk[1] : a_kk[0]) + (rj & 3) * CH_ASLOT;
#pragma unroll
            for (int q = 0; q < CH_TB; ++q) {
                const int hp = hp0[q] + sh;
                f.b[q] = hs + hp * 128 + ((hi4 ^ ((hp << 3) & 0x70)) ^ S);
            }
            const int adv = (rs & 3) == 3;                   // the next sub-step opens a new tap
            const int rt2 = rt + adv, wrap = rt2 == 9;
            rt = wrap ? 0 : rt2;
            rc += wrap;
            ++rs;
            return f;
        };
        auto rd = [&](half8_t* af, half8_t* bf, const Fa& f) {
#pragma unroll
            for (int q = 0; q < CH_TB; ++q) bf[q] = fz_lds_ld_h8(f.b[q], 0);
#pragma unroll
            for (int i = 0; i < CH_TA; ++i) af[i] = fz_lds_ld_h8(f.a, i * 32 * 64);
        };
        // one cluster (10 MFMAs on the fragments in af / bf) with the 7 reads of the NEXT fragments interleaved, one behind each of the first MFMAs
        auto mm_rd = [&](const half8_t* af, const half8_t* bf, half8_t* afn, half8_t* bfn, const Fa& fn) {
            rd(afn, bfn, fn);
#pragma unroll
            for (int i = 0; i < CH_TA; ++i)
#pragma unroll
                for (int q = 0; q < CH_TB; ++q) acc[i][q] = fz_mfma_32x32x16_f16(af[i], bf[q], acc[i][q]);
#ifndef FZ_EMU
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#endif
        };
        fz_barrier_raw();                                   // B(0)
        Fa f0 = next_addr(), f1 = next_addr();              // (0, 0), (0, 1)
        rd(af0, bf0, f0);
        for (int j = 0; j < nstep; ++j) {
            // (weight tile j + 1 and, at a chunk's end, the next halo tile landed before B(j): the fragments of (j + 1, 0) are read in step j)
            // (the reads are UNCONDITIONAL -- behind the last step they fetch LDS bytes nobody uses: a read under a branch makes hipcc's merged
            //  lgkmcnt wait in front of the next cluster drain half of the reads just issued)
#ifndef CH_FENCES
#define CH_FENCE() ((void)0)   /* (the address arithmetic of the next reads may sink into the clusters' shadow) */
#else
#define CH_FENCE() FZ_SCHED_FENCE()
#endif
            f0 = next_addr();                               // (j + 1, 0)
            CH_FENCE();
            mm_rd(af0, bf0, af1, bf1, f1);                  // cluster (j, 0) + the reads of (j, 1)
            FZ_SCHED_FENCE();
            f1 = next_addr();                               // (j + 1, 1)
            CH_FENCE();
            mm_rd(af1, bf1, af0, bf0, f0);                  // cluster (j, 1) + the reads of (j + 1, 0)
            FZ_SCHED_FENCE();
            {
                CH_T0();
                fz_barrier_raw();                           // B(j + 1)
                CH_T1(1);
            }
        }
    }
#ifdef CH_TIMING
    tacc[3] = clock64() - tstart;
    if (blockIdx.x == 0 && (tid == 0 || tid == 256 || tid == 384))
        for (int i = 0; i < 4; ++i) ch_timing[tid == 0 ? 0 : (tid == 256 ? 1 : 2)][i] = tacc[i];
#endif

    // ---- epilogue (igemm.hip's): + bias (fp32) -> fp16 tile through LDS [256 pixels][160 + 8] -> (+ temb row) (+ res) -> full-row 16-byte stores
    __syncthreads();
    half_t* Cs = reinterpret_cast<half_t*>(raw);
    if (wave < 4) {
#pragma unroll
        for (int i = 0; i < CH_TA; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int co = i * 32 + 8 * gq + 4 * hi;
                half4_t bv;
                if (g.bias != nullptr) {
                    bv = *reinterpret_cast<const half4_t*>(g.bias + a0 + co);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = (half_t)0.0f;
                }
#pragma unroll
                for (int q = 0; q < CH_TB; ++q) {
                    half4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (half_t)(acc[i][q][4 * gq + e] + (float)bv[e]);
                    *reinterpret_cast<half4_t*>(Cs + (64 * wave + 32 * q + l31) * CH_OSTR + co) = v;
                }
            }
    }
    __syncthreads();
    constexpr int OCH = CH_BA / 8;
    for (int id = tid; id < CH_BB * OCH; id += 512) {
        const int pl = id / OCH, ch = id - pl * OCH;
        const int64_t px = px0 + pl;
        const int co = a0 + ch * 8;
        const half8_t v = fz_ld_h8(Cs + pl * CH_OSTR + ch * 8);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
        if (g.temb != nullptr) {
            const half8_t t = fz_ld_h8(g.temb + (px / g.temb_group) * g.temb_stride + co);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += (float)t[e];
        }
        if (g.res != nullptr) {
            const half8_t r = fz_ld_h8(g.res + px * g.Cout + co);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
        }
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)f[e];
        fz_st_h8(g.y + px * g.Cout + co, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------
//                                                   host side
// ---------------------------------------------------------------------------------------------------------------
#ifdef CH_TIMING
extern "C" int fz_conv_halo_timing(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ch_timing), sizeof(long long) * 12) == hipSuccess ? FZ_OK : FZ_ERR_LAUNCH;
}
#endif

// shapes the kernel carries: whole image rows per 256-pixel tile, whole tiles per frame, 160-channel tiles, 64-channel chunks
int fz_conv_halo_ok(int n, int h, int w, int cin, int cout, int64_t temb_stride) {
    if (n <= 0 || w < 32 || w > 128 || CH_BB % w || (h * w) % CH_BB || cin % 64 || cin < 64 || cout % CH_BA) return 0;
    if (temb_stride % 8) return 0;
    const int np = ((CH_BB / w + 2) * (w + 2) + 7) / 8;
    return np <= 50 && (int64_t)n * h * w < (1ll << 31);
}

int fz_conv_halo_launch(const void* x, const void* wt, const void* bias, const void* temb, int64_t temb_stride, int64_t temb_group, const void* res,
                        void* y, int n, int h, int w, int cin, int cout, void* stream) {
    ChArgs g = {};
    g.x = (const half_t*)x;
    g.wt = (const half_t*)wt;
    g.bias = (const half_t*)bias;
    g.temb = (const half_t*)temb;
    g.res = (const half_t*)res;
    g.y = (half_t*)y;
    g.temb_stride = temb_stride;
    g.temb_group = temb_group;
    g.N = n; g.H = h; g.W = w; g.Cin = cin; g.Cout = cout;
    g.tiles_a = cout / CH_BA;
    g.np = ((CH_BB / w + 2) * (w + 2) + 7) / 8;
    g.hb_bytes = g.np * 1024;
    const size_t ring = (size_t)CH_NAS * CH_ASLOT + 2 * (size_t)g.hb_bytes, stage = (size_t)CH_BB * CH_OSTR * 2;
    const size_t lds = ring > stage ? ring : stage;
    if (lds > 160 * 1024) return FZ_ERR_UNSUPPORTED;
    const int64_t nwg = (int64_t)g.tiles_a * ((int64_t)n * h * w / CH_BB);
#ifndef FZ_EMU
    static std::atomic<uint64_t> attr_set_mask{0};  // LDS above 64 KB is an opt-in function attribute, per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FZ_ERR_LAUNCH;
    if (dev >= 64 || !(attr_set_mask.load(std::memory_order_relaxed) >> dev & 1)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return FZ_ERR_LAUNCH;
        if (dev < 64) attr_set_mask.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
#endif
    FZ_LAUNCH(conv_halo_kernel, dim3((unsigned)nwg), dim3(512), lds, stream, g);
    return fz_last_launch_status();
}
