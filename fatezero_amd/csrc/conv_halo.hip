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
//
// Forms of the one kernel (round 6):
//   * whole (fz_conv3x3's own choice from 160 tiles on) or in K SLICES (gridDim.y; slice ks contracts its 64-channel chunks under all nine taps into
//     igemm.hip's fp32 slab layout, igemm_reduce_kernel sums the slabs and applies bias / time embedding / residual) where the tiles alone do not fill
//     the chip: 8 frames x 32^2, the 16^2 and 8^2 levels;
//   * a tile = 256 / W whole image rows of one frame (W >= 32), one whole frame (16 x 16), or SEVERAL whole frames each with its own halo block (8 x 8:
//     four) -- the halo pixel index is (frame of the tile, halo row, halo column), every division a host-checked multiply-shift;
//   * conv_halo_kernel<4>: nearest-2x upsampling + 3x3 convolution as four 2x2 convolutions of the low-resolution input, one output parity per
//     blockIdx.z, on weights summed at pack time (fz_conv3x3_up2*);
//   * the eight XCDs either all walk every channel tile (pixels shared in L2) or split into groups along the channel tiles (weights the larger operand):
//     the host picks the split with the fewest bytes entering the L2s.
// The first halo chunk is fetched by all eight waves together with one more LDS-DMA piece carrying the tile's bias slice and time-embedding row, so the
// first MFMA issues 1.9 us after entry and the epilogue waits for no memory but the residual rows it requested before staging.
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
__device__ long long ch_timing3[16];   // wall stamps (10 ns) relative to kernel entry of workgroup 0: [0..2] pixel loader after setup / issue / wait; [4..6] weight loader same; [8..11] epilogue after sync / staging / sync / stores
__device__ long long ch_timing2[12];   // workgroup 0, thread 0: {wall, clock} at entry / loop start / loop end / exit; [8..9] min entry / max exit (wall) of all workgroups
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
    int64_t temb_stride;
    int temb_frames;       // frames per time-embedding row
    int N, H, W, Cin, Cout;
    int tiles_a;           // Cout / 160
    int xa;                // XCD groups along the channel tiles (1: every XCD walks all of them)
    int hb_bytes;          // bytes of one halo buffer (pieces of 1 KB)
    int np;                // LDS-DMA pieces of a halo tile
    float* part;           // split-K: fp32 partial slabs [ksplit][N H W][Cout] (igemm.hip's layout; igemm_reduce_kernel sums them and applies the tail), or null
    int ksplit;            // gridDim.y: slice ks contracts the 64-channel chunks [ks nchunk / ksplit, (ks + 1) nchunk / ksplit) under all nine taps
    int w_magic;           // the same for / W on the tile's 256 pixels (upsampling form: output rows)
    int fpt;               // frames per tile: 1 (a frame is one or several tiles), or 256 / (H W) whole frames (H W < 256: the 8 x 8 level), each with its own halo
    int fh_px;             // halo pixels per frame of the tile: (rows + 2)(W + 2), rows = 256 / W (fpt == 1) or H
    int fh_magic;          // ceil(2^20 / fh_px): halo pixel / fh_px == (halo pixel * fh_magic) >> 20 (host-checked)
    int w2_magic;          // ceil(65536 / (W + 2)): halo pixel / (W + 2) == (halo pixel * w2_magic) >> 16 for every halo pixel of a tile (host-checked)
};

// TAPS = 9: the 3x3 convolution.  TAPS = 4: nearest-2x upsampling + 3x3 convolution (UpsamplePseudo3D, resnet.py:145) as FOUR 2x2 convolutions
// of the low-resolution input, one per output parity (py, px) = blockIdx.z: U[y][x] = X[y >> 1][x >> 1] makes the three taps of a row hit only
// two input rows -- r - 1 and r for even output rows (weights w[0] and w[1] + w[2]), r and r + 1 for odd ones (w[0] + w[1], w[2]), columns
// alike: 4 instead of 9 MACs per output and input channel, on weights summed once at pack time (wt = [4 parities][Cout][4 taps][Cin]).
// TA = MFMA tiles of 32 output channels per workgroup: 5 (the 160 x 256 tile) or 2 (64 x 256: five times the workgroups at the same slab traffic
// where even eight K slices of the wide tile leave the chip idle -- the 8 x 8 level).
template <int TAPS, int TA = 5>
FZ_KERNEL void __launch_bounds__(512, 2) conv_halo_kernel(ChArgs g) {
    constexpr int CH_TA = TA, CH_BA = TA * 32;              // (shadow the namespace's wide-tile constants inside the kernel)
    constexpr int CH_ASLOT = CH_BA * 32 * 2, CH_OSTR = CH_BA + 8, APW = CH_BA / 32;   // APW: weight pieces (16 rows) per loader wave and tile
    constexpr int SPC = 2 * TAPS;                           // K-32 steps per 64-channel chunk
    const int py = TAPS == 4 ? (int)(blockIdx.z >> 1) : 0, pxp = TAPS == 4 ? (int)(blockIdx.z & 1) : 0;
    FZ_DYN_SMEM(raw);
    const int tid = threadIdx.x, wave = fz_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order (igemm.hip): blocks b, b + 8, .. share an XCD and get consecutive tiles; a-tile fastest: the two (or more) channel tiles of
    // one pixel tile sit next to each other and share its halo rows in L2
    const int nt = gridDim.x, bid = blockIdx.x;
    const int q8 = nt >> 3, r8 = nt & 7, xcd = bid & 7;
    int ta, tb;
    if (g.xa > 1) {
        // ... or, where the WEIGHTS are the larger operand (the 16^2 / 8^2 levels: 29.5 MB of them against 5-10 MB of pixels), the eight XCDs as
        // xa groups along the channel tiles x 8 / xa groups along the pixel tiles (host: the split with the fewest bytes entering the L2s --
        // with every XCD walking all channel tiles each of the eight L2s pulled the whole weight tensor: 235 MB per launch at 16 f x 16^2 x 1 280)
        const int idx = bid >> 3, xia = xcd % g.xa, xib = xcd / g.xa;
        const int ta_per = g.tiles_a / g.xa, tb_per = (nt / g.tiles_a) / (8 / g.xa);
        ta = xia * ta_per + idx % ta_per;
        tb = xib * tb_per + idx / ta_per;
    } else {
        const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        ta = lid % g.tiles_a;
        tb = lid / g.tiles_a;
    }
    const int a0 = ta * CH_BA;
    const int W = g.W, W2 = W + 2;
    const int64_t px0 = (int64_t)tb * CH_BB;                 // first output pixel of the tile (flattened n, y, x)
    const int hw = g.H * W;
    const int fn = (int)(px0 / hw), y0 = (int)((px0 - (int64_t)fn * hw) / W);
    unsigned char* const Aring = raw;
    unsigned char* const Hbuf = raw + CH_NAS * CH_ASLOT;
    unsigned char* const Cpre = Hbuf + 2 * g.hb_bytes;      // 1 KB: bias | time-embedding row of the tile (beyond the epilogue's staging area)
    uint32_t* const Ptab = reinterpret_cast<uint32_t*>(Cpre + 1024);   // np x 64 words: per (halo piece, lane) the byte offset of its chunk-0 source from the
                                                                        // frame's base, ~0 = the zero page (written in the prologue, read by the pixel loaders)
    const int nchunk_all = g.Cin >> 6, ks = blockIdx.y;
    const int c0 = ks * nchunk_all / g.ksplit, nchunk = (ks + 1) * nchunk_all / g.ksplit - c0;   // this slice's chunks: c0 .. c0 + nchunk
    const int nstep = nchunk * SPC;                         // a step = (chunk, tap, K half)
    const char* zero = reinterpret_cast<const char*>(ch_zero_page);

#ifdef CH_TIMING
    long long tacc[4] = {0, 0, 0, 0};
    const long long tstart = clock64();
    const long long w_entry = wall_clock64();
    long long w_ls = 0, c_ls = 0;
#endif
    f32x16 acc[CH_TA][CH_TB];
#pragma unroll
    for (int i = 0; i < CH_TA; ++i)
#pragma unroll
        for (int q = 0; q < CH_TB; ++q) acc[i][q] = fz_zero_f16v();

    // Halo piece p covers halo pixels [8 p, 8 p + 8): lane = (pixel 8 p + lane / 8, physical 16-byte chunk lane % 8) fetches logical chunk
    // (lane % 8) ^ ((pixel >> 1) & 7) of that pixel -- the swizzle the consumers' ds_read_b128 undo -- or zeros outside the image / the tile.
    // The per-lane source of a piece for chunk 0 (null: the zero page).  Cheap on purpose (the first version -- a division and 64-bit index
    // arithmetic per piece, all 25 pieces in the two pixel loaders -- took 3.6 us of set-up + 1.5 us of issue in front of the first MFMA:
    // profiles/r06_conv_halo_ab.txt): a multiply-shift for the row, 32-bit offsets from the frame's base.
    const half_t* const xf = g.x + (int64_t)fn * hw * g.Cin;
    auto piece_src = [&](int p) __attribute__((always_inline)) -> const char* {
        const int hp = 8 * p + (lane >> 3), pc = lane & 7;
        const int fi = (hp * g.fh_magic) >> 20, hr = hp - fi * g.fh_px;           // frame of the tile, halo pixel inside that frame's block
        const int yy = (hr * g.w2_magic) >> 16, xx = hr - yy * W2;
        const int iy = y0 + yy - 1, ix = xx - 1;
        const bool ok = fi < g.fpt && iy >= 0 && iy < g.H && ix >= 0 && ix < W;    // (yy < rows + 2 by construction: hr < fh_px)
        const int lc = pc ^ ((hp >> 1) & 7);
        return ok ? reinterpret_cast<const char*>(xf) + (uint32_t)((((fi * g.H + iy) * W + ix) * g.Cin + lc * 8) * 2) : nullptr;
    };
    // chunk 0 of the halo tile: ALL eight waves fetch it (pieces wave, wave + 8, ..: at most 7 each), then take their roles
    {
#pragma unroll 1   /* (rolled: straight-line code that runs once pays for its instruction fetches) */
        for (int p = wave; p < g.np; p += 8) {
            const char* s0 = piece_src(p);
            Ptab[p * 64 + lane] = s0 != nullptr ? (uint32_t)(s0 - reinterpret_cast<const char*>(xf)) : 0xffffffffu;
            fz_glds16(s0 != nullptr ? s0 + c0 * 128 : zero + lane * 16, Hbuf + (c0 & 1) * g.hb_bytes + p * 1024);
        }
        fz_lds_fence();                                     // (the table's words are in LDS before this wave reaches B(0))
        // one more piece behind the halo buffers: bias (the first CH_BA / 8 lanes) and the tile's time-embedding row (the next) for the epilogue
        if (wave == 3) {
            const char* s0 = zero + lane * 16;
            constexpr int BL = CH_BA / 8;   // 16-byte lanes of a channel-tile row
            if (lane < BL && g.bias != nullptr) s0 = reinterpret_cast<const char*>(g.bias + a0 + lane * 8);
            if (lane >= BL && lane < 2 * BL && g.temb != nullptr)
                s0 = reinterpret_cast<const char*>(g.temb + (int64_t)(fn / g.temb_frames) * g.temb_stride + a0 + (lane - BL) * 8);
            fz_glds16(s0, Cpre);
        }
    }

    if (wave >= 6) {
        // ================================================ pixel loaders ===================================================================
        // PPS pieces per step (two of the nine-tap form's 18 steps per chunk, five of the upsampler form's 8: room for the <= 25 pieces of the next
        // chunk's tile in the chunk's first SPC - 3 steps), their sources worked out on the spot: ~25 VALU per piece, in a wave that otherwise
        // waits at barriers (a per-piece table cost 3.6 us of set-up in front of the first MFMA).
        const int bl = wave - 6;
        // (sources from the prologue's table: worked out on the spot -- a dozen integer multiplies and a 64-bit select per piece -- the two pieces
        //  of a step kept a loader wave busy ~600 cycles per step, the floor of the narrow tile's step and next to the wide one's 760)
        auto fire = [&](int p, int chunk) __attribute__((always_inline)) {
            const uint32_t off = Ptab[p * 64 + lane];
            fz_glds16(off != 0xffffffffu ? reinterpret_cast<const char*>(xf) + off + chunk * 128 : zero + lane * 16, Hbuf + (chunk & 1) * g.hb_bytes + p * 1024);
        };
#ifdef CH_TIMING
        const long long w_p0 = wall_clock64(), w_p1 = w_p0;
#endif
        fz_wait_vm0();
#ifdef CH_TIMING
        if (blockIdx.x == 0 && tid == 384) { ch_timing3[0] = w_p0 - w_entry; ch_timing3[1] = w_p1 - w_entry; ch_timing3[2] = wall_clock64() - w_entry; }
#endif
        fz_barrier_raw();                                   // B(0)
        constexpr int PPS = TAPS == 9 ? 2 : 5;              // pieces per step and loader: (SPC - 3) PPS >= 25
        static_assert((SPC - 3) * PPS >= 25, "the next chunk's halo tile must fit its fire steps");
        for (int c = 0; c < nchunk; ++c) {
            const bool more = c + 1 < nchunk;
#pragma unroll 1
            for (int s = 0; s < SPC; ++s) {
#ifndef CH_NO_DMA   /* (trial flag: the steady state without a single LDS-DMA -- results are garbage) */
                if (more && s < SPC - 3) {
#pragma unroll
                    for (int u = 0; u < PPS; ++u) {         // this loader's pieces bl, bl + 2, ..: PPS per step
                        const int p = bl + 2 * (PPS * s + u);
                        if (p < g.np) fire(p, c0 + c + 1);
                    }
                }
#endif
                if (s == SPC - 2) {
                    CH_T0();
                    fz_wait_vm0();                          // the next chunk's tile is complete in front of the chunk's last barrier
                    CH_T1(0);
                }
                {
                    CH_T0();
                    fz_barrier_raw();                       // B(SPC c + s + 1)
                    CH_T1(1);
                }
            }
        }
    } else if (wave >= 4) {
        // ================================================ weight loaders ==================================================================
        // tile (chunk c, tap t, half h) = rows a0 .. a0 + 160 of wt, 32 halves at k = t Cin + 64 c + 32 h; piece p = rows [16 p, 16 p + 16) x 4
        // chunks of 16 B; lane = (row 16 p + lane / 4, physical chunk lane % 4) fetches logical chunk (lane % 4) ^ ((row >> 2) & 3)
        const int al = wave - 4;
        uint32_t aoff[APW];
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int row = 16 * (al + 2 * i) + (lane >> 2);
            const int lc = (lane & 3) ^ ((row >> 2) & 3);
            aoff[i] = (uint32_t)(((int64_t)row * TAPS * g.Cin + lc * 8) * 2);
        }
        const char* const a_tile = reinterpret_cast<const char*>(g.wt + ((int64_t)(TAPS == 4 ? blockIdx.z : 0) * g.Cout + a0) * TAPS * g.Cin);
        auto issue = [&](int j) {                           // step j = SPC c + 2 t + h
            const int c = j / SPC, r = j - SPC * c, t = r >> 1, h = r & 1;
            const char* base = a_tile + ((int64_t)t * g.Cin + 64 * (c0 + c) + 32 * h) * 2;
#pragma unroll
            for (int i = 0; i < APW; ++i) fz_glds16_so(base, aoff[i], Aring + (j & 3) * CH_ASLOT + (al + 2 * i) * 1024);
        };
#ifdef CH_TIMING
        const long long w_a0 = wall_clock64();
#endif
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nstep) issue(j);
#ifdef CH_TIMING
        const long long w_a1 = wall_clock64();
#endif
        if (nstep > 2) {
            fz_wait_vm<APW>();                                // tiles 0, 1 landed; tile 2 in flight
        } else {
            fz_wait_vm0();
        }
#ifdef CH_TIMING
        if (blockIdx.x == 0 && tid == 256) { ch_timing3[4] = w_a0 - w_entry; ch_timing3[5] = w_a1 - w_entry; ch_timing3[6] = wall_clock64() - w_entry; }
#endif
        fz_barrier_raw();                                   // B(0)
        for (int j = 0; j < nstep; ++j) {
            {
                CH_T0();
                if (j + 3 < nstep) {
#ifndef CH_NO_DMA
                    issue(j + 3);                           // into the slot tile j - 1 left (all its fragments were read before B(j))
#endif
                    CH_T1(2);
                    fz_wait_vm<APW>();                        // tile j + 2 landed
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
#if !defined(FZ_EMU) && defined(CH_CONSUMER_PRIO)
        __builtin_amdgcn_s_setprio(CH_CONSUMER_PRIO);        // (trial: the MFMA waves ahead of their SIMD partners' address arithmetic in the issue arbiter)
#endif
        // weight fragment (tile i, k sub-step kk): row 32 i + l31 of the slot, chunk (2 kk + hi) ^ ((row >> 2) & 3); a row = 64 bytes
        const int arow = l31 * 64, asw = (l31 >> 2) & 3;
        // pixel fragment (tile q, tap, chunk lc): halo pixel hp0[q] + tap shift, chunk lc ^ ((hp >> 1) & 7); a pixel = 128 bytes
        int hp0[CH_TB];
#pragma unroll
        for (int q = 0; q < CH_TB; ++q) {
            const int pl = 64 * wave + 32 * q + l31;
            const int fi = g.fpt > 1 ? pl / hw : 0, pr = pl - fi * hw, yl = pr / W, xl = pr - yl * W;
            hp0[q] = fi * g.fh_px + (yl + 1) * W2 + xl + 1;
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
        int rs = 0, rt = 0, rc = c0;                         // read cursor: k sub-step counter (4 per tap), tap, chunk (its parity = the halo buffer)
        auto next_addr = [&]() -> Fa {                      // the cursor's addresses; then one sub-step on
            const int rkk = rs & 1, rh = (rs >> 1) & 1, rj = rs >> 1;
            const int ky = TAPS == 9 ? rt / 3 : (rt >> 1) + py, kx = TAPS == 9 ? rt - 3 * ky : (rt & 1) + pxp, sh = (ky - 1) * W2 + (kx - 1);
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
            const int rt2 = rt + adv, wrap = rt2 == TAPS;
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
#ifndef CH_NO_READS   /* (trial: the MFMAs alone, on stale fragments) */
            rd(afn, bfn, fn);
#endif
#ifdef CH_NO_MFMA     /* (trial: the reads alone; each fragment is consumed by an empty asm) */
#pragma unroll
            for (int i = 0; i < CH_TA; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
            for (int q = 0; q < CH_TB; ++q) asm volatile("" ::"v"(bf[q]));
#else
#pragma unroll
            for (int i = 0; i < CH_TA; ++i)
#pragma unroll
                for (int q = 0; q < CH_TB; ++q) acc[i][q] = fz_mfma_32x32x16_f16(af[i], bf[q], acc[i][q]);
#endif
#ifndef FZ_EMU
            constexpr int NR = CH_TA + CH_TB, NM = CH_TA * CH_TB, NP = NR < NM ? NR : NM;   // reads, MFMAs, interleaved pairs
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read
            }
            if constexpr (NM > NP) __builtin_amdgcn_sched_group_barrier(0x008, NM - NP, 0);
#endif
        };
        fz_wait_vm0();                                      // (this wave's pieces of the first halo chunk)
        fz_barrier_raw();                                   // B(0)
#ifdef CH_TIMING
        w_ls = wall_clock64(); c_ls = clock64();
#endif
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
#ifdef CH_TIMING_CONSUMER   /* (two s_memtime + a full lgkmcnt drain per step: ~100 cycles of the consumers' 640) */
            {
                CH_T0();
                fz_barrier_raw();                           // B(j + 1)
                CH_T1(1);
            }
#else
            fz_barrier_raw();                               // B(j + 1)
#endif
        }
    }
#ifdef CH_TIMING
    const long long w_le = wall_clock64(), c_le = clock64();
    tacc[3] = clock64() - tstart;
    if (blockIdx.x == 0 && (tid == 0 || tid == 256 || tid == 384))
        for (int i = 0; i < 4; ++i) ch_timing[tid == 0 ? 0 : (tid == 256 ? 1 : 2)][i] = tacc[i];
#endif

    // ---- epilogue (igemm.hip's arithmetic): + bias (fp32) -> fp16 tile through LDS [256 pixels][160 + 8] -> (+ temb row) (+ res) -> full-row
    // 16-byte stores.  What is fetched from memory is on its way before it is needed: bias and the tile's time-embedding row came into LDS
    // with the first halo chunk (a tile lies inside one frame: one row for all of it), the residual rows are requested before the staging
    // (loader waves) / behind it (consumers: their accumulators are dead then) -- the first version's epilogue was 5.2 us, 2.8 of them the
    // staging waiting for its bias loads.
    if (g.part != nullptr) {   // split-K: this slice's fp32 tile as it is (igemm.hip's slab layout and store form); the tail kernel does the rest
        if (wave < 4) {
            float* const P = g.part + ((int64_t)ks * g.N * hw + px0) * g.Cout + a0;
#pragma unroll
            for (int i = 0; i < CH_TA; ++i)
#pragma unroll
                for (int q = 0; q < CH_TB; ++q) {
                    float* const prow = P + (uint32_t)((64 * wave + 32 * q + l31) * g.Cout + i * 32 + 4 * hi);
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][q][4 * gq + e];
                        *reinterpret_cast<f32x4*>(prow + 8 * gq) = v;
                    }
                }
        }
        return;
    }
    constexpr int OCH = CH_BA / 8, NOUT = CH_BB * OCH / 512;
    static_assert(CH_BB * OCH % 512 == 0, "the output loop's trip count");
    const half_t* const res_t = g.res != nullptr ? g.res + px0 * g.Cout + a0 : nullptr;
    // output row of the tile's pixel pl, in halves from y_t: the pixel itself -- or, upsampling, output pixel (2 y + py, 2 x + px) of the frame
    half_t* const y_t = TAPS == 9 ? g.y + px0 * g.Cout + a0
                                  : g.y + ((((int64_t)fn * 2 * g.H + 2 * y0 + py) * 2 * W + pxp)) * g.Cout + a0;
    auto out_off = [&](int pl) __attribute__((always_inline)) -> uint32_t {
        if constexpr (TAPS == 9) return (uint32_t)(pl * g.Cout);
        const int yl = (pl * g.w_magic) >> 16, xl = pl - yl * W;   // (row of the TILE: with several frames per tile, frame fi's rows start at fi H -- and so do its output rows, at 2 fi H)
        return (uint32_t)((4 * yl * W + 2 * xl) * g.Cout);
    };
    half8_t rv[NOUT];
    auto load_res = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const int id = tid + 512 * k, pl = id / OCH, ch = id - pl * OCH;
            rv[k] = fz_ld_h8(res_t + (uint32_t)(pl * g.Cout + ch * 8));
        }
    };
    if (wave >= 4 && res_t != nullptr) load_res();
    __syncthreads();
#ifdef CH_TIMING
    const long long w_e0 = wall_clock64();
#endif
    half_t* Cs = reinterpret_cast<half_t*>(raw);
    const half_t* const bias_l = reinterpret_cast<const half_t*>(Cpre);
    const half_t* const temb_l = reinterpret_cast<const half_t*>(Cpre) + CH_BA;
    if (wave < 4) {
        half4_t bvs[CH_TA][4];   // (all of them first: read one by one, each ds_read was waited for on the spot -- 4 TA LDS latencies in a row)
#pragma unroll
        for (int i = 0; i < CH_TA; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) bvs[i][gq] = *reinterpret_cast<const half4_t*>(bias_l + i * 32 + 8 * gq + 4 * hi);
        FZ_SCHED_FENCE();
#pragma unroll
        for (int i = 0; i < CH_TA; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int co = i * 32 + 8 * gq + 4 * hi;
                const half4_t bv = bvs[i][gq];                                        // (zeros without a bias: + 0.0f changes nothing)
#pragma unroll
                for (int q = 0; q < CH_TB; ++q) {
                    half4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (half_t)(acc[i][q][4 * gq + e] + (float)bv[e]);
                    *reinterpret_cast<half4_t*>(Cs + (64 * wave + 32 * q + l31) * CH_OSTR + co) = v;
                }
            }
        FZ_SCHED_FENCE();
        if (res_t != nullptr) load_res();
    }
#ifdef CH_TIMING
    const long long w_e1 = wall_clock64();
#endif
    __syncthreads();
#ifdef CH_TIMING
    const long long w_e2 = wall_clock64();
#endif
    half8_t cv[NOUT], tv[NOUT];   // (LDS reads first, arithmetic + stores behind them)
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        const int id = tid + 512 * k, pl = id / OCH, ch = id - pl * OCH;
        cv[k] = fz_ld_h8(Cs + pl * CH_OSTR + ch * 8);
        if (g.temb != nullptr) tv[k] = fz_ld_h8(temb_l + ch * 8);
    }
    FZ_SCHED_FENCE();
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        const int id = tid + 512 * k, pl = id / OCH, ch = id - pl * OCH;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)cv[k][e];
        if (g.temb != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += (float)tv[k][e];
        }
        if (res_t != nullptr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += (float)rv[k][e];
        }
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)f[e];
        fz_st_h8(y_t + out_off(pl) + ch * 8, o);
    }
#ifdef CH_TIMING
    if (tid == 0) {
        const long long w_x = wall_clock64(), c_x = clock64();
        atomicMin((unsigned long long*)&ch_timing2[8], (unsigned long long)w_entry);
        atomicMax((unsigned long long*)&ch_timing2[9], (unsigned long long)w_x);
        if (blockIdx.x == 0) {
            ch_timing3[8] = w_e0 - w_entry; ch_timing3[9] = w_e1 - w_entry; ch_timing3[10] = w_e2 - w_entry; ch_timing3[11] = w_x - w_entry;
            ch_timing2[0] = w_entry; ch_timing2[1] = tstart; ch_timing2[2] = w_ls; ch_timing2[3] = c_ls;
            ch_timing2[4] = w_le; ch_timing2[5] = c_le; ch_timing2[6] = w_x; ch_timing2[7] = c_x;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
//                                                   host side
// ---------------------------------------------------------------------------------------------------------------
#ifdef CH_TIMING
extern "C" int fz_conv_halo_timing3(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ch_timing3), sizeof(long long) * 16) == hipSuccess ? FZ_OK : FZ_ERR_LAUNCH;
}
extern "C" int fz_conv_halo_timing2(long long* out, int reset) {
    if (reset) {
        long long z[12] = {0}; z[8] = 0x7fffffffffffffffll;
        return hipMemcpyToSymbol(HIP_SYMBOL(ch_timing2), z, sizeof(z)) == hipSuccess ? FZ_OK : FZ_ERR_LAUNCH;
    }
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ch_timing2), sizeof(long long) * 12) == hipSuccess ? FZ_OK : FZ_ERR_LAUNCH;
}
extern "C" int fz_conv_halo_timing(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ch_timing), sizeof(long long) * 12) == hipSuccess ? FZ_OK : FZ_ERR_LAUNCH;
}
#endif

// shapes the kernel carries: whole image rows per 256-pixel tile, whole tiles per frame, 160-channel tiles, 64-channel chunks
static int conv_halo_ok_ba(int n, int h, int w, int cin, int cout, int64_t temb_stride, int ba) {
    if (n <= 0 || h <= 0 || w < 8 || w > 128 || CH_BB % w || cin % 64 || cin < 64 || cout % ba) return 0;
    if (temb_stride % 8) return 0;
    const int hw = h * w;
    int np;
    if (hw % CH_BB == 0) {                       // a frame is one or several tiles
        np = ((CH_BB / w + 2) * (w + 2) + 7) / 8;
    } else {                                     // several whole frames per tile, each with its own halo
        if (CH_BB % hw || n % (CH_BB / hw)) return 0;
        np = ((CH_BB / hw) * (h + 2) * (w + 2) + 7) / 8;
    }
    return np <= 50 && (int64_t)n * h * w < (1ll << 31) && (int64_t)h * w * cin * 2 * (hw < CH_BB ? CH_BB / hw : 1) < (1ll << 31);
}
int fz_conv_halo_ok(int n, int h, int w, int cin, int cout, int64_t temb_stride) { return conv_halo_ok_ba(n, h, w, cin, cout, temb_stride, CH_BA); }
// the narrow form (64 output channels per workgroup, conv_halo_kernel<9, 2>)
int fz_conv_halo64_ok(int n, int h, int w, int cin, int cout, int64_t temb_stride) { return conv_halo_ok_ba(n, h, w, cin, cout, temb_stride, 64); }

// part / ksplit: null / 1 = the whole convolution with its tail; else the fp32 slabs of ksplit K slices (the caller runs igemm_reduce_kernel)
static int conv_halo_launch_taps(int taps, int ta, const void* x, const void* wt, const void* bias, const void* temb, int64_t temb_stride, int64_t temb_group,
                                 const void* res, void* y, int n, int h, int w, int cin, int cout, float* part, int ksplit, void* stream) {
    ChArgs g = {};
    g.x = (const half_t*)x;
    g.wt = (const half_t*)wt;
    g.bias = (const half_t*)bias;
    g.temb = (const half_t*)temb;
    g.res = (const half_t*)res;
    g.y = (half_t*)y;
    g.temb_stride = temb_stride;
    if (temb != nullptr && (temb_group <= 0 || temb_group % ((int64_t)h * w))) return FZ_ERR_UNSUPPORTED;   // whole frames per row
    g.temb_frames = temb != nullptr ? (int)(temb_group / ((int64_t)h * w)) : 1;
    g.N = n; g.H = h; g.W = w; g.Cin = cin; g.Cout = cout;
    g.ksplit = ksplit > 1 ? ksplit : 1;
    g.part = g.ksplit > 1 ? part : nullptr;
    if (g.ksplit > 1 && (part == nullptr || g.ksplit > cin / 64)) return FZ_ERR_BAD_ARG;
    const int ba = ta * 32, aslot = ba * 32 * 2, ostr = ba + 8;   // (the kernel's per-instantiation constants)
    if ((ta != 5 && ta != 2) || (taps == 4 && ta != 5) || cout % ba) return FZ_ERR_UNSUPPORTED;
    g.tiles_a = cout / ba;
    g.fpt = h * w < CH_BB ? CH_BB / (h * w) : 1;
    g.fh_px = ((g.fpt > 1 ? h : CH_BB / w) + 2) * (w + 2);
    g.fh_magic = (1 << 20) / g.fh_px + 1;
    g.np = (g.fpt * g.fh_px + 7) / 8;
    g.hb_bytes = g.np * 1024;
    for (int hp = 0; hp < 8 * g.np; ++hp)
        if (((hp * g.fh_magic) >> 20) != hp / g.fh_px) return FZ_ERR_UNSUPPORTED;
    if (temb != nullptr && g.temb_frames % g.fpt) return FZ_ERR_UNSUPPORTED;      // one time-embedding row per tile
    g.w2_magic = 65536 / (w + 2) + 1;
    for (int hp = 0; hp < g.fh_px; ++hp)
        if (((hp * g.w2_magic) >> 16) != hp / (w + 2)) return FZ_ERR_UNSUPPORTED;   // (never for the widths fz_conv_halo_ok admits)
    const size_t ring = (size_t)CH_NAS * aslot + 2 * (size_t)g.hb_bytes + 1024 + (size_t)g.np * 256, stage = (size_t)CH_BB * ostr * 2;
    if ((size_t)CH_NAS * aslot + 2 * (size_t)g.hb_bytes < stage) return FZ_ERR_UNSUPPORTED;   // the bias / temb piece must lie beyond the staging area
    const size_t lds = ring > stage ? ring : stage;
    if (lds > 160 * 1024) return FZ_ERR_UNSUPPORTED;
    const int64_t nwg = (int64_t)g.tiles_a * ((int64_t)n * h * w / CH_BB);
    {   // bytes entering the eight L2s: every XCD group along the pixel tiles pulls its channel tiles' weights, every group along the channel tiles its pixels
        const int64_t tiles_b = (int64_t)n * h * w / CH_BB;
        const double wbytes = 2.0 * cout * taps * cin, xbytes = 2.0 * n * h * w * cin * (double)(g.fpt * g.fh_px) / CH_BB;
        g.xa = 1;
        if (nwg % 8 == 0) {
            const double base = wbytes * 8 + xbytes;   // (xa = 1: the walk of the first version)
            double best = 0.9 * base;                   // another split only where it saves >= 10 %
            for (int xa = 2; xa <= 8; xa *= 2) {
                if (g.tiles_a % xa || tiles_b % (8 / xa)) continue;
                const double cost = wbytes * (8 / xa) + xbytes * xa;
                if (cost < best) {
                    best = cost;
                    g.xa = xa;
                }
            }
        }
    }
    g.w_magic = 65536 / w + 1;
    for (int pl = 0; pl < CH_BB; ++pl)
        if (((pl * g.w_magic) >> 16) != pl / w) return FZ_ERR_UNSUPPORTED;
    if (taps == 4 && (g.ksplit > 1 || temb != nullptr || res != nullptr)) return FZ_ERR_UNSUPPORTED;   // the upsampler's convolution: bias only
    void (*kern)(ChArgs) = taps == 4 ? &conv_halo_kernel<4> : (ta == 2 ? &conv_halo_kernel<9, 2> : &conv_halo_kernel<9>);
#ifndef FZ_EMU
    static std::atomic<uint64_t> attr_set_mask[3] = {{0}, {0}, {0}};  // LDS above 64 KB is an opt-in function attribute, per device and instantiation
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FZ_ERR_LAUNCH;
    std::atomic<uint64_t>& mask = attr_set_mask[taps == 4 ? 1 : (ta == 2 ? 2 : 0)];
    if (dev >= 64 || !(mask.load(std::memory_order_relaxed) >> dev & 1)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return FZ_ERR_LAUNCH;
        if (dev < 64) mask.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
#endif
    if (taps == 4) {
        FZ_LAUNCH(conv_halo_kernel<4>, dim3((unsigned)nwg, 1, 4), dim3(512), lds, stream, g);
    } else if (ta == 2) {
        FZ_LAUNCH((conv_halo_kernel<9, 2>), dim3((unsigned)nwg, (unsigned)g.ksplit), dim3(512), lds, stream, g);
    } else {
        FZ_LAUNCH(conv_halo_kernel<9>, dim3((unsigned)nwg, (unsigned)g.ksplit), dim3(512), lds, stream, g);
    }
    return fz_last_launch_status();
}

int fz_conv_halo_launch(const void* x, const void* wt, const void* bias, const void* temb, int64_t temb_stride, int64_t temb_group, const void* res,
                        void* y, int n, int h, int w, int cin, int cout, float* part, int ksplit, void* stream) {
    return conv_halo_launch_taps(9, 5, x, wt, bias, temb, temb_stride, temb_group, res, y, n, h, w, cin, cout, part, ksplit, stream);
}
int fz_conv_halo64_launch(const void* x, const void* wt, const void* bias, const void* temb, int64_t temb_stride, int64_t temb_group, const void* res,
                          void* y, int n, int h, int w, int cin, int cout, float* part, int ksplit, void* stream) {
    return conv_halo_launch_taps(9, 2, x, wt, bias, temb, temb_stride, temb_group, res, y, n, h, w, cin, cout, part, ksplit, stream);
}

// ---- nearest-2x upsampling + 3x3 convolution as four 2x2 convolutions of the input (conv_halo_kernel<4>) ---------------------------------------
// wt_up[z = 2 py + px][co][t = 2 ty + tx][ci] = sum of w[co][ky][kx][ci] over ky in rows(py, ty), kx in rows(px, tx) with
// rows(0, 0) = {0}, rows(0, 1) = {1, 2}, rows(1, 0) = {0, 1}, rows(1, 1) = {2}; summed in fp32 in (ky, kx) order, rounded once to fp16.
FZ_KERNEL void __launch_bounds__(256) conv_up2_pack_kernel(const half_t* w9, half_t* wu, int cout, int cin) {
    const int64_t total = (int64_t)16 * cout * cin;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int ci = (int)(id % cin);
        int64_t r = id / cin;
        const int t = (int)(r & 3);
        r >>= 2;
        const int co = (int)(r % cout), z = (int)(r / cout);
        const int py = z >> 1, px = z & 1, ty = t >> 1, tx = t & 1;
        const int ky0 = py == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = py == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
        const int kx0 = px == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = px == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
        float acc = 0.0f;
        for (int ky = ky0; ky <= ky1; ++ky)
            for (int kx = kx0; kx <= kx1; ++kx) acc += (float)w9[((int64_t)co * 9 + ky * 3 + kx) * cin + ci];
        wu[id] = (half_t)acc;
    }
}

// C ABI (include/fatezero_hip.h)
extern "C" int fz_conv3x3_up2_ok(int n, int h, int w, int cin, int cout) { return fz_conv_halo_ok(n, h, w, cin, cout, 0); }
// where it is FASTER than fz_conv3x3(upsample = 1) on MI355X (profiles/r06_conv_up2_ab.txt): from 128 workgroups on (4 parities x tiles) -- the
// 8-frame launch of the 8 x 8 level (64 workgroups of 160 steps each) is the one UNet shape below
extern "C" int fz_conv3x3_up2_preferred(int n, int h, int w, int cin, int cout) {
    if (!fz_conv_halo_ok(n, h, w, cin, cout, 0)) return 0;
    return 4 * (int64_t)(cout / CH_BA) * ((int64_t)n * h * w / CH_BB) >= 128 ? 1 : 0;
}
extern "C" int64_t fz_conv3x3_up2_pack_halves(int cin, int cout) { return (int64_t)16 * cin * cout; }
extern "C" int fz_conv3x3_up2_pack(const void* wt, void* wt_up, int cin, int cout, void* stream) {
    if (!wt || !wt_up || cin <= 0 || cout <= 0) return FZ_ERR_BAD_ARG;
    const int64_t total = (int64_t)16 * cin * cout;
    FZ_LAUNCH(conv_up2_pack_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, stream, (const half_t*)wt,
              (half_t*)wt_up, cout, cin);
    return fz_last_launch_status();
}
extern "C" int fz_conv3x3_up2(const void* x, const void* wt_up, const void* bias, void* y, int n, int h, int w, int cin, int cout, void* stream) {
    if (!x || !wt_up || !y) return FZ_ERR_BAD_ARG;
    if (!fz_conv_halo_ok(n, h, w, cin, cout, 0)) return FZ_ERR_UNSUPPORTED;
    return conv_halo_launch_taps(4, 5, x, wt_up, bias, nullptr, 0, 0, nullptr, y, n, h, w, cin, cout, nullptr, 1, stream);
}
