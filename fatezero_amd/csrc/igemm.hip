// igemm.hip -- ONE MFMA implicit-GEMM kernel for every GEMM-shaped op of the pseudo-3D UNet:
//   * nn.Linear / 1x1 nn.Conv2d (models/attention.py:64-66,91-93 proj_in / proj_out, :199,216 to_q/k/v/out of
//     CrossAttention, :232 FeedForward GEGLU [3P diffusers], resnet.py:290 time_emb_proj, the 1x1 shortcuts)   -> fz_gemm
//   * the 3x3 spatial convolution of PseudoConv3d.forward (resnet.py:57-64; stride 2: :203; nearest-2x: :145) -> fz_conv3x3
//   * the k=3 temporal Conv1d pair of LoRALinearLayer (lora.py:31-54)                                        -> fz_temporal_conv3
// GEMM view:  D[a][b] = sum_k A[a][k] * B[b][k];  A rows are plain dense rows (weights: a = cout, k = tap * Cin + ci),
// B rows are gathered (pixel rows of x, shifted per tap, zero outside the image / clip); the output is y[b][a] (a
// contiguous) -- token-major activations on both sides, so there is never a layout conversion.  fz_gemm's transposed
// form (V^T = Wv X^T, the attention kernels' V operand) is the same kernel with the operands swapped.
//
// Structure (cdna_hip_programming.md section 5, "glds vs register staging"):  WA x WB waves, each owning a
// (TA*32) x (TB*32) block of 32x32x16 f16 MFMA tiles; K step = 64 halves of one tap.  Both operand tiles go HBM/L2 -> LDS
// by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write pass), double buffered, ONE barrier per K step:
//     wait vmcnt(0); barrier;  issue tile t+1 -> buf[~cur];  ds_read_b128 + MFMA on buf[cur]
// LDS-DMA writes lane-linear, so a tile is [row][8 chunks of 16 B] unpadded; the chunk index is XOR-swizzled with
// (row >> 1) & 7 on the per-lane SOURCE address and again on the fragment read (conflict-free ds_read_b128, rule 21).
// Zero padding (image border, clip ends, K tail) = lanes pointing at a device-global page of zeros.  (The same DMA through a
// buffer descriptor -- buffer_load_dwordx4 ... offen lds with an SGPR K offset and out-of-range lanes as zeros -- needs no
// VALU at all for addresses but measured 20-40 % SLOWER on MI355X: twice the VMEM instruction count in the counters,
// profiles/r02_pmc_conv64_bufferdma.json.)
// Epilogue: + bias (fp32, in registers) [* GEGLU gate] -> fp16 tile through LDS -> (+ temb[b]) (+ res) (+ res2) -> full-row
// 16-byte stores.  Split-K (small pyramid levels: too few tiles to fill 256 CUs) writes fp32 partial slabs that
// igemm_reduce_kernel combines with the same tail.
#include "fz_rt.h"
#include <type_traits>
#include <atomic>
#include <utility>
#include <stdlib.h>
#include "../../include/fatezero_hip.h"

// PP (template parameter of the kernel): 0 = ring loop; bit 0 = phase-interleaved ("ping-pong") loop, and its trial forms (only
// instantiated with -DFZ_IGEMM_TRIALS, scripts/igemm_ab.py): bit 1 = no s_setprio around the MFMA clusters, bit 2 = the two wave
// groups NOT staggered, bit 3 = the address VALU of the next phase runs at the START of that phase (in the read half, the round-3
// v1 form) instead of inside the MFMA cluster before it
#ifdef FZ_IGEMM_TRIALS
__attribute__((weak)) int fz_igemm_trial_no_pp = 0;
extern "C" { __attribute__((weak)) int fz_igemm_trial_no_kg2 = 0; __attribute__((weak)) int fz_igemm_trial_no_halo = 0; __attribute__((weak)) int fz_igemm_trial_no_halo_split = 0; }  // same-process A/B of the K-group substitution (scripts/ab_lib_flag.py)
__attribute__((weak)) int fz_igemm_trial_pp_splitk_min = 0;  // > 0: substitute under split-K as well when a K slice has at least this many K-64 steps
#endif
// The shipped library reads NO environment variable: the A/B switches of rounds 3-4 (tile order, K slices on XCDs, split-K launch cost)
// exist only in builds with -DFZ_IGEMM_TUNING (scripts/build_variant.sh), where they are read once per process.
#ifdef FZ_IGEMM_TUNING
#define FZ_TUNING_FLAG(name) (getenv(name) != nullptr)
#else
#define FZ_TUNING_FLAG(name) false
#endif
#ifndef FZ_A_AUX
#define FZ_A_AUX 0   /* cache policy of the weight operand's LDS-DMA (trial builds: 2 = nt, 16 = sc1: served by L2, the vector L1 left to the pixel rows) */
#endif
#define FZ_PP_ON 1
#define FZ_PP_NOPRIO 2
#define FZ_PP_NOSTAGGER 4
#define FZ_PP_PREP_IN_R 8
#define FZ_PP_K32 16  /* a phase is a whole K tile of 32: one barrier pair per tile, the sub-step-1 fragments re-read inside the cluster */
#define FZ_KG2 32     /* TWO K groups of WA x WB waves: group g contracts k sub-steps [2 g, 2 g + 2) of every K-64 tile (ring loop); merged through LDS */
#define FZ_LC 256      /* LOADER / CONSUMER waves: WA x WB consumer waves (one per SIMD) own the output tile and only read fragments + run MFMAs;
                          as many loader waves (their SIMD partners) issue ALL the LDS-DMA; K tiles of 32, 4 slots */
#define FZ_KGSPREAD 128  /* with FZ_KG2 (ring loop): the LDS-DMA pieces of the next tile go out one by one BETWEEN the MFMAs of the current one */
#define FZ_KGPP 64    /* with FZ_KG2: the two K groups in PING-PONG -- K tiles of 32, group g contracts sub-step g of every tile while the other group reads */
#ifdef FZ_IGEMM_TIMING  // scripts/igemm_timeline.hip: s_memtime totals per loop segment of waves 0 and 4 of workgroup 0 (never in the product)
__device__ long long fz_igemm_timing[2][8];
__device__ long long fz_igemm_timing2[2][6];   // [0..1] set-up / whole kernel (s_memtime ticks); [2..5] wall clock (10 ns) at entry / K loop start / K loop end / exit
#define FZ_TK_DECL() long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define FZ_TK(i)                              \
    __builtin_amdgcn_sched_barrier(0);        \
    const long long tk##i = clock64();        \
    __builtin_amdgcn_sched_barrier(0)
#define FZ_TK_ADD(slot, a, b) tacc_[slot] += (b) - (a)
#define FZ_TK_FLUSH()                                                                                      \
    do {                                                                                                   \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (tid & 255) == 0)                     \
            for (int s_ = 0; s_ < 8; ++s_) fz_igemm_timing[tid >> 8][s_] = tacc_[s_];                       \
    } while (0)
#else
#define FZ_TK_DECL() ((void)0)
#define FZ_TK(i) ((void)0)
#define FZ_TK_ADD(slot, a, b) ((void)0)
#define FZ_TK_FLUSH() ((void)0)
#endif

FZ_DEVICE_GLOBAL __attribute__((aligned(16))) half_t fz_zero_page[8192];  // 16 KB of zeros (K <= 8128 per tap)

struct IgArgs {
    const half_t* a;     // [Ma][lda]   (weights / the contiguous output dimension)
    const half_t* b;     // plain: [Nb][ldb];  conv: x[N][Hi][Wi][Cin]
    half_t* y;           // [Nb][ldy]
    float* part;         // split-K partial slabs [ksplit][batch][Nb][Ma] or null
    const half_t* bias;  // [Ma] or null
    const half_t* temb;  // rows of Ma values, one per temb_group consecutive B rows, or null
    const half_t* res;   // [Nb][ldres] or null
    const half_t* res2;
    int64_t lda, ldb, ldy, ldres, a_bs, b_bs, y_bs, res_bs, temb_stride;
    int64_t Nb;          // B rows (tokens / output pixels) per batch element
    int64_t temb_group;
    int Ma, Ma_store;    // A rows; rows [Ma, Ma_store) of the output are written as zeros (V^T padding)
    int Cin, taps, kchunks;
    int N, Hi, Wi, Ho, Wo, stride, upsample, fpb;
    int ksplit, tiles_a;
    int group_b;          // tile order inside an XCD's run of tiles: groups of `group_b` b-tiles, b-tile fastest inside a group, then the a-tiles,
                          // then the next group (1 = a-tile fastest: consecutive tiles share their B rows; tiles_b = b-tile fastest)
    int tiles_b;
    int nt_flat;          // > 0: split-K launch with a FLAT grid of nt_flat * ksplit workgroups, K slices mapped onto XCDs (ig_launch)
    // LayerNorm fused around the GEMM (fz_gemm_ln): the B rows are the RAW LayerNorm input, A holds gamma * W
    const float* ln_in;   // per B row: ln_blocks x (sum, sum of squares) of its 64-channel blocks, or null
    const float* ln_c1;   // [Ma]: sum_k (gamma W)[a][k]
    const float* ln_c0;   // [Ma]: sum_k beta_k W[a][k] + bias[a]
    float ln_eps;
    int ln_blocks;
    float* st_out;        // per output row: (Ma / 64) x (sum, sum of squares) of the STORED values, or null
    // fz_gemm_qkvt (the q | k | V^T projection of a self-attention in ONE launch): A rows [vt_split, Ma) are the V projection; the
    // tiles that hold them store TRANSPOSED, yt[row / vt_rows][a - vt_split][row % vt_rows] -- the attention kernels' V^T operand
    half_t* yt;           // [frames][Ma - vt_split][ldyt] or null
    int64_t yt_bs, ldyt;
    int vt_split, vt_rows;
    // GroupNorm statistics of the STORED output out of the epilogue (the GS instantiation): Welford partials (count, mean, M2) per
    // (frame, group, 128-row chunk) in the layout gn_finalize reads, gs_out[frame][gs_groups][gs_chunks][3]
    float* gs_out;
    int gs_cpg, gs_groups, gs_chunks, gs_rpf;  // channels per group, groups of the whole tensor, chunks per frame, B rows per frame
    // LayerNorm of the STORED rows out of the epilogue (the GS == -1 instantiation, fz_gemm_lnout): the tile spans the whole row (Ma == 320 ==
    // the tile width), so the row statistics are exact two-sweep sums over the staging tile; lno_y[row][0..Ma) = LN(y[row]) * gamma + beta
    half_t* lno_y;
    const half_t* lno_gamma;
    const half_t* lno_beta;
    int64_t lno_ld;
    float lno_eps;
};

template <int WA, int TA, int WB, int TB, int BK, int NS, bool GEGLU, int PP = 0>
struct IgCfg {
    static_assert(BK == 32 || BK == 64, "K step of 32 or 64 halves");
    // PP: the phase-interleaved ("ping-pong") K loop -- two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) staggered by one
    // barrier, so that one group's MFMA cluster runs while the other group issues fragment reads and LDS-DMA
    static_assert(!(PP & FZ_PP_ON) || (BK == 32 && NS == 4 && WA * WB == 8), "ping-pong loop: K step 32, 4 slots, 8 waves = two groups of 4");
    // K groups (FZ_KG2): the 320 x 128 tile as 2 x 2 waves of 5 x 2 MFMA tiles, twice -- each group contracts half of every K-64 step.  Why:
    // with 8 waves of 5 x 1 tiles the same output tile costs 6 fragment reads per 5 MFMAs and its LDS is busy ~94 % of the matrix time
    // (8 x 6 ds_read_b128 = 192 LDS cycles + ~110 of LDS-DMA landing per 320 cycles of MFMA per SIMD): the tile is LDS-bound.  5 x 2 tiles
    // read 7 fragments per 10 MFMAs (~69 %); the two groups' accumulators meet once, in the epilogue.
    static constexpr bool LC = (PP & FZ_LC) != 0;
    static_assert(!LC || (!(PP & (FZ_PP_ON | FZ_KG2)) && BK == 32 && NS == 4 && WA * WB == 4), "loader / consumer loop: K step 32, 4 slots, 4 + 4 waves");
    static constexpr int KG = (PP & FZ_KG2) ? 2 : 1;
    static constexpr bool KGPP = (PP & FZ_KGPP) != 0;
    static_assert(KG == 1 || (!(PP & FZ_PP_ON) && (KGPP ? (BK == 32 && NS == 4) : BK == 64)), "K groups: ring loop with K step 64, or their own ping-pong loop");
    static_assert(!KGPP || KG == 2, "the K-group ping-pong loop needs the two groups");
    static constexpr int NWG = WA * WB;          // waves of one K group = owners of the output tile
    static constexpr int NW = (LC ? 2 : KG) * NWG, T = 64 * NW;
    static constexpr int NDW = LC ? NWG : NW;     // waves that issue LDS-DMA (LC: the loader waves only)
    static constexpr int BA = WA * TA * 32, BB = WB * TB * 32;
    static constexpr int CPR = BK / 8;         // 16-byte chunks per tile row
    static constexpr int RPI = 64 / CPR;       // tile rows covered by one LDS-DMA wave instruction (1 KB)
    // LDS-DMA instructions per wave per K step; a tile whose rows do not split evenly over the waves is padded with
    // instructions that fetch (clamped) rows nobody reads, so that EVERY wave issues the same number per K step -- the
    // counted vmcnt of the ring depends on it
    static constexpr int ACH = (BA / RPI + NDW - 1) / NDW, BCH = (BB / RPI + NDW - 1) / NDW;
    static constexpr int PER = ACH + BCH;
    static constexpr int A_HALVES = ACH * NDW * 512, B_HALVES = BCH * NDW * 512;  // 1 KB = 512 halves per instruction
    static constexpr int STAGE = A_HALVES + B_HALVES;
    static constexpr int LDS_HALVES = NS * STAGE;
    static_assert(LDS_HALVES * 2 <= 160 * 1024, "LDS ring exceeds 160 KB");
    // Waves per SIMD the register allocation must leave room for: what the launch bound implies by itself (one workgroup per CU; the
    // explicit value leaves the code of those tiles bit for bit as it was) unless the tile is an 8-wave one whose ring fits the LDS twice
    // and whose accumulators fit a 128-register wave -- then TWO workgroups share a CU, and one's epilogue (VALU, stores) runs under the
    // other's K loop.  Measured on the epilogue-dominated short-K projections (profiles/r03_igemm_shortk_two_wg_per_cu.txt): 128 x 256
    // gains 5-10 % on the K = 320 / 640 GEGLU launches (64 gelu per lane in the epilogue) and loses 10-40 % on every plain projection;
    // 256 x 128, 128 x 128 and 320 x 128 lose everywhere.  Shipped for those GEGLU launches only (ig_run), the rest are trial ids.
    static constexpr int WAVES_PER_SIMD = (NW == 8 && LDS_HALVES * 2 * 2 <= 160 * 1024 && TA * TB * 16 <= 80) ? 4 : (NW + 3) / 4;
    static_assert((NS - 2) * PER < 64 && NS >= 2, "ring depth");
    static constexpr int CW = GEGLU ? BA / 2 : BA;  // output columns of the tile
    static constexpr int CSTR = CW + 8;
    static constexpr int wbp() {
        int w = WB;
        while (w > 1 && w * TB * 32 * CSTR > LDS_HALVES) w /= 2;
        return w;
    }
    static constexpr int WBP = wbp();  // B-direction waves staged per epilogue pass
    static constexpr int RP = WBP * TB * 32;
    static_assert(RP * CSTR <= LDS_HALVES, "epilogue staging does not fit");
    static_assert(!GEGLU || TA % 2 == 0, "GEGLU pairs MFMA tiles (h, gate)");
    // chunk swizzle of row r: conflict-free ds_read_b128 for 32 consecutive rows at one chunk column
    static FZ_DEVICE int swz(int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; }
};


// LN: the fz_gemm_ln form (LayerNorm correction of the B rows / row statistics of the output in the epilogue).  Its own
// instantiation: with the two blocks merely branched around, the 320- and 256-wide tiles of EVERY mode spilled 152-356 VGPRs.
// VT: the fz_gemm_qkvt form (column tiles at or beyond g.vt_split store transposed).  Its own instantiation for the same reason.
// GS > 0: the output's GroupNorm statistics (groups of GS channels) leave the epilogue as Welford partials -- the consumer's fz_groupnorm
// then skips its statistics kernel.  Own instantiations per group width (10 / 20: 320 / 640 channels over 32 groups), so that the
// statistics pass is straight-line code with its LDS loads in flight together.
template <int WA, int TA, int WB, int TB, int BK, int NS, int MODE, bool GEGLU, bool LN = false, int PP = 0, bool VT = false, int GS = 0>
FZ_KERNEL void __launch_bounds__((IgCfg<WA, TA, WB, TB, BK, NS, GEGLU, PP>::T), (IgCfg<WA, TA, WB, TB, BK, NS, GEGLU, PP>::WAVES_PER_SIMD)) igemm_kernel(IgArgs g) {
    typedef IgCfg<WA, TA, WB, TB, BK, NS, GEGLU, PP> C;
    FZ_DYN_SMEM(raw);
    half_t* smem = reinterpret_cast<half_t*>(raw);
    const int tid = threadIdx.x, wave = fz_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int kg = (C::KG > 1 || C::LC) ? wave / C::NWG : 0, wq_ = (C::KG > 1 || C::LC) ? wave - kg * C::NWG : wave;  // K group (LC: 1 = loader), wave inside it
    const int dwave = C::LC ? wq_ : wave;   // index among the waves that issue LDS-DMA
    const int wa = wq_ / WB, wb = wq_ % WB;
#ifdef FZ_IGEMM_TIMING
    const long long tw_entry = wall_clock64();
    const long long tk_entry = clock64();
#endif
    // ---- tile of this workgroup: XCD-aware (blocks b, b+8, b+16.. share an XCD and get consecutive tiles, which share
    //      their B rows: the activation panel is fetched once per XCD L2), a-tile fastest
    // Split-K: every XCD has its own L2, and with the K slices spread over all XCDs each of the eight L2s streams the WHOLE weight
    // matrix (16^2 conv 1280 -> 1280: 29.5 MB x 8 against 5 MB of input; in situ the 3x3 convolutions fetch 4.2x their algorithmic
    // bytes, profiles/r04_pmc_job.json).  The flat grid hands XCD x the work items [x, x + 1) * total / 8 of the K-slice-major order:
    // whole K slices when ksplit is a multiple of 8, a run of one slice's tiles otherwise -- an XCD then reads only ITS slices of the
    // weights and of the input channels.
    int lid, ks;
    if (g.nt_flat > 0) {
        const int per = (int)gridDim.x >> 3;
        const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        ks = item / g.nt_flat;
        lid = item - ks * g.nt_flat;
    } else {
        const int nt = gridDim.x, bid = blockIdx.x;
        const int q8 = nt >> 3, r8 = nt & 7, xcd = bid & 7;
        lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
        ks = blockIdx.y;
    }
    // tile order: which operand the consecutive tiles of an XCD share (ig_launch picks the one that moves fewer bytes into the L2s)
    int ta, tb;
    if (g.group_b <= 1) {
        ta = lid % g.tiles_a;
        tb = lid / g.tiles_a;
    } else {  // the 32 workgroups an XCD runs at a time then cover a 2-D block of group_b x (32 / group_b) tiles: both panels are re-used in L2
        const int per = g.group_b * g.tiles_a, gid = lid / per, first = gid * g.group_b;
        const int gs = g.tiles_b - first < g.group_b ? g.tiles_b - first : g.group_b, in = lid - gid * per;
        ta = in / gs;
        tb = first + in - ta * gs;
    }
    const int a0 = ta * C::BA;
    const int64_t b0 = (int64_t)tb * C::BB;
    const int z = blockIdx.z;
    const half_t* A = g.a + (int64_t)z * g.a_bs;
    const half_t* B = g.b + (int64_t)z * g.b_bs;
    const char* zero = reinterpret_cast<const char*>(fz_zero_page);

    // ---- per-lane sources.  Instruction (i, wave) of a tile covers rows RPI*(i*NW+wave) .. +RPI, lane -> (row = lane/CPR,
    //      chunk position lane%CPR); the lane fetches source chunk pos ^ swz(row).  A lane's pointer is fixed for the whole K
    //      loop of a GEMM and for one tap of a convolution; each K step only adds a wave-uniform byte offset to it (one 64-bit
    //      add per instruction -- everything else the loop needs is scalar).  Lanes that must read zeros (image border, clip
    //      ends) point into a device-global page of zeros; the K tail of a ragged Cin takes a separate, rarely executed path.
    const int pos = lane % C::CPR;
    const int ktail = g.Cin - (g.kchunks - 1) * BK;  // valid halves of the last K chunk of a tap (BK when Cin % BK == 0)
    const char* aptr[C::ACH];
    int asc[C::ACH];
    bool apad[C::ACH];
    uint32_t aoff[C::ACH], boff[C::BCH];
#pragma unroll
    for (int i = 0; i < C::ACH; ++i) {
        const int row = (i * C::NDW + dwave) * C::RPI + lane / C::CPR;
        asc[i] = pos ^ C::swz(row);
        int ar = a0 + row;
        ar = ar < g.Ma ? ar : g.Ma - 1;
        aptr[i] = reinterpret_cast<const char*>(A + (int64_t)ar * g.lda) + asc[i] * 16;
        apad[i] = row >= C::BA;  // an instruction that only pads the wave's count: fetch the (cache-resident) zero page
        // ping-pong loop: wave-uniform 64-bit base (the tile's first row + the K offset, scalar registers) + this per-lane 32-bit
        // byte offset -- the SGPR-base form of the LDS-DMA instruction, no address VALU per K tile.  Padding instructions (uniform
        // per wave: BA is a multiple of the rows per instruction) read the zero page at lane * 16.
        aoff[i] = apad[i] ? (uint32_t)lane * 16u : (uint32_t)(((int64_t)(ar - a0) * g.lda + asc[i] * 8) * 2);
    }
    // MODE: 0 = plain rows; 1 = 3x3 conv, K order (tap, Cin chunk); 3 = 3x3 conv, K order (Cin chunk, tap); 2 = temporal
    // 3-tap conv, K order (tap, Cin chunk) (chunk-outer measured 2-7 % slower there).  Chunk-outer order keeps the input window of a K chunk (a 128-byte slice of every
    // pixel row the tile and its halo touch) in the XCD's L2 while the taps sweep over it; with taps outermost the 32 tiles
    // running on an XCD re-read a > 4 MB window from MALL once per tap at 16 frames (profiles/r02_pmc_conv64_globaldma.json:
    // FETCH_SIZE 9x the input).  The per-tap pointer is then formed every K step from a per-lane centre pointer, six validity
    // bits and wave-uniform tap offsets (6 VALU per DMA instruction); tap-outer order (kept for the nearest-2x upsampling
    // conv, whose source pixel is not linear in the tap) recomputes the pointers once per tap.
    constexpr bool KORD = (MODE == 3);
    constexpr bool CONV = (MODE == 1 || MODE == 3);
    const char* bptr[C::BCH];   // !KORD: this tap's source (or the zero page); KORD: the centre tap's source
    int bsc[C::BCH];
    int bn[C::BCH], boy[C::BCH], box[C::BCH];  // conv: (frame, oy, ox) of the lane's pixel
    int bflag[C::BCH];          // KORD: bit ky = source row oy*stride+ky-1 exists, bit 3+kx = column; temporal: bit t = frame
    bool bok[C::BCH];
#pragma unroll
    for (int i = 0; i < C::BCH; ++i) {
        const int row = (i * C::NDW + dwave) * C::RPI + lane / C::CPR;
        bsc[i] = pos ^ C::swz(row);
        int64_t br = b0 + row;
        bok[i] = br < g.Nb;
        br = bok[i] ? br : g.Nb - 1;
        bflag[i] = 0;
        boff[i] = 0;
        if (MODE == 0) {
            bptr[i] = reinterpret_cast<const char*>(B + br * g.ldb) + bsc[i] * 16;
            boff[i] = (uint32_t)(((br - b0) * g.ldb + bsc[i] * 8) * 2);
            bn[i] = boy[i] = box[i] = 0;
        } else {
            // (pixel rows of a convolution launch fit 31 bits -- checked by conv_common --: unsigned 32-bit divisions, a third of the
            // instructions of the 64-bit one, in the set-up every lane of every conv / temporal-conv launch runs)
            const uint32_t hw = (uint32_t)(g.Ho * g.Wo), br32 = (uint32_t)br;
            bn[i] = (int)(br32 / hw);
            const uint32_t rem = br32 - (uint32_t)bn[i] * hw;
            boy[i] = (int)(rem / (uint32_t)g.Wo);
            box[i] = (int)(rem - (uint32_t)boy[i] * (uint32_t)g.Wo);
            bptr[i] = zero;
            if (KORD) {
                const int cy = boy[i] * g.stride, cx = box[i] * g.stride;  // centre tap (always inside the image)
                bptr[i] = reinterpret_cast<const char*>(B + (((int64_t)bn[i] * g.Hi + cy) * g.Wi + cx) * g.ldb) + bsc[i] * 16;
                if (bok[i]) {
                    bflag[i] = (cy > 0 ? 1 : 0) | 2 | (cy + 1 < g.Hi ? 4 : 0) | (cx > 0 ? 8 : 0) | 16 | (cx + 1 < g.Wi ? 32 : 0);
                }
            }
        }
    }
    const int Hu = g.upsample ? g.Hi * 2 : g.Hi, Wu = g.upsample ? g.Wi * 2 : g.Wi;
    auto retarget = [&](int tap) {  // tap-outer order: source pixel of every lane's chunk for tap `tap` (or the zero page)
#pragma unroll
        for (int i = 0; i < C::BCH; ++i) {
            bool inb;
            int64_t src;
            if (MODE == 2) {  // temporal: taps along the frame axis inside each clip of fpb frames
                const int f = bn[i] % g.fpb;
                const int fs = f + tap - 1;
                inb = fs >= 0 && fs < g.fpb;
                src = ((int64_t)(bn[i] - f + (inb ? fs : f)) * g.Hi + boy[i]) * g.Wi + box[i];
            } else {
                const int ky = tap / 3, kx = tap - 3 * ky;
                int iy = boy[i] * g.stride + ky - 1, ix = box[i] * g.stride + kx - 1;
                inb = iy >= 0 && iy < Hu && ix >= 0 && ix < Wu;
                iy = iy < 0 ? 0 : iy;
                ix = ix < 0 ? 0 : ix;
                if (g.upsample) {
                    iy >>= 1;
                    ix >>= 1;
                }
                iy = iy >= g.Hi ? g.Hi - 1 : iy;
                ix = ix >= g.Wi ? g.Wi - 1 : ix;
                src = ((int64_t)bn[i] * g.Hi + iy) * g.Wi + ix;
            }
            bptr[i] = (inb && bok[i]) ? reinterpret_cast<const char*>(B + src * g.ldb) + bsc[i] * 16 : zero;
        }
    };

    const int nkt = g.taps * g.kchunks;  // kchunks = ceil(Cin / BK)
    const int kt0 = (int)((int64_t)nkt * ks / g.ksplit), kt1 = (int)((int64_t)nkt * (ks + 1) / g.ksplit);
    // issue cursor: (tap, K chunk) of the next tile to fetch -- advanced incrementally, no division in the loop
    int itap, ikc;
    if (KORD) {
        ikc = kt0 / g.taps;
        itap = kt0 - ikc * g.taps;
    } else {
        itap = kt0 / g.kchunks;
        ikc = kt0 - itap * g.kchunks;
    }
    if (MODE == 1 || MODE == 2) retarget(itap);
    // one K tile = the A part (weights) + the B part (pixels) + the cursor step.  Each part is split into PREP (the per-lane source
    // addresses of the tile the cursor points at: VALU) and FIRE (the LDS-DMA instructions themselves: M0 + VMEM issue, no VALU); the
    // ring loops run them back to back, the ping-pong loop preps inside its own MFMA cluster and fires in the next read phase
    // (A first: the cursor moves with B's prep).
    const char* asrc[C::ACH];
    const char* bsrc[C::BCH];
    // ping-pong loop (Cin % BK == 0, checked by the launcher): scalar bases of the tile the cursor points at
    const char* const a_tile = reinterpret_cast<const char*>(A + (int64_t)a0 * g.lda);
    const char* const b_tile = reinterpret_cast<const char*>(B + b0 * g.ldb);
    const char* a_k = a_tile;
    const char* b_k = b_tile;
    constexpr bool PPL = (PP & FZ_PP_ON) != 0;  // the ping-pong loop (the other bits of PP: its trial forms, and FZ_KG2 for the ring loop)
    constexpr bool SADDR = PPL || (PP & (FZ_KGPP | FZ_LC)) != 0;  // LDS-DMA in the scalar-base form (no ragged K on these loops): no address VALU per K tile
    constexpr bool B_SADDR = SADDR && MODE == 0;  // plain B rows: scalar base + per-lane offset as well
    auto prep_a = [&]() {
        const int64_t ka = (int64_t)(itap * g.Cin + ikc * BK) * 2;  // wave-uniform byte offset along K
        if constexpr (SADDR) {
            a_k = a_tile + ka;
            return;
        }
        if (ikc == g.kchunks - 1 && ktail < BK) {  // wave-uniform: ragged last chunk of a tap, chunks past Cin read zeros
            FZ_COLD_PATH();
#pragma unroll
            for (int i = 0; i < C::ACH; ++i) asrc[i] = asc[i] * 8 < ktail && !(SADDR && apad[i]) ? aptr[i] + ka : zero;
        } else {
#pragma unroll
            for (int i = 0; i < C::ACH; ++i) {
                if (SADDR && (i + 1) * C::NDW * C::RPI > C::BA) {  // the only instruction slot that can be padding
                    asrc[i] = apad[i] ? zero : aptr[i] + ka;
                } else {
                    asrc[i] = aptr[i] + ka;
                }
            }
        }
    };
    auto fire_a = [&](int buf) {
        char* Ab = reinterpret_cast<char*>(smem + buf * C::STAGE);
#pragma unroll
        for (int i = 0; i < C::ACH; ++i) {
            if constexpr (SADDR) {
                const bool pad = (i + 1) * C::NDW * C::RPI > C::BA && (i * C::NDW + dwave) * C::RPI >= C::BA;  // wave-uniform
                fz_glds16_so_a(pad ? zero : a_k, aoff[i], Ab + (i * C::NDW + dwave) * 1024);
            } else {
                fz_glds16_aux<FZ_A_AUX>(asrc[i], Ab + (i * C::NDW + dwave) * 1024);
            }
        }
    };
    auto advance = [&]() {  // cursor -> next K tile
        if (KORD) {
            if (++itap == g.taps) {
                itap = 0;
                ++ikc;
            }
        } else if (++ikc == g.kchunks) {
            ikc = 0;
            ++itap;
            if ((MODE == 1 || MODE == 2) && itap < g.taps) retarget(itap);
        }
    };
    auto prep_b = [&]() {
        int64_t kb = ikc * BK * 2;
        if constexpr (B_SADDR) {
            b_k = b_tile + kb;
            advance();
            return;
        }
        int need = 0;  // KORD: validity bits this tap requires
        if (KORD) {
            const int ky = itap / 3, kx = itap - 3 * ky;
            kb += ((int64_t)(ky - 1) * g.Wi + (kx - 1)) * g.ldb * 2;
            need = (1 << ky) | (8 << kx);
        }
        if (!SADDR && ikc == g.kchunks - 1 && ktail < BK) {
            FZ_COLD_PATH();
#pragma unroll
            for (int i = 0; i < C::BCH; ++i) {
                const bool ok = bsc[i] * 8 < ktail && (!KORD || (bflag[i] & need) == need);
                bsrc[i] = ok ? bptr[i] + kb : zero;
            }
        } else {
#pragma unroll
            for (int i = 0; i < C::BCH; ++i) {
                if (KORD) {
                    bsrc[i] = (bflag[i] & need) == need ? bptr[i] + kb : zero;
                } else {
                    bsrc[i] = bptr[i] + kb;
                }
            }
        }
        advance();
    };
    auto fire_b = [&](int buf) {
        char* Bb = reinterpret_cast<char*>(smem + buf * C::STAGE) + C::A_HALVES * 2;
#pragma unroll
        for (int i = 0; i < C::BCH; ++i) {
            if constexpr (B_SADDR) {
                fz_glds16_so(b_k, boff[i], Bb + (i * C::NDW + dwave) * 1024);
            } else {
                fz_glds16(bsrc[i], Bb + (i * C::NDW + dwave) * 1024);
            }
        }
    };
    auto issue = [&](int buf) {
        prep_a();
        fire_a(buf);
        prep_b();
        fire_b(buf);
    };
    // (FZ_KGSPREAD) one piece of the prepared tile: p in [0, ACH) = A pieces, [ACH, PER) = B pieces -- the VGPR-address form of the ring loop
    auto fire_piece = [&](int buf, int p) {
        char* Ab = reinterpret_cast<char*>(smem + buf * C::STAGE);
#pragma unroll
        for (int i = 0; i < C::ACH; ++i)
            if (i == p) fz_glds16(asrc[i], Ab + (i * C::NDW + dwave) * 1024);
#pragma unroll
        for (int i = 0; i < C::BCH; ++i)
            if (C::ACH + i == p) fz_glds16(bsrc[i], Ab + C::A_HALVES * 2 + (i * C::NDW + dwave) * 1024);
    };
    auto pin_a = [&]() {};  // (A: scalar base in the ping-pong loop, nothing per lane)
    auto pin_b = [&]() {  // the prepared addresses exist as registers from here on (not re-derived next to the DMA instruction)
        if constexpr (!B_SADDR) {
#pragma unroll
            for (int i = 0; i < C::BCH; ++i) FZ_PIN_V(bsrc[i]);
        }
    };

    f32x16 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = fz_zero_f16v();

    // fragment read offsets (halves): row r, k chunk c -> r * BK + ((c ^ swz(r)) * 8); the tile rows of a lane are
    // l31 + multiples of 32, so the swizzle term depends on the lane only
    const int fsw = C::swz(l31);
    const int arow = (wa * TA * 32 + l31) * BK, brow = (wb * TB * 32 + l31) * BK;

    // ---- NS-deep LDS ring, tiles kt .. kt+NS-2 in flight while tile kt is consumed -------------------------------------
    //   iteration kt:  wait until this wave's DMA of tile kt landed (the NS-2 younger tiles may stay in flight) -> barrier
    //   (tile kt visible to all; everybody is done with tile kt-1, whose buffer is the one refilled next) -> issue tile
    //   kt+NS-1 -> MFMAs on tile kt.  The barrier does not drain vmcnt, so the loads span barriers (T3+T4 of the guide).
    const int ntile = kt1 - kt0;
    FZ_TK_DECL();
#ifdef FZ_IGEMM_TIMING
    const long long tw_loop = wall_clock64();
    const long long tk_loop = clock64();
#endif
    if constexpr (PPL) {
        // ---- phase-interleaved loop (cdna_hip_programming.md "The 256^2 8-phase template", T3+T4+T5) --------------------------------
        // K tiles of 32 in a 4-slot ring; a PHASE is one k sub-step of 16:   R: { ds_read the phase's fragments, issue a slice of
        // LDS-DMA }  s_barrier  M: { TA x TB MFMAs at raised priority }  s_barrier.  The waves of group wa = 1 (waves 4-7: the second
        // wave of every SIMD) pass ONE extra barrier before the loop and so run half a phase behind group 0: while one group's MFMA
        // cluster occupies the matrix pipe, the other group's wave on the same SIMD issues its fragment reads and DMA.
        // Tile j lives in slot j % 4 and is read in phases 2j, 2j+1.  Issue order per wave: A0 B0 A1 B1 A2 B2 (prologue), then
        // A(j+3) in phase 2j+1 and B(j+3) in phase 2j+2.
        //   RAW: phase 2j+1 waits (counted, before its first barrier) until this wave's part of tile j+1 has landed -- everything
        //        issued before tile j+2 -- and tile j+1 is first read in phase 2j+2: one phase after the wait, which is what two groups
        //        staggered by a barrier need (both groups' waits precede barrier 4j+3, both groups' reads follow it).
        //   WAR: slot (j+3) % 4 held tile j-1, last read in phase 2j-1 (reads retired at the start of M(2j-1), i.e. before barrier
        //        4j for both groups); it is restaged from phase 2j+1 on: two phases later (R(2j+1) starts after barrier 4j+1).
        // Up to 2 tiles (72-80 KB) stay in flight across 3-4 phases; vmcnt is never drained in the steady state.
        static_assert(2 * C::PER < 64, "vmcnt field");
        const bool late = wave >= 4;  // waves 4-7: the second wave of every SIMD
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (s < ntile) issue(s);
        if (ntile > 2) {
            fz_wait_vm<2 * C::PER>();
        } else if (ntile > 1) {
            fz_wait_vm<C::PER>();
        } else {
            fz_wait_vm0();
        }
        fz_barrier_raw();
        if (!(PP & FZ_PP_NOSTAGGER) && late) fz_barrier_raw();
        // One phase:  R: [counted wait] -> the fragment reads of this k sub-step + FIRE of the LDS-DMA slice prepared one phase ago --
        // LDS / VMEM / scalar instructions only -> barrier ->  M: the TA x TB MFMAs, and between them ALL the VALU of the next phase
        // (its LDS read addresses `ra` / `rb`, PREP of the next DMA slice, the cursor step) -> barrier.
        // Why the VALU lives in M (profiles/r03_igemm_timeline_v1.txt): while the other wave of the SIMD runs its MFMA cluster, a
        // VALU instruction of this wave gets the shared issue port about once per MFMA (~30 cycles each) -- 15 address instructions in
        // R made the read half of a phase 460-560 cycles long against 320 cycles of MFMA.  Inside its OWN cluster the wave's VALU
        // rides in the shadow of its MFMAs (each leaves seven free issue quads).
        const int oa0 = arow + (hi ^ fsw) * 8, oa1 = arow + ((2 + hi) ^ fsw) * 8;   // per-lane halves offsets of k sub-steps 0 / 1
        const int ob0 = C::A_HALVES + brow + (hi ^ fsw) * 8, ob1 = C::A_HALVES + brow + ((2 + hi) ^ fsw) * 8;
        fz_lds_addr ra = fz_lds_addr_of(smem + oa0), rb = fz_lds_addr_of(smem + ob0);
        FZ_PIN_V(ra);
        FZ_PIN_V(rb);
        auto pp_phase = [&](int wait_kind, auto&& fire, auto&& prep) {
            half8_t af[TA], bf[TB];
            FZ_TK(0);
            if (wait_kind == 1) {
                fz_wait_vm<C::PER>();
            } else if (wait_kind == 2) {
                fz_wait_vm0();
            }
            FZ_TK(1);
#pragma unroll
            for (int q = 0; q < TB; ++q) bf[q] = fz_lds_ld_h8(rb, q * 32 * BK * 2);
#pragma unroll
            for (int i = 0; i < TA; ++i) af[i] = fz_lds_ld_h8(ra, i * 32 * BK * 2);
            fire();
            FZ_TK(2);
            FZ_SCHED_FENCE();
            fz_barrier_raw();
            FZ_SCHED_FENCE();
            FZ_TK(3);
#ifdef FZ_IGEMM_TIMING
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // timing build only: the fragment reads' own latency, apart from the MFMAs
            FZ_SCHED_FENCE();
#endif
            FZ_TK(4);
            if (!(PP & FZ_PP_NOPRIO)) fz_setprio_hi();
#pragma unroll
            for (int i = 0; i < TA; ++i) {
#pragma unroll
                for (int q = 0; q < TB; ++q) acc[i][q] = fz_mfma_32x32x16_f16(af[i], bf[q], acc[i][q]);
                if (i == 0 && !(PP & FZ_PP_PREP_IN_R)) prep();  // VALU of the next phase, behind the first MFMAs of this one
            }
            if (!(PP & FZ_PP_NOPRIO)) fz_setprio_lo();
            FZ_TK(5);
            FZ_SCHED_FENCE();
            fz_barrier_raw();
            FZ_SCHED_FENCE();
            FZ_TK(6);
            if (PP & FZ_PP_PREP_IN_R) prep();
            FZ_TK_ADD(0, tk0, tk1);
            FZ_TK_ADD(1, tk1, tk2);
            FZ_TK_ADD(2, tk2, tk3);
            FZ_TK_ADD(3, tk3, tk4);
            FZ_TK_ADD(4, tk4, tk5);
            FZ_TK_ADD(5, tk5, tk6);
            FZ_TK_ADD(6, 0, 1);
        };
        int slot = 0;
        // one K tile = two phases.  STEADY: 1 <= j and j + 3 < ntile, every condition below is true -- the loop over those tiles
        // carries no branch and no flag; the first tile and the last three run the general form
        auto pp_tile = [&](int j, auto STEADY) {
            constexpr bool steady = decltype(STEADY)::value;
            // phase 2j (k sub-step 0): fires B(j+2) -> slot (j+2) % 4; its MFMA cluster prepares phase 2j+1: same slot, sub-step 1,
            // and A(j+3)
            pp_phase(0, [&]() {
                if (steady || (j >= 1 && j + 2 < ntile)) fire_b(slot ^ 2);
            }, [&]() {
                ra = fz_lds_addr_of(smem + slot * C::STAGE + oa1);
                rb = fz_lds_addr_of(smem + slot * C::STAGE + ob1);
                FZ_PIN_V(ra);
                FZ_PIN_V(rb);
                if (steady || j + 3 < ntile) {
                    prep_a();
                    pin_a();
                }
            });
            // phase 2j+1 (k sub-step 1): first the counted wait -- tile j+1 (first read in the next phase) has landed, tile j+2 may
            // stay in flight; fires A(j+3) -> slot (j+3) % 4; its cluster prepares phase 2j+2: next slot, sub-step 0, and B(j+3)
            pp_phase(steady ? 1 : (j + 1 < ntile ? (j + 2 < ntile ? 1 : 2) : 0), [&]() {
                if (steady || j + 3 < ntile) fire_a((slot + 3) & 3);
            }, [&]() {
                const int ns = (slot + 1) & 3;
                ra = fz_lds_addr_of(smem + ns * C::STAGE + oa0);
                rb = fz_lds_addr_of(smem + ns * C::STAGE + ob0);
                FZ_PIN_V(ra);
                FZ_PIN_V(rb);
                if (steady || j + 3 < ntile) {
                    prep_b();
                    pin_b();
                }
            });
            slot = (slot + 1) & 3;
        };
        if constexpr ((PP & FZ_PP_K32) != 0) {
            // ---- K-32 phases: ONE barrier pair per K tile (half the synchronisations per MFMA).  R: the fragments of k sub-step 0 +
            // FIRE of tile j+2 + the counted wait for tile j+1;  M: the MFMAs of sub-step 0, each A fragment register reloaded for
            // sub-step 1 right behind its last use (LDS instructions inside the cluster: the LDS port is not the VALU port), then
            // the MFMAs of sub-step 1.  Tile j is read in phase j (R and M); its slot is refilled from phase j+2 on (both groups have
            // passed the barrier behind M(j) by then) with tile j+4... i.e. phase p fires tile p+2 into slot (p+2) % 4, which is
            // waited for in phase p+1 and read in phase p+2.  Prologue (above) issued tiles 0, 1, 2: phase 0 fires nothing.
            auto pp_phase32 = [&](int j, auto STEADY) {
                constexpr bool steady = decltype(STEADY)::value;
                half8_t af[TA], bf[TB], bg[TB];
                // address VALU / scalar cursor work of this phase, at its very start
                const fz_lds_addr base = fz_lds_addr_of(smem + slot * C::STAGE);
                fz_lds_addr ra0 = base + 2 * oa0, rb0 = base + 2 * ob0, ra1 = base + 2 * oa1, rb1 = base + 2 * ob1;
                FZ_PIN_V(ra0);
                FZ_PIN_V(rb0);
                FZ_PIN_V(ra1);
                FZ_PIN_V(rb1);
                const bool fires = steady || (j >= 1 && j + 2 < ntile);
                if (fires) {
                    prep_a();
                    prep_b();
                    pin_b();
                }
#pragma unroll
                for (int q = 0; q < TB; ++q) bf[q] = fz_lds_ld_h8(rb0, q * 32 * BK * 2);
#pragma unroll
                for (int i = 0; i < TA; ++i) af[i] = fz_lds_ld_h8(ra0, i * 32 * BK * 2);
                if (fires) {
                    fire_a(slot ^ 2);
                    fire_b(slot ^ 2);
                }
                if (steady) {
                    fz_wait_vm<C::PER>();
                } else if (j + 1 < ntile) {  // tile j+1 (read in the next phase) has landed; tile j+2 stays in flight
                    if (j + 2 < ntile) {
                        fz_wait_vm<C::PER>();
                    } else {
                        fz_wait_vm0();
                    }
                }
                FZ_SCHED_FENCE();
                fz_barrier_raw();
                FZ_SCHED_FENCE();
                if (!(PP & FZ_PP_NOPRIO)) fz_setprio_hi();
#pragma unroll
                for (int q = 0; q < TB; ++q) bg[q] = fz_lds_ld_h8(rb1, q * 32 * BK * 2);
#pragma unroll
                for (int i = 0; i < TA; ++i) {
#pragma unroll
                    for (int q = 0; q < TB; ++q) acc[i][q] = fz_mfma_32x32x16_f16(af[i], bf[q], acc[i][q]);
                    af[i] = fz_lds_ld_h8(ra1, i * 32 * BK * 2);
                    FZ_SCHED_FENCE();
                }
#pragma unroll
                for (int i = 0; i < TA; ++i)
#pragma unroll
                    for (int q = 0; q < TB; ++q) acc[i][q] = fz_mfma_32x32x16_f16(af[i], bg[q], acc[i][q]);
                if (!(PP & FZ_PP_NOPRIO)) fz_setprio_lo();
                FZ_SCHED_FENCE();
                fz_barrier_raw();
                FZ_SCHED_FENCE();
                slot = (slot + 1) & 3;
            };
            pp_phase32(0, std::false_type());
            int j = 1;
            for (; j + 2 < ntile; ++j) pp_phase32(j, std::true_type());
            for (; j < ntile; ++j) pp_phase32(j, std::false_type());
        } else {
        pp_tile(0, std::false_type());
        int j = 1;
        for (; j + 3 < ntile; ++j) pp_tile(j, std::true_type());
        for (; j < ntile; ++j) pp_tile(j, std::false_type());
        }
        if (!(PP & FZ_PP_NOSTAGGER) && !late) fz_barrier_raw();  // every wave passes the same number of barriers
    } else if constexpr (C::LC) {
        // ---- loader / consumer waves ---------------------------------------------------------------------------------------------------
        // What the ingest microbenchmark says (scripts/ubench_ingest.hip, profiles/r06_ubench_ingest.txt): a wave's burst of LDS-DMA issues
        // (100-185 cycles a piece while the CU's address path is busy) and its MFMA cluster SERIALISE, and in the ring loop both waves of a SIMD
        // do both in lockstep -- 56 KB of L2-resident operands take 0.74 us alone, 1.16 us beside 20 MFMAs per wave.  Here the two kinds of
        // work live in different waves of a SIMD: waves 0-3 (one per SIMD) own the 320 x 128 tile as 2 x 2 waves of 5 x 2 MFMA tiles and do
        // nothing but fragment reads and MFMAs; waves 4-7 issue every LDS-DMA piece (7 per K-32 tile each) and their address arithmetic.
        // K tiles of 32 in a 4-slot ring, ONE barrier per tile.  Barrier B(j) -- in front of step j -- has tiles <= j + 1 landed (the
        // loaders' counted wait in front of it): a consumer contracts tile j with the fragments of (j, sub-step 0) already in registers
        // (read during step j - 1), reads (j, 1) and then (j + 1, 0) one cluster ahead of their use -- its matrix pipe never waits for an LDS
        // read behind a barrier.  The loaders refill the slot tile j - 1 left (all its fragments were read before B(j)) with tile j + 3.
        static_assert(2 * C::PER < 64, "vmcnt field");
        const bool loader = kg == 1;
        const int oa0 = arow + (hi ^ fsw) * 8, oa1 = arow + ((2 + hi) ^ fsw) * 8;   // per-lane halves offsets of k sub-steps 0 / 1
        const int ob0 = C::A_HALVES + brow + (hi ^ fsw) * 8, ob1 = C::A_HALVES + brow + ((2 + hi) ^ fsw) * 8;
        if (loader) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (t < ntile) issue(t);
            if (ntile > 2) {
                fz_wait_vm<C::PER>();     // tiles 0, 1 landed; tile 2 in flight
            } else {
                fz_wait_vm0();
            }
            fz_barrier_raw();             // B(0)
            for (int j = 0; j < ntile; ++j) {
                if (j + 3 < ntile) {
                    issue((j + 3) & 3);
                    fz_wait_vm<C::PER>();  // tile j + 2 landed (tile j + 3 in flight)
                } else {
                    fz_wait_vm0();
                }
                fz_barrier_raw();         // B(j + 1)
            }
        } else {
            half8_t af0[TA], bf0[TB], af1[TA], bf1[TB];
            auto rd = [&](half8_t* af, half8_t* bf, int slot, int oa, int ob) {
                const fz_lds_addr ra = fz_lds_addr_of(smem + slot * C::STAGE + oa), rb = fz_lds_addr_of(smem + slot * C::STAGE + ob);
#pragma unroll
                for (int q = 0; q < TB; ++q) bf[q] = fz_lds_ld_h8(rb, q * 32 * BK * 2);
#pragma unroll
                for (int i = 0; i < TA; ++i) af[i] = fz_lds_ld_h8(ra, i * 32 * BK * 2);
            };
            auto mm = [&](const half8_t* af, const half8_t* bf) {
#pragma unroll
                for (int i = 0; i < TA; ++i)
#pragma unroll
                    for (int q = 0; q < TB; ++q) acc[i][q] = fz_mfma_32x32x16_f16(af[i], bf[q], acc[i][q]);
            };
            fz_barrier_raw();             // B(0)
            rd(af0, bf0, 0, oa0, ob0);
            for (int j = 0; j < ntile; ++j) {
                rd(af1, bf1, j & 3, oa1, ob1);
                FZ_SCHED_FENCE();
                mm(af0, bf0);
                FZ_SCHED_FENCE();
                if (j + 1 < ntile) rd(af0, bf0, (j + 1) & 3, oa0, ob0);
                FZ_SCHED_FENCE();
                mm(af1, bf1);
                FZ_SCHED_FENCE();
                fz_barrier_raw();         // B(j + 1)
            }
        }
    } else if constexpr (C::KGPP) {
        // ---- the two K groups in ping-pong -----------------------------------------------------------------------------------------------
        // K tiles of 32 in a 4-slot ring (tile j in slot j % 4); group g (one wave of each group per SIMD) contracts k sub-step g of every
        // tile.  ONE barrier per sub-step: in interval 2 j - 1 group 0 READS its fragments of tile j (7 ds_read_b128) and fires its LDS-DMA
        // pieces of tile j + 3 while group 1 runs the 10 MFMAs of (tile j - 1, sub-step 1); in interval 2 j the roles swap -- the matrix pipe
        // of a SIMD always has one wave's cluster to run while the other wave does LDS / VMEM / address work, and nobody issues VALU beside
        // its own MFMAs.  Every wave fires its pieces of every tile (uniform counts: the counted vmcnt): tile j + 3 goes out in the wave's
        // read interval of tile j, i.e. after the barrier that follows the last read of tile j - 1, whose slot it takes (WAR); tile j + 1
        // must be complete at the barrier that ends interval 2 j (first read: interval 2 j + 1): every wave waits there with two younger
        // tiles in flight (RAW).  The fragments travel in registers across the barrier between a group's read and its MFMAs.
        static_assert(2 * C::PER < 64, "vmcnt field");
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (s < ntile) issue(s);
        if (ntile > 2) {
            fz_wait_vm<2 * C::PER>();
        } else if (ntile > 1) {
            fz_wait_vm<C::PER>();
        } else {
            fz_wait_vm0();
        }
        fz_barrier_nodrain();
        const int oa = arow + (((kg ? 2 : 0) + hi) ^ fsw) * 8;                  // this group's k sub-step of a tile: per-lane halves offsets
        const int ob = C::A_HALVES + brow + (((kg ? 2 : 0) + hi) ^ fsw) * 8;
        half8_t af[TA], bf[TB];
        // A wave's LDS-DMA pieces of tile j + 3 go out in TWO halves: the B pieces (with their address VALU: the per-lane im2col pointers)
        // in its read interval of tile j, the A pieces (scalar base + constant lane offsets: no VALU) between the MFMAs of the cluster that
        // follows -- four pieces in the read interval made it twice as long as a cluster (an LDS-DMA issue costs 100-185 cycles beside
        // LDS reads, ~60 among MFMAs: MI355X_MICROARCH.md).
        auto kp_read = [&](int j) {   // this group's fragments of tile j + the B pieces of tile j + 3 (the slot tile j - 1 left)
            const fz_lds_addr ra = fz_lds_addr_of(smem + (j & 3) * C::STAGE + oa), rb = fz_lds_addr_of(smem + (j & 3) * C::STAGE + ob);
#pragma unroll
            for (int q = 0; q < TB; ++q) bf[q] = fz_lds_ld_h8(rb, q * 32 * BK * 2);
#pragma unroll
            for (int i = 0; i < TA; ++i) af[i] = fz_lds_ld_h8(ra, i * 32 * BK * 2);
            if (j + 3 < ntile) {
                prep_a();
                prep_b();
                fire_b((j + 3) & 3);
            }
        };
        auto kp_mma = [&](int j) {    // the cluster of (tile j, this group's sub-step) + the A pieces of tile j + 3
            fz_setprio_hi();
            const bool fire = j + 3 < ntile;
#pragma unroll
            for (int i = 0; i < TA; ++i) {
#pragma unroll
                for (int q = 0; q < TB; ++q) acc[i][q] = fz_mfma_32x32x16_f16(af[i], bf[q], acc[i][q]);
                if (i == 1 && fire) fire_a((j + 3) & 3);
            }
            fz_setprio_lo();
        };
        // (one straight-line loop per group -- the same number of barriers in both: two per tile.  Bare barriers: a group's fragment reads stay
        //  in flight across the one behind its read interval -- its MFMAs wait for them -- and the slot they read is refilled two barriers later
        //  at the earliest, by an LDS-DMA that takes longer to land than any ds_read to return.  The counted wait at the end of interval 2 j --
        //  tile j + 1 complete -- leaves tiles j + 2 and j + 3 in flight; group 1 has only fired the B pieces of tile j + 3 by then.)
        if (kg == 0) {
            for (int j = 0; j < ntile; ++j) {
                kp_read(j);            // interval 2 j - 1
                FZ_SCHED_FENCE();
                fz_barrier_raw();
                FZ_SCHED_FENCE();
                kp_mma(j);             // interval 2 j
                if (j + 1 < ntile) {
                    if (j + 3 < ntile) {
                        fz_wait_vm<2 * C::PER>();
                    } else {
                        fz_wait_vm0();
                    }
                }
                FZ_SCHED_FENCE();
                fz_barrier_raw();
                FZ_SCHED_FENCE();
            }
        } else {
            for (int j = 0; j < ntile; ++j) {
                if (j >= 1) kp_mma(j - 1);  // interval 2 j - 1: (tile j - 1, sub-step 1) + the A pieces of tile j + 2
                FZ_SCHED_FENCE();
                fz_barrier_raw();
                FZ_SCHED_FENCE();
                kp_read(j);            // interval 2 j
                if (j + 1 < ntile) {
                    if (j + 3 < ntile) {
                        fz_wait_vm<C::PER + C::BCH>();
                    } else {
                        fz_wait_vm0();
                    }
                }
                FZ_SCHED_FENCE();
                fz_barrier_raw();
                FZ_SCHED_FENCE();
            }
            kp_mma(ntile - 1);         // interval 2 ntile - 1
        }
    } else {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < ntile) issue(s);
    int buf = 0;
    for (int it = 0; it < ntile; ++it) {
        FZ_TK(0);
        if (it + NS - 1 <= ntile) {
            fz_wait_vm<(NS - 2) * C::PER>();  // steady state: NS-2 younger tiles outstanding
        } else {
            fz_wait_vm0();                    // ring tail: fewer tiles were issued, drain
        }
        FZ_TK(1);
        fz_barrier_nodrain();
        FZ_TK(2);
        constexpr bool SPREAD = (PP & FZ_KGSPREAD) != 0;
        const bool more = it + NS - 1 < ntile;
        int nb = buf + NS - 1;
        nb = nb >= NS ? nb - NS : nb;
        if (more) {
            if constexpr (SPREAD) {   // the addresses now, the pieces between the MFMAs below
                prep_a();
                prep_b();
            } else {
                issue(nb);
            }
        }
        FZ_TK(3);
        const half_t* As = smem + buf * C::STAGE;
        const half_t* Bs = As + C::A_HALVES;
#pragma unroll
        for (int kq = 0; kq < BK / 16 / C::KG; ++kq) {
            const int kk = C::KG > 1 ? kg * (BK / 16 / C::KG) + kq : kq;  // (K groups: this group's half of the K step)
            const int co = ((2 * kk + hi) ^ fsw) * 8;
            half8_t af[TA], bf[TB];
#pragma unroll
            for (int i = 0; i < TA; ++i) af[i] = fz_ld_h8(As + arow + i * 32 * BK + co);
#pragma unroll
            for (int j = 0; j < TB; ++j) bf[j] = fz_ld_h8(Bs + brow + j * 32 * BK + co);
#pragma unroll
            for (int i = 0; i < TA; ++i) {
#pragma unroll
                for (int j = 0; j < TB; ++j) acc[i][j] = fz_mfma_32x32x16_f16(af[i], bf[j], acc[i][j]);
                if constexpr (SPREAD) {   // one LDS-DMA piece behind every TB MFMAs: 10 slots per K step for the PER = 7 pieces
                    if (more && kq * TA + i < C::PER) fire_piece(nb, kq * TA + i);
                }
            }
        }
        FZ_TK(4);
        FZ_TK_ADD(0, tk0, tk1);
        FZ_TK_ADD(2, tk1, tk2);
        FZ_TK_ADD(1, tk2, tk3);
        FZ_TK_ADD(4, tk3, tk4);
        FZ_TK_ADD(6, 0, 1);
        buf = buf + 1 == NS ? 0 : buf + 1;
    }
    }

#ifdef FZ_IGEMM_TIMING
    tacc_[7] = clock64() - tk_loop;   // the K loop as a whole (prologue fetch included)
    const long long tw_loop_end = wall_clock64();
    tacc_[3] += 0;
    const long long tk_setup = tk_loop - tk_entry;  // pointer set-up before the loop
#endif
    FZ_TK_FLUSH();

    // ---- K groups: the accumulators of group 1 join those of group 0 through LDS (fixed order: group 0 + group 1), as many tiles per pass
    //      as the ring holds; from here on group 0 owns the output tile, group 1 only lends its threads to the cooperative store loops
    if constexpr (C::KG > 1) {
        constexpr int NACC = TA * TB, PERP = (C::LDS_HALVES * 2) / (C::NWG * 4096) < NACC ? (C::LDS_HALVES * 2) / (C::NWG * 4096) : NACC;
        static_assert(PERP >= 1, "K-group merge staging");
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int p0 = 0; p0 < NACC; p0 += PERP) {
            __syncthreads();  // the ring (p0 == 0) / the previous pass's readers are done with the LDS
            if (kg == 1) {
#pragma unroll
                for (int a = p0; a < p0 + PERP && a < NACC; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((wq_ * PERP + (a - p0)) * 16 + r) * 64 + lane] = acc[a / TB][a % TB][r];
            }
            __syncthreads();
            if (kg == 0) {
#pragma unroll
                for (int a = p0; a < p0 + PERP && a < NACC; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a / TB][a % TB][r] += red[((wq_ * PERP + (a - p0)) * 16 + r) * 64 + lane];
            }
        }
    }
    const bool owner = (C::KG == 1 && !C::LC) || kg == 0;

    // ---- split-K: fp32 partial slab, reduced by igemm_reduce_kernel ---------------------------------------------
    if (g.part != nullptr && !owner) return;
    if (g.part != nullptr) {
        float* P = g.part + ((int64_t)(ks * gridDim.z + z) * g.Nb) * g.Ma;
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const int64_t px = b0 + (wb * TB + j) * 32 + l31;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int co = a0 + (wa * TA + i) * 32 + 8 * gq + 4 * hi;
                    if (px < g.Nb && co < g.Ma) {  // Ma % 4 == 0 on this path (checked by the launcher)
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * gq + e];
                        *reinterpret_cast<f32x4*>(P + px * g.Ma + co) = v;
                    }
                }
            }
        return;
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------
    // LN instantiation: LayerNorm of the B rows applied to the products, y = rstd (x . gamma W - mean c1) + c0 (lane <-> B row),
    // and the plain bias as its degenerate case (mean 0, rstd 1, c1 0) -- ONE straight-line pass over the accumulators: with a
    // branch per form the 320 live accumulators merge from two paths and the register allocator spills 150-360 of them.
    if constexpr (LN) {
        const bool has_ln = g.ln_in != nullptr;
        float mu[TB], rs[TB];
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            int64_t px = b0 + (wb * TB + j) * 32 + l31;
            px = px < g.Nb ? px : g.Nb - 1;
            const int nblk = has_ln ? g.ln_blocks : 0;
            const float* sp = g.ln_in + ((int64_t)z * g.Nb + px) * nblk * 2;
            float s1 = 0.0f, s2 = 0.0f;
            for (int t = 0; t < nblk; ++t) {  // fixed order: the statistics do not depend on who produced the partials
                const f32x2 v = *reinterpret_cast<const f32x2*>(sp + 2 * t);
                s1 += v[0];
                s2 += v[1];
            }
            const float inv_n = 1.0f / (float)(g.ln_blocks * 64);
            const float m = s1 * inv_n;
            const float var = fmaxf(s2 * inv_n - m * m, 0.0f);
            mu[j] = has_ln ? m : 0.0f;
            rs[j] = has_ln ? fz_rsqrt(var + g.ln_eps) : 1.0f;
        }
        const float* zf = reinterpret_cast<const float*>(fz_zero_page);
        const float* c1p = has_ln ? g.ln_c1 : zf;
        const float* c0p = has_ln ? g.ln_c0 : zf;
        const half_t* bp = (!has_ln && g.bias != nullptr) ? g.bias : fz_zero_page;
        const int ma_f = has_ln ? g.Ma : 0, ma_b = (!has_ln && g.bias != nullptr) ? g.Ma : 0;
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int co = a0 + (wa * TA + i) * 32 + 8 * gq + 4 * hi;  // Ma % 4 == 0 on this path: whole groups of 4
                const int cf = co < ma_f ? co : 0, cb = co < ma_b ? co : 0;  // absent / out of range -> the zero page's first words
                const f32x4 c1 = *reinterpret_cast<const f32x4*>(c1p + cf);
                f32x4 c0 = *reinterpret_cast<const f32x4*>(c0p + cf);
                const half4_t bv = *reinterpret_cast<const half4_t*>(bp + cb);
                const bool live_f = co < ma_f, live_b = co < ma_b;
#pragma unroll
                for (int e = 0; e < 4; ++e) c0[e] = (live_f ? c0[e] : 0.0f) + (live_b ? (float)bv[e] : 0.0f);
#pragma unroll
                for (int j = 0; j < TB; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j][4 * gq + e] = rs[j] * (acc[i][j][4 * gq + e] - (live_f ? mu[j] * c1[e] : 0.0f)) + c0[e];
                FZ_SCHED_FENCE();  // keeps the TA x 4 coefficient loads from being issued up front (160 VGPRs next to the accumulators)
            }
    } else
    // bias in fp32 on the accumulators (lane <-> B row, register group gq <-> 4 consecutive A rows 8*gq + 4*hi)
    if (g.bias != nullptr) {
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int co = a0 + (wa * TA + i) * 32 + 8 * gq + 4 * hi;
                half4_t bv;
                if (co + 3 < g.Ma) {
                    bv = *reinterpret_cast<const half4_t*>(g.bias + co);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = co + e < g.Ma ? g.bias[co + e] : (half_t)0.0f;
                }
#pragma unroll
                for (int j = 0; j < TB; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * gq + e] += (float)bv[e];
            }
    }
    if constexpr (VT) {
        // The accumulators are rounded to fp16 HERE, before the two store forms part ways: element pairs (4 gq, 4 gq + 1) of every
        // register group then carry the four halves of the group as raw bits and the other two die -- 80 live registers per wave of
        // the 5 x 2 tile instead of 160 where the two epilogue paths meet (with fp32 accumulators live on both sides of the branch
        // that tile spilled 91 VGPRs).
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    half4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (half_t)acc[i][j][4 * gq + e];
                    const f32x2 pk = __builtin_bit_cast(f32x2, v);
                    acc[i][j][4 * gq] = pk[0];
                    acc[i][j][4 * gq + 1] = pk[1];
                }
        // ---- transposed tile (fz_gemm_qkvt): the V columns of the fused q | k | v projection leave as V^T[frame][channel][token].
        // The accumulator layout already has the token (B row) on the lane: lanes 0..31 of a register hold 32 CONSECUTIVE tokens of
        // one channel, so the tile is staged channel-major -- Cs[channel][token], 2-byte LDS writes, 64 contiguous bytes per half
        // wave -- and leaves in 16-byte pieces of 8 tokens.  A piece never straddles a frame (vt_rows % 8 == 0, checked by the host).
        if (a0 >= g.vt_split) {  // workgroup-uniform
            constexpr int RSTR = C::RP + 8;
            static_assert(C::CW * RSTR <= C::LDS_HALVES, "transposed epilogue staging does not fit");
            half_t* Ct = smem;
            for (int ps = 0; ps < WB / C::WBP; ++ps) {
                __syncthreads();
                if (wb / C::WBP == ps) {
                    const int rl = (wb % C::WBP) * TB * 32 + l31;
#pragma unroll
                    for (int j = 0; j < TB; ++j)
#pragma unroll
                        for (int i = 0; i < TA; ++i)
#pragma unroll
                            for (int gq = 0; gq < 4; ++gq) {
                                f32x2 pk;
                                pk[0] = acc[i][j][4 * gq];
                                pk[1] = acc[i][j][4 * gq + 1];
                                const half4_t v = __builtin_bit_cast(half4_t, pk);
#pragma unroll
                                for (int e = 0; e < 4; ++e) Ct[((wa * TA + i) * 32 + 8 * gq + 4 * hi + e) * RSTR + rl + j * 32] = v[e];
                            }
                }
                __syncthreads();
                constexpr int RCH = C::RP / 8;
                for (int id = tid; id < C::CW * RCH; id += C::T) {
                    const int col = id / RCH, rc = id - col * RCH;
                    const int64_t px = b0 + ps * C::RP + rc * 8;
                    const int a = a0 + col;
                    if (px >= g.Nb || a >= g.Ma) continue;
                    const int64_t fr = px / g.vt_rows;
                    const int64_t tok = px - fr * g.vt_rows;
                    fz_st_h8(g.yt + fr * g.yt_bs + (int64_t)(a - g.vt_split) * g.ldyt + tok, fz_ld_h8(Ct + col * RSTR + rc * 8));
                }
            }
            return;
        }
    }
    const int Mo = GEGLU ? g.Ma / 2 : g.Ma;          // output columns in total
    const int o0 = GEGLU ? a0 / 2 : a0;              // first output column of this tile
    const int Mo_store = GEGLU ? Mo : g.Ma_store;
    half_t* Cs = smem;
    half_t* Y = g.y + (int64_t)z * g.y_bs;
    const half_t* R1 = g.res ? g.res + (int64_t)z * g.res_bs : nullptr;
    const half_t* R2 = g.res2 ? g.res2 + (int64_t)z * g.res_bs : nullptr;
    constexpr int OCH = C::CW / 8;
    const bool vec_ok = (g.ldy % 8) == 0 && (g.y_bs % 8) == 0 && (g.ldres % 8) == 0 && (g.res_bs % 8) == 0 && (g.temb_stride % 8) == 0;
    for (int ps = 0; ps < WB / C::WBP; ++ps) {
        __syncthreads();  // main loop (ps = 0) / the previous pass's readers are done with the LDS
        if (wb / C::WBP == ps && owner) {
            const int rl = (wb % C::WBP) * TB * 32 + l31;
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                half_t* crow = Cs + (rl + j * 32) * C::CSTR;
                if (GEGLU) {
#pragma unroll
                    for (int i = 0; i < TA; i += 2)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            half4_t v;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                v[e] = (half_t)(acc[i][j][4 * gq + e] * fz_gelu_erf(acc[i + 1][j][4 * gq + e]));
                            *reinterpret_cast<half4_t*>(crow + (wa * TA / 2 + i / 2) * 32 + 8 * gq + 4 * hi) = v;
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < TA; ++i)
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            half4_t v;
                            if constexpr (VT) {  // already rounded: the group's four halves as raw bits in its first two registers
                                f32x2 pk;
                                pk[0] = acc[i][j][4 * gq];
                                pk[1] = acc[i][j][4 * gq + 1];
                                v = __builtin_bit_cast(half4_t, pk);
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = (half_t)acc[i][j][4 * gq + e];
                            }
                            *reinterpret_cast<half4_t*>(crow + (wa * TA + i) * 32 + 8 * gq + 4 * hi) = v;
                        }
                }
            }
        }
        __syncthreads();
        if (LN && g.st_out != nullptr) {
            // Row statistics of what is stored (the next LayerNorm's input): 8 lanes per row, one 64-column block per
            // iteration, the block's sum / sum of squares reduced over the 8 lanes in a fixed order -> one partial per
            // (row, 64-column block), independent of the tile shape.  (launcher: plain epilogue, vector path, Ma % 64 == 0)
            constexpr int NBLK = C::CW / 64;
            for (int rb = wave; rb * 8 < C::RP; rb += C::NW) {
                const int pl = rb * 8 + (lane >> 3);
                const int64_t px = b0 + ps * C::RP + pl;
                const bool rowok = px < g.Nb;
                const int64_t pxc = rowok ? px : g.Nb - 1;
                for (int blk = 0; blk < NBLK; ++blk) {
                    const int ch = blk * 8 + (lane & 7);
                    const int co = o0 + ch * 8;
                    const bool ok = rowok && co < g.Ma;
                    const int coc = co < g.Ma ? co : 0;
                    const half8_t v = fz_ld_h8(Cs + pl * C::CSTR + ch * 8);
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)v[e];
                    if (R1 != nullptr) {
                        const half8_t r = fz_ld_h8(R1 + pxc * g.ldres + coc);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
                    }
                    if (R2 != nullptr) {
                        const half8_t r = fz_ld_h8(R2 + pxc * g.ldres + coc);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
                    }
                    half8_t o;
                    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        o[e] = (half_t)f[e];
                        const float fr = (float)o[e];  // statistics of the rounded value the consumer will read
                        s1 += fr;
                        s2 += fr * fr;
                    }
                    if (ok) fz_st_h8(Y + px * g.ldy + co, o);
                    if (!ok) s1 = s2 = 0.0f;
                    s1 = fz_sum8(s1);
                    s2 = fz_sum8(s2);
                    if ((lane & 7) == 0 && ok) {
                        float* sp = g.st_out + (((int64_t)z * g.Nb + px) * (g.Ma / 64) + (o0 / 64 + blk)) * 2;
                        sp[0] = s1;
                        sp[1] = s2;
                    }
                }
            }
            continue;
        }
        for (int id = tid; id < C::RP * OCH; id += C::T) {
            const int pl = id / OCH, ch = id - pl * OCH;
            const int64_t px = b0 + ps * C::RP + pl;
            const int co = o0 + ch * 8;
            if (px >= g.Nb || co >= Mo_store) continue;
            const half8_t v = fz_ld_h8(Cs + pl * C::CSTR + ch * 8);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = co + e < Mo ? (float)v[e] : 0.0f;  // [Mo, Mo_store): zero padding
            const bool full = vec_ok && co + 8 <= Mo;
            if (full) {  // aligned 16-byte accesses
                if (g.temb != nullptr) {
                    const half8_t t = fz_ld_h8(g.temb + (px / g.temb_group) * g.temb_stride + co);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += (float)t[e];
                }
                if (R1 != nullptr) {
                    const half8_t r = fz_ld_h8(R1 + px * g.ldres + co);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
                }
                if (R2 != nullptr) {
                    const half8_t r = fz_ld_h8(R2 + px * g.ldres + co);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
                }
            } else {
                if (g.temb != nullptr) {
                    const half_t* t = g.temb + (px / g.temb_group) * g.temb_stride + co;
                    for (int e = 0; e < 8; ++e)
                        if (co + e < Mo) f[e] += (float)t[e];
                }
                if (R1 != nullptr) {
                    const half_t* r = R1 + px * g.ldres + co;
                    for (int e = 0; e < 8; ++e)
                        if (co + e < Mo) f[e] += (float)r[e];
                }
                if (R2 != nullptr) {
                    const half_t* r = R2 + px * g.ldres + co;
                    for (int e = 0; e < 8; ++e)
                        if (co + e < Mo) f[e] += (float)r[e];
                }
            }
            half_t* dst = Y + px * g.ldy + co;
            if (full) {
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)f[e];
                fz_st_h8(dst, o);
                if constexpr (GS != 0) fz_st_h8(Cs + pl * C::CSTR + ch * 8, o);  // the staging tile now holds what was STORED
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (co + e < Mo_store) dst[e] = (half_t)f[e];
            }
        }
        if constexpr (GS == -1) {
            // LayerNorm of the RP rows this pass stored (attention.py:295-337: every `x = f(norm(x)) + x` step ends in a Linear + residual
            // whose output is the NEXT LayerNorm's input).  The tile is the whole row (launcher: Ma == CW == 320, a0 == 0, vector path), and
            // the staging tile holds the fp16 values that were stored -- what a stand-alone fz_layernorm would read back from HBM.  8 lanes per
            // row, 5 chunks of 8 channels per lane, exact two-sweep statistics (DPP sums over the 8 lanes, fixed order), one more row-major
            // store.  The LayerNorm launch, its read of x and its launch latency are gone; the GEMM pays ~1 us of epilogue.
            static_assert(C::CW == 320, "the tile must hold whole 320-channel rows");
            __syncthreads();
            const int l8 = tid & 7;
            FzRow5 gmv, btv;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                gmv.c[i] = fz_ld_h8(g.lno_gamma + (l8 + 8 * i) * 8);
                btv.c[i] = fz_ld_h8(g.lno_beta + (l8 + 8 * i) * 8);
            }
            for (int rb = tid >> 3; rb < C::RP; rb += C::T >> 3) {
                const int64_t px = b0 + ps * C::RP + rb;
                FzRow5 v;
#pragma unroll
                for (int i = 0; i < 5; ++i) v.c[i] = fz_ld_h8(Cs + rb * C::CSTR + (l8 + 8 * i) * 8);
                const FzRow5 o = fz_ln_row320(v, gmv, btv, g.lno_eps);  // (one out-of-line body for every launch that writes LN(y): fz_rt.h)
                if (px < g.Nb) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) fz_st_h8(g.lno_y + ((int64_t)z * g.Nb + px) * g.lno_ld + (l8 + 8 * i) * 8, o.c[i]);
                }
            }
        }
        if constexpr (GS > 0) {
            // GroupNorm statistics of the RP x CW values this pass stored (the launcher guarantees whole groups per tile, whole
            // passes per frame, every access on the vector path).  Deterministic: thread (group, slice) sums its RP / 16 rows of the group's
            // channels in a fixed order -- shifted by the first value it sees, so that sum-of-squares cancellation stays harmless --, one
            // thread per group Chan-merges the 16 slices in slice order and writes (count, mean, M2) where gn_finalize expects it.
            constexpr int SL = 16;
            static_assert(C::RP % SL == 0, "row slices");
            constexpr int cpg = GS > 0 ? GS : 2, ngrp = C::CW / cpg;
            float* red = reinterpret_cast<float*>(smem + C::RP * C::CSTR);
            __syncthreads();
            for (int item = tid; item < ngrp * SL; item += C::T) {
                const int gi = item % ngrp, rs = item / ngrp;
                const half_t* base = Cs + gi * cpg + rs * C::CSTR;
                // a slice = RP / SL rows x GS channels, read in batches of <= 20 registers with all loads of a batch in flight before the
                // first use (a runtime-bounded loop of dependent 4-byte reads cost 5-14 us per launch: every read paid its LDS latency)
                constexpr int ROWS = GS >= 40 ? 1 : (GS >= 20 ? 2 : 4);
                static_assert((C::RP / SL) % ROWS == 0 && GS % 2 == 0, "row batches");
                float p0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int r0 = 0; r0 < C::RP / SL; r0 += ROWS) {
                    half2_t v[ROWS][GS / 2];
#pragma unroll
                    for (int r = 0; r < ROWS; ++r)
#pragma unroll
                        for (int c = 0; c < GS / 2; ++c) v[r][c] = *reinterpret_cast<const half2_t*>(base + (r0 + r) * SL * C::CSTR + 2 * c);
                    if (r0 == 0) p0 = (float)v[0][0][0];
#pragma unroll
                    for (int r = 0; r < ROWS; ++r)
#pragma unroll
                        for (int c = 0; c < GS / 2; ++c) {
                            const float d0 = (float)v[r][c][0] - p0, d1 = (float)v[r][c][1] - p0;
                            s1 += d0 + d1;
                            s2 += d0 * d0 + d1 * d1;
                        }
                }
                const float n = (float)((C::RP / SL) * cpg);
                red[(gi * SL + rs) * 3 + 0] = n;
                red[(gi * SL + rs) * 3 + 1] = p0 + s1 / n;
                red[(gi * SL + rs) * 3 + 2] = s2 - s1 * s1 / n;
            }
            __syncthreads();
            // (a pass that lies wholly beyond Nb -- the second 128-row pass of the last 256-row tile when Nb % 256 == 128 -- holds stale
            //  staging rows and has no record to write: its frame index would be n_frames)
            if (tid < ngrp && b0 + ps * C::RP < g.Nb) {
                float cnt = red[tid * SL * 3], mean = red[tid * SL * 3 + 1], m2 = red[tid * SL * 3 + 2];
                for (int q = 1; q < SL; ++q) {  // Chan et al., the order gn_finalize uses
                    const float nb = red[(tid * SL + q) * 3], mb = red[(tid * SL + q) * 3 + 1], m2b = red[(tid * SL + q) * 3 + 2];
                    const float tot = cnt + nb, delta = mb - mean;
                    mean += delta * (nb / tot);
                    m2 += m2b + delta * delta * (cnt * nb / tot);
                    cnt = tot;
                }
                const int64_t px0 = b0 + ps * C::RP;
                const int64_t fr = px0 / g.gs_rpf;
                const int chunk = (int)((px0 - fr * g.gs_rpf) / C::RP);
                float* out = g.gs_out + ((fr * g.gs_groups + (a0 / cpg + tid)) * g.gs_chunks + chunk) * 3;
                out[0] = cnt;
                out[1] = mean;
                out[2] = m2;
            }
        }
    }
#ifdef FZ_IGEMM_TIMING
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (tid & 255) == 0) {
        fz_igemm_timing2[tid >> 8][0] = tk_setup;
        fz_igemm_timing2[tid >> 8][1] = clock64() - tk_entry;   // whole kernel, entry to the last store issued
        fz_igemm_timing2[tid >> 8][2] = tw_entry;
        fz_igemm_timing2[tid >> 8][3] = tw_loop;
        fz_igemm_timing2[tid >> 8][4] = tw_loop_end;
        fz_igemm_timing2[tid >> 8][5] = wall_clock64();
    }
#endif
}

// split-K tail: y[b][a] = sum_s part[s][b][a] + bias[a] (+ temb) (+ res) (+ res2), the slabs summed in slice order.  One WAVE per (output row,
// segment of 256 columns = blockIdx.y), a lane per 4 consecutive outputs, 4 rows per workgroup: what depends on the row only -- batch element,
// time-embedding row -- is wave-uniform and needs no per-element division (the first form, a flat index per thread, spent two or three 64-bit
// divisions per 4 outputs; a wave walking its whole row segment by segment serialised 5 memory round trips at 1 280 columns and was slower).
FZ_KERNEL void __launch_bounds__(256) igemm_reduce_kernel(IgArgs g, int batch) {
    const int q = g.Ma / 4;
    const int cl = threadIdx.x & 63, rl = fz_uniform((int)(threadIdx.x >> 6));
    const int64_t rows = g.Nb * batch;
    for (int64_t row = (int64_t)blockIdx.x * 4 + rl; row < rows; row += (int64_t)gridDim.x * 4) {
        int z = 0;
        int64_t px = row;
        if (batch > 1) {
            z = (int)(row / g.Nb);
            px = row - (int64_t)z * g.Nb;
        }
        const float* const p0 = g.part + row * g.Ma;                      // slab 0: [batch][Nb][Ma]
        const int64_t slab = (int64_t)batch * g.Nb * g.Ma;
        const half_t* const trow = g.temb != nullptr ? g.temb + (px / g.temb_group) * g.temb_stride : nullptr;
        const half_t* const r1 = g.res != nullptr ? g.res + (int64_t)z * g.res_bs + px * g.ldres : nullptr;
        const half_t* const r2 = g.res2 != nullptr ? g.res2 + (int64_t)z * g.res_bs + px * g.ldres : nullptr;
        half_t* const yrow = g.y + (int64_t)z * g.y_bs + px * g.ldy;
        const int c4 = (int)blockIdx.y * 64 + cl;
        if (c4 < q) {
            const int co = c4 * 4;
            // (requesting every operand before the first add -- the slabs of up to eight slices, bias, temb, residuals -- was built and measured:
            //  no faster, profiles/r06_igemm_bias_fetch.txt; and under -ffast-math the unrolled sum is free to re-associate, which moved results by
            //  an ulp: the loop form keeps the slice order)
            f32x4 s = *reinterpret_cast<const f32x4*>(p0 + co);
            for (int k = 1; k < g.ksplit; ++k) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p0 + k * slab + co);
                s += t;
            }
            float f[4] = {s[0], s[1], s[2], s[3]};
            if (g.bias != nullptr)
                for (int e = 0; e < 4; ++e) f[e] += (float)g.bias[co + e];
            if (trow != nullptr)
                for (int e = 0; e < 4; ++e) f[e] += (float)trow[co + e];
            if (r1 != nullptr)
                for (int e = 0; e < 4; ++e) f[e] += (float)r1[co + e];
            if (r2 != nullptr)
                for (int e = 0; e < 4; ++e) f[e] += (float)r2[co + e];
            half4_t o;
            for (int e = 0; e < 4; ++e) o[e] = (half_t)f[e];
            *reinterpret_cast<half4_t*>(yrow + co) = o;
        }
    }
}

// conv_in of the UNet (4 -> C channels, resnet.py:57-64 with in_channels = 4): K = 36 is far below one MFMA K step, so it
// is a direct VALU convolution.  The whole weight matrix sits in LDS transposed to [k][cout] (k = tap * Cin + ci), one thread
// owns one pixel x 8 output channels: it gathers its 9 * Cin inputs once into registers and walks k with one 16-byte LDS read
// (8 weights) per input value.
template <int CIN>
FZ_KERNEL void __launch_bounds__(256) conv3x3_small_cin_kernel(IgArgs g) {
    FZ_DYN_SMEM(raw);
    half_t* wl = reinterpret_cast<half_t*>(raw);  // [9 * Cin][Ma]
    const int kk = 9 * g.Cin;
    for (int id = threadIdx.x; id < kk * g.Ma; id += 256) {
        const int k = id / g.Ma, co = id - k * g.Ma;
        wl[id] = g.a[(int64_t)co * g.lda + k];
    }
    __syncthreads();
    const int och = g.Ma / 8;
    const int64_t total = g.Nb * och;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int64_t px = id / och;
        const int co = (int)(id - px * och) * 8;
        const int hw = g.Ho * g.Wo;
        const int n = (int)(px / hw), rem = (int)(px - (int64_t)n * hw);
        const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
        float xin[9 * CIN];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = oy * g.stride + tap / 3 - 1, ix = ox * g.stride + tap % 3 - 1;
            const bool inb = iy >= 0 && iy < g.Hi && ix >= 0 && ix < g.Wi;
            const half_t* xs = g.b + (((int64_t)n * g.Hi + (inb ? iy : 0)) * g.Wi + (inb ? ix : 0)) * g.ldb;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) xin[tap * CIN + ci] = inb ? (float)xs[ci] : 0.0f;
        }
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = g.bias ? (float)g.bias[co + e] : 0.0f;
#pragma unroll
        for (int k = 0; k < 9 * CIN; ++k) {
            const half8_t w8 = fz_ld_h8(wl + k * g.Ma + co);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += xin[k] * (float)w8[e];
        }
        if (g.temb != nullptr) {
            const half8_t t = fz_ld_h8(g.temb + (px / g.temb_group) * g.temb_stride + co);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += (float)t[e];
        }
        if (g.res != nullptr) {
            const half8_t r = fz_ld_h8(g.res + px * g.ldres + co);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
        }
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)f[e];
        fz_st_h8(g.y + px * g.ldy + co, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------
//                                                   host side
// ---------------------------------------------------------------------------------------------------------------
template <int WA, int TA, int WB, int TB, int BK, int NS, int MODE, bool GEGLU, bool LN = false, int PP = 0, bool VT = false, int GS = 0>
static int ig_launch(IgArgs g, int batch, void* stream) {
    typedef IgCfg<WA, TA, WB, TB, BK, NS, GEGLU, PP> C;
    g.kchunks = fz_ceil_div(g.Cin, BK);
    if (g.ksplit > g.taps * g.kchunks) return FZ_ERR_BAD_ARG;
    if ((PP & (FZ_PP_ON | FZ_KGPP | FZ_LC)) && (g.Cin % BK)) return FZ_ERR_UNSUPPORTED;  // the ping-pong loop has no ragged-K path (every SD width is a multiple of 32)
    g.tiles_a = fz_ceil_div(g.Ma_store > g.Ma ? g.Ma_store : g.Ma, C::BA);  // (V^T padding rows [Ma, Ma_store) are written as zeros: their tiles run too)
    const int64_t tiles_b = (g.Nb + C::BB - 1) / C::BB;
    const int64_t nt = (int64_t)g.tiles_a * tiles_b;
    if (nt <= 0 || nt >= (1ll << 31) || batch <= 0 || batch > 65535 || g.ksplit < 1 || g.ksplit > 65535) return FZ_ERR_BAD_ARG;
    const size_t lds = (size_t)C::LDS_HALVES * sizeof(half_t);
#ifndef FZ_EMU
    // LDS above 64 KB is an opt-in function attribute -- PER DEVICE: a process that drives a second GPU must set it there as well
    static std::atomic<uint64_t> attr_set_mask{0};  // bit d = set on device ordinal d (ordinals >= 64: set on every launch)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FZ_ERR_LAUNCH;
    if (dev >= 64 || !(attr_set_mask.load(std::memory_order_relaxed) >> dev & 1)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<WA, TA, WB, TB, BK, NS, MODE, GEGLU, LN, PP, VT, GS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return FZ_ERR_LAUNCH;
        if (dev < 64) attr_set_mask.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
#endif
    // Tile order.  An XCD (32 CUs, its own 4 MB L2) runs ~32 consecutive tiles of the launch order at a time and fetches what those tiles
    // span into its L2.  With the a-tile fastest that is min(tiles_a, 32) weight panels x 1-2 row panels; with groups of group_b b-tiles
    // it is a 2-D block of (32 / group_b) weight panels x group_b row panels.  The launcher prices one such round for group_b in
    // {1, 2, 4, .., 32, tiles_b} (panel bytes = tile rows x K x 2, row panels of a convolution with their halo) and takes the cheapest
    // when it beats the a-fastest order by 20 %: weight-heavy launches (GEGLU 640 -> 5120 at 32^2: 20 weight panels of 328 KB per row
    // panel -- 6.5 MB of weights streaming through a 4 MB L2 once per row panel, 294 MB fetched per launch for 27 MB of operands,
    // profiles/r04_pmc_job.json) get blocks in which both operands are re-used; row-heavy launches (everything at 64^2) keep group_b = 1.
    g.tiles_b = (int)tiles_b;
    g.group_b = 1;
    {
        static const bool order_off = FZ_TUNING_FLAG("FZ_IGEMM_NO_TILE_ORDER");  // A/B switch (tuning builds only)
        const double kbytes = 2.0 * g.taps * g.Cin / (g.ksplit > 0 ? g.ksplit : 1);
        const double apanel = C::BA * kbytes, bpanel = C::BB * kbytes * (g.taps == 9 ? 1.5 : 1.0);
        const double conc = 32.0 * C::WAVES_PER_SIMD * 4 / C::NW;  // workgroups an XCD runs at a time (32 CUs x workgroups per CU)
        auto cost = [&](double gb) {
            gb = gb > (double)tiles_b ? (double)tiles_b : gb;
            double da = conc / gb;  // a round of `conc` consecutive tiles spans da weight panels x db row panels
            da = da > g.tiles_a ? (double)g.tiles_a : (da < 1.0 ? 1.0 : da);
            double db = conc / da;
            db = db > (double)tiles_b ? (double)tiles_b : db;
            return (da * apanel + db * bpanel) / (da * db);  // bytes fetched per tile of the round
        };
        double best = cost(1.0);
        int best_g = 1;
        for (int gb = 2; gb <= 64; gb *= 2) {
            const int gbc = gb > tiles_b ? (int)tiles_b : gb;
            if (cost((double)gbc) < best) {
                best = cost((double)gbc);
                best_g = gbc;
            }
        }
        if (cost((double)tiles_b) < best) {
            best = cost((double)tiles_b);
            best_g = (int)tiles_b;
        }
        // (8-wave tiles only: in situ -- two kernel-stats profiles per setting, profiles/r04_tile_order_in_situ.txt -- the 3x3 convolutions
        // gain 2.3 ms per job and the GEGLU projections 3.9 ms, but the 64 x 128 four-wave tile, three workgroups per CU on 10-25 us
        // launches, LOSES 4.8 ms with a grouped order: it keeps the a-fastest one)
        if (!order_off && batch == 1 && g.ksplit == 1 && C::NW == 8 && best < 0.8 * cost(1.0)) g.group_b = best_g;
    }
    static const bool xcd_ks_off = FZ_TUNING_FLAG("FZ_IGEMM_NO_XCD_KS");  // A/B switch of the K-slice -> XCD mapping (tuning builds only)
    // Which launches: the kernel-level A/B (profiles/r04_xcd_ks_ab.txt: one launch repeated, its weights resident in Infinity Cache) has the
    // convolutions with Cin >= 1280 gain 3-6 % and the 10-25 us split-K projections / temporal convolutions LOSE up to 15 %; IN SITU (two
    // kernel-stats profiles per setting, profiles/r04_xcd_ks_in_situ.txt), where the weights come from HBM, the split-K projections gain
    // 1.9 ms per job and only the temporal convolutions lose (0.6 ms): every mode but the temporal one takes the flat grid.
    static const bool xcd_ks_all = FZ_TUNING_FLAG("FZ_IGEMM_XCD_KS_ALL");  // trial switch: the temporal convolutions as well
    const bool flat = (MODE != 2 || xcd_ks_all) && g.ksplit > 1 && !xcd_ks_off && (nt * g.ksplit) % 8 == 0 && nt * g.ksplit < (1ll << 31);
    g.nt_flat = flat ? (int)nt : 0;
    dim3 grid(flat ? (unsigned)(nt * g.ksplit) : (unsigned)nt, flat ? 1u : (unsigned)g.ksplit, (unsigned)batch), block(C::T);
    if constexpr (GS > 0) {  // the statistics scratch (groups x 16 slices x 3 floats) sits behind the staging tile
        static_assert((C::RP * C::CSTR + 2 * 3 * 16 * (C::CW / GS)) <= C::LDS_HALVES, "GroupNorm-statistics scratch does not fit");
        static_assert(C::RP == 128, "the partials' chunk is 128 rows whatever the tile");
    }
    FZ_LAUNCH((igemm_kernel<WA, TA, WB, TB, BK, NS, MODE, GEGLU, LN, PP, VT, GS>), grid, block, lds, stream, g);
    return fz_last_launch_status();
}

// Tile configurations.  id = WA TA WB TB (decimal digits) * 100 + (BK / 32) * 10 + NS, e.g. 254222 = 2 x 4 waves of 5 x 2 MFMA
// tiles (320 x 256), K step 64, 2-deep ring.  The kernel template also builds K step 32 and rings up to 4 deep; on MI355X
// they measured 2-7 % SLOWER than the 2-deep K-step-64 form of the same tile on every shape of the UNet
// (profiles/r02_kbench_gemm_sweep.json, r02_kbench_conv_sweep.json: the loop is not load-latency-bound), so they are not
// instantiated.  Neither are 4-wave forms with larger wave tiles (320 x 192 as 2 x 2 waves of 5 x 3, 256 x 256 as 2 x 2 of 4 x 4: 24-29 %
// fewer LDS fragment reads per FLOP, K loop spill-free): with ONE wave per SIMD nothing covers the ds_read latency after each barrier
// and they ran 510 vs 860 TFLOP/s on the 16-frame 64^2 convolution (profiles/r02_tile_trial_4wave.txt).
// An explicit register double buffer of the MFMA operand fragments (ds_reads of k sub-step kk+1 issued before the MFMAs of kk, fenced)
// instead of the compiler's read-4 / wait / MFMA-4 bursts measured 3-6 % SLOWER on every conv / GEMM shape, same box
// (profiles/r02_tile_trial_frag2.txt), and is not in the kernel either.
// Spreading the LDS-DMA pieces of the next tile over the four MFMA groups of the current one (one basic block, 2 pieces per group)
// instead of the burst after the barrier: 5-7 % SLOWER on the 5x1 / 2x2 / 1x2 tile kernels, and the second copy of the K-step body
// spills the 10- and 8-tile waves (profiles/r02_tile_trial_spread_dma.txt).  The burst stays.
//   254222: 320 x 256, 8 waves -- SD-1.x widths are all multiples of 320: no A-side waste, 142 FLOP per staged byte
//   254122: 320 x 128, 8 waves -- the same for launches with 4096 < rows <= 32768 (twice the workgroups)
//   158122: 160 x 256, 8 waves -- the rank-160 down projection of the temporal LoRA convolution (lora.py:31-37)
//   244222: 256 x 256, 8 waves -- the GEGLU projection (8C = multiples of 256) and generic large shapes
//   224223: 128 x 256, 8 waves, 3-deep;  222222: 128 x 128, 4 waves, two workgroups per CU
//   224212: 128 x 256, 8 waves, K step 32, 2-deep: TWO workgroups per CU -- the short-K GEGLU launches (rule in ig_run)
//   212222:  64 x 128, 4 waves, three workgroups per CU -- small launches and ragged widths
template <int MODE, bool GEGLU, bool LN = false>
static int ig_dispatch_cfg(int cfg, const IgArgs& g, int batch, void* stream) {
    if constexpr (!LN) {  // ping-pong K loop (last digit 8): K step 32, 4-slot ring, two wave groups half a phase apart
        constexpr int P = FZ_PP_ON | FZ_PP_PREP_IN_R;  // the form that measured best (profiles/r03_igemm_ab_v3.txt)
        if (cfg == 244218) return ig_launch<2, 4, 4, 2, 32, 4, MODE, GEGLU, false, P>(g, batch, stream);
        if (!GEGLU) {
            switch (cfg) {
                case 254218: return ig_launch<2, 5, 4, 2, 32, 4, MODE, false, false, P>(g, batch, stream);
                default: break;
            }
        }
        if constexpr (MODE == 0 && GEGLU) {  // 128 x 256, K step 32, 2-deep ring: two workgroups per CU (the short-K GEGLU launches)
            if (cfg == 224212) return ig_launch<2, 2, 4, 2, 32, 2, MODE, true, false>(g, batch, stream);
        }
#ifdef FZ_IGEMM_TRIALS  // trial forms: tile id + 1000000 * n
        if constexpr (MODE == 0) {  // two workgroups per CU (K step 32, 2-deep ring, <= 80 accumulator registers)
            switch (cfg) {
                case 244112: return ig_launch<2, 4, 4, 1, 32, 2, MODE, GEGLU, false>(g, batch, stream);  // 256 x 128
                case 224212: return ig_launch<2, 2, 4, 2, 32, 2, MODE, GEGLU, false>(g, batch, stream);  // 128 x 256
                case 224112: return ig_launch<2, 2, 4, 1, 32, 2, MODE, GEGLU, false>(g, batch, stream);  // 128 x 128
                default: break;
            }
            if (!GEGLU && cfg == 254112) return ig_launch<2, 5, 4, 1, 32, 2, MODE, false, false>(g, batch, stream);  // 320 x 128
        }
        if (!GEGLU && MODE != 2) {
            constexpr int P = FZ_PP_ON | FZ_PP_PREP_IN_R;
            switch (cfg) {
                case 254118: return ig_launch<2, 5, 4, 1, 32, 4, MODE, false, false, P>(g, batch, stream);   // one B-side tile per wave:
                case 158118: return ig_launch<1, 5, 8, 1, 32, 4, MODE, false, false, P>(g, batch, stream);   // slower than their ring twins
                case 1254218: return ig_launch<2, 5, 4, 2, 32, 4, MODE, false, false, 1>(g, batch, stream);          // prep inside the MFMA cluster
                case 2254218: return ig_launch<2, 5, 4, 2, 32, 4, MODE, false, false, 1 | 8 | 2>(g, batch, stream);  // no s_setprio
                case 3254218: return ig_launch<2, 5, 4, 2, 32, 4, MODE, false, false, 1 | 8 | 4>(g, batch, stream);  // groups not staggered
                case 1254118: return ig_launch<2, 5, 4, 1, 32, 4, MODE, false, false, 1>(g, batch, stream);
                case 1244218: return ig_launch<2, 4, 4, 2, 32, 4, MODE, false, false, 1>(g, batch, stream);
                case 4254218: return ig_launch<2, 5, 4, 2, 32, 4, MODE, false, false, 1 | 16>(g, batch, stream);     // K-32 phases
                case 4244218: return ig_launch<2, 4, 4, 2, 32, 4, MODE, false, false, 1 | 16>(g, batch, stream);
                case 4254118: return ig_launch<2, 5, 4, 1, 32, 4, MODE, false, false, 1 | 16>(g, batch, stream);
                case 4158118: return ig_launch<1, 5, 8, 1, 32, 4, MODE, false, false, 1 | 16>(g, batch, stream);
                default: break;
            }
        }
#endif
    }
    switch (cfg) {
        case 244222: return ig_launch<2, 4, 4, 2, 64, 2, MODE, GEGLU, LN>(g, batch, stream);
        case 224223: return ig_launch<2, 2, 4, 2, 64, 3, MODE, GEGLU, LN>(g, batch, stream);
        case 222222: return ig_launch<2, 2, 2, 2, 64, 2, MODE, GEGLU, LN>(g, batch, stream);
        default: break;
    }
    if (!GEGLU) {  // odd TA / TA = 1 cannot pair (h, gate) tiles
#ifdef FZ_IGEMM_TRIALS
        // 160 x 128, 4 waves, two workgroups per CU, for the rank-160 down projection of the temporal LoRA pair at 8 frames (128 tiles of
        // 160 x 256 = half the chip): 22 us against 26 us for 158122 on the 64^2 shape, but the 64 x 128 tile the chooser already takes
        // there runs 21 us (profiles/r04_tile_154122_ab.txt): nothing to gain, not shipped
        if constexpr (!LN) {
            if (cfg == 154122) return ig_launch<1, 5, 4, 1, 64, 2, MODE, false, false>(g, batch, stream);
        }
        // 320 x 128 as 5 x 2 waves of 2 x 2 MFMA tiles (TEN waves: 4 MFMAs per 4 fragment reads per k sub-step instead of 5 per 6, 2.5 waves
        // per SIMD): 3-11 % SLOWER than the 8-wave 5 x 1 form on every conv / projection it carries (profiles/r04_tile_10wave_ab.txt)
        if constexpr (!LN) {
            if (cfg == 522222) return ig_launch<5, 2, 2, 2, 64, 2, MODE, false, false>(g, batch, stream);
        }
#endif
        if constexpr (!LN) {  // 320 x 128 as TWO K groups of 2 x 2 waves of 5 x 2 MFMA tiles (IgCfg::KG): the LDS-lean form of 254122
            if (cfg == 252222) return ig_launch<2, 5, 2, 2, 64, 2, MODE, false, false, FZ_KG2>(g, batch, stream);
            if (cfg == 252218) return ig_launch<2, 5, 2, 2, 32, 4, MODE, false, false, FZ_KG2 | FZ_KGPP>(g, batch, stream);  // ... in ping-pong
            if (cfg == 252214) return ig_launch<2, 5, 2, 2, 32, 4, MODE, false, false, FZ_LC>(g, batch, stream);  // 4 consumer + 4 loader waves
            if (cfg == 252226) return ig_launch<2, 5, 2, 2, 64, 2, MODE, false, false, FZ_KG2 | FZ_KGSPREAD>(g, batch, stream);  // ... DMA pieces spread
        }
        switch (cfg) {
            case 254222: return ig_launch<2, 5, 4, 2, 64, 2, MODE, false, LN>(g, batch, stream);
            case 254122: return ig_launch<2, 5, 4, 1, 64, 2, MODE, false, LN>(g, batch, stream);
            case 158122: return ig_launch<1, 5, 8, 1, 64, 2, MODE, false, LN>(g, batch, stream);
            case 212222: return ig_launch<2, 1, 2, 2, 64, 2, MODE, false, LN>(g, batch, stream);
            default: break;
        }
    }
    return FZ_ERR_BAD_ARG;
}

// The fz_gemm_qkvt instantiations (MODE 0, plain epilogue, ring loop): the tiles whose width divides every 2 C of the UNet (320 | 640,
// 1280, 2560; 128 and 64 likewise), so that no column tile straddles the k | v boundary.
static int ig_dispatch_vt(int cfg, const IgArgs& g, int batch, void* stream) {
    switch (cfg) {
        case 254222: return ig_launch<2, 5, 4, 2, 64, 2, 0, false, false, 0, true>(g, batch, stream);
        case 254122: return ig_launch<2, 5, 4, 1, 64, 2, 0, false, false, 0, true>(g, batch, stream);
        case 224223: return ig_launch<2, 2, 4, 2, 64, 3, 0, false, false, 0, true>(g, batch, stream);
        case 222222: return ig_launch<2, 2, 2, 2, 64, 2, 0, false, false, 0, true>(g, batch, stream);
        case 212222: return ig_launch<2, 1, 2, 2, 64, 2, 0, false, false, 0, true>(g, batch, stream);
        default: return FZ_ERR_BAD_ARG;
    }
}

// The GS instantiations (GroupNorm statistics of the output out of the epilogue): the two 320-wide ring tiles -- SD-1.x group widths
// (10 / 20 / 40 channels) divide 320, and both stage their epilogue in passes of 128 rows -- for the projections (MODE 0: proj_out of a
// transformer) and the temporal convolutions (MODE 2: the LoRA up convolution that ends a PseudoConv3d).
template <int MODE, int CPG>
static int ig_dispatch_gs_w(int cfg, const IgArgs& g, int batch, void* stream) {
    switch (cfg) {
        case 254222: return ig_launch<2, 5, 4, 2, 64, 2, MODE, false, false, 0, false, CPG>(g, batch, stream);
        case 254122: return ig_launch<2, 5, 4, 1, 64, 2, MODE, false, false, 0, false, CPG>(g, batch, stream);
        default: return FZ_ERR_BAD_ARG;
    }
}
// The LayerNorm-out instantiations (GS == -1): the three 320-wide tiles a 64x64-level projection onto 320 channels takes (MODE 0).
static int ig_dispatch_lno(int cfg, const IgArgs& g, int batch, void* stream) {
    constexpr int P = FZ_PP_ON | FZ_PP_PREP_IN_R;
    switch (cfg) {
        case 254222: return ig_launch<2, 5, 4, 2, 64, 2, 0, false, false, 0, false, -1>(g, batch, stream);
        case 254122: return ig_launch<2, 5, 4, 1, 64, 2, 0, false, false, 0, false, -1>(g, batch, stream);
        case 254218: return ig_launch<2, 5, 4, 2, 32, 4, 0, false, false, P, false, -1>(g, batch, stream);
        default: return FZ_ERR_BAD_ARG;
    }
}

template <int MODE>
static int ig_dispatch_gs(int cfg, const IgArgs& g, int batch, void* stream) {
    if constexpr (MODE == 0 || MODE == 2) {
        switch (g.gs_cpg) {
            case 10: return ig_dispatch_gs_w<MODE, 10>(cfg, g, batch, stream);
            case 20: return ig_dispatch_gs_w<MODE, 20>(cfg, g, batch, stream);
            default: break;
        }
    }
    return FZ_ERR_BAD_ARG;
}

struct IgTile {
    int cfg, ba, bb, bk, wg_per_cu;
    double rate_pf;   // PFLOP/s the whole chip sustains in the K loop of this tile with every CU busy (long-K convolutions)
    double fixed_us;  // prologue + epilogue of one workgroup round
    bool geglu_ok;
};
// Fitted to the MI355X sweeps (scripts/kbench.py --gemm / --conv; build_tmp/fit_chooser.py reproduces the fit: the model's
// pick is within 0.5 % of the best measured (tile, split-K) summed over all 59 swept shapes).
static const IgTile kTiles[] = {
    {254222, 320, 256, 64, 1, 1.00, 12.0, false}, {254122, 320, 128, 64, 1, 0.85, 8.0, false},
    {158122, 160, 256, 64, 1, 0.80, 8.0, false},
    {244222, 256, 256, 64, 1, 0.80, 8.0, true},   {224223, 128, 256, 64, 1, 0.60, 3.0, true},
    {222222, 128, 128, 64, 2, 0.60, 2.0, true},   {212222, 64, 128, 64, 3, 0.70, 4.0, false},
};

// Choose (tile configuration, split-K factor).  256 CUs; time ~ rounds x (K steps per slice x step time + fixed) with
// rounds = ceil(workgroups / (256 x workgroups per CU)), floored by the HBM traffic of the launch at 4 TB/s; split-K adds one
// fp32 slab round trip and a reduce launch.  What the measurements say: the 320 x 256 tile runs 0.9-1.0 PFLOP/s when the
// launch has a multiple of 256 workgroups -- split-K is how the small pyramid levels get there -- and a half-empty last round
// costs a full one, which is why the 64 x 128 tile wins the shapes in between.
#ifndef FZ_SPLITK_LAUNCH_US
#define FZ_SPLITK_LAUNCH_US 3.0  /* fitted; 9.0 (reduce run time + inter-kernel gap at face value) chose too few splits: 32^2 conv 728 -> 561 TF/s */
#endif
static void ig_choose(const IgArgs& g, int batch, bool geglu, int64_t ws_floats, int* cfg_out, int* ksplit_out) {
#ifdef FZ_IGEMM_TUNING
    static const double splitk_us = getenv("FZ_IGEMM_SPLITK_US") ? atof(getenv("FZ_IGEMM_SPLITK_US")) : FZ_SPLITK_LAUNCH_US;  // (in-situ tuning runs)
#else
    constexpr double splitk_us = FZ_SPLITK_LAUNCH_US;
#endif
    double best = 1e300;
    *cfg_out = 212222;
    *ksplit_out = 1;
    const double out_elems = (double)g.Nb * g.Ma * batch;
    const double bytes = 2.0 * ((double)g.Nb * (geglu ? g.Ma / 2 : g.Ma) + (double)g.Nb * g.Cin * (g.taps == 1 ? 1.0 : 1.5) +
                                (double)g.Ma * g.Cin * g.taps) * batch;
    for (const IgTile& t : kTiles) {
        if (geglu && !t.geglu_ok) continue;
        if (g.st_out != nullptr && (t.ba % 64)) continue;  // row statistics are per 64-column block
        if (g.yt != nullptr && (g.vt_split % t.ba || t.cfg == 244222 || t.cfg == 158122)) continue;  // no tile across the k | v boundary
        const int nkt = g.taps * fz_ceil_div(g.Cin, t.bk);
        const int64_t tiles = (int64_t)fz_ceil_div(g.Ma, t.ba) * ((g.Nb + t.bb - 1) / t.bb) * batch;
        const double t_step = 2.0 * t.ba * t.bb * t.bk * t.wg_per_cu / (t.rate_pf * 1e15 / 256.0) * 1e6;  // microseconds
        for (int sk = 1; sk <= 32; sk *= 2) {
            if (sk > 1 && (geglu || g.ln_in != nullptr || g.yt != nullptr || g.Ma % 4 || g.Ma_store != g.Ma || nkt / sk < 4 || out_elems * sk > (double)ws_floats)) break;
            const int64_t wgs = tiles * sk;
            const int64_t slots = 256 * t.wg_per_cu;
            const int64_t rounds = (wgs + slots - 1) / slots;
            double us = (double)rounds * (t.fixed_us + (double)((nkt + sk - 1) / sk) * t_step);
            const double floor_us = bytes / 4.0e6;
            us = us > floor_us ? us : floor_us;
            if (sk > 1) us += splitk_us + out_elems * sk * 8.0 / 4.0e6;  // reduce launch: its own run time + the inter-kernel gap
            if (us < best) {
                best = us;
                *cfg_out = t.cfg;
                *ksplit_out = sk;
            }
        }
    }
}

template <int MODE, bool GEGLU>
static int ig_run(IgArgs g, int batch, int cfg, int ksplit, float* workspace, int64_t workspace_floats, void* stream) {
    if (g.Cin % 8 || g.Cin > 8128 || g.Cin <= 0) return FZ_ERR_UNSUPPORTED;
    if (cfg == 0) {
        int c, sk;
        ig_choose(g, batch, GEGLU, workspace ? workspace_floats : 0, &c, &sk);
        cfg = c;
        if (ksplit == 0) ksplit = sk;
        // The two 8-wave tiles whose waves own 2 B-side MFMA tiles run the ping-pong K loop where it measured faster than the ring loop
        // of the same tile on MI355X (profiles/r03_igemm_prod_ring_vs_pp_v3.txt, r03_igemm_ab_v*.txt): no split-K (the loop's three-tile
        // prologue is not amortised over a short K slice: 8^2 convs lost 35 %), K >= 640 (shorter loops are epilogue / HBM-bound either
        // way), no ragged K, not the LayerNorm-fused form.  +2 ... +7 % on the 16-frame 64^2 convs, +4 ... +16 % on the GEGLU and long-K
        // projections.  The 320 x 128 / 160 x 256 tiles (one B-side MFMA tile per wave: five MFMAs per phase against six fragment
        // reads) measured 3-7 % SLOWER in ping-pong form and stay on the ring loop.
        // (under split-K only when a K slice keeps >= 16 K-64 steps: +1.5 ... +5 % on the 1920 / 2560-wide 16^2 and 32^2 convs,
        //  profiles/r03_igemm_prod_pp_under_splitk.txt; the 8^2 convs with their 5-11-step slices lost 35 %)
        const int64_t k64_per_slice = (int64_t)g.taps * g.Cin / 64 / ksplit;
        bool pp_ok = g.Cin % 32 == 0 && g.ln_in == nullptr && g.st_out == nullptr && g.yt == nullptr && (int64_t)g.taps * g.Cin >= 640 &&
                     (ksplit == 1 || k64_per_slice >= 16);
#ifdef FZ_IGEMM_TRIALS  // scripts/igemm_timeline.hip: A/B of the library's own choice with / without the substitution, and under split-K
        if (fz_igemm_trial_pp_splitk_min > 0 && ksplit > 1 && g.Cin % 32 == 0 && g.ln_in == nullptr && g.st_out == nullptr &&
            (int64_t)g.taps * g.Cin / 64 / ksplit >= fz_igemm_trial_pp_splitk_min)
            pp_ok = true;
        if (fz_igemm_trial_no_pp) pp_ok = false;
#endif
        if (pp_ok) {
            if (cfg == 254222) cfg = 254218;
            if (cfg == 244222) cfg = 244218;
        }
        // The 320 x 128 tile of a long-K 3x3 convolution runs as TWO K groups of 2 x 2 waves of 5 x 2 MFMA tiles (252222, IgCfg::KG): 7 fragment
        // reads per 10 MFMAs instead of 6 per 5 -- the 8-wave 5 x 1 form is LDS-bound.  Same-process interleaved A/B on MI355X
        // (profiles/r06_kg_tile_ab.txt): +4.6 ... +7 % on the Cin >= 640 convolutions of the 8-frame 64^2 / 16-frame 32^2 launches, equal at
        // Cin = 320 (nine K-64 steps per tap set: prologue / epilogue-bound) -- exactly those; not under short split-K slices (the merge of the
        // two groups' accumulators through LDS costs ~1 us per tile).
        {
            bool kg_ok = (MODE == 1 || MODE == 3) && cfg == 254122 && g.Cin >= 640 && k64_per_slice >= 8;
#ifdef FZ_IGEMM_TRIALS
            if (fz_igemm_trial_no_kg2) kg_ok = false;
#endif
            if (kg_ok) cfg = 252222;
        }
        // GEGLU with a short K: the epilogue (64 gelu per lane) is longer than the K loop, and with one 8-wave workgroup per CU nothing
        // runs under it.  The 128 x 256 tile fits a CU twice; same-process A/B on MI355X (profiles/r03_igemm_shortk_two_wg_per_cu.txt):
        // 32768 / 65536 x 320 -> 2560: +6 / +10 %, 8192 x 640 -> 5120: +5 % (16384 rows: -2 %, K = 1280: -3 ... -20 %) -- exactly those.
        if (GEGLU && MODE == 0 && g.ln_in == nullptr && g.st_out == nullptr && ksplit == 1 && cfg != 0 &&
            ((g.Cin == 320 && g.Nb >= 32768) || (g.Cin == 640 && g.Nb >= 4096 && g.Nb <= 8192)))
            cfg = 224212;
    }
    if (ksplit == 0) ksplit = 1;
    if (g.ln_in != nullptr) ksplit = 1;  // the LayerNorm correction lives in the GEMM's own epilogue
    bool gs_dropped = false;
    if (g.lno_y != nullptr) {
        // the LayerNorm leaves the epilogue only where the launch the library would pick ANYWAY is a 320-wide tile that IS the whole row
        const bool ok = MODE == 0 && !GEGLU && ksplit == 1 && (cfg == 254222 || cfg == 254122 || cfg == 254218) && g.Ma == 320 &&
                        g.ln_in == nullptr && g.st_out == nullptr && g.yt == nullptr && g.gs_out == nullptr && batch == 1 &&
                        (g.ldy % 8) == 0 && (g.ldres % 8) == 0 && (g.lno_ld % 8) == 0;
        if (ok) {
            g.ksplit = 1;
            g.part = nullptr;
            return ig_dispatch_lno(cfg, g, batch, stream);
        }
        // (a split-K launch could hand the LayerNorm to its tail kernel -- one wave per row summing the slabs -- and that form was built and
        //  measured: it costs the job +0.5 % where the whole-row epilogue alone gains 0.9 %, profiles/r05_ln_from_producer_ab.txt)
        g.lno_y = nullptr;
        gs_dropped = true;  // reported like a dropped statistics request: FZ_GEMM_NO_STATS, y complete
    }
    if (g.gs_out != nullptr) {
        // Statistics leave the epilogue only where the launch the library would pick ANYWAY is a 320-wide ring tile without split-K
        // (forcing such a tile onto a launch that wants another one costs more than the statistics kernel it saves): otherwise the
        // launch runs as usual and the caller is told to compute the statistics itself (FZ_GEMM_NO_STATS).
        const bool ok = (MODE == 0 || MODE == 2) && !GEGLU && ksplit == 1 && (cfg == 254222 || cfg == 254122) && g.ln_in == nullptr &&
                        g.st_out == nullptr && g.yt == nullptr && batch == 1;
        if (ok) {
            g.ksplit = 1;
            g.part = nullptr;
            return ig_dispatch_gs<MODE>(cfg, g, batch, stream);
        }
        g.gs_out = nullptr;
        gs_dropped = true;
    }
    if (g.yt != nullptr) {               // fz_gemm_qkvt: transposed tiles leave from the GEMM's own epilogue as well
        if (GEGLU || MODE != 0 || ksplit != 1 || g.ln_in != nullptr || g.st_out != nullptr) return FZ_ERR_UNSUPPORTED;
        g.ksplit = 1;
        g.part = nullptr;
        return ig_dispatch_vt(cfg, g, batch, stream);
    }
    bool stats_dropped = false;
    if (g.st_out != nullptr && ksplit > 1) {  // the split-K tail does not compute row statistics: tell the caller
        g.st_out = nullptr;
        stats_dropped = true;
    }
    g.ksplit = ksplit;
    g.part = nullptr;
    if (ksplit > 1) {
        if (GEGLU || workspace == nullptr || (g.Ma % 4) || g.Ma_store != g.Ma || (g.ldy % 4)) return FZ_ERR_UNSUPPORTED;
        if ((int64_t)ksplit * batch * g.Nb * g.Ma > workspace_floats) return FZ_ERR_BAD_ARG;
        g.part = workspace;
    }
    int rc;
    if constexpr (MODE == 0) {
        rc = (g.ln_in != nullptr || g.st_out != nullptr) ? ig_dispatch_cfg<0, GEGLU, true>(cfg, g, batch, stream)
                                                         : ig_dispatch_cfg<0, GEGLU>(cfg, g, batch, stream);
    } else {
        if (g.ln_in != nullptr || g.st_out != nullptr) return FZ_ERR_UNSUPPORTED;
        rc = ig_dispatch_cfg<MODE, GEGLU>(cfg, g, batch, stream);
    }
    if (rc != FZ_OK || ksplit == 1) return rc != FZ_OK ? rc : (gs_dropped ? FZ_GEMM_NO_STATS : FZ_OK);
    const int64_t rblocks = (g.Nb * batch + 3) / 4;   // one wave per row, 4 rows per workgroup
    dim3 grid((unsigned)(rblocks < 16384 ? rblocks : 16384), (unsigned)((g.Ma / 4 + 63) / 64)), block(256);
    FZ_LAUNCH(igemm_reduce_kernel, grid, block, 0, stream, g, batch);
    const int rc2 = fz_last_launch_status();
    return rc2 != FZ_OK ? rc2 : ((stats_dropped || gs_dropped) ? FZ_GEMM_NO_STATS : FZ_OK);
}

extern "C" int64_t fz_gemm_workspace_floats(int64_t rows, int out_features, int batch) {
    // enough for the largest split the library would choose on its own: 8 slabs, capped at 256 MB of scratch
    const int64_t out = rows * (int64_t)out_features * (batch > 0 ? batch : 1);
    const int64_t cap = 64ll << 20;
    return 8 * out < cap ? 8 * out : (2 * out < cap ? cap : 2 * out);
}

static int gemm_impl(const FzGemmDesc* d, const FzGemmLn* ln, const void* x, const void* w, const void* bias, const void* res,
                     const void* res2, void* y, void* workspace, void* stream);

extern "C" int fz_gemm(const FzGemmDesc* d, const void* x, const void* w, const void* bias, const void* res, const void* res2,
                       void* y, void* workspace, void* stream) {
    return gemm_impl(d, nullptr, x, w, bias, res, res2, y, workspace, stream);
}

extern "C" int fz_gemm_ln(const FzGemmDesc* d, const FzGemmLn* ln, const void* x, const void* w, const void* bias, const void* res,
                          const void* res2, void* y, void* workspace, void* stream) {
    if (!ln) return FZ_ERR_BAD_ARG;
    return gemm_impl(d, ln, x, w, bias, res, res2, y, workspace, stream);
}

static int gemm_impl(const FzGemmDesc* d, const FzGemmLn* ln, const void* x, const void* w, const void* bias, const void* res,
                     const void* res2, void* y, void* workspace, void* stream) {
    if (!d || !x || !w || !y || d->rows <= 0 || d->in_features <= 0 || d->out_features <= 0) return FZ_ERR_BAD_ARG;
    const int batch = d->batch > 0 ? d->batch : 1;
    IgArgs g = {};
    g.taps = 1;
    g.fpb = 1;
    g.Cin = d->in_features;
    g.temb = nullptr;
    g.temb_group = 1;
    g.res = (const half_t*)res;
    g.res2 = (const half_t*)res2;
    g.ldres = d->ldres ? d->ldres : d->ldy;
    g.res_bs = d->res_batch_stride;
    g.y = (half_t*)y;
    g.ldy = d->ldy;
    g.y_bs = d->y_batch_stride;
    if (d->ldx < d->in_features || d->ldw < d->in_features || (d->ldx % 8) || (d->ldw % 8)) return FZ_ERR_BAD_ARG;
    const bool geglu = d->epilogue == FZ_GEMM_GEGLU;
    if (d->epilogue != FZ_GEMM_PLAIN && !geglu) return FZ_ERR_BAD_ARG;
    if (!d->transpose_out) {  // y[row][out]:  A = W (out features), B = x rows
        g.a = (const half_t*)w;
        g.lda = d->ldw;
        g.a_bs = d->w_batch_stride;  // 0: shared weights
        g.Ma = d->out_features;
        g.Ma_store = d->out_features;
        g.b = (const half_t*)x;
        g.ldb = d->ldx;
        g.b_bs = d->x_batch_stride;
        g.Nb = d->rows;
        g.bias = (const half_t*)bias;
        const int outw = geglu ? d->out_features / 2 : d->out_features;
        if (d->ldy < outw) return FZ_ERR_BAD_ARG;
        if (ln != nullptr) {
            if (ln->stats_in != nullptr) {  // x holds the raw LayerNorm input, w holds gamma * W
                if (!ln->c1 || !ln->c0 || d->in_features % 64 || d->out_features % 64 || d->x_batch_stride % d->ldx) return FZ_ERR_UNSUPPORTED;
                g.ln_in = ln->stats_in;
                g.ln_c1 = ln->c1;
                g.ln_c0 = ln->c0;
                g.ln_eps = ln->eps;
                g.ln_blocks = d->in_features / 64;
                g.bias = nullptr;  // folded into c0
            }
            if (ln->stats_out != nullptr) {
                if (geglu || d->out_features % 64 || (d->ldy % 8) || (d->y_batch_stride % 8) || (g.ldres % 8) || (g.res_bs % 8))
                    return FZ_ERR_UNSUPPORTED;
                g.st_out = ln->stats_out;
            }
        }
        if (geglu) {
            if (d->out_features % 64 || res || res2) return FZ_ERR_UNSUPPORTED;
            return ig_run<0, true>(g, batch, d->tile_cfg, 1, nullptr, 0, stream);
        }
        return ig_run<0, false>(g, batch, d->tile_cfg, d->split_k, (float*)workspace, d->workspace_floats, stream);
    }
    // y[b][out][row] (V^T): A = x rows of the batch element (row index contiguous in the output), B = W rows
    if (geglu || bias || res || res2 || ln) return FZ_ERR_UNSUPPORTED;
    if (d->rows >= (1ll << 31)) return FZ_ERR_UNSUPPORTED;
    g.a = (const half_t*)x;
    g.lda = d->ldx;
    g.a_bs = d->x_batch_stride;
    g.Ma = (int)d->rows;
    g.Ma_store = d->rows_store > d->rows ? (int)d->rows_store : (int)d->rows;
    g.b = (const half_t*)w;
    g.ldb = d->ldw;
    g.b_bs = 0;
    g.Nb = d->out_features;
    g.bias = nullptr;
    if (d->ldy < g.Ma_store) return FZ_ERR_BAD_ARG;
    return ig_run<0, false>(g, batch, d->tile_cfg, 1, nullptr, 0, stream);
}

extern "C" int fz_gemm_qkvt(const FzGemmDesc* d, const void* x, const void* w, void* y, void* yt, int split_col, int64_t rows_per_frame,
                            int64_t yt_frame_stride, int64_t ldyt, void* stream) {
    if (!d || !x || !w || !y || !yt || d->rows <= 0 || d->in_features <= 0 || d->out_features <= 0) return FZ_ERR_BAD_ARG;
    if (d->epilogue != FZ_GEMM_PLAIN || d->transpose_out || (d->batch > 1) || d->w_batch_stride) return FZ_ERR_UNSUPPORTED;
    if (split_col <= 0 || split_col >= d->out_features || split_col % 64 || rows_per_frame <= 0 || rows_per_frame % 8 ||
        d->rows % rows_per_frame || ldyt < rows_per_frame || (ldyt % 8) || (yt_frame_stride % 8) || d->ldy < split_col || (d->ldy % 8))
        return FZ_ERR_BAD_ARG;
    if (d->ldx < d->in_features || d->ldw < d->in_features || (d->ldx % 8) || (d->ldw % 8) || d->rows >= (1ll << 31)) return FZ_ERR_BAD_ARG;
    IgArgs g = {};
    g.taps = 1;
    g.fpb = 1;
    g.Cin = d->in_features;
    g.temb_group = 1;
    g.a = (const half_t*)w;
    g.lda = d->ldw;
    g.Ma = g.Ma_store = d->out_features;
    g.b = (const half_t*)x;
    g.ldb = d->ldx;
    g.Nb = d->rows;
    g.y = (half_t*)y;
    g.ldy = g.ldres = d->ldy;
    g.yt = (half_t*)yt;
    g.yt_bs = yt_frame_stride;
    g.ldyt = ldyt;
    g.vt_split = split_col;
    g.vt_rows = (int)rows_per_frame;
    if (d->tile_cfg != 0) {  // a pinned tile must not straddle the k | v boundary either (ig_choose checks this for its own picks only)
        int ba = 0;
        switch (d->tile_cfg) {
            case 254222: case 254122: ba = 320; break;
            case 224223: case 222222: ba = 128; break;
            case 212222: ba = 64; break;
            default: return FZ_ERR_BAD_ARG;
        }
        if (split_col % ba) return FZ_ERR_BAD_ARG;
    }
    return ig_run<0, false>(g, 1, d->tile_cfg, 1, nullptr, 0, stream);
}

static int conv_common(IgArgs& g, const void* x, const void* wt, const void* bias, const void* temb, int64_t temb_stride,
                       const void* res, const void* res2, void* y, int cin, int cout) {
    g.a = (const half_t*)wt;
    g.lda = (int64_t)g.taps * cin;
    g.Ma = g.Ma_store = cout;
    g.b = (const half_t*)x;
    g.ldb = cin;
    g.Cin = cin;
    g.bias = (const half_t*)bias;
    g.temb = (const half_t*)temb;
    g.temb_stride = temb_stride ? temb_stride : cout;
    g.temb_group = (int64_t)g.fpb * g.Ho * g.Wo;
    g.res = (const half_t*)res;
    g.res2 = (const half_t*)res2;
    g.y = (half_t*)y;
    g.ldy = g.ldres = cout;
    g.Nb = (int64_t)g.N * g.Ho * g.Wo;
    return g.Nb < (1ll << 31) ? FZ_OK : FZ_ERR_UNSUPPORTED;
}

// Temporal k=3 convolution with fewer than 8 channels on a side: conv_out's rank-2 LoRA pair (4 -> 2 -> 4, lora.py:26-28 caps the
// rank at min(in, out) / 2) and the 4 -> 4 Conv1d of conv_out in configs without a `lora` key (resnet.py:42-55).  A few FLOPs per
// byte: one thread per (frame, token), weights in LDS, plain VALU.
FZ_KERNEL void __launch_bounds__(256) temporal_conv3_small_kernel(IgArgs g) {
    FZ_SHARED float wl[3 * 8 * 8];
    const int cin = g.Cin, cout = g.Ma;
    for (int id = threadIdx.x; id < cout * 3 * cin; id += 256) wl[id] = (float)g.a[id];  // [cout][3][cin]
    __syncthreads();
    const int64_t tokens = g.Wo;
    const int64_t total = (int64_t)g.N * tokens;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int n = (int)(id / tokens);
        const int f = n % g.fpb;
        float xin[3][8];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int fs = f + t - 1;
            const bool inb = fs >= 0 && fs < g.fpb;
            const half_t* xs = g.b + (id + (int64_t)(inb ? t - 1 : 0) * tokens) * cin;
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) xin[t][ci] = (inb && ci < cin) ? (float)xs[ci < cin ? ci : 0] : 0.0f;
        }
        half_t* yo = g.y + id * cout;
        for (int co = 0; co < cout; ++co) {
            float acc = 0.0f;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int ci = 0; ci < 8; ++ci)
                    if (ci < cin) acc += xin[t][ci] * wl[(co * 3 + t) * cin + ci];
            if (g.temb != nullptr) acc += (float)g.temb[(id / g.temb_group) * g.temb_stride + co];
            if (g.res != nullptr) acc += (float)g.res[id * cout + co];
            if (g.res2 != nullptr) acc += (float)g.res2[id * cout + co];
            yo[co] = (half_t)acc;
        }
    }
}

// GroupNorm-statistics request of fz_gemm_gn / fz_temporal_conv3_gn -> IgArgs (false: the shape cannot carry it; run without)
static bool ig_gs_setup(IgArgs& g, float* partial, int groups, int64_t rows_per_frame) {
    if (partial == nullptr || groups <= 0 || groups > 64 || rows_per_frame <= 0) return false;
    if (g.Ma % groups || g.Ma % 320 || g.Nb % rows_per_frame || rows_per_frame % 128 || rows_per_frame >= (1ll << 31)) return false;
    const int cpg = g.Ma / groups;
    if ((cpg != 10 && cpg != 20) ||  // the instantiated group widths (40 -- 1280 channels -- spilled, and its launches split K anyway)
        (g.ldy % 8) || (g.ldres % 8) || (g.temb != nullptr && (g.temb_stride % 8))) return false;
    g.gs_out = partial;
    g.gs_cpg = cpg;
    g.gs_groups = groups;
    g.gs_rpf = (int)rows_per_frame;
    g.gs_chunks = (int)(rows_per_frame / 128);
    return true;
}

extern "C" int fz_gn_epilogue_chunks(int64_t rows_per_frame) { return rows_per_frame % 128 == 0 ? (int)(rows_per_frame / 128) : 0; }

extern "C" int fz_gemm_gn(const FzGemmDesc* d, const void* x, const void* w, const void* bias, const void* res, const void* res2, void* y,
                          float* gn_partial, int gn_groups, int64_t rows_per_frame, void* stream) {
    if (!d || !x || !w || !y || !gn_partial || d->rows <= 0 || d->in_features <= 0 || d->out_features <= 0) return FZ_ERR_BAD_ARG;
    if (d->epilogue != FZ_GEMM_PLAIN || d->transpose_out || d->batch > 1 || d->w_batch_stride) return FZ_ERR_UNSUPPORTED;
    if (d->ldx < d->in_features || d->ldw < d->in_features || (d->ldx % 8) || (d->ldw % 8) || d->ldy < d->out_features) return FZ_ERR_BAD_ARG;
    IgArgs g = {};
    g.taps = 1;
    g.fpb = 1;
    g.Cin = d->in_features;
    g.temb_group = 1;
    g.a = (const half_t*)w;
    g.lda = d->ldw;
    g.Ma = g.Ma_store = d->out_features;
    g.b = (const half_t*)x;
    g.ldb = d->ldx;
    g.Nb = d->rows;
    g.bias = (const half_t*)bias;
    g.res = (const half_t*)res;
    g.res2 = (const half_t*)res2;
    g.y = (half_t*)y;
    g.ldy = d->ldy;
    g.ldres = d->ldres ? d->ldres : d->ldy;
    const bool want = ig_gs_setup(g, gn_partial, gn_groups, rows_per_frame);
    const int rc = ig_run<0, false>(g, 1, d->tile_cfg, 1, nullptr, 0, stream);
    return rc != FZ_OK ? rc : (want ? FZ_OK : FZ_GEMM_NO_STATS);
}

extern "C" int fz_gemm_lnout(const FzGemmDesc* d, const void* x, const void* w, const void* bias, const void* res, const void* res2, void* y,
                             const void* gamma, const void* beta, float eps, void* y_ln, int64_t ld_ln, void* workspace, void* stream) {
    if (!d || !x || !w || !y || !gamma || !beta || !y_ln || d->rows <= 0 || d->in_features <= 0 || d->out_features <= 0) return FZ_ERR_BAD_ARG;
    if (d->epilogue != FZ_GEMM_PLAIN || d->transpose_out || d->batch > 1 || d->w_batch_stride) return FZ_ERR_UNSUPPORTED;
    if (d->ldx < d->in_features || d->ldw < d->in_features || (d->ldx % 8) || (d->ldw % 8) || d->ldy < d->out_features || ld_ln < d->out_features)
        return FZ_ERR_BAD_ARG;
    IgArgs g = {};
    g.taps = 1;
    g.fpb = 1;
    g.Cin = d->in_features;
    g.temb_group = 1;
    g.a = (const half_t*)w;
    g.lda = d->ldw;
    g.Ma = g.Ma_store = d->out_features;
    g.b = (const half_t*)x;
    g.ldb = d->ldx;
    g.Nb = d->rows;
    g.bias = (const half_t*)bias;
    g.res = (const half_t*)res;
    g.res2 = (const half_t*)res2;
    g.y = (half_t*)y;
    g.ldy = d->ldy;
    g.ldres = d->ldres ? d->ldres : d->ldy;
    g.lno_y = (half_t*)y_ln;
    g.lno_gamma = (const half_t*)gamma;
    g.lno_beta = (const half_t*)beta;
    g.lno_ld = ld_ln;
    g.lno_eps = eps;
    return ig_run<0, false>(g, 1, d->tile_cfg, d->split_k, (float*)workspace, d->workspace_floats, stream);
}

static int temporal_conv3_impl(const void* x, const void* wt, const void* res, const void* res2, const void* temb, int64_t temb_stride,
                               void* y, int n, int tokens, int cin, int cout, int clip_len, void* workspace, int64_t workspace_floats,
                               float* gn_partial, int gn_groups, void* stream);

extern "C" int fz_temporal_conv3_gn(const void* x, const void* wt, const void* res, const void* res2, const void* temb,
                                    int64_t temb_stride, void* y, int n, int tokens, int cin, int cout, int clip_len, void* workspace,
                                    int64_t workspace_floats, float* gn_partial, int gn_groups, void* stream) {
    if (!gn_partial || gn_groups <= 0) return FZ_ERR_BAD_ARG;
    return temporal_conv3_impl(x, wt, res, res2, temb, temb_stride, y, n, tokens, cin, cout, clip_len, workspace, workspace_floats,
                               gn_partial, gn_groups, stream);
}

extern "C" int fz_temporal_conv3(const void* x, const void* wt, const void* res, const void* res2, const void* temb,
                                int64_t temb_stride, void* y, int n, int tokens, int cin, int cout, int clip_len,
                                void* workspace, int64_t workspace_floats, void* stream) {
    return temporal_conv3_impl(x, wt, res, res2, temb, temb_stride, y, n, tokens, cin, cout, clip_len, workspace, workspace_floats, nullptr,
                               0, stream);
}

static int temporal_conv3_impl(const void* x, const void* wt, const void* res, const void* res2, const void* temb, int64_t temb_stride,
                               void* y, int n, int tokens, int cin, int cout, int clip_len, void* workspace, int64_t workspace_floats,
                               float* gn_partial, int gn_groups, void* stream) {
    if (!x || !wt || !y || n <= 0 || tokens <= 0 || clip_len <= 0 || n % clip_len) return FZ_ERR_BAD_ARG;
    IgArgs g = {};
    g.taps = 3;
    g.N = n; g.Hi = 1; g.Wi = tokens; g.Ho = 1; g.Wo = tokens; g.stride = 1; g.upsample = 0; g.fpb = clip_len;
    if (conv_common(g, x, wt, nullptr, temb, temb_stride, res, res2, y, cin, cout) != FZ_OK) return FZ_ERR_UNSUPPORTED;
    if (cin % 8 || cout < 8) {  // conv_out's 4 / 2-channel temporal convolutions: direct VALU kernel
        if (cin > 8 || cout > 8 || cin <= 0 || cout <= 0) return FZ_ERR_UNSUPPORTED;
        const int64_t total = (int64_t)n * tokens;
        dim3 grid((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), block(256);
        FZ_LAUNCH(temporal_conv3_small_kernel, grid, block, 0, stream, g);
        const int rc = fz_last_launch_status();
        return rc != FZ_OK ? rc : (gn_partial ? FZ_GEMM_NO_STATS : FZ_OK);
    }
    const bool want = gn_partial != nullptr && ig_gs_setup(g, gn_partial, gn_groups, tokens);
    const int rc = ig_run<2, false>(g, 1, 0, 0, (float*)workspace, workspace_floats, stream);
    return rc != FZ_OK ? rc : ((gn_partial != nullptr && !want) ? FZ_GEMM_NO_STATS : FZ_OK);
}

// csrc/conv_halo.hip: the stride-1 convolution with the pixel rows + halo resident in LDS across the nine taps (tile id FZ_TILE_CONV_HALO)
int fz_conv_halo_ok(int n, int h, int w, int cin, int cout, int64_t temb_stride);
int fz_conv_halo_launch(const void* x, const void* wt, const void* bias, const void* temb, int64_t temb_stride, int64_t temb_group, const void* res,
                        void* y, int n, int h, int w, int cin, int cout, float* part, int ksplit, void* stream);
int fz_conv_halo64_ok(int n, int h, int w, int cin, int cout, int64_t temb_stride);
int fz_conv_halo64_launch(const void* x, const void* wt, const void* bias, const void* temb, int64_t temb_stride, int64_t temb_group, const void* res,
                          void* y, int n, int h, int w, int cin, int cout, float* part, int ksplit, void* stream);
// the halo kernel, whole (ksplit <= 1) or as ksplit K slices into fp32 slabs + the split-K tail kernel (bias / temb / residual applied there)
static int conv_halo_run(IgArgs& g, const void* x, const void* wt, const void* bias, const void* temb, const void* res, void* y, int n, int hi, int wi,
                         int cin, int cout, int ksplit, float* workspace, int64_t workspace_floats, void* stream, bool narrow = false) {
    auto launch = narrow ? fz_conv_halo64_launch : fz_conv_halo_launch;
    if (ksplit <= 1)
        return launch(x, wt, bias, temb, g.temb_stride, g.temb_group, res, y, n, hi, wi, cin, cout, nullptr, 1, stream);
    if (workspace == nullptr || (g.Ma % 4) || (g.ldy % 4) || ksplit > cin / 64) return FZ_ERR_UNSUPPORTED;
    if ((int64_t)ksplit * g.Nb * g.Ma > workspace_floats) return FZ_ERR_BAD_ARG;
    const int rc = launch(x, wt, nullptr, nullptr, g.temb_stride, g.temb_group, nullptr, y, n, hi, wi, cin, cout, workspace, ksplit, stream);
    if (rc != FZ_OK) return rc;
    g.ksplit = ksplit;
    g.part = workspace;
    const int64_t rblocks = (g.Nb + 3) / 4;
    dim3 grid((unsigned)(rblocks < 16384 ? rblocks : 16384), (unsigned)((g.Ma / 4 + 63) / 64)), block(256);
    FZ_LAUNCH(igemm_reduce_kernel, grid, block, 0, stream, g, 1);
    return fz_last_launch_status();
}
#define FZ_TILE_CONV_HALO 154299
#define FZ_TILE_CONV_HALO64 154264   /* the same kernel with 64 output channels per workgroup (conv_halo_kernel<9, 2>) */

extern "C" int fz_conv3x3(const void* x, const void* wt, const void* bias, const void* temb, int64_t temb_stride, const void* res,
                          void* y, int n, int hi, int wi, int cin, int cout, int stride, int upsample, int frames_per_batch,
                          void* workspace, int64_t workspace_floats, int tile_cfg, int split_k, void* stream) {
    if (!x || !wt || !y || n <= 0 || hi <= 0 || wi <= 0) return FZ_ERR_BAD_ARG;
    if ((stride != 1 && stride != 2) || (upsample && stride != 1)) return FZ_ERR_UNSUPPORTED;
    IgArgs g = {};
    g.taps = 9;
    g.N = n; g.Hi = hi; g.Wi = wi; g.stride = stride; g.upsample = upsample;
    g.fpb = frames_per_batch > 0 ? frames_per_batch : 1;
    const int hu = upsample ? 2 * hi : hi, wu = upsample ? 2 * wi : wi;
    g.Ho = (hu + 2 - 3) / stride + 1;
    g.Wo = (wu + 2 - 3) / stride + 1;
    if (conv_common(g, x, wt, bias, temb, temb_stride, res, nullptr, y, cin, cout) != FZ_OK) return FZ_ERR_UNSUPPORTED;
    if (tile_cfg == FZ_TILE_CONV_HALO64) {   // the narrow form of the halo kernel (64 output channels per workgroup), pinned
        if (stride != 1 || upsample || !fz_conv_halo64_ok(n, hi, wi, cin, cout, g.temb_stride)) return FZ_ERR_UNSUPPORTED;
        return conv_halo_run(g, x, wt, bias, temb, res, y, n, hi, wi, cin, cout, split_k, (float*)workspace, workspace_floats, stream, true);
    }
    if (tile_cfg == FZ_TILE_CONV_HALO) {
        if (stride != 1 || upsample || !fz_conv_halo_ok(n, hi, wi, cin, cout, g.temb_stride)) return FZ_ERR_UNSUPPORTED;
        return conv_halo_run(g, x, wt, bias, temb, res, y, n, hi, wi, cin, cout, split_k, (float*)workspace, workspace_floats, stream);
    }
    // The library's own choice takes the halo form where its 160 x 256 tiles fill the chip one to four times (160-1 024 workgroups: the 8- and 16-frame
    // launches of the 64^2 level, the 16-frame ones of 32^2): +26 % / +21 ... +28 % at one wave of tiles, +1 ... +6 % at two on MI355X
    // (profiles/r06_conv_halo_ab.txt) -- and, in K slices with the split-K tail kernel, where they do not: ~128 tiles (8 frames x 32^2, 16 x 16^2)
    // in two slices, ~64 (8 frames x 16^2) in four: -8 ... -15 % of the launch against the split-K implicit GEMM (profiles/r06_halo_split_ab.txt);
    // a slice wants >= 2.5 chunks of 64 channels (Cin = 320 stays whole).
    if (tile_cfg == 0 && split_k <= 1 && stride == 1 && !upsample && fz_conv_halo_ok(n, hi, wi, cin, cout, g.temb_stride)) {
        const int64_t tiles = (int64_t)(cout / 160) * (g.Nb / 256);
        const int nchunk = cin / 64;
        int ks = tiles >= 160 ? 1 : (tiles >= 96 ? 2 : (tiles >= 48 ? 4 : (tiles >= 24 && nchunk >= 20 ? 8 : 0)));   // (8: 16 frames x 8^2: -6 ... -9 %)
        if (ks == 4 && nchunk < 10) ks = 2;
        if (ks == 2 && nchunk < 6) ks = tiles >= 96 ? 1 : 0;
        if (ks > 1 && (workspace == nullptr || (int64_t)ks * g.Nb * g.Ma > workspace_floats || (g.Ma % 4) || (g.ldy % 4))) ks = tiles >= 96 ? 1 : 0;
        bool take = ks >= 1 && tiles <= 1024;   // (768 tiles, 24 frames x 64^2: -17 ... -22 % of the launch; 1 024, 32 frames: -5 ... -6 %; profiles/r06_halo_split_ab_large.txt)
#ifdef FZ_IGEMM_TRIALS
        if (fz_igemm_trial_no_halo) take = false;
        if (fz_igemm_trial_no_halo_split && (ks > 1 || tiles < 200 || tiles > 512)) take = false;   // (= the rule of the first halo commit)
#endif
        if (take) return conv_halo_run(g, x, wt, bias, temb, res, y, n, hi, wi, cin, cout, ks, (float*)workspace, workspace_floats, stream);
        // fewer than 24 wide tiles (8 frames x 8^2: 16): the NARROW form -- 64 output channels per workgroup, five times the workgroups at the same
        // slab traffic -- in 5 (Cin 1 280) / 6 (2 560) K slices: -16 / -12 % against the split-K implicit GEMM (profiles/r06_halo_narrow_ab.txt)
        if (ks == 0 && tiles >= 8 && tiles < 24 && nchunk >= 20 && fz_conv_halo64_ok(n, hi, wi, cin, cout, g.temb_stride)) {
            const int ks64 = nchunk >= 40 ? 6 : 5;
            bool take64 = workspace != nullptr && (int64_t)ks64 * g.Nb * g.Ma <= workspace_floats && (g.Ma % 4) == 0 && (g.ldy % 4) == 0;
#ifdef FZ_IGEMM_TRIALS
            if (fz_igemm_trial_no_halo || fz_igemm_trial_no_halo_split) take64 = false;
#endif
            if (take64) return conv_halo_run(g, x, wt, bias, temb, res, y, n, hi, wi, cin, cout, ks64, (float*)workspace, workspace_floats, stream, true);
        }
    }
    if (cin % 8) {  // conv_in (4 input channels): direct VALU convolution
        if (cout % 8 || upsample || cin != 4 || (int64_t)9 * cin * cout * 2 > 64 * 1024 || (temb && g.temb_stride % 8))
            return FZ_ERR_UNSUPPORTED;
        const int64_t total = g.Nb * (cout / 8);
        dim3 grid((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), block(256);
        FZ_LAUNCH(conv3x3_small_cin_kernel<4>, grid, block, (size_t)9 * cin * cout * 2, stream, g);
        return fz_last_launch_status();
    }
    // K order.  With taps outermost the ~32 tiles running on one XCD read, per tap, a window of 32 x (pixels per tile) x Cin
    // input values; when that exceeds the XCD's 4 MB L2 (16-frame launches at 64^2, Cin >= 960 at 8 frames) every tap re-reads it
    // from MALL, and the Cin-chunk-outer order wins (+3 % at 64^2 x 16 f: 901 TF/s); below that, tap-outer is 3 % faster.  The
    // nearest-2x upsampling conv always runs tap-outer (its source pixel is not linear in the tap).
    const int64_t rows_per_tile = g.Nb >= 65536 ? 256 : 128;  // what ig_choose picks for launches that fill the chip
    const bool chunk_outer = !upsample && 32 * rows_per_tile * (int64_t)cin * 2 > (3ll << 20);
    if (!chunk_outer) return ig_run<1, false>(g, 1, tile_cfg, split_k, (float*)workspace, workspace_floats, stream);
    return ig_run<3, false>(g, 1, tile_cfg, split_k, (float*)workspace, workspace_floats, stream);
}
