// xattn_chain.hip -- the CROSS-ATTENTION CHAIN of the 64x64-level transformer block in ONE launch:
//     hidden_states = attn2(norm2(hidden_states), encoder_hidden_states) + hidden_states ;  norm3(hidden_states)   (attention.py:303-311)
//   and, in the FRONT form, the step in front of it as well:
//     hidden_states = attn1.to_out(attention) + hidden_states ;  norm2(hidden_states)                               (attention.py:295-301)
//   attn2 = diffusers CrossAttention [3P] through the reference's patched forward (attention_register.py:71-128):
//     q = xn Wq^T ;  S = q_h K_h^T * scale ;  P = softmax(S) over the 77 text keys ;  o_h = P V_h ;  y = o Wo^T + bo + res.
//   Per op this was fz_gemm (to_q) + fz_attn_cross + fz_gemm_lnout (to_out + residual + norm3) -- and fz_gemm_lnout (attn1.to_out + residual +
//   norm2) in front: three / four short-K launches of 20-55 us that pay a launch floor and their own weight ingest each, with q, o and the
//   LayerNorm output making a round trip through HBM between them (profiles/r06_job_breakdown_by_op_and_shape.txt).  No controller touches
//   maps of more than 32^2 queries (attention_store.py:83, attention_util.py:104), so at this level the cross attention is plain.
//
// One workgroup = 128 rows of ONE frame = four PAIRS of waves, each pair one 32-row strip; wave w (QA) and wave w + 4 (O) share a SIMD:
//   QA wave  holds the strip's LayerNorm'ed rows as the 20 B fragments of the q projection (80 VGPRs, lane = row + 32 * k half) and walks the
//            heads in PAIRS: q of two heads (3 MFMA tiles: every head padded to 48 units so that a head's q IS three B fragments of its
//            QK^T -- the accumulator layout of the MFMA is the B-fragment layout when the weight rows are packed in the order the registers
//            want, csrc/ff_chain.hip), then per head S^T = K_h q_h^T (9 MFMAs, all 96 key slots in registers), the exact softmax of
//            csrc/attn_cross.hip, O^T = V_h^T P^T (12 MFMAs).  V^T's rows are packed so that 8 consecutive channels of o land in one lane
//            half: the head pair's 80 output channels leave as the five B fragments of the output projection (2.5 k steps per head: the
//            middle fragment takes its lower lane half from the even head and its upper one from the odd head);
//   O wave   accumulates y^T = Wo o^T for the strip (10 tiles, 160 registers) one head pair behind: its 50 MFMAs per pair run beside the QA
//            wave's softmax (VALU) and fragment traffic;
//   weights  Wq, Wo (fz_xattn_chain_pack, once per weight set) and the text context's K / V^T (fz_xattn_chain_kv_pack, once per context)
//            are pre-packed into the exact 1 KB MFMA A fragments the loops consume, in consumption order; LDS-DMA copies them lane-linear
//            into a three-slot ring of 41 KB sub-steps (fetched two sub-steps ahead), one barrier per sub-step, a fragment read is one conflict-free ds_read_b128 at lane *
//            16.  The stream (~610 KB, + 200 KB with FRONT) is the same for every workgroup of a frame: it lives in L2;
//   FRONT    the QA wave first runs attn1's output projection on the strip of attention outputs (passes of 3 + 3 + 2 + 2 tiles), adds bias and
//            residual, stores hidden_states, and normalises the rows IN ITS REGISTERS with the summation order of fz_gemm_lnout's epilogue
//            (the eight partial sums of a row live four per lane half; the DPP tree of fz_sum8 becomes four cross-half exchanges);
//   epilogue y + bo -> fp16 -> LDS tile -> (+ res) -> full-row 16-byte stores of y AND of LayerNorm(y), all eight waves (csrc/ff_chain.hip).
// Arithmetic = the separate launches': fp32 accumulation over k ascending with the same MFMA and the same 16 k values per instruction, the
// scores scaled in fp32, max / sum / exp2 / reciprocal in attn_cross.hip's order, every intermediate rounded to fp16 where the launches
// store it -- the results are BIT-IDENTICAL to fz_gemm + fz_attn_cross + fz_gemm_lnout (tests/kernel_cases.py case_xattn_chain).
#include "fz_rt.h"
#include <atomic>
#include <type_traits>
#include "../../include/fatezero_hip.h"

namespace {
constexpr int XC_C = 320;                      // channels (the 64x64 level of SD-1.x)
constexpr int XC_H = 8, XC_D = 40;             // heads x head dim
constexpr int XC_KS = XC_C / 16;               // 20 k steps of a 320-wide projection
constexpr int XC_CT = XC_C / 32;               // 10 output tiles of a 320-wide projection
constexpr int XC_ROWS = 128;                   // rows per workgroup
constexpr int XC_FRAG = 1024;                  // bytes of one A fragment (64 lanes x 16 B)
constexpr int XC_SLOTF = 41;                   // fragments per ring slot (the largest sub-step)
constexpr int XC_SLOT = XC_SLOTF * XC_FRAG;
constexpr int XC_NSLOT = 3;                    // ring depth: a sub-step is fetched two sub-steps ahead
constexpr int XC_HOFF = XC_NSLOT * XC_SLOT;    // hand-over area: [pair 4][5 fragments] = the head pair's o as B fragments
constexpr int XC_HPAIR = 8;                    // fragments per pair: 5 (o of a head pair); FRONT: up to 8 residual fragments of a pass
constexpr int XC_HBYTES = 4 * XC_HPAIR * XC_FRAG;
constexpr int XC_COFF = XC_HOFF + XC_HBYTES;   // FRONT: bias of attn1.to_out | gamma | beta of norm2, 320 halves each (2 fragments of the stream)
constexpr int XC_CBYTES = 2 * XC_FRAG;
constexpr size_t XC_LDS_BYTES = XC_COFF + XC_CBYTES;
constexpr int XC_OSTR = XC_C + 8;              // staging row stride of the epilogue (halves)
constexpr int XC_BLK = 110;                    // weight-pack fragments per head pair: Wq 30 + 30 | Wo 10 + 20 + 20 (of the pair before)
constexpr int XC_NBLK = 5;                     // 4 head pairs + the output projection's tail
constexpr int XC_KVH = 21;                     // context-pack fragments per head: K 3 x 3, V^T 2 x 3 x 2
constexpr int XC_FRONTW = XC_KS * XC_CT;       // front projection: 20 k steps x 10 tiles (passes of 3, 3, 2, 2 tiles)
constexpr int XC_FRONTF = 2 + XC_FRONTW;       // + the constants block in front of them
constexpr int XC_FRONT_STEPS = 6;
constexpr int XC_MAIN_STEPS = 4 * XC_NBLK;
static_assert(4 * 32 * XC_OSTR * 2 <= XC_NSLOT * XC_SLOT, "epilogue staging fits the ring");
static_assert(XC_LDS_BYTES <= 160 * 1024, "LDS");
static_assert(XC_H * XC_D == XC_C, "heads");

// unit (within a tile of 32) that A row i carries so that the MFMA accumulator registers [8 b, 8 b + 8) of lane half hi hold the units
// 16 b + 8 hi + [0, 8) -- a B fragment of the NEXT contraction (csrc/ff_chain.hip fc_unit_of_arow, csrc/attn_cross.hip fz_pi_x)
FZ_HOST_DEVICE int xc_unit_of_arow(int i) {
    const int gq = i >> 3, hi = (i >> 2) & 1, e = i & 3;
    return 16 * (gq >> 1) + 8 * hi + 4 * (gq & 1) + e;
}
// head-local channel that A row i of V^T tile t carries (or -1: padding).  Group g = 2 t + b = accumulator registers [8 b, 8 b + 8) of tile
// t; lane half hi.  Even head: group g holds chunk 2 g + hi (chunks 0..4 real).  Odd head: groups 0, 1 hold chunk 2 g + 1 + hi, group 2 holds
// chunk 0 in its UPPER half only -- the head pair's 10 chunks then sit where the output projection's B fragments want them (see header)
FZ_HOST_DEVICE int xc_vchan_of_arow(int odd, int t, int i) {
    const int gq = i >> 3, hi = (i >> 2) & 1, e = i & 3;
    const int g = 2 * t + (gq >> 1);
    int chunk;
    if (!odd) {
        chunk = 2 * g + hi;
    } else if (g < 2) {
        chunk = 2 * g + 1 + hi;
    } else {
        chunk = (g == 2 && hi == 1) ? 0 : 5;
    }
    return chunk < 5 ? 8 * chunk + 4 * (gq & 1) + e : -1;
}
}  // namespace

// a pointer the program knows to be wave-uniform (the DMA's scalar base and LDS destination: functions of the wave id and the sub-step)
template <typename T>
FZ_DEVICE T* xc_uniform_ptr(T* p) {
#ifdef FZ_EMU
    return p;
#else
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T*>((uintptr_t)(((uint64_t)hi << 32) | lo));
#endif
}

#ifdef XC_TIMING  // trial build (scripts/xattn_chain_variants.sh): cycle totals of the QA and the O wave of pair 0, workgroup 0
__device__ long long xc_timing[2][8];
#define XC_TK(i) __builtin_amdgcn_sched_barrier(0); const long long tk##i = clock64(); __builtin_amdgcn_sched_barrier(0)
#define XC_TK_ADD(slot, a, b) tacc[slot] += (b) - (a)
#else
#define XC_TK(i) ((void)0)
#define XC_TK_ADD(slot, a, b) ((void)0)
#endif

struct XcArgs {
    const half_t* xn;       // [rows][320]  !FRONT: LayerNorm'ed input of attn2;  FRONT: attention output of attn1 (input of its to_out)
    const half_t* res;      // [rows][320]  !FRONT: residual of attn2 (hidden_states);  FRONT: residual of attn1
    const char* wpack;      // fz_xattn_chain_pack's stream: [FRONT: 2 + 200 fragments][5 x 110 fragments]
    const char* kvpack;     // fz_xattn_chain_kv_pack's stream: [batch][8 heads][21 fragments]
    const half_t* bo;       // [320] or null: bias of attn2.to_out
    half_t* y;              // [rows][320]
    half_t* yln;            // [rows][320] or null: LayerNorm(y; gamma, beta)
    const half_t* gamma;
    const half_t* beta;
    // FRONT only
    half_t* y1;             // [rows][320]: attn1's result (hidden_states in front of attn2)
    int64_t rows, rows_per_frame;
    int frames_per_batch, lk;
    float cs;               // softmax scale * log2(e)
    float eps, eps1;
};

template <bool FRONT>
FZ_KERNEL void __launch_bounds__(512, 2) xattn_chain_kernel(XcArgs g) {
    FZ_DYN_SMEM(raw);
    const int tid = threadIdx.x, wave = fz_uniform(tid >> 6), lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const bool is_qa = wave < 4;
    const int pair = wave & 3;
    // XCD-aware order (csrc/ff_chain.hip): each XCD takes a contiguous run of row blocks -- one frame's context pack per L2 where it can
    const int nt = gridDim.x, bid = blockIdx.x;
    const int q8 = nt >> 3, r8 = nt & 7, xcd = bid & 7;
    const int blk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int64_t row0 = (int64_t)blk * XC_ROWS + pair * 32;
    const int batch = (int)(((int64_t)blk * XC_ROWS / g.rows_per_frame) / g.frames_per_batch);
    const char* const kv = g.kvpack + (int64_t)batch * (XC_H * XC_KVH * XC_FRAG);
    const char* const wmain = g.wpack + (FRONT ? XC_FRONTF * XC_FRAG : 0);
    unsigned char* const hbase = raw + XC_HOFF + pair * XC_HPAIR * XC_FRAG;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    constexpr int NSTEP = (FRONT ? XC_FRONT_STEPS : 0) + XC_MAIN_STEPS;

    // ---- the stream: sub-step n -> ring slot n % 3, fetched TWO sub-steps ahead.  The QA waves issue the even sub-steps, the O waves the odd
    //      ones (wave w of a role copies fragments w, w + 4, ... of each range): a wave then only ever waits for ALL of its own LDS-DMA
    //      (vmcnt(0), no counted wait whose count would depend on the sub-step and the wave) while the other role's fetch stays in flight
    const int w4 = wave & 3;
    auto dma = [&](const char* src, int n, unsigned char* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const int j = w4 + 4 * i;
            if (j < n) fz_glds16_so(xc_uniform_ptr(src + j * XC_FRAG), lane_off, xc_uniform_ptr(dst + j * XC_FRAG));
        }
    };
    auto slot_of = [&](int n) __attribute__((always_inline)) -> unsigned char* { return raw + (n % XC_NSLOT) * XC_SLOT; };
    auto issue_step = [&](int n) __attribute__((always_inline)) {
        unsigned char* dst = slot_of(n);
        if (FRONT && n < XC_FRONT_STEPS) {   // passes of 3, 3, 2, 2 tiles: sub-steps of 30, 30 | 30, 30 | 40 | 40 fragments
            dma(g.wpack + (2 + (n < 4 ? 30 * n : 120 + 40 * (n - 4))) * XC_FRAG, n < 4 ? 30 : 40, dst);
            return;
        }
        const int m = n - (FRONT ? XC_FRONT_STEPS : 0), hp = m >> 2, k = m & 3;
        const char* b = wmain + hp * (XC_BLK * XC_FRAG);
        if (k == 0) {
            if (hp < 4) dma(b, 30, dst);
        } else if (k == 1) {
            if (hp < 4) dma(b + 30 * XC_FRAG, 30, dst);
            if (hp >= 1) dma(b + 60 * XC_FRAG, 10, dst + 30 * XC_FRAG);
        } else {
            if (hp < 4) dma(kv + (2 * hp + (k - 2)) * (XC_KVH * XC_FRAG), XC_KVH, dst);
            if (hp >= 1) dma(b + (k == 2 ? 70 : 90) * XC_FRAG, 20, dst + XC_KVH * XC_FRAG);
        }
    };
    const int my_parity = is_qa ? 0 : 1;
    int step = 0;
    // the skeleton every wave runs once per sub-step: the issuing role's part of sub-step `step` has landed, then everybody's; the slot of
    // sub-step step - 1 is free (its readers passed this barrier after their last MFMA took its operands) and takes sub-step step + 2
#ifdef XC_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long tk_start = clock64();
#endif
    auto sync_step = [&]() __attribute__((always_inline)) {
        XC_TK(0);
        if ((step & 1) == my_parity) fz_wait_vm0();
        XC_TK(1);
        fz_barrier_nodrain();
        XC_TK(2);
        if ((step & 1) == my_parity && step + 2 < NSTEP) issue_step(step + 2);
        XC_TK(3);
        XC_TK_ADD(0, tk0, tk1);   // wait for this role's LDS-DMA
        XC_TK_ADD(1, tk1, tk2);   // barrier
        XC_TK_ADD(2, tk2, tk3);   // DMA issue
        ++step;
    };
    auto slot_base = [&](int n) __attribute__((always_inline)) -> fz_lds_addr { return fz_lds_addr_of(slot_of(n)) + lane_off; };

    f32x16 yacc[XC_CT];  // (O waves)

    if (is_qa) {
        // ================================================ QA waves ==========================================================================
        half8_t xb[XC_KS];  // the strip as B fragments: lane (row l31, k half hi) holds x[row][16 s + 8 hi .. + 8)
        const int64_t row = row0 + l31;
        {
            const half_t* src = g.xn + row * XC_C + hi * 8;
#pragma unroll
            for (int s = 0; s < XC_KS; ++s) xb[s] = fz_ld_h8(src + s * 16);
        }
        // (retired HERE for the compiler's scoreboard: csrc/ff_chain.hip)
#pragma unroll
        for (int s = 0; s < XC_KS; ++s) asm volatile("" : "+v"(xb[s]));
        issue_step(0);

        if constexpr (FRONT) {
            // ---- attn1.to_out + bias + residual -> hidden_states (stored), norm2 -> xb -------------------------------------------------------
            // Wo1's rows are packed in register order: accumulator registers [8 b, 8 b + 8) of tile c are channels 32 c + 16 b + 8 hi + [0, 8)
            // = chunk 2 s + hi of fragment s = 2 c + b.
            half8_t hv[XC_KS];  // hidden_states of the strip, fp16, fragment layout
            // Four passes over the output tiles (3 + 3 + 2 + 2: the accumulators of a pass, the input strip and the finished part of hv share
            // 256 registers); a pass of 3 tiles is two sub-steps of 10 k steps, a pass of 2 tiles one sub-step of 20.  This wave issues NO
            // ordinary global load in here (one pending makes hipcc drain the LDS-DMA queue at its first use): the residual chunks come
            // through the hand-over area from the O wave of the pair, which has nothing else to do yet, a sub-step ahead (`rv`: read right
            // behind the barrier that publishes them); bias, gamma, beta sit in LDS (the constants block of the stream).
            const unsigned char* const cst = raw + XC_COFF + (hi * 8) * 2;
            half8_t rv[8];
            auto take_rv = [&](int nf) __attribute__((always_inline)) {
#pragma unroll
                for (int f = 0; f < 8; ++f)
                    if (f < nf) rv[f] = *reinterpret_cast<const half8_t*>(hbase + f * XC_FRAG + lane_off);
            };
            auto front_pass = [&](auto T0c, auto NTc, auto RV0c, auto TAKEc) __attribute__((always_inline)) {
                constexpr int T0 = decltype(T0c)::value, NT = decltype(NTc)::value, RV0 = decltype(RV0c)::value, TAKE = decltype(TAKEc)::value;
                constexpr int NSUB = NT == 3 ? 2 : 1, KPS = XC_KS / NSUB;
                f32x16 acc[NT];
#pragma unroll
                for (int c = 0; c < NT; ++c) acc[c] = fz_zero_f16v();
#pragma unroll
                for (int q = 0; q < NSUB; ++q) {
                    const int n = step;
                    sync_step();
                    // behind the first barrier of passes 0, 1, 2: the chunks the O wave left in front of the first barrier / during sub-step
                    // 1 / during sub-step 3 (passes 2 AND 3)
                    if (q == 0 && TAKE > 0) take_rv(TAKE);
                    const fz_lds_addr base = slot_base(n);
                    // batches of 2 k steps x NT tiles; the fragment reads of batch j + 1 are issued before the MFMAs of batch j
                    half8_t fa[2 * NT], fb[2 * NT];
                    auto ldf = [&](half8_t* f, int kk) __attribute__((always_inline)) {
#pragma unroll
                        for (int i = 0; i < 2 * NT; ++i) f[i] = fz_lds_ld_h8(base, (kk * NT + i) * XC_FRAG);
                    };
                    auto mmf = [&](const half8_t* f, int kk) __attribute__((always_inline)) {
#pragma unroll
                        for (int i = 0; i < 2 * NT; ++i) acc[i % NT] = fz_mfma_32x32x16_f16(f[i], xb[KPS * q + kk + i / NT], acc[i % NT]);
                    };
                    ldf(fa, 0);
                    FZ_SCHED_FENCE();
#pragma unroll
                    for (int kk = 0; kk < KPS; kk += 4) {
                        if (kk + 2 < KPS) ldf(fb, kk + 2);
                        mmf(fa, kk);
                        FZ_SCHED_FENCE();
                        if (kk + 2 < KPS) {
                            if (kk + 4 < KPS) ldf(fa, kk + 4);
                            mmf(fb, kk + 2);
                            FZ_SCHED_FENCE();
                        }
                    }
                }
                // + bias (fp32) -> fp16 -> + residual (fp32) -> fp16: fz_gemm_lnout's epilogue, in place on the accumulator layout
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int s = 2 * (T0 + c) + b;
                        const half8_t bv = *reinterpret_cast<const half8_t*>(cst + 16 * s * 2);
                        half8_t o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const half_t t = (half_t)(acc[c][8 * b + e] + (float)bv[e]);
                            o[e] = (half_t)((float)t + (float)rv[RV0 + 2 * c + b][e]);
                        }
                        hv[s] = o;
                    }
            };
            typedef std::integral_constant<int, 0> I0;
            front_pass(I0(), std::integral_constant<int, 3>(), I0(), std::integral_constant<int, 6>());
            front_pass(std::integral_constant<int, 3>(), std::integral_constant<int, 3>(), I0(), std::integral_constant<int, 6>());
            front_pass(std::integral_constant<int, 6>(), std::integral_constant<int, 2>(), I0(), std::integral_constant<int, 8>());
            front_pass(std::integral_constant<int, 8>(), std::integral_constant<int, 2>(), std::integral_constant<int, 4>(), I0());
            // hidden_states of the strip leave here, in one burst, under the LayerNorm arithmetic below
#pragma unroll
            for (int s = 0; s < XC_KS; ++s) fz_st_h8(g.y1 + row * XC_C + 16 * s + 8 * hi, hv[s]);
            // norm2 of the row in fz_gemm_lnout's summation order: its lane l8 (of 8 per row) sums chunks l8 + 8 i, i = 0..4 -- lane half hi
            // here holds those of l8 = 2 u + hi (fragments s = u + 4 i) -- even and odd elements apart, then fz_sum8 = xor 1 (the other lane
            // half), xor 2 (u ^ 1), mirror (3 - u, the other half): ((t0 + t1) + (t2 + t3)) with t_u = own_u + other_u
            {
#ifndef FZ_EMU
#pragma clang fp reassociate(off)  // (the written order IS fz_ln_row320's: the two must agree bit for bit)
#endif
            float pu[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        s0 += (float)hv[u + 4 * i][e];
                        s1 += (float)hv[u + 4 * i][e + 1];
                    }
                pu[u] = s0 + s1;
            }
            float tu[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) tu[u] = pu[u] + fz_shfl_xor(pu[u], 32);
            // (the pairing is pinned with opaque moves: hipcc's SLP vectoriser turned ((t0 + t1) + (t2 + t3)) into ((t0 + t2) + (t1 + t3)) in
            //  spite of the pragma -- one element in 21 million then rounded the other way, found on MI355X)
            auto pair_sum = [&](const float* t) __attribute__((always_inline)) -> float {
                float a = t[0] + t[1], b = t[2] + t[3];
                asm volatile("" : "+v"(a), "+v"(b));
                return a + b;
            };
            const float mean = pair_sum(tu) * (1.0f / 320.0f);
            // (the row stays 80 registers of packed halves between the sweeps: without the pins the compiler keeps all 160 fp32 conversions
            //  of the first sweep alive for the other two and spills)
#pragma unroll
            for (int s = 0; s < XC_KS; ++s) asm volatile("" : "+v"(hv[s]));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float q0 = 0.0f, q1 = 0.0f;
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const float d0 = (float)hv[u + 4 * i][e] - mean, d1 = (float)hv[u + 4 * i][e + 1] - mean;
                        q0 += d0 * d0;
                        q1 += d1 * d1;
                    }
                pu[u] = q0 + q1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) tu[u] = pu[u] + fz_shfl_xor(pu[u], 32);
            const float rstd = 1.0f / sqrtf(pair_sum(tu) * (1.0f / 320.0f) + g.eps1);
#pragma unroll
            for (int s = 0; s < XC_KS; ++s) asm volatile("" : "+v"(hv[s]));
#pragma unroll
            for (int s = 0; s < XC_KS; ++s) {
                const half8_t gm = *reinterpret_cast<const half8_t*>(cst + (XC_C + 16 * s) * 2);
                const half8_t bt = *reinterpret_cast<const half8_t*>(cst + (2 * XC_C + 16 * s) * 2);
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)hv[s][e] - mean) * rstd * (float)gm[e] + (float)bt[e]);
                xb[s] = o;
                if (s % 4 == 3) FZ_SCHED_FENCE();  // (keeps the 40 coefficient reads from being issued up front: 160 VGPRs)
            }
            }
#pragma unroll
            for (int s = 0; s < XC_KS; ++s) asm volatile("" : "+v"(xb[s]));
#ifdef XC_DEBUG_XN  // trial build: the LayerNorm output leaves through y_ln (the epilogue then skips its own)
#pragma unroll
            for (int s = 0; s < XC_KS; ++s) fz_st_h8(g.yln + row * XC_C + 16 * s + 8 * hi, xb[s]);
#endif
        }

        // ---- head pairs ------------------------------------------------------------------------------------------------------------------
        auto key_of = [&](int sub, int r) __attribute__((always_inline)) -> int { return 32 * sub + (r < 8 ? 8 * hi + r : 8 + 8 * hi + r); };
        // one head: S^T = K q^T over all 96 key slots, exact softmax (attn_cross.hip's order), O^T = V^T P^T; the head's o leaves as groups of
        // 8 consecutive channels per lane half: og[0], og[1] = tile 0, og[2] = registers [0, 8) of tile 1
        auto head = [&](const half8_t* qh, fz_lds_addr base, half8_t* og) __attribute__((always_inline)) {
            float s[48];
            half8_t kf[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) kf[i] = fz_lds_ld_h8(base, i * XC_FRAG);
            FZ_SCHED_FENCE();
            f32x16 sacc[3];
#pragma unroll
            for (int sub = 0; sub < 3; ++sub) {
                sacc[sub] = fz_zero_f16v();
#pragma unroll
                for (int c = 0; c < 3; ++c) sacc[sub] = fz_mfma_32x32x16_f16(kf[sub * 3 + c], qh[c], sacc[sub]);
            }
            // the V^T fragments of the first channel tile travel while the softmax runs
            half8_t vf[6], vg[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) vf[i] = fz_lds_ld_h8(base, (9 + i) * XC_FRAG);
            FZ_SCHED_FENCE();
#pragma unroll
            for (int sub = 0; sub < 3; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[16 * sub + r] = (key_of(sub, r) < g.lk) ? sacc[sub][r] * g.cs : -INFINITY;
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
            half8_t pf[3][2];
#pragma unroll
            for (int sub = 0; sub < 3; ++sub)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[sub][mm][e] = (half_t)s[16 * sub + 8 * mm + e];
            FZ_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < 6; ++i) vg[i] = fz_lds_ld_h8(base, (15 + i) * XC_FRAG);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x16 oacc = fz_zero_f16v();
#pragma unroll
                for (int sub = 0; sub < 3; ++sub)
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) oacc = fz_mfma_32x32x16_f16(t == 0 ? vf[sub * 2 + mm] : vg[sub * 2 + mm], pf[sub][mm], oacc);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (t == 1 && b == 1) continue;  // padding
#pragma unroll
                    for (int e = 0; e < 8; ++e) og[2 * t + b][e] = (half_t)oacc[8 * b + e];
                }
            }
        };
        for (int hp = 0; hp < XC_NBLK; ++hp) {
            const bool live = hp < 4;
            f32x16 qacc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) qacc[c] = fz_zero_f16v();
            // sub-steps 0, 1: q of the head pair: 3 tiles x 20 k steps, 10 k steps each
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int n = step;
                sync_step();
                if (!live) continue;
                const fz_lds_addr base = slot_base(n);
                // 5 batches of 2 k steps x 3 tiles; the fragment reads of batch j + 1 are issued before the MFMAs of batch j
                half8_t fa[6], fb[6];
                auto ldq = [&](half8_t* f, int kk) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) f[i] = fz_lds_ld_h8(base, (kk * 3 + i) * XC_FRAG);
                };
                auto mmq = [&](const half8_t* f, int kk) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) qacc[i % 3] = fz_mfma_32x32x16_f16(f[i], xb[10 * k + kk + i / 3], qacc[i % 3]);
                };
                XC_TK(4);
                ldq(fa, 0);
                FZ_SCHED_FENCE();
                ldq(fb, 2);
                mmq(fa, 0);
                FZ_SCHED_FENCE();
                ldq(fa, 4);
                mmq(fb, 2);
                FZ_SCHED_FENCE();
                ldq(fb, 6);
                mmq(fa, 4);
                FZ_SCHED_FENCE();
                ldq(fa, 8);
                mmq(fb, 6);
                FZ_SCHED_FENCE();
                mmq(fa, 8);
                XC_TK(5);
                XC_TK_ADD(5, tk4, tk5);   // q projection: 30 fragment reads + MFMAs
            }
            half8_t qf[6];  // fragment 2 c + b = units [32 c + 16 b, + 16): head a = fragments 0..2, head b = 3..5
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 8; ++e) qf[2 * c + b][e] = (half_t)qacc[c][8 * b + e];
            half8_t oga[3], ogb[3];
            {   // sub-step 2: the even head; fragments 0, 1 of the pair's o are complete
                const int n = step;
                sync_step();
                if (live) {
                    XC_TK(6);
                    head(qf, slot_base(n), oga);
                    XC_TK(7);
                    XC_TK_ADD(6, tk6, tk7);   // one head: QK^T, softmax, PV
                    *reinterpret_cast<half8_t*>(hbase + lane_off) = oga[0];
                    *reinterpret_cast<half8_t*>(hbase + XC_FRAG + lane_off) = oga[1];
                }
            }
            {   // sub-step 3: the odd head; fragment 2 = (even head, chunk 4 | odd head, chunk 0), fragments 3, 4
                const int n = step;
                sync_step();
                if (live) {
                    XC_TK(6);
                    head(qf + 3, slot_base(n), ogb);
                    XC_TK(7);
                    XC_TK_ADD(6, tk6, tk7);
                    half8_t mid;
#pragma unroll
                    for (int e = 0; e < 8; ++e) mid[e] = hi ? ogb[2][e] : oga[2][e];
                    *reinterpret_cast<half8_t*>(hbase + 2 * XC_FRAG + lane_off) = mid;
                    *reinterpret_cast<half8_t*>(hbase + 3 * XC_FRAG + lane_off) = ogb[0];
                    *reinterpret_cast<half8_t*>(hbase + 4 * XC_FRAG + lane_off) = ogb[1];
                }
            }
        }
    } else {
        // ================================================ O waves ===========================================================================
        fz_setprio_hi();  // its MFMAs and fragment reads go first: they fit beside the partner's softmax, not the other way round
#pragma unroll
        for (int c = 0; c < XC_CT; ++c) yacc[c] = fz_zero_f16v();
        if constexpr (FRONT) {
            // This wave has nothing to compute yet: it fetches the RESIDUAL chunks of the strip for its partner, a pass ahead, and leaves them
            // in the hand-over area as fragments (lane = row + 32 * k half, as the partner's accumulators want them).  Its loads are retired
            // by the vmcnt(0) in front of the barriers of ITS sub-steps (the odd ones) -- pinned there, in front of the next DMA issue.
            const int64_t row = row0 + l31;
            const half_t* rs = g.res + row * XC_C + hi * 8;
            half8_t ra[8];
#pragma unroll
            for (int f = 0; f < 6; ++f) ra[f] = fz_ld_h8(rs + f * 16);            // pass 0: tiles 0..2 = fragments 0..5
            if (w4 == 0) {   // the constants block (bias | gamma | beta): two fragments, once
                fz_glds16_so(xc_uniform_ptr(g.wpack), lane_off, xc_uniform_ptr(raw + XC_COFF));
                fz_glds16_so(xc_uniform_ptr(g.wpack + XC_FRAG), lane_off, xc_uniform_ptr(raw + XC_COFF + XC_FRAG));
            }
            issue_step(1);
            fz_wait_vm0();
#pragma unroll
            for (int f = 0; f < 6; ++f) *reinterpret_cast<half8_t*>(hbase + f * XC_FRAG + lane_off) = ra[f];
            sync_step();                                                            // sub-step 0: the partner takes pass 0's
#pragma unroll
            for (int f = 0; f < 6; ++f) ra[f] = fz_ld_h8(rs + (6 + f) * 16);       // pass 1: tiles 3..5 = fragments 6..11
            // sub-step 1 (this wave's: vmcnt(0) + barrier + DMA of sub-step 3), with the loads pinned between the wait and the DMA issue
            fz_wait_vm0();
#pragma unroll
            for (int f = 0; f < 6; ++f) asm volatile("" : "+v"(ra[f]));
            fz_barrier_nodrain();
            if (step + 2 < NSTEP) issue_step(step + 2);
            ++step;
#pragma unroll
            for (int f = 0; f < 6; ++f) *reinterpret_cast<half8_t*>(hbase + f * XC_FRAG + lane_off) = ra[f];
            sync_step();                                                            // sub-step 2: the partner takes pass 1's
#pragma unroll
            for (int f = 0; f < 8; ++f) ra[f] = fz_ld_h8(rs + (12 + f) * 16);      // passes 2, 3: tiles 6..9 = fragments 12..19
            fz_wait_vm0();                                                          // sub-step 3
#pragma unroll
            for (int f = 0; f < 8; ++f) asm volatile("" : "+v"(ra[f]));
            fz_barrier_nodrain();
            if (step + 2 < NSTEP) issue_step(step + 2);
            ++step;
#pragma unroll
            for (int f = 0; f < 8; ++f) *reinterpret_cast<half8_t*>(hbase + f * XC_FRAG + lane_off) = ra[f];
            sync_step();                                                            // sub-step 4: the partner takes passes 2 and 3's
            sync_step();                                                            // sub-step 5
        } else {
            issue_step(1);
        }
        for (int hp = 0; hp < XC_NBLK; ++hp) {
            const bool live = hp >= 1;
            sync_step();  // sub-step 0: nothing for this wave
            half8_t hf[5];
            {   // sub-step 1: the pair's o of the head pair before (written before the last two barriers) + k step 0
                const int n = step;
                sync_step();
                if (live) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) hf[i] = *reinterpret_cast<const half8_t*>(hbase + i * XC_FRAG + lane_off);
                    const fz_lds_addr base = slot_base(n);
                    half8_t fa[10];
#pragma unroll
                    for (int c = 0; c < XC_CT; ++c) fa[c] = fz_lds_ld_h8(base, (30 + c) * XC_FRAG);
#pragma unroll
                    for (int c = 0; c < XC_CT; ++c) yacc[c] = fz_mfma_32x32x16_f16(fa[c], hf[0], yacc[c]);
                }
            }
#pragma unroll
            for (int k = 2; k < 4; ++k) {   // sub-steps 2, 3: k steps 1, 2 and 3, 4
                const int n = step;
                sync_step();
                if (!live) continue;
                const fz_lds_addr base = slot_base(n);
                half8_t fa[10], fb[10];
#pragma unroll
                for (int c = 0; c < XC_CT; ++c) fa[c] = fz_lds_ld_h8(base, (XC_KVH + c) * XC_FRAG);
                FZ_SCHED_FENCE();
#pragma unroll
                for (int c = 0; c < XC_CT; ++c) fb[c] = fz_lds_ld_h8(base, (XC_KVH + 10 + c) * XC_FRAG);
#pragma unroll
                for (int c = 0; c < XC_CT; ++c) yacc[c] = fz_mfma_32x32x16_f16(fa[c], hf[2 * k - 3], yacc[c]);
                FZ_SCHED_FENCE();
#pragma unroll
                for (int c = 0; c < XC_CT; ++c) yacc[c] = fz_mfma_32x32x16_f16(fb[c], hf[2 * k - 2], yacc[c]);
            }
        }
    }

    // ---- epilogue (csrc/ff_chain.hip): + bo -> fp16 -> LDS tile [128 rows][320 + 8] (the O waves own the accumulators) -> (+ res) -> y and
    //      LayerNorm(y), full rows, ALL eight waves.  FRONT: the residual is the hidden_states the QA waves of THIS workgroup stored above
    //      (same CU: visible after their vmcnt wait and the barrier)
#ifdef XC_TIMING
    tacc[3] = clock64() - tk_start;  // prologue + all sub-steps
#endif
    fz_wait_vm0();
    __syncthreads();
    half_t* Call = reinterpret_cast<half_t*>(raw);
    if (!is_qa) {
        half_t* Cs = Call + pair * 32 * XC_OSTR;
#pragma unroll
        for (int c = 0; c < XC_CT; ++c)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int co = c * 32 + 8 * gq + 4 * hi;
                half4_t bv;
                if (g.bo != nullptr) {
                    bv = *reinterpret_cast<const half4_t*>(g.bo + co);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = (half_t)0.0f;
                }
                half4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (half_t)(yacc[c][4 * gq + e] + (float)bv[e]);
                *reinterpret_cast<half4_t*>(Cs + l31 * XC_OSTR + co) = v;
            }
    }
    __syncthreads();
    const half_t* resp = FRONT ? g.y1 : g.res;
    const int l8 = lane & 7;
    FzRow5 gmv, btv;
    if (g.yln != nullptr) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            gmv.c[i] = fz_ld_h8(g.gamma + (l8 + 8 * i) * 8);
            btv.c[i] = fz_ld_h8(g.beta + (l8 + 8 * i) * 8);
        }
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {  // 8 rows per pass: 8 lanes per row, 5 chunks of 8 channels per lane
        const int rl = wave * 16 + it * 8 + (lane >> 3);
        const int64_t px = (int64_t)blk * XC_ROWS + rl;
        FzRow5 v;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const half8_t a = fz_ld_h8(Call + rl * XC_OSTR + (l8 + 8 * i) * 8);
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)a[e];
            if (resp != nullptr) {
                const half8_t r = fz_ld_h8(resp + px * XC_C + (l8 + 8 * i) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += (float)r[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v.c[i][e] = (half_t)f[e];
            fz_st_h8(g.y + px * XC_C + (l8 + 8 * i) * 8, v.c[i]);
        }
#ifdef XC_DEBUG_XN
        continue;
#endif
        if (g.yln == nullptr) continue;
        // LayerNorm of the stored row: the out-of-line body igemm.hip's GS == -1 epilogue calls (fz_rt.h): the same bits
        const FzRow5 o = fz_ln_row320(v, gmv, btv, g.eps);
        {
#pragma unroll
            for (int i = 0; i < 5; ++i) fz_st_h8(g.yln + px * XC_C + (l8 + 8 * i) * 8, o.c[i]);
        }
    }
#ifdef XC_TIMING
    tacc[4] = clock64() - tk_start;  // whole kernel
    if (blockIdx.x == 0 && (tid == 0 || tid == 256))
        for (int i = 0; i < 8; ++i) xc_timing[tid >> 8][i] = tacc[i];
#endif
}

// ---- packing: one thread per 16 bytes of a stream ------------------------------------------------------------------------------------------
struct XcPackArgs {
    const half_t* wq;    // [320][320]  attn2.to_q.weight
    const half_t* wo;    // [320][320]  attn2.to_out[0].weight
    const half_t* wo1;   // [320][320]  attn1.to_out[0].weight, or null (no FRONT part)
    const half_t* bo1;   // [320] or null: attn1.to_out[0].bias       }
    const half_t* g1;    // [320]: norm2.weight                       } FRONT: the constants block
    const half_t* b1;    // [320]: norm2.bias                         }
    char* out;
};

FZ_KERNEL void __launch_bounds__(256) xattn_chain_pack_kernel(XcPackArgs g) {
    const int nfront = g.wo1 != nullptr ? XC_FRONTF : 0;
    const int64_t total = (int64_t)(nfront + XC_NBLK * XC_BLK) * 64;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int f = (int)(id >> 6), ln = (int)(id & 63), l31 = ln & 31, hi = ln >> 5;
        half8_t v = fz_zero_h8();
        if (f < 2 && nfront) {   // constants: 1024 halves = bias[320] | gamma[320] | beta[320] | 0
            const int h0 = (f * 64 + ln) * 8;
            const half_t* src = h0 < XC_C ? g.bo1 : (h0 < 2 * XC_C ? g.g1 : (h0 < 3 * XC_C ? g.b1 : nullptr));
            if (src != nullptr) v = fz_ld_h8(src + h0 % XC_C);
        } else if (f < nfront) {   // Wo1: passes of 3, 3, 2, 2 tiles; inside a pass (k step s, tile c); output channel of A row i = 32 tile + unit(i)
            const int fw = f - 2;
            int t0, ntl, r;
            if (fw < 120) {
                t0 = 3 * (fw / 60), ntl = 3, r = fw % 60;
            } else {
                t0 = 6 + 2 * ((fw - 120) / 40), ntl = 2, r = (fw - 120) % 40;
            }
            const int s = r / ntl, c = r - ntl * s;
            v = fz_ld_h8(g.wo1 + (int64_t)(32 * (t0 + c) + xc_unit_of_arow(l31)) * XC_C + 16 * s + 8 * hi);
        } else {
            const int m = f - nfront, hp = m / XC_BLK, r = m - hp * XC_BLK;
            if (r < 60) {   // Wq of head pair hp: (k step s, tile c of the pair); unit u of the pair = head 2 hp + u / 48, channel u % 48 (< 40)
                if (hp < 4) {
                    const int s = r / 3, c = r - 3 * s;
                    const int u = 32 * c + xc_unit_of_arow(l31), h = 2 * hp + u / 48, w = u % 48;
                    if (w < XC_D) v = fz_ld_h8(g.wq + (int64_t)(h * XC_D + w) * XC_C + 16 * s + 8 * hi);
                }
            } else if (hp >= 1) {   // Wo: k steps 5 (hp - 1) + [0, 5) x 10 tiles, natural row order
                const int q = r - 60, s = 5 * (hp - 1) + q / XC_CT, c = q % XC_CT;
                v = fz_ld_h8(g.wo + (int64_t)(32 * c + l31) * XC_C + 16 * s + 8 * hi);
            }
        }
        *reinterpret_cast<half8_t*>(g.out + id * 16) = v;
    }
}

struct XcKvArgs {
    const half_t* k;     // [batch][>= lk][>= 320]: K of the text context
    const half_t* vt;    // [batch][320][>= 96]: V^T
    char* out;
    int64_t k_batch_stride, k_row_stride, vt_batch_stride, vt_chan_stride;
    int batch, lk;
};

FZ_KERNEL void __launch_bounds__(256) xattn_chain_kv_pack_kernel(XcKvArgs g) {
    const int64_t total = (int64_t)g.batch * XC_H * XC_KVH * 64;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int f = (int)(id >> 6), ln = (int)(id & 63), l31 = ln & 31, hi = ln >> 5;
        const int b = f / (XC_H * XC_KVH), r0 = f - b * (XC_H * XC_KVH), h = r0 / XC_KVH, r = r0 - h * XC_KVH;
        half8_t v = fz_zero_h8();
        if (r < 9) {   // K: (key sub-tile, k step c): A row i = key 32 sub + unit(i), 8 channels 16 c + 8 hi of the head
            const int sub = r / 3, c = r - 3 * sub;
            const int key = 32 * sub + xc_unit_of_arow(l31), dd = 16 * c + 8 * hi;
            if (key < g.lk && dd < XC_D) v = fz_ld_h8(g.k + (int64_t)b * g.k_batch_stride + (int64_t)key * g.k_row_stride + h * XC_D + dd);
        } else {       // V^T: (channel tile t, key sub-tile, k step mm): A row i = channel xc_vchan_of_arow, 8 keys 32 sub + 16 mm + 8 hi
            const int q = r - 9, t = q / 6, sub = (q - 6 * t) / 2, mm = q & 1;
            const int ch = xc_vchan_of_arow(h & 1, t, l31), key0 = 32 * sub + 16 * mm + 8 * hi;
            if (ch >= 0) {
                const half8_t raw8 = fz_ld_h8(g.vt + (int64_t)b * g.vt_batch_stride + (int64_t)(h * XC_D + ch) * g.vt_chan_stride + key0);
                for (int e = 0; e < 8; ++e) v[e] = key0 + e < g.lk ? raw8[e] : (half_t)0.0f;
            }
        }
        *reinterpret_cast<half8_t*>(g.out + id * 16) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
//                                                   host side
// ---------------------------------------------------------------------------------------------------------------
#ifdef XC_TIMING
extern "C" int fz_xattn_chain_timing(long long* out) {  // trial builds only
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(xc_timing), sizeof(long long) * 16) == hipSuccess ? FZ_OK : FZ_ERR_LAUNCH;
}
#endif

extern "C" int fz_xattn_chain_ok(int64_t rows, int64_t rows_per_frame, int channels, int heads, int lk) {
    return rows > 0 && rows < (1ll << 38) && rows_per_frame > 0 && rows_per_frame % XC_ROWS == 0 && rows % rows_per_frame == 0 &&
           channels == XC_C && heads == XC_H && lk > 0 && lk <= FZ_CROSS_MAX_KEYS;
}

// Where is the one launch the faster form on MI355X?  A workgroup streams the whole weight + context set (0.6-0.8 MB) for its 128 rows, so
// the launch wants the chip full (profiles/r06_xattn_chain_ab.txt).
extern "C" int fz_xattn_chain_preferred(int64_t rows, int64_t rows_per_frame, int channels, int heads, int lk) {
    return fz_xattn_chain_ok(rows, rows_per_frame, channels, heads, lk) && rows >= 128 * 192;
}

extern "C" int64_t fz_xattn_chain_pack_bytes(int with_front) { return (int64_t)((with_front ? XC_FRONTF : 0) + XC_NBLK * XC_BLK) * XC_FRAG; }

extern "C" int64_t fz_xattn_chain_kv_pack_bytes(int batch) { return batch > 0 ? (int64_t)batch * XC_H * XC_KVH * XC_FRAG : 0; }

extern "C" int fz_xattn_chain_pack(const void* wq, const void* wo, const void* wo1, const void* bias_out1, const void* ln1_gamma,
                                   const void* ln1_beta, void* packed, void* stream) {
    if (!wq || !wo || !packed) return FZ_ERR_BAD_ARG;
    if (wo1 != nullptr && (!ln1_gamma || !ln1_beta)) return FZ_ERR_BAD_ARG;
    XcPackArgs g = {(const half_t*)wq, (const half_t*)wo, (const half_t*)wo1, (const half_t*)bias_out1, (const half_t*)ln1_gamma,
                    (const half_t*)ln1_beta, (char*)packed};
    const int64_t total = fz_xattn_chain_pack_bytes(wo1 != nullptr) / 16;
    FZ_LAUNCH(xattn_chain_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, g);
    return fz_last_launch_status();
}

extern "C" int fz_xattn_chain_kv_pack(const void* k, int64_t k_batch_stride, int64_t k_row_stride, const void* vt, int64_t vt_batch_stride,
                                      int64_t vt_chan_stride, int batch, int lk, void* packed, void* stream) {
    if (!k || !vt || !packed || batch <= 0 || lk <= 0 || lk > FZ_CROSS_MAX_KEYS) return FZ_ERR_BAD_ARG;
    if ((k_batch_stride | k_row_stride | vt_batch_stride | vt_chan_stride) & 7) return FZ_ERR_BAD_ARG;
    XcKvArgs g = {(const half_t*)k, (const half_t*)vt, (char*)packed, k_batch_stride, k_row_stride, vt_batch_stride, vt_chan_stride, batch, lk};
    const int64_t total = fz_xattn_chain_kv_pack_bytes(batch) / 16;
    FZ_LAUNCH(xattn_chain_kv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, g);
    return fz_last_launch_status();
}

extern "C" int fz_xattn_chain(const FzXattnChain* d, void* stream) {
    if (!d || !d->x || !d->packed || !d->kv_packed || !d->y) return FZ_ERR_BAD_ARG;
    if (!fz_xattn_chain_ok(d->rows, d->rows_per_frame, d->channels, d->heads, d->lk)) return FZ_ERR_UNSUPPORTED;
    if (d->frames_per_batch <= 0) return FZ_ERR_BAD_ARG;
    if (d->y_ln != nullptr && (!d->ln_gamma || !d->ln_beta)) return FZ_ERR_BAD_ARG;
    const bool front = d->front != 0;
    if (front && (!d->res || !d->y1)) return FZ_ERR_BAD_ARG;
    XcArgs g = {};
    g.xn = (const half_t*)d->x;
    g.res = (const half_t*)d->res;
    g.wpack = (const char*)d->packed;
    g.kvpack = (const char*)d->kv_packed;
    g.bo = (const half_t*)d->bias_out;
    g.y = (half_t*)d->y;
    g.yln = (half_t*)d->y_ln;
    g.gamma = (const half_t*)d->ln_gamma;
    g.beta = (const half_t*)d->ln_beta;
    g.y1 = (half_t*)d->y1;
    g.rows = d->rows;
    g.rows_per_frame = d->rows_per_frame;
    g.frames_per_batch = d->frames_per_batch;
    g.lk = d->lk;
    g.cs = d->scale * 1.4426950408889634f;
    g.eps = d->ln_eps;
    g.eps1 = d->ln1_eps;
    const int64_t nwg = d->rows / XC_ROWS;
#ifndef FZ_EMU
    static std::atomic<uint64_t> attr_set_mask[2] = {{0}, {0}};  // LDS above 64 KB is an opt-in function attribute, per device and kernel
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FZ_ERR_LAUNCH;
    if (dev >= 64 || !(attr_set_mask[front].load(std::memory_order_relaxed) >> dev & 1)) {
        const void* fn = front ? reinterpret_cast<const void*>(&xattn_chain_kernel<true>) : reinterpret_cast<const void*>(&xattn_chain_kernel<false>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)XC_LDS_BYTES) != hipSuccess) return FZ_ERR_LAUNCH;
        if (dev < 64) attr_set_mask[front].fetch_or(1ull << dev, std::memory_order_relaxed);
    }
#endif
    if (front) {
        FZ_LAUNCH(xattn_chain_kernel<true>, dim3((unsigned)nwg), dim3(512), XC_LDS_BYTES, stream, g);
    } else {
        FZ_LAUNCH(xattn_chain_kernel<false>, dim3((unsigned)nwg), dim3(512), XC_LDS_BYTES, stream, g);
    }
    return fz_last_launch_status();
}
