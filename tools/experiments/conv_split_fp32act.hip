// The SPLIT engines' conv kernel (bmi_model_desc.dtype = BMI_DTYPE_F16X2 / BMI_DTYPE_BF16X3): fp32 activations in the workspace
// (the exact engine's layout and its stem / site / max-pool / head kernels), every conv on the 16-bit matrix pipe with BOTH operands
// split into a 16-bit head and tail,
//
//     v = hi + lo,   hi = rn16(v),   lo = rn16(v - hi)        (fp16: 22 significant bits, bf16: 16)
//     w . x  =  w_lo . x_hi  +  w_hi . x_lo  +  w_hi . x_hi    (three v_mfma_f32_32x32x16_{f16,bf16} per K-step, fp32 accumulate;
//                                                              each 16-bit product is exact in fp32; lo . lo — 2^-22 / 2^-16 relative — is dropped)
//
// i.e. the dense layers' trick (dense_f32.hip:dense_split_kernel) for the convolutions: the arithmetic of the reference's fp32 CPU
// path (ATen conv2d, SA/models/resnet18/resnet18.py:32-48; the converted nets of Hardware_Artifact/converter/pytorch/nn2bnn.py:32-45
// on SA/models/vgg19/vgg19.py:256-324, whose peaky logits put plain fp16 at 1.7e-3) to ~1e-6 (f16x2) / ~1e-5 (bf16x3) at 3/16 instead
// of 1/16 of the 16-bit MFMA rate's cost: north_star's 1e-3 with a margin of two to three orders where fp16 / bf16 have none, and
// BASELINE configs[1] ("bf16") inside 1e-3 on the bf16 pipe.  bf16x3 has fp32's exponent range; f16x2 needs |v| < 65504.
//
// One generic per-tap implicit GEMM (as conv_exact.hip / conv_igemm.hip):
//     D[cout][pixel] = sum_k W[cout][k] X[k][pixel],   k = (ky*ks + kx)*Cin + ci
//   tile     = CT = 64 TI channels (TI = 1, 2, 4 by Cout) x 256 pixels x 32 deep (one tap, 32 channels), 512 threads: wave w =
//              channel half w >> 2, pixel quarter w & 3; wave tile 32 TI ch x 64 px = TI x 2 accumulators of 32 x 32
//   weights  = 16-bit [2][Cout][k*k*Cin] (plane 0: heads, plane 1: tails; split ONCE by the host when the engine is built), fetched by
//              LDS-DMA (global_load_lds, 16 B per lane) one K-step ahead
//   input    = fp32 NHWC, fetched into registers one K-step ahead (8 consecutive channels of a pixel per item: four lanes cover a
//              pixel's 128-byte line), split on the VALU and written as head / tail planes under the MFMAs of the current step
//   LDS      = two stages of [W hi | W lo | X hi | X lo], 64-byte rows (32 k of one channel / pixel); the 16-byte chunk c of row r
//              sits at chunk c ^ ((r >> 2) & 3): conflict-free for the fragments' ds_read_b128 (lane groups of 16 rows) and for the
//              staging writes; the DMA writes lane-linearly, so the permutation is applied to the per-lane SOURCE address
//   epilogue = conv_exact.hip's, on the accumulator quads (a lane holds pixel lane & 31 and, per quad q, channels 8q + 4(lane >> 5)..):
//              folded BN, inner / outer site of every kind, fp32 residual, ReLU, fp32 store
// Two wave groups in ping-pong (below).  The fp16-engine launch forms that exist for speed only (fused shortcut, pair, pooling, lazy sites,
// split-K, dynamic-exit row tables) are not built for this dtype (bmi_create keeps them out, as for the exact engine).
#include <type_traits>

#include "conv_epilogue.h"
#include "kernels.h"

typedef float f32x16_s __attribute__((ext_vector_type(16)));
typedef float f32x4_s __attribute__((ext_vector_type(4)));

#define SP_PT 256
// LDS-DMA, 16 B per lane from SBASE (wave-uniform, 64-bit) + VOFF (per lane, 32-bit) to LDSPTR + 16 lane.  Inline asm, not
// __builtin_amdgcn_global_load_lds: to hipcc's waitcnt pass the builtin is a load AND a store ("mixed events": no in-order counting),
// and while one is in flight every register dependency on a plain global load becomes s_waitcnt vmcnt(0).  The DMA writes no register,
// so hiding it is safe; the K loop's counted waits cover it (it is older than what they leave in flight).
#define SP_GLDS16S(VOFF, SBASE, LDSPTR)                                                                                      \
    {                                                                                                                        \
        const uint64_t b_ = (uint64_t)(uintptr_t)(SBASE);       /* (readfirstlane: the base must sit in SGPRs whatever hipcc thinks of its uniformity) */ \
        const uint64_t sb_ = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b_ >> 32)) << 32) |                   \
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b_);                                    \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                                         \
                     :: "s"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(LDSPTR)), "v"(VOFF), "s"(sb_) : "m0", "memory"); \
    }

static __device__ float g_split_zero[64];   // zeros: what an out-of-image tap / a tile row beyond M fetches (256 B)

// v[0..7] -> heads and tails.  fp16: hi = rn16(v); bf16: hi = v truncated to bf16 (one AND instead of a rounding convert: |lo| < 2^-7 |v|
// instead of <= 2^-8 |v|, the pair still carries 16 significant bits); lo = rn16(v - hi) either way (v - hi is exact in fp32).
template <bool BF>
__device__ __forceinline__ void split8(const f32x4_s& x0, const f32x4_s& x1, half8_t& hi, half8_t& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? x0[e] : x1[e - 4];
        if constexpr (BF) {
            const uint32_t hb = __builtin_bit_cast(uint32_t, v) & 0xffff0000u;
            hi[e] = __builtin_bit_cast(_Float16, (uint16_t)(hb >> 16));
            lo[e] = a16_from_f32<true>(v - __builtin_bit_cast(float, hb));
        } else {
            const _Float16 h = (_Float16)v;
            hi[e] = h;
            lo[e] = (_Float16)__builtin_fmaf((float)h, -1.0f, v);
        }
    }
}

template <bool BF, int TI>
__global__ __launch_bounds__(512) void conv_split_kernel(ConvArgs a) {
    constexpr int CT = 64 * TI;
    constexpr int WPL = CT * 64;                       // bytes of one weight plane of a stage
    constexpr int XPL = SP_PT * 64;
    // LDS: a ring of NW weight slots [hi | lo] and two input stages [hi | lo].  NW = 3 (TI <= 2): the weights of K-step ks + 2 are
    // fetched during K-step ks and have a whole K-step to land — with two slots they have half of one (two barrier intervals: 2 x 6 TI
    // MFMAs), which TI = 4 covers (measured) and TI = 2 does not: its K-step took 3900 cycles for 1536 of MFMA, the rest waiting for
    // the DMA.  (TI = 4 with three slots would need all 160 KB of the CU's LDS.)
    constexpr int NW = TI == 4 ? 2 : 3;
    constexpr int WSLOT = 2 * WPL, XST = 2 * XPL;
    constexpr int XBASE = NW * WSLOT;
    __shared__ __attribute__((aligned(16))) char smem[NW * WSLOT + 2 * XST];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kq = lane >> 5;
    const int wc = wave >> 2, wp = wave & 3;
    const int sw = (r >> 2) & 3;

    const int n_ct = a.Cout / CT;
    int ptile, ctile;
    xcd_tile_map(blockIdx.x, (a.M + SP_PT - 1) / SP_PT, n_ct, ptile, ctile, 1);
    const int ch0 = ctile * CT;
    const int pix0 = ptile * SP_PT;
    const int HoWo = a.Ho * a.Wo;
    const int Ktot = a.ksize * a.ksize * a.Cin;
    const float* const in = (const float*)a.in;

    // ---- weight DMA: piece q = tid + 512 i of the stage's [hi | lo] image: plane q / (4 CT), row (q >> 2) % CT, slot q & 3.  A block of
    // 512 pieces is 128 rows (TI = 1: a plane is 256 pieces = 64 rows, waves 0-3 / 4-7 take plane 0 / 1) — a multiple of the swizzle's
    // period, so every piece of a lane has the same row-in-block and source slot: ONE 32-bit lane offset serves them all, and the
    // piece's plane / row block / K-step go into a scalar base (the vaddr + saddr form of global_load_lds: 1 VGPR instead of 2 TI)
    const uint32_t woff_l = (uint32_t)(((tid >> 2) & (TI == 1 ? 63 : 127)) * Ktot + (((tid & 3) ^ ((tid >> 4) & 3)) << 3)) * 2u;
    const char* wbase[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int q0 = 512 * i + (TI == 1 ? 256 * (wave >> 2) : 0), plane = q0 / (4 * CT), row0 = (q0 - plane * 4 * CT) >> 2;
        wbase[i] = (const char*)(a.wgt + ((size_t)plane * a.Cout + ch0 + row0) * Ktot);
    }
#define SP_ISSUE_W1(I, KOFF, ST) SP_GLDS16S(woff_l, wbase[I] + 2 * (KOFF), (ST) + ((I) * 512 + wave * 64) * 16)
#if SP_ABL_NOW                  // timing probe (wrong results): every K-step fetches the weights of K-step 0
#define SP_ISSUE_W(KOFF, ST)                                                                             \
    _Pragma("unroll") for (int i = 0; i < TI; ++i) SP_GLDS16S(woff_l, wbase[i], (ST) + (i * 512 + wave * 64) * 16);
#else
#define SP_ISSUE_W(KOFF, ST)                                                                             \
    _Pragma("unroll") for (int i = 0; i < TI; ++i) SP_GLDS16S(woff_l, wbase[i] + 2 * (KOFF), (ST) + (i * 512 + wave * 64) * 16);
#endif

    // ---- input staging: item f = tid + 512 i: tile row (pixel) (tid >> 2) + 128 i, channels 8 (tid & 3) .. of the K-step ----
    // xorg = address of the item's channels at tap (0, 0) — outside the image where the padding says so, never dereferenced there: a
    // K-step adds its wave-uniform offset ((ky W + kx) Cin + c0) and fetches from a page of zeros instead when its tap is out of bounds
    const float* xorg[2];
    int iy0[2], ix0[2], xdst[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 2) + 128 * i;
        const int m = pix0 + row;
        const bool vm = m < a.M;
        const int mm = vm ? m : 0;
        const int n = mm / HoWo, rem = mm - n * HoWo;
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0[i] = vm ? oy * a.stride - a.pad : -0x10000;      // a row beyond M never passes the bounds test
        ix0[i] = ox * a.stride - a.pad;
        xorg[i] = in + ((long)(n % a.in_mod) * a.H * a.W + (long)(oy * a.stride - a.pad) * a.W + ix0[i]) * a.Cin + 8 * (tid & 3);
        xdst[i] = XBASE + row * 64 + (((tid & 3) ^ ((row >> 2) & 3)) << 4);
    }
    // X registers of one K-step: two items of 8 channels (two float4).  ONE set: the fetches of K-step ks + 2 are issued into it right
    // after the split of K-step ks + 1 has read it, and stay in flight for a whole K-step.
    struct XRegs { f32x4_s v[2][2]; };
    XRegs xa;
    auto load_x_item = [&](XRegs& R, int i, int ky, int kx, int c0) {
        const long soff = (long)(ky * a.W + kx) * a.Cin + min(c0, a.Cin - 32);        // wave-uniform (c0 = Cin: past the last K-step)
        const bool ok = (unsigned)(iy0[i] + ky) < (unsigned)a.H && (unsigned)(ix0[i] + kx) < (unsigned)a.W;
        // (a select between two addresses, not a branch: a branch splits the K loop into basic blocks and hipcc then waits
        //  vmcnt(0) where a counted wait would do)
        const float* p_ = ok ? xorg[i] + soff : g_split_zero;
#if SP_ABL_NOX                  // timing probe (wrong results): every K-step fetches the page of zeros (an L1 hit)
        p_ = g_split_zero;
#endif
        R.v[i][0] = *(const f32x4_s*)p_;
        R.v[i][1] = *(const f32x4_s*)(p_ + 4);
    };
    auto load_x = [&](XRegs& R, int ky, int kx, int c0) {
        load_x_item(R, 0, ky, kx, c0);
        load_x_item(R, 1, ky, kx, c0);
    };
    auto write_x = [&](const XRegs& R, char* st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            half8_t hi_, lo_;
            split8<BF>(R.v[i][0], R.v[i][1], hi_, lo_);
            *(half8_t*)(st + xdst[i]) = hi_;
            *(half8_t*)(st + xdst[i] + XPL) = lo_;
        }
    };

    f32x16_s acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int a_off = (wc * 32 * TI + r) * 64;
    const int b_off = XBASE + (wp * 64 + r) * 64;
    // fragments of one 16-deep sub-step SS of the K-step in stage ST ((TI + 2) x 2 ds_read_b128), and its TI x 2 x 3 MFMAs: the small
    // products first (lo . hi, hi . lo), then hi . hi
    half8_t ah[TI], al[TI], bh[2], bl[2];
#define SP_READ(WS, XS, SS)                                                                              \
    if (!SP_ABL_NOREAD || ks == 0) {                                                                     \
        const int coff = ((2 * (SS) + kq) ^ sw) << 4;                                                    \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) {                                                 \
            ah[i] = *(const half8_t*)((WS) + a_off + i * 32 * 64 + coff);                                \
            al[i] = *(const half8_t*)((WS) + WPL + a_off + i * 32 * 64 + coff);                          \
        }                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                  \
            bh[j] = *(const half8_t*)((XS) + b_off + j * 32 * 64 + coff);                                \
            bl[j] = *(const half8_t*)((XS) + XPL + b_off + j * 32 * 64 + coff);                          \
        }                                                                                                \
    }
    // the 6 TI MFMAs of a sub-step in NG groups, HOOK(g) behind group g: the staging work that needs no LDS result — the weight DMA's
    // issue, the input fetches' address arithmetic and loads — rides in the MFMA parts, in the issue slots the matrix pipe leaves free
    // (an MFMA holds the SIMD's issue port for 8 of its 32 cycles), instead of lengthening the LOAD parts: those ran ~900 cycles
    // against 768 of MFMA at TI = 4 (12 fragment reads + four DMA pieces at 100-185 cycles each inside a phase that also reads LDS,
    // MI355X_MICROARCH.md).  The scheduler is fenced around every hook, so the order below is the order issued.
    constexpr int NG = TI == 4 ? 4 : 2, GS = 6 * TI / NG;
#define SP_MFMA(HOOK)                                                                                    \
    _Pragma("unroll") for (int m = 0; m < 6 * TI; ++m) {                                                 \
        const int pr = m / (2 * TI), rm = m - pr * 2 * TI, i = rm >> 1, j = rm & 1;                      \
        if (!SP_ABL_NOMFMA || m == 0) acc[i][j] = mfma_32x32x16<BF>(pr == 0 ? al[i] : ah[i], pr == 1 ? bl[j] : bh[j], acc[i][j]);      \
        if ((m + 1) % GS == 0) {                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                           \
            HOOK((m + 1) / GS - 1);                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }                                                                                                \
    }
    // raw s_barrier with the scheduler fenced off on both sides (a __syncthreads() would drain the fetches in flight: vmcnt(0))
// (no s_setprio around the MFMA parts: with the partner wave prioritised, the VALU-heavy LOAD parts — split, address arithmetic — issue at
// a fraction of their rate, MI355X_MICROARCH.md 'Two waves per SIMD' item 2; -DSP_SETPRIO=1 restores it for an A/B)
// timing probes (tools/ab_split.sh name:-DSP_ABL_...=1; wrong results by construction, never in the product build)
#ifndef SP_ABL_NOBAR
#define SP_ABL_NOBAR 0
#endif
#ifndef SP_ABL_NOSPLIT
#define SP_ABL_NOSPLIT 0
#endif
#ifndef SP_ABL_NOREAD
#define SP_ABL_NOREAD 0
#endif
#ifndef SP_ABL_NOMFMA
#define SP_ABL_NOMFMA 0
#endif
#ifndef SP_ABL_NOX
#define SP_ABL_NOX 0
#endif
#ifndef SP_ABL_NOW
#define SP_ABL_NOW 0
#endif
#ifndef SP_TAP_MAJOR
#define SP_TAP_MAJOR 0
#endif
#ifndef SP_SETPRIO
#define SP_SETPRIO 0
#endif
#define SP_PRIO(P) { if (SP_SETPRIO) __builtin_amdgcn_s_setprio(P); }
#define SP_BARRIER()                                   \
    {                                                  \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("" ::: "memory");                 \
        if (!SP_ABL_NOBAR) __builtin_amdgcn_s_barrier();   \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_sched_barrier(0);             \
    }

    const int nK = a.ksize * a.ksize * (a.Cin / 32);
    // (ky, kx, c0) of the K-step whose input is fetched next.  Every step fetches (at the end: clamped, unused), so the number of
    // loads in flight — what the counted vmcnt waits below rely on — never changes
    int ky = 0, kx = 0, c0 = 0;
    auto advance = [&]() {            // scalar selects, no branch; runs past the last K-step (c0 = Cin) at the end: the fetches of
#if SP_TAP_MAJOR                      // those steps are clamped to valid addresses and never used
        const int c1 = c0 + 32;
        const bool wrapc = c1 == a.Cin;
        c0 = wrapc ? 0 : c1;
        const int kx1 = kx + (wrapc ? 1 : 0);
        const bool wrapx = kx1 == a.ksize;
        kx = wrapx ? 0 : kx1;
        ky += wrapx ? 1 : 0;
#else
        // K order: 32-channel chunk outer, tap inner.  The k*k K-steps of a chunk read the SAME 128-byte lines of the tile's pixels
        // and their halo (a tap is a shift by whole pixels), so all but the first find them in L1 / L2; tap-major, a line came back
        // Cin / 32 K-steps later, behind 32 KB x Cin / 32 of other input per workgroup — beyond an XCD's L2 share: every K-step then
        // streamed its 32 KB at the far-memory rate (~10 B/clk/CU: 3300 cycles, whatever the MFMA work of the step)
        const int kx1 = kx + 1;
        const bool wrapx = kx1 == a.ksize;
        kx = wrapx ? 0 : kx1;
        const int ky1 = ky + (wrapx ? 1 : 0);
        const bool wrapy = ky1 == a.ksize;
        ky = wrapy ? 0 : ky1;
        c0 += wrapy ? 32 : 0;
#endif
    };
    // (the weight offset of (ky, kx, c0), clamped for the K-steps past the end)
#define SP_KOFF() min((ky * a.ksize + kx) * a.Cin + c0, Ktot - 32)
    // K-step 0 -> weight slot 0 / input stage 0 (NW = 3: K-step 1's weights -> slot 1 too); the input fetches of K-step 1 -> xa
    SP_ISSUE_W(0, smem);
    load_x(xa, 0, 0, 0);
    write_x(xa, smem);
    advance();
    if constexpr (NW == 3) { SP_ISSUE_W(SP_KOFF(), smem + WSLOT); }
    load_x(xa, ky, kx, c0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         // all but the four loads just issued: this wave's weight DMA has landed
    lds_barrier();
    // ---- ping-pong main loop (the schedule of conv_igemm_wide.hip) ----------------------------------------------------------------
    // A K-step is four intervals between barriers: LOAD(0) | MFMA(0) | LOAD(1) | MFMA(1), one 16-deep sub-step each.  A LOAD part
    // reads the sub-step's fragments and does the step's staging work (sub-step 0: the weight DMA of the next K-step; sub-step 1:
    // split + ds_write of the next K-step's input, fetched a whole step ago, then the fetches of the K-step after it into the same
    // registers); an MFMA part is 6 TI MFMAs at raised priority.  The two wave groups (waves 0-3 / 4-7 = the two channel halves: one
    // wave per SIMD each) run ONE BARRIER APART, so on every SIMD one wave is in an MFMA part while the other reads LDS, fetches and
    // splits: the matrix pipe does not wait for the staging, and a barrier is the hand-over between the two waves of a SIMD.
    // Intervals, K-step T: group 0 = 4T .. 4T+3, group 1 = 4T+1 .. 4T+4.  Hazards, by construction:
    //   WAR  what K-step T+1 will read (input stage (T+1)&1; weight slot (T+1) % NW for NW = 2, (T+2) % 3 = (T-1) % 3 for NW = 3) was
    //        last read for K-step T-1, in intervals 4T-2 (group 0) and 4T-1 (group 1); every LOAD part retires its reads (lgkmcnt(0))
    //        before the barrier that ends it; the first write of K-step T is group 0's DMA in interval 4T (NW = 2) / 4T+2 (NW = 3).
    //   RAW  the writers: DMA issued in 4T / 4T+1 (NW = 2) or a K-step earlier (NW = 3), ds_writes in 4T+2 / 4T+3; every wave retires
    //        its own DMA (the vmcnt(0) of its LOAD(1)) and ds_writes before the barrier that ends its LOAD(1), i.e. by 4T+3; the
    //        first reads of K-step T+1 are in 4T+4 (group 0) and 4T+5 (group 1).
    // Both groups execute the same number of barriers: group 1 one extra before the loop, group 0 one extra after it.
    const int g = wc;
    if (g == 1) SP_BARRIER();
    // K-step ks: weight slot ks % NW and input stage ks & 1 are complete; xa holds the input of K-step ks + 1, in flight since the
    // LOAD(1) of K-step ks - 1; (ky, kx, c0) = K-step ks + 1
    int wslot = 0;
    for (int ks = 0; ks < nK; ++ks) {
        const char* const ws = smem + wslot * WSLOT;
        const char* const xs = smem + (ks & 1) * XST;
        char* const nxs = smem + ((ks & 1) ^ 1) * XST;
        wslot = wslot + 1 == NW ? 0 : wslot + 1;              // slot of K-step ks + 1
        char* const wnext = smem + (NW == 2 ? wslot : (wslot + 1 == NW ? 0 : wslot + 1)) * WSLOT;   // NW = 2: K-step ks + 1's slot; NW = 3: K-step ks + 2's
        const int koff1 = SP_KOFF();                         // (ky, kx, c0) = K-step ks + 1 until MFMA(1)'s first hook
        // LOAD(0): the sub-step's fragments
        SP_READ(ws, xs, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SP_BARRIER();
        // MFMA(0); NW = 2 (TI = 4): K-step ks + 1's weight DMA, one piece behind each group of six MFMAs
#define SP_HOOK0(G) { if constexpr (NW == 2) { SP_ISSUE_W1(G, koff1, wnext); } }
        SP_PRIO(1);
        SP_MFMA(SP_HOOK0);
        SP_PRIO(0);
#undef SP_HOOK0
        SP_BARRIER();
        // LOAD(1): fragments; then the input of K-step ks + 1 — fetched in MFMA(1) of K-step ks - 1, three intervals ago — is split and written.
        // vmcnt(0): xa's fetches (hipcc waits for them too) and every weight DMA issued so far (which it does not know): NW = 2:
        // K-step ks + 1's, issued in the interval before; NW = 3: K-step ks + 1's, issued a whole K-step ago
        SP_READ(ws, xs, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!SP_ABL_NOSPLIT || ks == 0) write_x(xa, nxs);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SP_BARRIER();
        // MFMA(1); the fetches of K-step ks + 2 into the registers the split has just read (and, NW = 3, its weight DMA into the slot
        // K-step ks - 1 used, free since interval 4 ks - 1) behind the first groups
#define SP_HOOK1(G)                                                                                      \
    {                                                                                                    \
        if ((G) == 0) { advance(); load_x_item(xa, 0, ky, kx, c0); }                                     \
        if ((G) == 1) {                                                                                  \
            load_x_item(xa, 1, ky, kx, c0);                                                              \
            if constexpr (NW == 3) { SP_ISSUE_W(SP_KOFF(), wnext); }                                     \
        }                                                                                                \
    }
        SP_PRIO(1);
        SP_MFMA(SP_HOOK1);
        SP_PRIO(0);
#undef SP_HOOK1
        SP_BARRIER();
    }
    if (g == 0) SP_BARRIER();
#undef SP_ISSUE_W
#undef SP_ISSUE_W1
#undef SP_READ
#undef SP_MFMA
#undef SP_BARRIER
#undef SP_KOFF
#undef SP_PRIO

    // ---- epilogue, coalesced through LDS (the stages are dead: every wave is behind the loop's last barrier) ---------------------------
    // Straight from the accumulators a lane owns 4 channels of 32 different pixels: 16-byte stores 4 Cout bytes apart, and as many
    // scattered residual loads (rocprofv3 WRITE_SIZE: 2.0x the tensor on the Cout = 128 launches).  Instead, in two rounds (pixel tile
    // j = 0, 1 of every wave: 128 of the tile's pixels x all CT channels = CT / 2 KB of fp32): the raw accumulators go to LDS as
    // [pixel][channel] rows (16-byte chunk c of pixel row p at c ^ (p & 31 & (chunks - 1)): conflict-free both ways), then consecutive
    // lanes take consecutive chunks of a row — folded BN, site, residual, ReLU on the quad (epilogue_quad_f32: arithmetic and order of
    // every other conv kernel) and whole contiguous rows to and from HBM.  The second phase is a ROLLED loop (it does not touch the
    // accumulators, so nothing lands in scratch) — the site arithmetic is compiled once instead of 8 TI times.
    float* const out = (float*)a.out;
    const float* const res = (const float*)a.res;
    constexpr int CHUNKS = CT / 4;                             // 16-byte chunks per pixel row
    constexpr int ROWB = CT * 4;                               // bytes per row
    static_assert(128 * ROWB <= (int)sizeof(smem), "one round's tile fits the dead stages");
    auto round = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        __syncthreads();                                       // the previous round's rows have been read
        {
            const int pl = wp * 32 + r;                        // this lane's pixel among the round's 128
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int chunk = (wc * 32 * TI + 32 * i + 8 * q + 4 * kq) >> 2;
                    *(f32x4_s*)(smem + pl * ROWB + ((chunk ^ (pl & 31 & (CHUNKS - 1))) << 4)) =
                        f32x4_s{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                }
        }
        __syncthreads();
#pragma unroll 1
        for (int it = 0; it < 128 * CHUNKS / 512; ++it) {
            const int qi = tid + 512 * it, pl = qi / CHUNKS, chunk = qi - pl * CHUNKS;
            const int m_o = pix0 + (pl >> 5) * 64 + 32 * j + (pl & 31);
            if (m_o >= a.M) continue;
            const f32x4_s raw = *(const f32x4_s*)(smem + pl * ROWB + ((chunk ^ (pl & 31 & (CHUNKS - 1))) << 4));
            const int n = m_o / HoWo, rem = m_o - n * HoWo;
            PixelCtx p;
            p.out_off = ((size_t)n * HoWo + rem) * a.Cout;
            p.resp = nullptr;
            const int tl = n / a.B;
            p.b = n - tl * a.B;
            p.t = a.t0 + tl;
            p.e_pix = p.b * HoWo + rem;
            p.mrow = a.site.kind == BMI_SITE_MASKSEMBLE ? a.site.masks + (size_t)((a.site.cnt0 + p.t) % a.site.num_masks) * a.Cout : nullptr;
            const float* resp = res ? res + ((size_t)(n % a.res_mod) * HoWo + rem) * a.Cout : nullptr;
            const int c4 = ch0 + 4 * chunk;
            float v[4] = {raw[0], raw[1], raw[2], raw[3]};
            epilogue_quad_f32(a, p, resp, v, c4);
            *(f32x4_s*)(out + p.out_off + c4) = f32x4_s{v[0], v[1], v[2], v[3]};
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the last K-steps' (unused) weight DMA must land BEFORE the rows overwrite its slot
    round(std::integral_constant<int, 0>{});
    round(std::integral_constant<int, 1>{});
}

template <bool BF>
static int launch_split_t(const ConvArgs& a, hipStream_t s) {
    const int ct = a.Cout % 256 == 0 ? 256 : (a.Cout % 128 == 0 ? 128 : 64);
    const long blocks = (long)((a.M + SP_PT - 1) / SP_PT) * (a.Cout / ct);
    if (blocks <= 0 || blocks > 0x7fffffffL) return BMI_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)blocks), block(512);
    if (ct == 256) hipLaunchKernelGGL((conv_split_kernel<BF, 4>), grid, block, 0, s, a);
    else if (ct == 128) hipLaunchKernelGGL((conv_split_kernel<BF, 2>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv_split_kernel<BF, 1>), grid, block, 0, s, a);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

bool conv_takes_split_kernel(int cin, int cout) { return cin % 32 == 0 && cout % 64 == 0; }

int launch_conv_split(const ConvArgs& a, int bf16, hipStream_t s) {
    if (!conv_takes_split_kernel(a.Cin, a.Cout)) return BMI_ERR_UNSUPPORTED;
    if (a.in2 || a.wgt_b || a.in_bits || a.in2_bits || a.pool || a.pool_b || a.partial || a.imap) return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.M <= 0 || a.in_mod <= 0 || a.B <= 0 || (a.res && a.res_mod <= 0)) return BMI_ERR_INVALID;
    return bf16 ? launch_split_t<true>(a, s) : launch_split_t<false>(a, s);
}
