// The SPLIT engines' conv kernel (bmi_model_desc.dtype = BMI_DTYPE_F16X2 / BMI_DTYPE_BF16X3): every conv on the 16-bit matrix pipe with
// BOTH operands a 16-bit head and tail,
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
// Activations live in the workspace as pair32 tensors (conv_epilogue.h): per pixel, 32-channel blocks [hi x 32 | lo x 32] — the operand
// form, encoded ONCE by whatever produced the tensor (this kernel's epilogue, the stem, the site and max-pool kernels).  Round 5's first
// form (tools/experiments/conv_split_fp32act.hip) kept fp32 activations and split them while staging: timing probes put that staging
// (fetch into registers, ~50 VALU of conversions per thread and K-step, ds_write) at 21 % (Cout = 256) to 35 % (Cout = 128) of a launch,
// nine times per element (once per tap) and not overlapped by the ping-pong (a wave alone issues its LOAD part at ~10 cycles per
// instruction).  Now both operands are fetched by LDS-DMA and a LOAD part is the fragment reads.
//
// One generic per-tap implicit GEMM (as conv_exact.hip / conv_igemm.hip):
//     D[cout][pixel] = sum_k W[cout][k] X[k][pixel],   k = (ky*ks + kx)*Cin + ci
//   tile     = CT = 64 TI channels (TI = 1, 2, 4 by Cout) x 256 pixels x 32 deep (one tap, 32 channels), 512 threads: wave w =
//              channel half w >> 2, pixel quarter w & 3; wave tile 32 TI ch x 64 px = TI x 2 accumulators of 32 x 32
//   weights  = 16-bit [2][Cout][k*k*Cin] (plane 0: heads, plane 1: tails; split ONCE by the host when the engine is built)
//   input    = pair32: a K-step's operand of one pixel (32 channels, head + tail) is ONE contiguous 128-byte line
//   LDS      = rings of N = D + 1 weight slots [hi rows | lo rows] (64-byte rows, 16-byte chunk c of row r at c ^ ((r >> 2) & 3)) and
//              input slots (128-byte rows [hi | lo], chunk s of row r at s ^ ((r >> 1) & 7)): conflict-free for the fragments'
//              ds_read_b128; the DMA writes lane-linearly, so both permutations are applied to the per-lane SOURCE address.
//              D = the fetch distance in K-steps: 1 (TI = 4: 2 x 32 + 2 x 32 KB) or 2 (TI <= 2)
//   K order  = 32-channel chunk outer, tap inner: the k*k K-steps of a chunk read the same lines of the tile's pixels and their halo
//   epilogue = through LDS: [pixel][channel] fp32 rows, then 8 consecutive channels per lane — folded BN, inner / outer site of every
//              kind, residual (a pair32 tensor too), ReLU (epilogue_quad_f32v: arithmetic and order of every other conv kernel) — and
//              the result ENCODED into the pair32 output: whole contiguous rows to and from HBM
// Two wave groups in ping-pong (below).  The fp16-engine launch forms that exist for speed only (fused shortcut, pair, pooling, lazy
// sites) are not built for this dtype (bmi_create keeps them out, as for the exact engine); the fused shortcut, pair launches, split-K and the
// dynamic-exit row tables (IMAP) are.
#include <type_traits>

#include "conv_epilogue.h"
#include "kernels.h"

typedef float f32x16_s __attribute__((ext_vector_type(16)));
typedef float f32x4_s __attribute__((ext_vector_type(4)));

#define SP_PT 256
// LDS-DMA, 16 B per lane to LDSPTR + 16 lane.  Inline asm, not __builtin_amdgcn_global_load_lds: to hipcc's waitcnt pass the builtin is a
// load AND a store ("mixed events": no in-order counting) and every later register dependency on a global load becomes s_waitcnt vmcnt(0);
// written out, the K loop has NO compiler-visible vector-memory operation and its counted waits are exactly the ones below.  The DMA
// writes no register, so hiding it is safe.  S form: SBASE (wave-uniform, 64-bit) + VOFF (per lane, 32-bit); V form: a 64-bit lane address.
#define SP_GLDS16S(VOFF, SBASE, LDSPTR)                                                                                      \
    {                                                                                                                        \
        const uint64_t b_ = (uint64_t)(uintptr_t)(SBASE);       /* (readfirstlane: the base must sit in SGPRs whatever hipcc thinks of its uniformity) */ \
        const uint64_t sb_ = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b_ >> 32)) << 32) |                   \
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b_);                                    \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"                                         \
                     :: "s"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(LDSPTR)), "v"(VOFF), "s"(sb_) : "m0", "memory"); \
    }
#define SP_GLDS16V(VPTR, LDSPTR)                                                                                             \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"                                            \
                 :: "s"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(LDSPTR)), "v"(VPTR) : "m0", "memory")

typedef unsigned int u32x4_sp __attribute__((ext_vector_type(4)));
static __device__ unsigned int g_split_zero[64];   // zeros: what an out-of-image tap / a tile row beyond M fetches (256 B)

// timing probes (tools/ab_split.sh name:-DSP_ABL_...=1; wrong results by construction, never in the product build)
#ifndef SP_ABL_NOBAR
#define SP_ABL_NOBAR 0
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
#ifndef SP_ABL_SHIFT0
#define SP_ABL_SHIFT0 0      // timing probe: the shared slot is always read straight (loop-invariant fragment addresses); wrong results
#endif
#ifndef SP_ABL_X3
#define SP_ABL_X3 0          // timing probe: the pixel tile is fetched for the kx = 0 tap of a row of taps only (what a padded-row slot shared by the three
#endif                       // taps of a row would land in LDS); wrong results
#ifndef SP_TAP_MAJOR
#define SP_TAP_MAJOR 0
#endif

// SHX (3x3, stride 1, pad 1): the three taps of a tap row read ONE pixel slot.  The slot holds the row's centre column (kx = 1: pixel p's line is
// input (oy + ky - 1, ox)); the kx = 0 / 2 taps read the slot rows of the tile pixels p - 1 / p + 1 — the same image row unless p sits on the
// map's left / right edge, where the tap is padding: those lanes' fragments are cleared in registers.  The swizzle key of a row is a function of the
// row index, so a shifted read is as conflict-free as a straight one.  Two thirds of the pixel DMA — the larger part of what a K-step lands in
// LDS — is not issued at all: same-box probe (pixel DMA on the kx = 0 taps only, wrong results) 3x3 stride-1 launches -13 % (128 channels) ...
// -22 % (256 / 512), the headline step -17 %.  Bit-identical to the unshared loop (the same operands in the same order).
// SHX = 2 (map rows of 8, 16, 32 ... pixels; the swizzle key stays that of the PIXEL — pixels 2 m and 2 m + 1 share a key and sit in neighbouring
// cells of one map row, so the eight keys x two cell parities of a ds_read_b128 lane group are still 16 distinct slots; keyed by the cell index the
// zero cells shifted the parities and the reads conflicted two ways): no clearing at all — the slot keeps one ZERO cell in front of every map row (cell of tile pixel p =
// p + p / Wo + 1; written once at kernel start, never by the DMA, whose eight-pixel pieces never straddle a row), so a shifted read at an edge lands
// on padding by itself.  (SHX = 1 clears 16 registers per 16-deep sub-step behind the fragment reads, in the LOAD part the other wave group's MFMA
// part has to cover: measured +0.8 % instead of the probe's 17 %.)
// IMAP (dynamic early exit, bmi_forward_mcd_exit): the launch covers the still-active images only — compact image n of the launch is row
// a.imap[n] of every tensor and of the Philox index space (ConvArgs::imap; a kernel template parameter like everywhere else: see map_image)
template <bool BF, int TI, int SHX, bool IMAP = false>
__global__ __launch_bounds__(512) void conv_split_kernel(ConvArgs a) {
    constexpr bool PADX = SHX == 2;
    constexpr int CT = 64 * TI;
    constexpr int WPL = CT * 64;                       // bytes of one weight plane of a slot
    constexpr int WSLOT = 2 * WPL;
    constexpr int XSLOT = PADX ? 296 * 128 : SP_PT * 128;   // bytes of one input slot: 256 rows [hi 64 B | lo 64 B] (PADX: + up to 33 zero cells)
    // fetch distance: the operands of K-step ks + D are requested during K-step ks.  D = 2 needs rings of three: 96 + 96 KB at TI = 4
    // — more than the CU has — so TI = 4 fetches one K-step ahead (its K-step is 2 x 768 cycles of MFMA per wave group: enough for a
    // line that sits in L2, which all but the first tap of a chunk do) and TI <= 2 two (their K-steps are 384 / 192 cycles per group)
    constexpr int D = TI == 4 ? 1 : 2, NS = D + 1;
    constexpr int XBASE = NS * WSLOT;
    __shared__ __attribute__((aligned(16))) char smem[NS * (WSLOT + XSLOT) + (SHX == 1 ? 128 : 0)];      // (SHX = 1: the last slot's row 256, read by edge lanes and cleared)
    static_assert(sizeof(smem) <= 163840, "one workgroup's LDS");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, kq = lane >> 5;
    const int wc = wave >> 2, wp = wave & 3;
    const int sw = (r >> 2) & 3, sx = (r >> 1) & 7;

    const int n_ct = a.Cout / CT;
    int ptile, ctile;
    xcd_tile_map(blockIdx.x, (a.M + SP_PT - 1) / SP_PT, n_ct, ptile, ctile, 1);
    const int ch0 = ctile * CT;
    const int pix0 = ptile * SP_PT;
    const int HoWo = a.Ho * a.Wo;
    const int Ktot = a.ksize * a.ksize * a.Cin;

    // ---- weight DMA: piece q = tid + 512 i of the slot's [hi | lo] image: plane q / (4 CT), row (q >> 2) % CT, slot q & 3.  A block of
    // 512 pieces is 128 rows (TI = 1: a plane is 256 pieces = 64 rows, waves 0-3 / 4-7 take plane 0 / 1) — a multiple of the swizzle's
    // period, so every piece of a lane has the same row-in-block and source slot: ONE 32-bit lane offset serves them all, and the
    // piece's plane / row block / K-step go into a scalar base (the vaddr + saddr form of global_load_lds: 1 VGPR instead of 2 TI)
    const uint32_t woff_l = (uint32_t)(((tid >> 2) & (TI == 1 ? 63 : 127)) * Ktot + (((tid & 3) ^ ((tid >> 4) & 3)) << 3)) * 2u;
    const char* wbase[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int q0 = 512 * i + (TI == 1 ? 256 * (wave >> 2) : 0), plane = q0 / (4 * CT), row0 = (q0 - plane * 4 * CT) >> 2;
        // pair mode (two convs on one input in one launch, kernels.h): the rows from a.split on are the second conv's, from its own planes
        const int rowg = ch0 + row0;
        if (a.wgt_b && rowg >= a.split) wbase[i] = (const char*)(a.wgt_b + ((size_t)plane * (a.Cout - a.split) + rowg - a.split) * Ktot);
        else wbase[i] = (const char*)(a.wgt + ((size_t)plane * (a.wgt_b ? a.split : a.Cout) + rowg) * Ktot);
    }
    // fused 1x1 strided shortcut (ConvArgs::in2 / wgt2: the BasicBlock downsample path as Cin2 / 32 extra K-steps behind the conv's own, both BN
    // scales folded into the weight planes by the host): the same pieces from wgt2's planes [2][Cout][Cin2] — another row pitch, so another
    // lane offset; which of the two a K-step uses is a scalar select (SP_SC(): the request state has run past the conv's own K-steps)
    const uint32_t woff2_l = (uint32_t)(((tid >> 2) & (TI == 1 ? 63 : 127)) * a.Cin2 + (((tid & 3) ^ ((tid >> 4) & 3)) << 3)) * 2u;
    const char* wbase2[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int q0 = 512 * i + (TI == 1 ? 256 * (wave >> 2) : 0), plane = q0 / (4 * CT), row0 = (q0 - plane * 4 * CT) >> 2;
        wbase2[i] = a.in2 ? (const char*)(a.wgt2 + ((size_t)plane * a.Cout + ch0 + row0) * a.Cin2) : wbase[i];
    }
#define SP_ISSUE_W1(I, KOFF, ST)                                                                         \
    SP_GLDS16S(SP_SC() ? woff2_l : woff_l, (SP_SC() ? wbase2[I] : wbase[I]) + (SP_ABL_NOW ? 0 : 2 * (KOFF)), (ST) + ((I) * 512 + wave * 64) * 16)

    // ---- input DMA: piece q = tid + 512 i (i < 4) of the slot's image: tile row (pixel) (tid >> 3) + 64 i, 16-byte slot tid & 7 of its
    // 128-byte line (slots 0-3: the heads of the K-step's 32 channels, 4-7: the tails), read from source slot (tid & 7) ^ ((row >> 1) & 7)
    // — 64 i does not move the swizzle, so the source slot is the lane's own.  xorg = address of that slot in the pixel's line at tap
    // (0, 0), chunk 0 — outside the image where the padding says so, never dereferenced there: a K-step adds its wave-uniform offset and
    // fetches from a page of zeros instead when its tap is out of bounds (a select, no branch)
    const _Float16* xorg[4];
    const _Float16* x2org[4];         // (fused shortcut) the pixel's line in in2 at chunk 0, or the page of zeros for a row beyond M
    int iy0[4], ix0[4];
    // PADX: piece i of a wave is the eight tile pixels 8 rb .. 8 rb + 7 (rb = 8 i + wave), one map row segment -> eight consecutive cells from
    // xcell0[i]; the swizzle key of a line is that of its CELL
    const int wlog = 31 - __builtin_clz(a.Wo);
    int xcell0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xcell0[i] = 8 * (8 * i + wave) + ((8 * (8 * i + wave)) >> wlog) + 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 64 * i;
        const int m = pix0 + row;
        const bool vm = m < a.M;
        const int mm = vm ? m : 0;
        const int nc = mm / HoWo, rem = mm - nc * HoWo;
        const int n = map_image<IMAP>(a, nc);               // (IMAP: the tensors' row of compact image nc)
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0[i] = vm ? oy * a.stride - a.pad : -0x10000;      // a row beyond M never passes the bounds test
        ix0[i] = ox * a.stride - a.pad;
        const int key = (tid >> 4) & 7;      // (row >> 1) & 7 of the tile pixel — PADX too: the key follows the PIXEL, not its cell (below)
        xorg[i] = a.in + ((long)(n % a.in_mod) * a.H * a.W + (long)(oy * a.stride - a.pad) * a.W + ix0[i]) * (2 * a.Cin) + (((tid & 7) ^ key) << 3);
        x2org[i] = (a.in2 && vm) ? a.in2 + ((long)(n % a.in2_mod) * a.H2 * a.W2 + (long)(oy * a.stride2) * a.W2 + ox * a.stride2) * (2 * a.Cin2) +
                                       (((tid & 7) ^ key) << 3)
                                 : nullptr;
    }
#define SP_ISSUE_X1(I, KY, KX, C0, ST)                                                                   \
    {                                                                                                    \
        const int kxe_ = SHX ? 1 : (KX);                /* SHX: the centre column, whatever tap of the row asks */                   \
        const bool ok_ = (unsigned)(iy0[I] + (KY)) < (unsigned)a.H && (unsigned)(ix0[I] + kxe_) < (unsigned)a.W;   \
        const long so_ = (long)((KY) * a.W + kxe_) * (2 * a.Cin) + 2 * min((C0), a.Cin - 32);          /* (C0 = Cin: past the last K-step) */ \
        const _Float16* p_ = (ok_ && !SP_ABL_NOX) ? xorg[I] + so_ : (const _Float16*)g_split_zero + ((tid & 7) << 3);                  \
        if (SP_SC()) p_ = x2org[I] ? x2org[I] + 2 * min(c2, a.Cin2 - 32) : (const _Float16*)g_split_zero + ((tid & 7) << 3);           \
        if (!SP_ABL_X3 || (KX) == 0) SP_GLDS16V(p_, (ST) + (PADX ? xcell0[I] * 128 : ((I) * 512 + wave * 64) * 16));   \
    }

    f32x16_s acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int a_off = (wc * 32 * TI + r) * 64;
    const int b_off = XBASE + (wp * 64 + r) * 128;
    // fragments of one 16-deep sub-step SS of a K-step (weight slot WS, input slot XS): (TI + 2) x 2 ds_read_b128
    half8_t ah[TI], al[TI], bh[2], bl[2];
    // SHX: per fragment j, is this lane's pixel on the left / right edge of its map row (the kx = 0 / 2 tap is padding there)
    bool edge_l[2] = {false, false}, edge_r[2] = {false, false};
    int bcell[2] = {0, 0};            // PADX: the cell of this lane's pixel of fragment j
    if constexpr (PADX) {
#pragma unroll
        for (int j = 0; j < 2; ++j) bcell[j] = (wp * 64 + 32 * j + r) + ((wp * 64 + 32 * j + r) >> wlog) + 1;
    }
    if constexpr (SHX == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ox = (pix0 + wp * 64 + 32 * j + r) % a.Wo;
            edge_l[j] = ox == 0;
            edge_r[j] = ox == a.Wo - 1;
        }
    }
    // (SHIFT = kx - 1 of a SHX tap: the slot row of tile pixel p + SHIFT, with THAT row's swizzle key; 0 otherwise)
#define SP_READ(WS, XS, SS, SHIFT)                                                                       \
    if (!SP_ABL_NOREAD || ks == 0) {                                                                     \
        const int coff = ((2 * (SS) + kq) ^ sw) << 4;                                                    \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) {                                                 \
            ah[i] = *(const half8_t*)((WS) + a_off + i * 32 * 64 + coff);                                \
            al[i] = *(const half8_t*)((WS) + WPL + a_off + i * 32 * 64 + coff);                          \
        }                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                  \
            const int cell_ = bcell[j] + (SHIFT);                                                        \
            const int sxs_ = SHX ? (((r + (SHIFT)) >> 1) & 7) : sx;     /* key of tile pixel p + SHIFT (a zero cell reads the same under any key) */ \
            const char* const row_ = PADX ? (XS) + XBASE + cell_ * 128 : (XS) + b_off + (j * 32 + (SHX ? (SHIFT) : 0)) * 128;       \
            bh[j] = *(const half8_t*)(row_ + (((2 * (SS) + kq) ^ sxs_) << 4));                           \
            bl[j] = *(const half8_t*)(row_ + (((4 + 2 * (SS) + kq) ^ sxs_) << 4));                       \
        }                                                                                                \
        if (SHX == 1 && (SHIFT) != 0) {      /* (wave-uniform) */                                        \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                              \
                const bool z_ = (SHIFT) < 0 ? edge_l[j] : edge_r[j];                                     \
                u32x4_sp vh_ = __builtin_bit_cast(u32x4_sp, bh[j]), vl_ = __builtin_bit_cast(u32x4_sp, bl[j]);   \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                          \
                    vh_[e] = z_ ? 0u : vh_[e];                                                           \
                    vl_[e] = z_ ? 0u : vl_[e];                                                           \
                }                                                                                        \
                bh[j] = __builtin_bit_cast(half8_t, vh_);                                                \
                bl[j] = __builtin_bit_cast(half8_t, vl_);                                                \
            }                                                                                            \
        }                                                                                                \
    }
    // the 6 TI MFMAs of a sub-step in NG groups — the small products first (lo . hi, hi . lo), then hi . hi — with HOOK(g) behind group
    // g: DMA issue rides in the issue slots the matrix pipe leaves free (an MFMA holds the SIMD's issue port for 8 of its 32 cycles).
    // The scheduler is fenced around every hook, so the order below is the order issued.
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
    // raw s_barrier with the scheduler fenced off on both sides (a __syncthreads() would drain the DMA in flight: vmcnt(0))
#define SP_BARRIER()                                   \
    {                                                  \
        __builtin_amdgcn_sched_barrier(0);             \
        asm volatile("" ::: "memory");                 \
        if (!SP_ABL_NOBAR) __builtin_amdgcn_s_barrier();   \
        asm volatile("" ::: "memory");                 \
        __builtin_amdgcn_sched_barrier(0);             \
    }

    // split-K (ConvArgs::partial / nsplit; plain convs on small grids): blockIdx.y takes a contiguous range of the K-steps and the epilogue
    // stores the raw fp32 accumulators [split][M][Cout]; splitk_finish_pair_kernel adds them in split order and applies BN / ReLU
    const int nK_all = a.ksize * a.ksize * (a.Cin / 32) + (a.in2 ? a.Cin2 / 32 : 0);
    const int ks_begin = a.partial ? (int)((long)blockIdx.y * nK_all / a.nsplit) : 0;
    const int nK = a.partial ? (int)((long)(blockIdx.y + 1) * nK_all / a.nsplit) - ks_begin : nK_all;
    // (ky, kx, c0) of the K-step whose operands are requested next.  Every step requests (at the end: clamped, unused), so the number of
    // DMA pieces in flight — what the counted vmcnt waits below rely on — never changes
    int ky = 0, kx = 0, c0 = 0, c2 = 0;        // c2: channel offset of the fused shortcut's K-step, once c0 has reached Cin
    if (!SP_TAP_MAJOR && ks_begin) {           // (split-K: the state of K-step ks_begin in the chunk-major order)
        const int kk = a.ksize * a.ksize, tap = ks_begin % kk;
        c0 = (ks_begin / kk) * 32;
        ky = tap / a.ksize;
        kx = tap - ky * a.ksize;
    }
#define SP_SC() (a.in2 != nullptr && c0 >= a.Cin)
    auto advance = [&]() {
        const bool past = c0 >= a.Cin;          // the conv's own K-steps are all requested: the shortcut's chunks follow (or clamped, unused requests)
        c2 += past ? 32 : 0;            // scalar selects, no branch; runs past the last K-step (c0 = Cin) at the end: the requests of
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
        // Cin / 32 K-steps later, behind 32 KB x Cin / 32 of other input per workgroup — beyond an XCD's L2 share
        const int kx1 = kx + (past ? 0 : 1);
        const bool wrapx = kx1 == a.ksize;
        kx = wrapx ? 0 : kx1;
        const int ky1 = ky + (wrapx ? 1 : 0);
        const bool wrapy = ky1 == a.ksize;
        ky = wrapy ? 0 : ky1;
        c0 += wrapy ? 32 : 0;
#endif
    };
    // (the weight offset of (ky, kx, c0), clamped for the K-steps past the end)
#define SP_KOFF() (SP_SC() ? min(c2, a.Cin2 - 32) : min((min(ky, a.ksize - 1) * a.ksize + kx) * a.Cin + min(c0, a.Cin - 32), Ktot - 32))
    // all of one K-step's pieces: TI of weights, 4 of input
    // SHX: the pixel slots are a ring of their own — filled for the first tap of a tap row (and for every K-step of a fused shortcut), so the
    // request state says whether this request carries pixel pieces (SP_NEEDX) and xq is the slot they go to
    int xq = 0, xc = 0;                                  // pixel slot of the next fill | of the K-step being computed
#define SP_NEEDX() (!SHX || SP_SC() || kx == 0)
#define SP_XDST(SLOT) (smem + XBASE + (SHX ? xq : (SLOT)) * XSLOT)
#define SP_ISSUE_ALL(SLOT, NEEDX_)                                                                       \
    {                                                                                                    \
        const int koff_ = SP_KOFF();                                                                     \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) SP_ISSUE_W1(i, koff_, smem + (SLOT) * WSLOT);     \
        if (NEEDX_) { _Pragma("unroll") for (int i = 0; i < 4; ++i) SP_ISSUE_X1(i, ky, kx, c0, SP_XDST(SLOT)); }   \
    }
    // (after a request: the fill slot moves on if the request carried pixel pieces; then the request state)
#define SP_NEXT_REQUEST(NEEDX_)                                                                          \
    {                                                                                                    \
        if (SHX && (NEEDX_)) xq = xq + 1 == NS ? 0 : xq + 1;                                             \
        advance();                                                                                       \
    }
    constexpr int PIECES = TI + 4;                       // DMA pieces a wave issues per K-step
    if constexpr (PADX) {             // the zero cells of every slot: cell 0 and the cell in front of every further map row + the one behind the last
        const int npad = (SP_PT >> wlog) + 1;
        for (int idx = tid; idx < NS * npad * 8; idx += 512) {
            const int sl = idx / (npad * 8), rem = idx - sl * npad * 8, k = rem >> 3;
            *(u32x4_sp*)(smem + XBASE + sl * XSLOT + k * (a.Wo + 1) * 128 + (rem & 7) * 16) = u32x4_sp{0u, 0u, 0u, 0u};
        }
    }
    // K-steps 0 .. D-1 -> slots 0 .. D-1
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const bool nx = SP_NEEDX();
        SP_ISSUE_ALL(d, nx);
        SP_NEXT_REQUEST(nx);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    // ---- ping-pong main loop (the schedule of conv_igemm_wide.hip) ----------------------------------------------------------------
    // A K-step is four intervals between barriers: LOAD(0) | MFMA(0) | LOAD(1) | MFMA(1), one 16-deep sub-step each.  A LOAD part is the
    // sub-step's fragment reads; an MFMA part 6 TI MFMAs.  The DMA requests of K-step ks + D are issued in LOAD(0) (D = 1) or behind the
    // MFMA groups of MFMA(0) (D = 2).  The two wave groups (waves 0-3 / 4-7 = the two channel halves: one wave per SIMD each) run ONE
    // BARRIER APART, so on every SIMD one wave is in an MFMA part while the other reads LDS: the matrix pipe does not wait for a
    // fragment, and a barrier is the hand-over between the two waves of a SIMD.
    // Intervals, K-step T: group 0 = 4T .. 4T+3, group 1 = 4T+1 .. 4T+4.  Hazards, by construction:
    //   WAR  the slot K-step T+D will use (index (T+D) % (D+1) = (T-1) % (D+1)) was last read for K-step T-1, in intervals 4T-2 (group 0)
    //        and 4T-1 (group 1); every LOAD part retires its reads (lgkmcnt(0)) before the barrier that ends it; the first request
    //        into it is group 0's in interval 4T.
    //   RAW  every wave retires the requests it issued for K-step T+1 — all but the (D-1) PIECES youngest of its DMA — before the barrier
    //        that ends its LOAD(1) of K-step T, i.e. by 4T+3; the first reads of K-step T+1 are in 4T+4 (group 0) and 4T+5 (group 1).
    // Both groups execute the same number of barriers: group 1 one extra before the loop, group 0 one extra after it.
    const int g = wc;
    if (g == 1) SP_BARRIER();
    int slot = 0;                                        // ring index of K-step ks
    const int nK_own = a.ksize * a.ksize * (a.Cin / 32);  // the conv's own K-steps (then the fused shortcut's)
    // One K-step.  SHIFT_ = kx - 1 of a shared-slot tap (0: a straight read), NEEDX_ = the request issued during this K-step (for K-step ks + D)
    // carries pixel pieces.  With literal arguments everything they decide folds at compile time (the unrolled tap rows below).
#define SP_HOOK0(G)                                                                                      \
    {                                                                                                    \
        if constexpr (D == 2) {          /* two groups: half of the K-step's pieces behind each */       \
            const int koff_ = SP_KOFF();                                                                 \
            if ((G) == 0) { if (needx_) { SP_ISSUE_X1(0, ky, kx, c0, SP_XDST(nslot)); SP_ISSUE_X1(1, ky, kx, c0, SP_XDST(nslot)); } SP_ISSUE_W1(0, koff_, smem + nslot * WSLOT); } \
            if ((G) == 1) { if (needx_) { SP_ISSUE_X1(2, ky, kx, c0, SP_XDST(nslot)); SP_ISSUE_X1(3, ky, kx, c0, SP_XDST(nslot)); } \
                            if constexpr (TI == 2) { SP_ISSUE_W1(TI - 1, koff_, smem + nslot * WSLOT); } } \
        }                                                                                                \
    }
#define SP_HOOK1(G) {}
#define SP_STEP(SHIFT_, NEEDX_)                                                                          \
    {                                                                                                    \
        const char* const ws = smem + slot * WSLOT;                                                      \
        const char* const xs = smem + (SHX ? xc : slot) * XSLOT;      /* (+ XBASE inside b_off) */       \
        const int nslot = slot + D >= NS ? slot + D - NS : slot + D;  /* ring index of K-step ks + D */  \
        const bool needx_ = (NEEDX_);                                                                    \
        /* LOAD(0) */                                                                                    \
        SP_READ(ws, xs, 0, SHIFT_);                                                                      \
        if constexpr (D == 1) { SP_ISSUE_ALL(nslot, needx_); }                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        SP_BARRIER();                                                                                    \
        /* MFMA(0) */                                                                                    \
        SP_MFMA(SP_HOOK0);                                                                               \
        SP_NEXT_REQUEST(needx_);                                                                         \
        SP_BARRIER();                                                                                    \
        /* LOAD(1) */                                                                                    \
        SP_READ(ws, xs, 1, SHIFT_);                                                                      \
        /* K-step ks + 1's operands have landed (this wave's pieces): all but the pieces of the request issued during this K-step (D = 2) */ \
        if (SHX && !needx_) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * TI) : "memory");         \
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * PIECES) : "memory");                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
        SP_BARRIER();                                                                                    \
        /* MFMA(1) */                                                                                    \
        SP_MFMA(SP_HOOK1);                                                                               \
        SP_BARRIER();                                                                                    \
        slot = slot + 1 == NS ? 0 : slot + 1;                                                            \
    }
    int ks0 = 0;                                         // K-steps done by the unrolled rows
    if constexpr (SHX != 0) {
        // Whole tap rows, the three taps unrolled — which tap reads the slot shifted which way and which K-step's request
        // carries the next row's pixel pieces (the one D K-steps ahead of a row's first tap) are compile-time, the loop is ONE basic block again
        // (with run-time decisions it was 124 scalar and 8 lane-spill instructions and 9 branches per K-step against 72 / 0 / 1, and kept 4-10 %
        // of the 16-24 % the probe promised).  Requests past the last K-step follow the same pattern (clamped addresses, never read).  With a
        // fused shortcut the last tap row and the shortcut's K-steps (a slot each) stay with the run-time loop below.
        if (!SP_ABL_SHIFT0) {
            const int rows = nK_own / 3 - (a.in2 ? 1 : 0);
            ks0 = 3 * rows;
            for (int row = 0; row < rows; ++row) {
                const int ks = row;                              // (SP_READ's probe switch only)
                SP_STEP(-1, false);                              // kx = 0: requests kx = 1 (D = 1) | kx = 2 (D = 2) of this row
                SP_STEP(0, D == 2);                              // kx = 1: requests kx = 2 | the next row's first tap
                SP_STEP(1, D == 1);                              // kx = 2: requests the next row's first tap | its second
                xc = xc + 1 == NS ? 0 : xc + 1;
            }
        }
    }
    {
        int ckx = 0;                                     // SHX: kx of K-step ks (taps inner, three per row)
        for (int ks = ks0; ks < nK; ++ks) {
            const bool own = ks < nK_own;                // (SHX: a shortcut K-step reads its own slot straight)
            const int shift = SP_ABL_SHIFT0 ? 0 : (SHX && own ? ckx - 1 : 0);
            const bool needx = SP_NEEDX();               // does the request issued during this K-step carry pixel pieces
            SP_STEP(shift, needx);
            if constexpr (SHX != 0) {                    // the pixel slot moves on behind the third tap of a row (behind every shortcut K-step)
                if (!own || ckx == 2) xc = xc + 1 == NS ? 0 : xc + 1;
                ckx = ckx == 2 ? 0 : ckx + 1;
            }
        }
    }
#undef SP_STEP
#undef SP_HOOK0
#undef SP_HOOK1
    if (g == 0) SP_BARRIER();
#undef SP_ISSUE_W1
#undef SP_ISSUE_X1
#undef SP_ISSUE_ALL
#undef SP_NEEDX
#undef SP_XDST
#undef SP_NEXT_REQUEST
#undef SP_READ
#undef SP_MFMA
#undef SP_BARRIER
#undef SP_KOFF
#undef SP_SC

    // ---- epilogue, coalesced through LDS (the slots are dead: every wave is behind the loop's last barrier) ---------------------------
    // In two rounds (pixel tile j = 0, 1 of every wave: 128 of the tile's pixels x all CT channels = CT / 2 KB of fp32): the raw
    // accumulators go to LDS as [pixel][channel] rows (16-byte chunk c of pixel row p at c ^ (p & 31 & (chunks - 1)): conflict-free both
    // ways), then a lane takes 8 consecutive channels of a row — folded BN, site, residual (decoded from its pair32 tensor), ReLU on
    // the two quads (epilogue_quad_f32v: arithmetic and order of every other conv kernel) — and ENCODES them into the pair32 output:
    // consecutive lanes write consecutive 16 bytes of heads and of tails, whole lines to and from HBM.  The second phase is a ROLLED
    // loop (it does not touch the accumulators, so nothing lands in scratch; the site arithmetic is compiled once).
    constexpr int CHUNKS = CT / 4;                             // 16-byte chunks per fp32 pixel row
    constexpr int ROWB = CT * 4;                               // bytes per row
    static_assert(128 * ROWB <= (int)sizeof(smem), "one round's tile fits the dead slots");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the last K-steps' (unused) DMA must land BEFORE the rows overwrite its slot
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
        for (int it = 0; it < 128 * (CT / 8) / 512; ++it) {
            const int qi = tid + 512 * it, pl = qi / (CT / 8), c8l = (qi - pl * (CT / 8)) * 8;
            const int m_o = pix0 + (pl >> 5) * 64 + 32 * j + (pl & 31);
            if (m_o >= a.M) continue;
            const int x_ = pl & 31 & (CHUNKS - 1);
            const f32x4_s r0 = *(const f32x4_s*)(smem + pl * ROWB + (((c8l >> 2) ^ x_) << 4));
            const f32x4_s r1 = *(const f32x4_s*)(smem + pl * ROWB + ((((c8l >> 2) + 1) ^ x_) << 4));
            const int nc = m_o / HoWo, rem = m_o - nc * HoWo;
            const int n = map_image<IMAP>(a, nc);
            PixelCtx p;
            p.out_off = ((size_t)n * HoWo + rem) * a.Cout;
            p.resp = nullptr;
            const int tl = n / a.B;
            p.b = n - tl * a.B;
            p.t = a.t0 + tl;
            p.e_pix = p.b * HoWo + rem;
            p.mrow = a.site.kind == BMI_SITE_MASKSEMBLE ? a.site.masks + (size_t)((a.site.cnt0 + p.t) % a.site.num_masks) * a.Cout : nullptr;
            const int c8 = ch0 + c8l;
            if (a.partial) {                                   // split-K: the raw sums of this K range
                float* const pp = a.partial + ((size_t)blockIdx.y * a.M + m_o) * a.Cout + c8;
                *(f32x4_s*)pp = r0;
                *(f32x4_s*)(pp + 4) = r1;
                continue;
            }
            // (two separate quads, not halves of one array: pointer arithmetic on a local array sends it to scratch)
            float ra[4] = {0.f, 0.f, 0.f, 0.f}, rb[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.res) {
                const _Float16* rp = a.res + pair32_off((size_t)(n % a.res_mod) * HoWo + rem, a.Cout, c8);
                pair_decode<BF, 4>(rp, ra);
                pair_decode<BF, 4>(rp + 4, rb);
            }
            float va[4] = {r0[0], r0[1], r0[2], r0[3]}, vb[4] = {r1[0], r1[1], r1[2], r1[3]};
            // pair mode: channels from a.split on belong to the second conv (its BN vectors moved back by the split, its own output tensor)
            const bool second = a.wgt_b && c8 >= a.split;
            const float* const sc = second ? a.scale_b - a.split : a.scale;
            const float* const bi = second ? a.bias_b - a.split : a.bias;
            epilogue_quad_f32sb(a, sc, bi, p, ra, a.res != nullptr, va, c8);
            epilogue_quad_f32sb(a, sc, bi, p, rb, a.res != nullptr, vb, c8 + 4);
            _Float16* op = second ? a.out_b + pair32_off((size_t)n * HoWo + rem, a.Cout - a.split, c8 - a.split)
                                  : a.out + pair32_off((size_t)n * HoWo + rem, a.wgt_b ? a.split : a.Cout, c8);
            const float v8[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
            pair_encode<BF, 8>(op, v8);
        }
    };
    round(std::integral_constant<int, 0>{});
    round(std::integral_constant<int, 1>{});
}

// out (pair32) = encode(relu?(bn(sum over the splits, in split order))): conv_split's epilogue arithmetic on the added partial sums
template <bool BF>
__global__ __launch_bounds__(256) void splitk_finish_pair_kernel(ConvArgs a) {
    const long total = (long)a.M * (a.Cout >> 3);
    const int HoWo = a.Ho * a.Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % (a.Cout >> 3)) * 8;
        const long m = i / (a.Cout >> 3);
        const float* pp = a.partial + (size_t)m * a.Cout + c8;
        f32x4_s s0 = *(const f32x4_s*)pp, s1 = *(const f32x4_s*)(pp + 4);
        for (int sp = 1; sp < a.nsplit; ++sp) {
            const float* q = pp + (size_t)sp * a.M * a.Cout;
            s0 += *(const f32x4_s*)q;
            s1 += *(const f32x4_s*)(q + 4);
        }
        PixelCtx p;
        p.out_off = 0; p.resp = nullptr; p.b = 0; p.t = 0; p.e_pix = 0; p.mrow = nullptr;       // (plain epilogue: no site, no residual)
        float va[4] = {s0[0], s0[1], s0[2], s0[3]}, vb[4] = {s1[0], s1[1], s1[2], s1[3]};
        const float zero[4] = {0.f, 0.f, 0.f, 0.f};
        epilogue_quad_f32sb(a, a.scale, a.bias, p, zero, false, va, c8);
        epilogue_quad_f32sb(a, a.scale, a.bias, p, zero, false, vb, c8 + 4);
        const float v8[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
        pair_encode<BF, 8>(a.out + pair32_off((size_t)m, a.Cout, c8), v8);
    }
    (void)HoWo;
}

template <bool BF>
static int launch_split_t(const ConvArgs& a, hipStream_t s) {
    // Channel tile: the widest that still fills the chip.  A 256-channel tile shares a pixel tile among the most MFMAs, but the B-image prefix
    // launches (VGG-11: 250 images of 8x8 ... 2x2 maps) are a few dozen workgroups of 72-144 K-steps each with it; narrower tiles give 2-4x
    // the workgroups at the same K order per accumulator (the same bits: tests/test_split_engine.py).  Decided on the engine's full-chunk image
    // count (n_ref), never on this launch's, like every other kernel selection; a pair launch keeps the 256-channel tile it was merged for.
    static const int n_cu = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0;
        return cu > 0 ? cu : 256;
    }();
    const long ptiles_sel = ((long)(a.n_ref > 0 ? a.n_ref : a.N) * a.Ho * a.Wo + SP_PT - 1) / SP_PT;
    int ct = a.Cout % 256 == 0 ? 256 : (a.Cout % 128 == 0 ? 128 : 64);
    while (!a.wgt_b && opt_split_tile() && ct > 64 && ptiles_sel * (a.Cout / ct) < 3L * n_cu / 4) ct >>= 1;     // (250 workgroups of 128 channels beat 500 of 64)
    const long blocks = (long)((a.M + SP_PT - 1) / SP_PT) * (a.Cout / ct);
    if (blocks <= 0 || blocks > 0x7fffffffL) return BMI_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)blocks, a.partial ? (unsigned)a.nsplit : 1u), block(512);
    // the three taps of a tap row share a pixel slot (SHX): 3x3, stride 1, pad 1, whole K range per workgroup
    // (a tile must start and end on a map row boundary, so that a pixel's horizontal neighbours are in its own tile: 256 % Wo == 0)
    const bool shx = opt_split_shx() && !SP_TAP_MAJOR && a.ksize == 3 && a.stride == 1 && a.pad == 1 && !a.partial && SP_PT % a.Wo == 0;
    const bool padx = shx && a.Wo % 8 == 0 && opt_split_shx() != 2;      // ("split_shx" = 2: the clearing form everywhere: A/B, tests)
#define SP_LAUNCH_I(TI_, IM_)                                                                                    \
    {                                                                                                            \
        if (padx) hipLaunchKernelGGL((conv_split_kernel<BF, TI_, 2, IM_>), grid, block, 0, s, a);                \
        else if (shx) hipLaunchKernelGGL((conv_split_kernel<BF, TI_, 1, IM_>), grid, block, 0, s, a);            \
        else hipLaunchKernelGGL((conv_split_kernel<BF, TI_, 0, IM_>), grid, block, 0, s, a);                     \
    }
#define SP_LAUNCH(TI_)                                                                                           \
    {                                                                                                            \
        if (a.imap) SP_LAUNCH_I(TI_, true) else SP_LAUNCH_I(TI_, false)                                          \
    }
    if (ct == 256) SP_LAUNCH(4) else if (ct == 128) SP_LAUNCH(2) else SP_LAUNCH(1)
#undef SP_LAUNCH
#undef SP_LAUNCH_I
    BMI_CHECK_LAUNCH();
    if (a.partial) {
        const long total = (long)a.M * (a.Cout >> 3);
        long fb = (total + 255) / 256;
        if (fb > 256 * 8) fb = 256 * 8;
        hipLaunchKernelGGL(splitk_finish_pair_kernel<BF>, dim3((unsigned)fb), dim3(256), 0, s, a);
        BMI_CHECK_LAUNCH();
    }
    return BMI_OK;
}

bool conv_takes_split_kernel(int cin, int cout) { return cin % 32 == 0 && cout % 64 == 0; }

// a.in / a.res / a.out: pair32 tensors (conv_epilogue.h); a.wgt: 16-bit [2][Cout][k*k*Cin] head / tail planes
int launch_conv_split(const ConvArgs& a, int bf16, hipStream_t s) {
    if (!conv_takes_split_kernel(a.Cin, a.Cout)) return BMI_ERR_UNSUPPORTED;
    if (a.in_bits || a.in2_bits || a.pool || a.pool_b || (a.imap && a.partial)) return BMI_ERR_UNSUPPORTED;
    // fused 1x1 shortcut: in2 a pair32 tensor of Cin2 channels read at stride2 (no padding), wgt2 the planes [2][Cout][Cin2]
    if (a.in2 && (!a.wgt2 || a.wgt_b || a.Cin2 % 32 != 0 || a.Cin2 <= 0 || a.stride2 < 1 || a.in2_mod <= 0 || (a.Ho - 1) * a.stride2 >= a.H2 ||
                  (a.Wo - 1) * a.stride2 >= a.W2))
        return BMI_ERR_UNSUPPORTED;
    // pair mode: both convs plain (BN + ReLU), the split on a 128-row block of the weight DMA, both outputs whole 32-channel blocks
    if (a.wgt_b && (!a.out_b || a.split <= 0 || a.split >= a.Cout || a.split % 128 != 0 || (a.Cout - a.split) % 128 != 0 || a.res || a.site.kind != BMI_SITE_NONE ||
                    !a.scale || !a.bias || !a.scale_b || !a.bias_b))
        return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.M <= 0 || a.in_mod <= 0 || a.B <= 0 || (a.res && a.res_mod <= 0)) return BMI_ERR_INVALID;
    // split-K: plain convs only (BN + ReLU in the finishing pass), K ranges of at least one K-step
    if (a.partial && (a.nsplit < 2 || a.nsplit > a.ksize * a.ksize * (a.Cin / 32) || a.res || a.in2 || a.wgt_b || a.site.kind != BMI_SITE_NONE || SP_TAP_MAJOR))
        return BMI_ERR_UNSUPPORTED;
    return bf16 ? launch_split_t<true>(a, s) : launch_split_t<false>(a, s);
}
