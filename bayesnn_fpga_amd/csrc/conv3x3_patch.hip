// 3x3 / pad-1 convolution with the INPUT PATCH resident in LDS (gfx950, fp16 -> fp32 accumulate).
//
// Why: in the per-tap implicit GEMM (conv_igemm.hip) every K-step re-stages the activation tile,
// so LDS write traffic equals LDS read traffic equals MFMA time and the kernel is LDS-bound.  A
// 3x3 stride-1 conv reads each input pixel for 9 taps: here a workgroup stages the input patch of
// its output tile (with halo) ONCE per 64-channel chunk and runs all 9 taps out of LDS by
// shifting the per-lane fragment address; only the 16 KB weight tile streams per K-step.
//
//   work-group  = IMGS images x (TH x TW) output pixels (BP = 64*TJ pixels) x 128 output channels
//   4 waves     = 2 (channels) x 2 (pixels); wave tile 64 channels x 32*TJ pixels of 32x32x16 MFMAs
//   LDS         = patch [cells][64 ch] fp16 (128 B per cell) + 2 x weight tile [128][64]
//   staging     = patch: global_load_lds_dwordx4 (LDS-DMA), no VGPR round trip, no ds_write; padding and
//                 halo cells are DMA'd from a zero page, so there is no bounds logic in the loop.
//                 weights: LDS-DMA one K-step ahead (or register-staged, BMI_PATCH_WDMA=0)
//   swizzle     = 16-byte chunk c of a cell is stored at chunk (c ^ ((key >> 1) & 7)) with
//                 key = patch_x + KA * patch_y chosen per tile shape so that the 16 lanes of every
//                 ds_read_b128 group hit 16 distinct 16-byte slots; LDS-DMA writes linearly, so the
//                 permutation is applied to the per-lane SOURCE address (cdna guide rule 21)
//   pipeline    = per K-step (tap, chunk): wait DMA + barrier, issue next weight tile, MFMA.
//   epilogue    = coalesced through LDS (conv_epilogue.h: epilogue_coalesced).
//
// Orientation, fragment layouts and the fused epilogue are those of conv_igemm.hip
// (channels on the MFMA row axis -> one Philox call / one 8-byte store per accumulator quad).
// Reference semantics: BasicBlock.forward SA/models/resnet18/resnet18.py:32-48.
#include <cstdlib>

#include "conv_epilogue.h"
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifdef BMI_PATCH_STAMPS
// Diagnostic build only (tools/ab_build.py stamps:-DBMI_PATCH_STAMPS): per-workgroup phase timestamps of
// wave 0, read back with bmi_debug_stamps().  No stamp executes in the product build.
__device__ unsigned long long g_stamps[8192 * 8];
#define STAMP(SLOT)                                                                                  \
    if (tid == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (SLOT)] = __builtin_readcyclecounter();
#define STAMP_ADD(SLOT, T0)                                                                          \
    if (tid == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (SLOT)] += __builtin_readcyclecounter() - (T0);
extern "C" int bmi_debug_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}
extern "C" int bmi_debug_stamps_clear() {
    static unsigned long long z[8192 * 8];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) == hipSuccess ? 0 : -5;
}
#else
#define STAMP(SLOT)
#define STAMP_ADD(SLOT, T0)
#endif

__device__ unsigned int g_zero_page[64];  // 256 B of zeros: DMA source for padding / halo / tail cells

#define GLDS16(SRC, LDSPTR)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),       \
                                     (__attribute__((address_space(3))) void*)(LDSPTR), 16, 0, 0)

template <int S, int TH, int TW, int IMGS, int TJ, int NB = 2>
struct PatchGeom {
    static constexpr int BC = 128, TI = 2;
    static constexpr int BP = IMGS * TH * TW;
    static_assert(BP == 64 * TJ, "pixel tile must be 2 waves x TJ x 32");
    static constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
    static constexpr int HALF = S == 1 ? 0 : (((PW + 1) / 2 + 1) & ~1);        // even
    static constexpr int PWP = S == 1 ? ((PW + 1) & ~1) : 2 * HALF;            // even pitch (cells)
    static constexpr int KA = TW >= 16 ? 0 : (TW == 8 ? 8 : 4);
    static constexpr int CELLS = IMGS * PH * PWP;
    static constexpr int PIECES = CELLS * 8;
    static constexpr int ITER_P = (PIECES + 255) / 256;
    static constexpr int PATCH_BYTES = ITER_P * 256 * 16;
    static constexpr int WTILE = BC * 128;
    static constexpr int MAIN_BYTES = PATCH_BYTES + NB * WTILE;
    static constexpr int LDS_BYTES = MAIN_BYTES > BMI_EPILOGUE_LDS_BYTES ? MAIN_BYTES : BMI_EPILOGUE_LDS_BYTES;
};

#ifndef BMI_PATCH_WDMA
#define BMI_PATCH_WDMA 1
#endif
#ifndef BMI_PATCH_NB_S3
#define BMI_PATCH_NB_S3 2
#endif
#ifndef BMI_PATCH_SWPIPE
#define BMI_PATCH_SWPIPE 0
#endif
#ifndef BMI_PATCH_SETPRIO   // raise the wave's priority while it issues a k-substep's MFMAs (the other workgroup's wave loads meanwhile)
#define BMI_PATCH_SETPRIO 0
#endif
#ifndef BMI_PATCH_RDEARLY   // fragment reads of k-substep kk+1 right behind the FIRST MFMA of kk (hipcc puts them behind the last)
#define BMI_PATCH_RDEARLY 0
#endif
#ifndef BMI_PATCH_WSPREAD   // issue the next weight tile's 4 DMA pieces one per k-substep, behind its MFMAs (not as a burst)
#define BMI_PATCH_WSPREAD 0
#endif

template <int S, int TH, int TW, int IMGS, int TJ, int NB, bool PLAIN>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(ConvArgs a) {
    constexpr bool WDMA = BMI_PATCH_WDMA != 0;
    static_assert(NB == 2 || WDMA, "deeper weight prefetch is implemented for the LDS-DMA path");
    using G = PatchGeom<S, TH, TW, IMGS, TJ, NB>;
    constexpr int BC = G::BC, TI = G::TI, PH = G::PH, PW = G::PW, PWP = G::PWP, HALF = G::HALF, KA = G::KA;
    __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];
    char* const patch = smem;
    char* const wbuf = smem + G::PATCH_BYTES;

    const int tid = threadIdx.x;
    STAMP(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wc = wave >> 1, wp = wave & 1;

    // ---- tile coordinates (channel tile fastest) ------------------------------------------------
    const int n_ctiles = a.Cout / BC;
    const int tiles_x = a.Wo / TW, tiles_y = a.Ho / TH;
    int bid, ctile;
    xcd_tile_map(blockIdx.x, (int)(gridDim.x / n_ctiles), n_ctiles, bid, ctile);
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n0 = (bid / tiles_y) * IMGS;
    const int ch0 = ctile * BC;
    const int y0 = ty * TH, x0 = tx * TW;
    const int Ktot = 9 * a.Cin;

    // ---- per-thread DMA sources ----------------------------------------------------------------
    // piece q = tid + 256*i  ->  LDS slot (cell = q >> 3, physical chunk = q & 7)
    // element offset of the piece's source at chunk 0 (relative to a.in), or -1 -> zero page
    int psrc[G::ITER_P];
#pragma unroll
    for (int i = 0; i < G::ITER_P; ++i) {
        const int q = tid + 256 * i;
        const int cell = q >> 3, cp = q & 7;
        const int rowc = cell / PWP, col = cell - rowc * PWP;
        const int img = rowc / PH, py = rowc - img * PH;
        int px, key;
        if (S == 1) { px = col; key = px + KA * py; }
        else { const int par = col / HALF, hx = col - par * HALF; px = 2 * hx + par; key = hx + KA * py; }
        const int c = cp ^ ((key >> 1) & 7);
        const int n = n0 + img;
        const int iy = y0 * S - 1 + py, ix = x0 * S - 1 + px;
        const bool ok = cell < G::CELLS && px < PW && n < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        psrc[i] = ok ? (int)((((size_t)(n % a.in_mod) * a.H + iy) * a.W + ix) * a.Cin + c * 8) : -1;
    }
    // weight tile, build-time variants (same-box A/B on S2/S3/S4, tools/ab_run.sh): LDS-DMA one K-step
    // ahead with plain per-substep fragment reads is the default (910 TF/s); register staging
    // (BMI_PATCH_WDMA=0) -10 %, hand-pipelined fragment reads (BMI_PATCH_SWPIPE=1) -3 %, a third
    // weight buffer with counted vmcnt (NB=3) -4 % on S3: per-phase stamps (tools/stamps.py) show the
    // weight-DMA wait is only ~70 cycles per K-step; the losses are the ~680-cycle barrier skew per
    // K-step and the prologue / epilogue phases.
    const int w_row = tid >> 3;
    const int w_sw = (w_row >> 1) & 7;
    const _Float16* wsrc = a.wgt + (size_t)(ch0 + w_row) * Ktot + (WDMA ? ((tid & 7) ^ w_sw) : (tid & 7)) * 8;
    const int w_st = w_row * 128 + (((tid & 7) ^ w_sw) << 4);
    u32x4 wreg[4];

#define ISSUE_PATCH(C0)                                                                           \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < G::ITER_P; ++i)                                     \
            GLDS16(psrc[i] >= 0 ? a.in + (size_t)(unsigned)psrc[i] + (C0) : (const _Float16*)g_zero_page,   \
                   patch + (i * 256 + wave * 64) * 16);                                           \
    }
#define LOAD_W(KOFF, BUF)                                                                         \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
            if constexpr (WDMA)                                                                   \
                GLDS16(wsrc + (size_t)(32 * i) * Ktot + (KOFF),                                   \
                       wbuf + (BUF) * G::WTILE + (i * 256 + wave * 64) * 16);                     \
            else                                                                                  \
                wreg[i] = *(const u32x4*)(wsrc + (size_t)(32 * i) * Ktot + (KOFF));               \
        }                                                                                         \
    }
#define STORE_W(BUF)                                                                              \
    {                                                                                             \
        if constexpr (!WDMA) {                                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
                *(u32x4*)(wbuf + (BUF) * G::WTILE + w_st + i * 32 * 128) = wreg[i];               \
        }                                                                                         \
    }

    // ---- per-lane fragment geometry --------------------------------------------------------------
    int bcell[TJ], bkey[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int p = wp * (32 * TJ) + 32 * j + r;
        const int img = p / (TH * TW), rem = p - img * (TH * TW);
        const int oy = rem / TW, ox = rem - oy * TW;
        bcell[j] = (img * PH + oy * S) * PWP + (S == 1 ? ox : ox);   // tap (0,0); S=2: column part added per tap
        bkey[j] = (S == 1 ? ox : ox) + KA * (oy * S);
    }
    const int a_off = (wc * 64 + r) * 128;
    const int a_sw = (r >> 1) & 7;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nchunks = a.Cin / 64;
    const int nK = 9 * nchunks;
    // K-step s = chunk * 9 + tap uses weight buffer s % NB.  The weight tiles of steps s+1 .. s+NB-1
    // are in flight while step s computes (L2 latency under load is about one whole K-step, so
    // NB = 3 keeps two tiles in flight and waits with a COUNTED vmcnt).
    auto w_koff = [&](int st) { const int c = st / 9, t = st - 9 * c; return t * a.Cin + c * 64; };
    ISSUE_PATCH(0);
    LOAD_W(0, 0);
    STORE_W(0);
    if constexpr (NB == 3) {
        if (nK > 1) LOAD_W(w_koff(1), 1);
    }
    int step = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        for (int tap = 0; tap < 9; ++tap, ++step) {
            const int buf = step % NB;
            if constexpr (NB == 3) {
                // all but the youngest weight tile (4 DMA instructions per wave) must have landed;
                // at a chunk start the patch DMA (issued last) must have landed too
                if (tap == 0 || step + 1 >= nK) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                lds_barrier();
                if (step + 2 < nK) LOAD_W(w_koff(step + 2), (step + 2) % NB);
            } else {
#ifdef BMI_PATCH_STAMPS
                const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                STAMP_ADD(4, tw0);
                __syncthreads();   // patch landed, W[step&1] written; every wave is done with W[(step+1)&1]
                STAMP_ADD(5, tw0);
                if (step == 0) { STAMP(1); }
                if (!(BMI_PATCH_WSPREAD && WDMA) && step + 1 < nK) LOAD_W(w_koff(step + 1), buf ^ 1);
            }
            const bool more = step + 1 < nK;
            const int ky = tap / 3, kx = tap - 3 * ky;
            // per-tap cell shift and swizzle key shift
            int coff, koff;
            if (S == 1) { coff = ky * PWP + kx; koff = kx + KA * ky; }
            else { coff = ky * PWP + (kx & 1) * HALF + (kx >> 1); koff = (kx >> 1) + KA * ky; }
            const char* wt = wbuf + buf * G::WTILE;
            int boff[TJ], bsw[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                boff[j] = (bcell[j] + coff) * 128;
                bsw[j] = ((bkey[j] + koff) >> 1) & 7;
            }
#if BMI_PATCH_SWPIPE
            // Fragment reads are software-pipelined by hand: the reads for k-substep kk+1 are issued
            // between the MFMAs of kk, each into the registers whose last MFMA use has just been issued,
            // so the LDS latency hides under the remaining MFMAs instead of stalling every substep.
            half8 af[TI], bf[TJ];
#define RD_A(I, KK) af[I] = *(const half8*)(wt + a_off + (I) * 32 * 128 + (((2 * (KK) + hh) ^ a_sw) << 4))
#define RD_B(J, KK) bf[J] = *(const half8*)(patch + boff[J] + (((2 * (KK) + hh) ^ bsw[J]) << 4))
#pragma unroll
            for (int i = 0; i < TI; ++i) RD_A(i, 0);
#pragma unroll
            for (int j = 0; j < TJ; ++j) RD_B(j, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[j], acc[0][j], 0, 0, 0);
                if (kk < 3) RD_A(0, kk + 1);
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1], bf[j], acc[1][j], 0, 0, 0);
                    if (kk < 3) RD_B(j, kk + 1);
                }
                if (kk < 3) RD_A(1, kk + 1);
            }
#undef RD_A
#undef RD_B
            // pin the interleave (LLVM sched groups: 0x008 = MFMA, 0x100 = DS read)
            __builtin_amdgcn_sched_group_barrier(0x100, TI + TJ, 0);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                __builtin_amdgcn_sched_group_barrier(0x008, TJ, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TJ, 0);
#elif BMI_PATCH_RDEARLY
            {
                half8 fa[2][TI], fb[2][TJ];
#define RDF(KK, SET)                                                                                              \
    {                                                                                                             \
        const int ch_ = 2 * (KK) + hh;                                                                            \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) fa[SET][i] = *(const half8*)(wt + a_off + i * 32 * 128 + ((ch_ ^ a_sw) << 4)); \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j) fb[SET][j] = *(const half8*)(patch + boff[j] + ((ch_ ^ bsw[j]) << 4));      \
    }
                RDF(0, 0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int cur = kk & 1;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
                    if (kk < 3) RDF(kk + 1, cur ^ 1);
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int j = 0; j < TJ; ++j)
                            if (i + j > 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
                }
#undef RDF
                // pin: per substep 1 MFMA, then the next substep's reads, then the remaining MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, TI + TJ, 0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (kk < 3) __builtin_amdgcn_sched_group_barrier(0x100, TI + TJ, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, TI * TJ - 1, 0);
                }
            }
#else
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ch = 2 * kk + hh;
                half8 af[TI], bf[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *(const half8*)(wt + a_off + i * 32 * 128 + ((ch ^ a_sw) << 4));
#pragma unroll
                for (int j = 0; j < TJ; ++j) bf[j] = *(const half8*)(patch + boff[j] + ((ch ^ bsw[j]) << 4));
                if (BMI_PATCH_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                if (BMI_PATCH_SETPRIO) __builtin_amdgcn_s_setprio(0);
                if constexpr (BMI_PATCH_WSPREAD && WDMA && NB == 2) {
                    if (more)
                        GLDS16(wsrc + (size_t)(32 * kk) * Ktot + w_koff(step + 1), wbuf + (buf ^ 1) * G::WTILE + (kk * 256 + wave * 64) * 16);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#endif
            if (more) STORE_W(buf ^ 1);
            if (tap == 8 && chunk + 1 < nchunks) {
                lds_barrier();                 // every wave is done reading this chunk's patch
                ISSUE_PATCH((chunk + 1) * 64);
            }
        }
    }
#undef ISSUE_PATCH
#undef LOAD_W
#undef STORE_W

    // ---- fused 1x1 strided shortcut: extra K-steps on a halo-free "patch" of the block input -----
    // (the downsample conv of BasicBlock: as its own launch it is a K = 64..256 GEMM that costs 10 % of
    // the conv time for 1.2 % of the FLOPs; here it is Cin2/64 more K-steps and no residual traffic)
    if (a.in2) {
        const int nch2 = a.Cin2 / 64;
        for (int c2 = 0; c2 < nch2; ++c2) {
            lds_barrier();   // every wave is done with the patch and the weight buffers
#pragma unroll
            for (int i = 0; i < 2 * TJ; ++i) {   // BP cells x 8 pieces = 256 threads x 2*TJ
                const int q = tid + 256 * i;
                const int p = q >> 3, cp = q & 7;
                const int img = p / (TH * TW), rem = p - img * (TH * TW);
                const int oy = rem / TW, ox = rem - oy * TW;
                const int c = cp ^ (((ox + KA * oy) >> 1) & 7);
                const int n = n0 + img;
                const int nm = n < a.in2_mod ? n : n % a.in2_mod;
                const _Float16* src = n < a.N ? a.in2 + (((size_t)nm * a.H2 + (size_t)(y0 + oy) * a.stride2) * a.W2 +
                                                        (size_t)(x0 + ox) * a.stride2) * a.Cin2 + c2 * 64 + c * 8
                                              : (const _Float16*)g_zero_page;
                GLDS16(src, patch + (i * 256 + wave * 64) * 16);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                GLDS16(a.wgt2 + (size_t)(ch0 + w_row + 32 * i) * a.Cin2 + c2 * 64 + ((tid & 7) ^ w_sw) * 8,
                       wbuf + (i * 256 + wave * 64) * 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ch = 2 * kk + hh;
                half8 af[TI], bf[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *(const half8*)(wbuf + a_off + i * 32 * 128 + ((ch ^ a_sw) << 4));
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    const int p = wp * (32 * TJ) + 32 * j + r;
                    const int rem = p % (TH * TW);
                    const int sw = (((rem % TW) + KA * (rem / TW)) >> 1) & 7;
                    bf[j] = *(const half8*)(patch + p * 128 + ((ch ^ sw) << 4));
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    STAMP(2);
    // ---- epilogue (coalesced through LDS) ----------------------------------------------------------
    auto pixmap = [&](int p, int& n, int& rem) -> bool {
        const int img = p / (TH * TW), q = p - img * (TH * TW);
        const int oy = q / TW, ox = q - oy * TW;
        n = n0 + img;
        rem = (y0 + oy) * a.Wo + x0 + ox;
        return n < a.N;
    };
    auto offmap = [&](int p, size_t& off) -> bool {
        int n, rem;
        const bool ok = pixmap(p, n, rem);
        off = ((size_t)n * (a.Ho * a.Wo) + rem) * a.Cout;
        return ok;
    };
    epilogue_coalesced<TJ, PLAIN>(a, acc, smem, tid, ch0, pixmap, offmap);
    STAMP(3);
}

template <int S, int TH, int TW, int IMGS, int TJ, int NB = 2>
static int launch_patch(const ConvArgs& a, hipStream_t s) {
    const long tiles = (long)((a.N + IMGS - 1) / IMGS) * (a.Ho / TH) * (a.Wo / TW) * (a.Cout / 128);
    if (tiles <= 0 || tiles > 0x7fffffffL) return BMI_ERR_INVALID;
    if (conv_epilogue_is_plain(a)) hipLaunchKernelGGL((conv3x3_patch_kernel<S, TH, TW, IMGS, TJ, NB, true>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<S, TH, TW, IMGS, TJ, NB, false>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// Shapes the patch kernel takes (kept in sync with launch_conv3x3_patch; the engine uses it to decide
// which consumers of a site can read keep bits instead of a materialised masked tensor).
bool conv_takes_patch_kernel(int ksize, int stride, int pad, int cin, int cout, int ho, int wo) {
    if (ksize != 3 || pad != 1 || stride != 1 || cin % 64 != 0 || cout % 128 != 0) return false;
    return (ho == 16 && wo == 16) || (ho == 8 && wo == 8) || (ho == 4 && wo == 4) || (ho % 8 == 0 && wo == 32);
}

// Returns BMI_ERR_UNSUPPORTED when no patch configuration fits (the caller falls back to conv_igemm).
int launch_conv3x3_patch(const ConvArgs& a, hipStream_t s) {
    if (a.in_bits) return BMI_ERR_UNSUPPORTED;   // the patch is filled by DMA: no place to apply keep bits
    if (a.in2 && (!a.wgt2 || a.Cin2 % 64 != 0 || a.in2_mod <= 0 || a.stride2 < 1)) return BMI_ERR_INVALID;
    if (a.ksize != 3 || a.pad != 1 || a.Cin % 64 != 0 || a.Cout % 128 != 0) return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.in_mod <= 0 || a.B <= 0 || (a.res && a.res_mod <= 0)) return BMI_ERR_INVALID;
    if ((size_t)a.in_mod * a.H * a.W * a.Cin >= 0x7fffffffull) return BMI_ERR_UNSUPPORTED;   // 31-bit DMA source offsets
    if (a.stride == 1) {
        if (a.Ho == 16 && a.Wo == 16) return launch_patch<1, 16, 16, 1, 4>(a, s);
        if (a.Ho == 8 && a.Wo == 8) return launch_patch<1, 8, 8, 2, 2, BMI_PATCH_NB_S3>(a, s);
        if (a.Ho == 4 && a.Wo == 4) return launch_patch<1, 4, 4, 8, 2>(a, s);
        if (a.Ho % 8 == 0 && a.Wo == 32) return launch_patch<1, 8, 32, 1, 4>(a, s);
    }
    static const int s2 = [] { const char* v = std::getenv("BMI_PATCH_S2"); return v ? std::atoi(v) : 0; }();
    if (a.stride == 2 && s2) {   // experimental: stride-2 through the patch kernel (1 workgroup per CU)
        if (a.Ho == 8 && a.Wo == 8) return launch_patch<2, 8, 8, 2, 2>(a, s);
        if (a.Ho == 4 && a.Wo == 4) return launch_patch<2, 4, 4, 8, 2>(a, s);
        if (a.Ho == 16 && a.Wo == 16) return launch_patch<2, 8, 16, 1, 2>(a, s);
    }
    return BMI_ERR_UNSUPPORTED;
}
