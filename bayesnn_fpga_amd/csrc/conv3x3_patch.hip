// 3x3 / pad-1 convolution with the INPUT PATCH resident in LDS (gfx950, fp16 -> fp32 accumulate).
//
// Why: in the per-tap implicit GEMM (conv_igemm.hip) every K-step re-stages the activation tile,
// so LDS write traffic equals LDS read traffic equals MFMA time and the kernel is LDS-bound.  A
// 3x3 stride-1 conv reads each input pixel for 9 taps: here a workgroup stages the input patch of
// its output tile (with halo) ONCE per 64-channel chunk and runs all 9 taps out of LDS by
// shifting the per-lane fragment address; only the 16 KB weight tile streams per K-step.
//
//   work-group  = IMGS images x (TH x TW) output pixels (BP = 64*TJ pixels) x 128 output channels
//   4 waves     = 2 (channels) x 2 (pixels); wave tile 64 channels x 32*TJ pixels of 32x32x16 or 16x16x32 MFMAs
//   LDS         = patch [cells][64 ch] fp16 (128 B per cell) + 2 x weight tile [128][64]
//   staging     = patch: global_load_lds_dwordx4 (LDS-DMA), no VGPR round trip, no ds_write; padding and
//                 halo cells are DMA'd from a zero page, so there is no bounds logic in the loop.
//                 weights: LDS-DMA one K-step ahead
//   swizzle     = 16-byte chunk c of a cell is stored at chunk c ^ sw(key), key = patch_x + KA * patch_y (KA chosen per
//                 tile shape so that the 16 / 32 pixels of a fragment have consecutive keys), so that the 16 lanes of
//                 every ds_read_b128 group — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) —
//                 hit 16 distinct 16-byte slots for every tap shift.  32x32x16 fragments (a group = 16 rows, one chunk):
//                 sw = (key >> 1) & 7.  16x16x32 fragments (lane = row + 16 kq: a group reads chunk c of rows 0-3 and
//                 12-15 and chunk c ^ 1 of rows 4-11): sw = 2 ((key >> 1) & 3) — bit 0 of the chunk is left alone, so the two
//                 chunk classes of a group never meet, and inside a class the four same-parity cells differ in key >> 1
//                 mod 4.  (With the 32-row swizzle the 16-row fragments conflicted on the kx = 1, 2 taps: SQ_LDS_BANK_CONFLICT
//                 = 32 % of SQ_LDS_IDX_ACTIVE on the 16x16-map convs, profiles/experiments/r3_lds_bank_conflicts_before.txt.)
//                 LDS-DMA writes linearly, so the permutation is applied to the per-lane SOURCE address (cdna guide rule 21)
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

template <int TH, int TW, int IMGS, int TJ, int BCT = 128>
struct PatchGeom {
    static constexpr int BC = BCT;
    static constexpr int BP = IMGS * TH * TW;
    static_assert(BP == 64 * TJ, "pixel tile must be 2 waves x TJ x 32");
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int PWP = (PW + 1) & ~1;                                  // even pitch (cells)
    static constexpr int KA = TW >= 16 ? 0 : (TW == 8 ? 8 : 4);
    static constexpr int CELLS = IMGS * PH * PWP;
    static constexpr int PIECES = CELLS * 8;
    static constexpr int ITER_P = (PIECES + 255) / 256;
    static constexpr int PATCH_BYTES = ITER_P * 256 * 16;
    static constexpr int WTILE = BC * 128;
    static constexpr int MAIN_BYTES = PATCH_BYTES + 2 * WTILE;
    static constexpr int LDS_BYTES = MAIN_BYTES > BMI_EPILOGUE_LDS_BYTES ? MAIN_BYTES : BMI_EPILOGUE_LDS_BYTES;
};

// MS = MFMA shape: 32 -> v_mfma_f32_32x32x16_f16 (wave tile = 2 x TJ tiles), 16 -> v_mfma_f32_16x16x32_f16 (4 x 2TJ
// tiles).  Same wave tile (64 channels x 32*TJ pixels), same LDS bytes per FLOP (a ds_read_b128 is a 32 x 16 or a
// 16 x 32 fragment), same LDS layout and swizzle (the 16 lanes of a read group are 16 consecutive rows either way).
// Which one is faster is a clock question, not a cycle question (MI355X_MICROARCH.md, DVFS give-back item 7): both are
// built and launch_conv3x3_patch picks by measured wall time (BMI_MFMA_SHAPE overrides).
// BCT = 64 (round 6; 16x16x32 MFMAs only): the 64-channel tile for the 64 -> 64 convs on 32x32 maps — the stem's BasicBlocks: once per batch in every
// ResNet-18 / VGG configuration (four of the 27 launches of the paper's exit-only step), and per SAMPLE behind a "layer" site, where the per-tap
// conv_igemm took 4.2 + 6.5 ms of a 34 ms step at 447 / 289 TFLOP/s.  All four waves sit on the pixel axis (wave tile 64 channels x 16*TJ pixels),
// the 8 KB weight stage is read by every wave, and the launch finishes in the per-quad epilogue (conv_epilogue.h: epilogue_quad — every site kind,
// residual, ReLU; no LDS): same K order per accumulator as the 128-channel tile.  No fused shortcut (a 64-channel block has no downsample path).
// DIRECT (round 6; the 16x16 class, 16x16x32 MFMAs, plain epilogue and the two BasicBlock tails): the epilogue runs on the accumulator registers and
// stores straight to HBM — what conv3x3_pw's persistent kernel does since round 4 (PWP_DIRECT).  The MFMA rows of a wave's four channel tiles are a
// PERMUTATION of its 64 channels — LDS weight row 16 i + r holds channel 32 (i >> 1) + 8 (r >> 2) + 4 (i & 1) + (r & 3), applied where the weight DMA picks
// its source row, so nothing on the LDS side moves — and a lane's sixteen accumulators of a pixel are two runs of 8 consecutive channels (8 q + 0..7 and
// 32 + 8 q + 0..7, q = lane >> 4): two 16-byte stores per pixel tile (the four lanes of a pixel write 64 contiguous bytes), the residual as two 16-byte loads,
// no LDS trip, no barrier, no residual DMA.  A row permutation does not touch a row's arithmetic: the same bits as epilogue_plain / epilogue_lite.
template <int TH, int TW, int IMGS, int TJ, int EPI, int MS, bool BF, bool IMAP = false, int BCT = 128, bool DIRECT = false>
__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(ConvArgs a) {
    using G = PatchGeom<TH, TW, IMGS, TJ, BCT>;
    static_assert(BCT == 128 || (BCT == 64 && MS == 16), "64-channel tiles: the 16x16x32 form");
    static_assert(!DIRECT || (BCT == 128 && MS == 16 && !IMAP && TJ == 4 && (EPI == BMI_EPI_PLAIN || EPI == BMI_EPI_LITE_RES || EPI == BMI_EPI_LITE_RES_MC)),
                  "the register epilogue: 128-channel tiles, 16x16x32 accumulators, plain / residual / residual + 2-bit site");
    constexpr int BC = G::BC, PH = G::PH, PW = G::PW, PWP = G::PWP, KA = G::KA;
    constexpr int NWP = BCT == 128 ? 2 : 4;       // waves on the pixel axis (x 4 / NWP on the channel axis)
    constexpr int PPW = G::BP / NWP;              // pixels per wave: 32 TJ | 16 TJ
    constexpr int TI = MS == 32 ? 2 : 4;          // channel tiles per wave
    constexpr int TP = PPW / MS;                  // pixel tiles per wave
    constexpr int RW = MS;                        // rows (channels / pixels) per MFMA tile
    __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];
    char* const patch = smem;
    char* const wbuf = smem + G::PATCH_BYTES;

    const int tid = threadIdx.x;
    STAMP(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & (RW - 1);                // row of the lane inside an MFMA tile
    const int kq = lane / RW;                     // which 8-element k group of the MFMA's K the lane holds (0..64/RW-1)
    constexpr int KSUB = MS == 32 ? 4 : 2;        // MFMA k-substeps per 64-deep K-step
    constexpr int KQ = 64 / RW;                   // 16-byte chunks of a 128-byte row consumed per substep
    const int wc = wave / NWP, wp = wave % NWP;

    // ---- tile coordinates (channel tile fastest) ------------------------------------------------
    const int n_ctiles = a.Cout / BC;
    const int tiles_x = a.Wo / TW, tiles_y = a.Ho / TH;
    int bid, ctile;
    xcd_tile_map(blockIdx.x, (int)(gridDim.x / n_ctiles), n_ctiles, bid, ctile, a.xcd_split);
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n0 = (bid / tiles_y) * IMGS;
    const int ch0 = ctile * BC;
    const int y0 = ty * TH, x0 = tx * TW;
    const int Ktot = 9 * a.Cin;

    // patch-cell swizzle (see the header): chunk c of a cell with key k lives at chunk c ^ PSW(k)
#define PSW(KEY) (MS == 32 ? (((KEY) >> 1) & 7) : ((((KEY) >> 1) & 3) << 1))
    // ---- per-thread DMA sources ----------------------------------------------------------------
    // piece q = tid + 256*i  ->  LDS slot (cell = q >> 3, physical chunk = q & 7)
    // element offset of the piece's source at chunk 0 (relative to a.in), or -1 -> zero page
    int psrc[G::ITER_P];
    // 64-channel tile, input read through a lazy site (ConvArgs::in_bits; round 6: the first "layer" site feeds 64 -> 64 stride-1 convs): byte offset
    // of the piece's eight keep bits in the FOLDED tensor's bit image (the piece itself comes from the B scaled images: in_mod = B), or -1
    long pbit[BCT == 64 ? G::ITER_P : 1];
#pragma unroll
    for (int i = 0; i < G::ITER_P; ++i) {
        const int q = tid + 256 * i;
        const int cell = q >> 3, cp = q & 7;
        const int rowc = cell / PWP, px = cell - rowc * PWP;
        const int img = rowc / PH, py = rowc - img * PH;
        const int key = px + KA * py;
        const int c = cp ^ PSW(key);
        const int n = n0 + img;
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = cell < G::CELLS && px < PW && n < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        psrc[i] = ok ? (int)((((size_t)(map_image<IMAP>(a, n) % a.in_mod) * a.H + iy) * a.W + ix) * a.Cin + c * 8) : -1;
        if constexpr (BCT == 64) pbit[i] = (ok && a.in_bits) ? (long)(((((size_t)map_image<IMAP>(a, n) * a.H + iy) * a.W + ix) * a.Cin + c * 8) >> 3) : -1L;
    }
    // Weight tile: LDS-DMA one K-step ahead, double buffered, issued as one burst behind the barrier.  Recorded
    // negatives (round 1, same-box A/B on S2/S3/S4): register staging -10 %, hand-pipelined fragment reads -3 %, a third
    // buffer with counted vmcnt -4 %, one DMA piece per k-substep -2..-7 %, s_setprio around the MFMAs -2 %,
    // stride-2 convs through this kernel (1 workgroup per CU) -20..-35 % vs the wide per-tap kernel.
    const int w_row = tid >> 3;
    const int w_sw = (w_row >> 1) & 7;
    const _Float16* wsrc = a.wgt + (size_t)(ch0 + w_row) * Ktot + ((tid & 7) ^ w_sw) * 8;
    // DIRECT: LDS weight row rho = w_row + 32 i holds channel sigma(rho) (see the kernel's header); identity otherwise
    int wch[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rho = w_row + 32 * i, t = (rho >> 4) & 3, r16 = rho & 15;
        wch[i] = DIRECT ? (rho & 64) + 32 * (t >> 1) + 8 * (r16 >> 2) + 4 * (t & 1) + (r16 & 3) : rho;
    }

#define ISSUE_PATCH(C0)                                                                           \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < G::ITER_P; ++i)                                     \
            GLDS16(psrc[i] >= 0 ? a.in + (size_t)(unsigned)psrc[i] + (C0) : (const _Float16*)g_zero_page,   \
                   patch + (i * 256 + wave * 64) * 16);                                           \
    }
#define LOAD_W(KOFF, BUF)                                                                         \
    {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < BC / 32; ++i)                                       \
            GLDS16(wsrc + (size_t)(wch[i] - w_row) * Ktot + (KOFF),                               \
                   wbuf + (BUF) * G::WTILE + (i * 256 + wave * 64) * 16);                         \
    }

    // ---- per-lane fragment geometry --------------------------------------------------------------
    int bcell[TP], bkey[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) {
        const int p = wp * PPW + RW * j + r;
        const int img = p / (TH * TW), rem = p - img * (TH * TW);
        const int oy = rem / TW, ox = rem - oy * TW;
        bcell[j] = (img * PH + oy) * PWP + ox;   // tap (0,0)
        bkey[j] = ox + KA * oy;
    }
    const int a_off = (wc * 64 + r) * 128;
    const int a_sw = (r >> 1) & 7;

    typedef float accv __attribute__((ext_vector_type(MS == 32 ? 16 : 4)));
    accv acc[TI][TP];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int e = 0; e < (MS == 32 ? 16 : 4); ++e) acc[i][j][e] = 0.f;

#define PATCH_MFMA(AFR, BFR, ACC)                                                                 \
    if constexpr (MS == 32) ACC = mfma_32x32x16<BF>(AFR, BFR, ACC);                               \
    else ACC = mfma_16x16x32<BF>(AFR, BFR, ACC);

    const int nchunks = a.Cin / 64;
    const int nK = 9 * nchunks;
    // K-step s = chunk * 9 + tap uses weight buffer s & 1; the tile of step s+1 is in flight while step s computes
    auto w_koff = [&](int st) { const int c = st / 9, t = st - 9 * c; return t * a.Cin + c * 64; };
    uint8_t kbp[BCT == 64 ? G::ITER_P : 1];   // (64-channel tile with in_bits: the keep bits of this thread's patch pieces)
    ISSUE_PATCH(0);
    LOAD_W(0, 0);
    int step = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        for (int tap = 0; tap < 9; ++tap, ++step) {
            const int buf = step & 1;
#ifdef BMI_PATCH_STAMPS
            const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
            if constexpr (BCT == 64) {
                // keep bits of this chunk's pieces: requested now, used behind the wait (their latency rides with the patch DMA's)
                if (tap == 0 && a.in_bits) {
#pragma unroll
                    for (int i = 0; i < G::ITER_P; ++i) kbp[i] = pbit[i] >= 0 ? a.in_bits[pbit[i] + chunk * 8] : (uint8_t)0xff;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            STAMP_ADD(4, tw0);
            if constexpr (BCT == 64) {
                // ... each thread clears the dropped elements of the pieces IT fetched (its own DMA has landed: no barrier needed before), the
                // barrier below publishes the masked patch — what the fused shortcut's operand does on the 16x16 maps (below)
                if (tap == 0 && a.in_bits) {
#pragma unroll
                    for (int i = 0; i < G::ITER_P; ++i) {
                        u32x4* const pp = (u32x4*)(patch + (i * 256 + tid) * 16);
                        u32x4 v = *pp;
                        const int b = kbp[i];
#pragma unroll
                        for (int jq = 0; jq < 4; ++jq) {
                            const unsigned lo = (unsigned)__builtin_amdgcn_sbfe(b, 2 * jq, 1), hi = (unsigned)__builtin_amdgcn_sbfe(b, 2 * jq + 1, 1);
                            v[jq] &= (lo & 0xffffu) | (hi & 0xffff0000u);
                        }
                        *pp = v;
                    }
                }
            }
            __syncthreads();   // patch landed, W[step&1] written; every wave is done with W[(step+1)&1]
            STAMP_ADD(5, tw0);
            if (step == 0) { STAMP(1); }
            if (step + 1 < nK) LOAD_W(w_koff(step + 1), buf ^ 1);
            const int ky = tap / 3, kx = tap - 3 * ky;
            // per-tap cell shift and swizzle key shift
            const int coff = ky * PWP + kx, koff = kx + KA * ky;
            const char* wt = wbuf + buf * G::WTILE;
            int boff[TP], bsw[TP];
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                boff[j] = (bcell[j] + coff) * 128;
                bsw[j] = PSW(bkey[j] + koff);
            }
#pragma unroll
            for (int kk = 0; kk < KSUB; ++kk) {
                const int ch = KQ * kk + kq;
                half8 af[TI], bf[TP];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *(const half8*)(wt + a_off + i * RW * 128 + ((ch ^ a_sw) << 4));
#pragma unroll
                for (int j = 0; j < TP; ++j) bf[j] = *(const half8*)(patch + boff[j] + ((ch ^ bsw[j]) << 4));
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) { PATCH_MFMA(af[i], bf[j], acc[i][j]); }
            }
            if (tap == 8 && chunk + 1 < nchunks) {
                lds_barrier();                 // every wave is done reading this chunk's patch
                ISSUE_PATCH((chunk + 1) * 64);
            }
        }
    }
#undef ISSUE_PATCH
#undef LOAD_W

    // ---- fused 1x1 strided shortcut: extra K-steps on a halo-free "patch" of the block input -----
    // (the downsample conv of BasicBlock: as its own launch it is a K = 64..256 GEMM that costs 10 % of
    // the conv time for 1.2 % of the FLOPs; here it is Cin2/64 more K-steps and no residual traffic)
    if (BCT == 128 && a.in2) {
        const int nch2 = a.Cin2 / 64;
        for (int c2 = 0; c2 < nch2; ++c2) {
            lds_barrier();   // every wave is done with the patch and the weight buffers
            uint8_t kb[2 * TJ];                  // in2_bits: keep flags of this thread's pieces (8 channels each)
#pragma unroll
            for (int i = 0; i < 2 * TJ; ++i) {   // BP cells x 8 pieces = 256 threads x 2*TJ
                const int q = tid + 256 * i;
                const int p = q >> 3, cp = q & 7;
                const int img = p / (TH * TW), rem = p - img * (TH * TW);
                const int oy = rem / TW, ox = rem - oy * TW;
                const int c = cp ^ PSW(ox + KA * oy);
                const int n = n0 + img;
                const int row = map_image<IMAP>(a, n);
                const int nm = row % a.in2_mod;
                const size_t pix = ((size_t)(y0 + oy) * a.stride2) * a.W2 + (size_t)(x0 + ox) * a.stride2;
                // element offset inside the image: NHWC, or the lazy site's planar layout (kernels.h lazy_planar_off; only with keep bits)
                const size_t eoff = (MS == 16 && a.in2_bits && a.lazy_planar)
                                        ? (size_t)lazy_planar_off(c2 * 64 + c * 8, (y0 + oy) * a.stride2, (x0 + ox) * a.stride2, a.H2 * a.W2, a.W2)
                                        : pix * a.Cin2 + c2 * 64 + c * 8;
                const size_t img_e = (size_t)a.H2 * a.W2 * a.Cin2;
                const _Float16* src = n < a.N ? a.in2 + (size_t)nm * img_e + eoff : (const _Float16*)g_zero_page;
                GLDS16(src, patch + (i * 256 + wave * 64) * 16);
                kb[i] = 0xff;
                if (MS == 16 && a.in2_bits && n < a.N) kb[i] = a.in2_bits[((size_t)row * img_e + eoff) >> 3];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                GLDS16(a.wgt2 + (size_t)(ch0 + wch[i]) * a.Cin2 + c2 * 64 + ((tid & 7) ^ w_sw) * 8,
                       wbuf + (i * 256 + wave * 64) * 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MS == 16 && a.in2_bits) {        // each thread clears the dropped elements of the pieces it fetched itself (launcher: 16x16x32 form only)
#pragma unroll
                for (int i = 0; i < 2 * TJ; ++i) {
                    u32x4* const pp = (u32x4*)(patch + (i * 256 + tid) * 16);
                    u32x4 v = *pp;
                    const int b = kb[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned lo = (unsigned)__builtin_amdgcn_sbfe(b, 2 * j, 1), hi = (unsigned)__builtin_amdgcn_sbfe(b, 2 * j + 1, 1);
                        v[j] &= (lo & 0xffffu) | (hi & 0xffff0000u);
                    }
                    *pp = v;
                }
            }
            lds_barrier();
#pragma unroll
            for (int kk = 0; kk < KSUB; ++kk) {
                const int ch = KQ * kk + kq;
                half8 af[TI], bf[TP];
#pragma unroll
                for (int i = 0; i < TI; ++i) af[i] = *(const half8*)(wbuf + a_off + i * RW * 128 + ((ch ^ a_sw) << 4));
#pragma unroll
                for (int j = 0; j < TP; ++j) {
                    const int p = wp * PPW + RW * j + r;
                    const int rem = p % (TH * TW);
                    const int sw = PSW((rem % TW) + KA * (rem / TW));
                    bf[j] = *(const half8*)(patch + p * 128 + ((ch ^ sw) << 4));
                }
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TP; ++j) { PATCH_MFMA(af[i], bf[j], acc[i][j]); }
            }
        }
    }
#undef PATCH_MFMA
#undef PSW

    STAMP(2);
    // ---- epilogue (coalesced through LDS) ----------------------------------------------------------
    auto pixmap = [&](int p, int& n, int& rem) -> bool {
        const int img = p / (TH * TW), q = p - img * (TH * TW);
        const int oy = q / TW, ox = q - oy * TW;
        n = n0 + img;
        rem = (y0 + oy) * a.Wo + x0 + ox;
        const bool ok = n < a.N;
        n = map_image<IMAP>(a, n);
        return ok;
    };
    auto offmap = [&](int p, size_t& off) -> bool {
        int n, rem;
        const bool ok = pixmap(p, n, rem);
        off = ((size_t)n * (a.Ho * a.Wo) + rem) * a.Cout;
        return ok;
    };
    if constexpr (BCT == 64) {
        // per-quad epilogue: lane = pixel (lane & 15) of pixel tile j, its register quad of channel tile i = channels 16 i + 4 (lane >> 4) ..
        // Arithmetic and order of epilogue_quad.  The common stochastic launch — an elementwise site drawn at 2 bits per element (p = 0.25) —
        // shares its Philox calls: ONE call masks the 64 channels of a pixel (the whole tile row), lane (pixel, q4) computes the call of
        // pixel tile j = q4 and the four lanes of a pixel fetch its words with __shfl (epilogue_lite's scheme) instead of sixteen calls per pixel.
        (void)offmap;
        static_assert(TP <= 4, "one shared Philox call per lane");
        const int l16 = lane & 15, q4 = lane >> 4;
        const bool fast_site = a.site.kind == BMI_SITE_ELEMENTWISE && a.site.log2_bits == 1 && !a.site.drop_all && !a.site_inner;   // launch-uniform
        // launch-uniform: the register form below (BN vectors loaded once per lane, ALL residual quads of the lane requested before the first
        // is used — 16 eight-byte loads in flight instead of one exposed round trip per quad —, then arithmetic and stores) takes the launches
        // without a site and with the 2-bit elementwise site; every other site kind goes quad by quad through epilogue_quad.
        const bool regs_form = !a.site_inner && (a.site.kind == BMI_SITE_NONE || fast_site);
        philox4 mine = {{0u, 0u, 0u, 0u}};
        if (fast_site) {
            int n, rem;
            pixmap(wp * PPW + 16 * q4 + l16, n, rem);
            const int tl = n / a.B;
            mine = philox_site_call(a.site, (uint64_t)((n - tl * a.B) * (a.Ho * a.Wo) + rem) * a.Cout + ch0, (uint32_t)(a.t0 + tl));
        }
        if (regs_form) {
            f32x4_e sc[TI], bi[TI];
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int c4 = ch0 + 16 * i + 4 * q4;
                sc[i] = f32x4_e{1.f, 1.f, 1.f, 1.f};
                bi[i] = f32x4_e{0.f, 0.f, 0.f, 0.f};
                if (a.scale) sc[i] = *(const f32x4_e*)(a.scale + c4);
                if (a.bias) bi[i] = *(const f32x4_e*)(a.bias + c4);
            }
            bool okp[TP];
            size_t ooff[TP];
            half4 rq[TP][TI];
            uint8_t rkb[TP][TI];
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                int n, rem;
                okp[j] = pixmap(wp * PPW + 16 * j + l16, n, rem);
                ooff[j] = ((size_t)n * (a.Ho * a.Wo) + rem) * a.Cout;
                const size_t roff = a.res ? ((size_t)(n % a.res_mod) * (a.Ho * a.Wo) + rem) * a.Cout : 0;
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    rq[j][i] = half4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                    rkb[j][i] = 0xff;
                    if (a.res && okp[j]) rq[j][i] = *(const half4*)(a.res + roff + ch0 + 16 * i + 4 * q4);
                    // residual read through a lazy site: its keep bits sit at the element's index in the FOLDED tensor (row n, not n % res_mod)
                    if (a.res_bits && okp[j]) rkb[j][i] = a.res_bits[(ooff[j] + ch0 + 16 * i + 4 * q4) >> 3];
                }
            }
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                if (fast_site) {
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) w[wd] = (uint32_t)__shfl((int)mine.w[wd], l16 + 16 * j, 64);
                }
                if (!okp[j]) continue;
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    const uint32_t fields = w[i] >> (8 * q4);      // the four 2-bit fields of channels 16 i + 4 q4 ..
                    half4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // epilogue_quad's arithmetic and order: (acc * (scale * out_mul)) + bias, + residual, ReLU, site
                        float x = acc[i][j][e];
                        if (a.scale) x *= sc[i][e] * a.out_mul;
                        else if (a.out_mul != 1.f) x *= a.out_mul;
                        if (a.bias) x += bi[i][e];
                        if (a.res) x += ((rkb[j][i] >> (4 * (q4 & 1) + e)) & 1) ? a16_to_f32<BF>(rq[j][i][e]) : 0.f;
                        if (a.relu) x = fmaxf(x, 0.f);
                        if (fast_site) x = ((fields >> (2 * e)) & 3u) >= a.site.thresh ? x * a.site.scale : 0.f;
                        o[e] = a16_from_f32<BF>(x);
                    }
                    *(half4*)(a.out + ooff[j] + ch0 + 16 * i + 4 * q4) = o;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TP; ++j) {
                int n, rem;
                if (!pixmap(wp * PPW + 16 * j + l16, n, rem)) continue;
                const PixelCtx px = make_pixel_ctx(a, n, rem);
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    epilogue_quad<BF>(a, px, v, ch0 + 16 * i + 4 * q4);
                }
            }
        }
    } else if constexpr (DIRECT) {
        // ---- register epilogue (see the kernel's header): lane (l16, q4) holds, per pixel tile j, channels cA .. cA + 7 (tiles 0, 1) and cA + 32 .. (tiles 2, 3)
        const int l16 = lane & 15, q4 = lane >> 4;
        const int cA = ch0 + wc * 64 + 8 * q4;
        constexpr bool HAS_RES = EPI != BMI_EPI_PLAIN, MASKED = EPI == BMI_EPI_LITE_RES_MC;
        f32x4_e sc[4], bi[4];           // tile i: channels cA + 32 (i >> 1) + 4 (i & 1) ..
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = cA + 32 * (i >> 1) + 4 * (i & 1);
            sc[i] = f32x4_e{1.f, 1.f, 1.f, 1.f};
            bi[i] = f32x4_e{0.f, 0.f, 0.f, 0.f};
            if (a.scale) sc[i] = *(const f32x4_e*)(a.scale + c4);
            if (a.bias) bi[i] = *(const f32x4_e*)(a.bias + c4);
            sc[i] *= a.out_mul;
        }
        philox4 mine[2] = {{{0u, 0u, 0u, 0u}}, {{0u, 0u, 0u, 0u}}};
        if constexpr (MASKED) {         // one Philox call masks the 64 channels of (pixel, wave channel half): 128 pixels per wave = two calls per lane
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                int n, rem;
                pixmap(wp * PPW + 16 * (q4 + 4 * r2) + l16, n, rem);
                const int tl = n / a.B;
                const uint64_t e0 = (uint64_t)((n - tl * a.B) * (a.Ho * a.Wo) + rem) * a.Cout + ch0 + wc * 64;
                mine[r2] = philox_site_call(a.site, e0, (uint32_t)(a.t0 + tl));
            }
        }
#pragma unroll
        for (int jb = 0; jb < TP; jb += 4) {       // four pixel tiles at a time: eight 16-byte residual loads in flight
            size_t off[4];
            bool okp[4];
            half8_e rA[4], rB[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                okp[jj] = offmap(wp * PPW + 16 * (jb + jj) + l16, off[jj]);
                if constexpr (HAS_RES) {
                    if (okp[jj]) {
                        rA[jj] = *(const half8_e*)(a.res + off[jj] + cA);
                        rB[jj] = *(const half8_e*)(a.res + off[jj] + cA + 32);
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int j = jb + jj;
                uint32_t wA = 0u, wB = 0u;
                if constexpr (MASKED) {     // (every lane takes part in the exchange, valid pixel or not)
                    uint32_t w[4];
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) w[wd] = (uint32_t)__shfl((int)mine[j >> 2].w[wd], l16 + 16 * (j & 3), 64);
                    // channel k of the wave's 64 <-> word k >> 4, bits 2 (k & 15): cA's run = word q4 >> 1 from bit 16 (q4 & 1), cA + 32's = word 2 + (q4 >> 1)
                    wA = ((q4 & 2) ? w[1] : w[0]) >> (16 * (q4 & 1));
                    wB = ((q4 & 2) ? w[3] : w[2]) >> (16 * (q4 & 1));
                }
                if (!okp[jj]) continue;
                half8_e oA, oB;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][e] * sc[i][e] + bi[i][e];
                        const int e8 = 4 * (i & 1) + e;
                        if constexpr (HAS_RES) {
                            v += a16_to_f32<BF>((i >> 1) ? rB[jj][e8] : rA[jj][e8]);
                            v = fmaxf(v, 0.f);                       // (the specialised tails: ReLU compiled in)
                        } else {
                            if (a.relu) v = fmaxf(v, 0.f);
                        }
                        if constexpr (MASKED) {
                            const uint32_t f = (((i >> 1) ? wB : wA) >> (2 * e8)) & 3u;
                            v = f >= a.site.thresh ? v * a.site.scale : 0.f;
                            asm("" : "+v"(v));                       // (epilogue_lite: keep the fp32 product — one rounding, like every other epilogue)
                        }
                        if (i >> 1) oB[e8] = a16_from_f32<BF>(v);
                        else oA[e8] = a16_from_f32<BF>(v);
                    }
                *(half8_e*)(a.out + off[jj] + cA) = oA;
                *(half8_e*)(a.out + off[jj] + cA + 32) = oB;
            }
        }
    } else {
        epilogue_coalesced<TJ, EPI, MS, BF>(a, acc, smem, tid, ch0, pixmap, offmap);
    }
    STAMP(3);
}

template <int TH, int TW, int IMGS, int TJ>
static int launch_patch(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    a.xcd_split = xcd_split_for(a.Cout / 128, (size_t)a.Cout * 9 * a.Cin * 2);
    const long tiles = (long)((a.N + IMGS - 1) / IMGS) * (a.Ho / TH) * (a.Wo / TW) * (a.Cout / 128);
    if (tiles <= 0 || tiles > 0x7fffffffL) return BMI_ERR_INVALID;
    const int ms = (a.imap || a.bf16) ? 16 : opt_mfma_shape_patch();   // bf16 / dynamic exit: the 16x16x32 shape only
    if (a.in2_bits && (ms != 16 || TW != 16)) return BMI_ERR_UNSUPPORTED;   // keep bits on the shortcut's input: the 16x16-map, 16x16x32 form
    if (a.lazy_planar && (!a.in2_bits || (a.W2 & 1) || a.Cin2 % 32 != 0)) return BMI_ERR_UNSUPPORTED;
    const int epi = opt_epilogue_lite() ? conv_epilogue_kind(a, ms) : (conv_epilogue_is_plain(a) ? BMI_EPI_PLAIN : BMI_EPI_GENERAL);
    const dim3 grid((unsigned)tiles), block(256);
#define PATCH_LAUNCH(EPI_, MS_, BF_, IMAP_) \
    hipLaunchKernelGGL((conv3x3_patch_kernel<TH, TW, IMGS, TJ, EPI_, MS_, BF_, IMAP_>), grid, block, 0, s, a)
#define PATCH_LAUNCH_DIRECT(EPI_, BF_) \
    hipLaunchKernelGGL((conv3x3_patch_kernel<TH, TW, IMGS, TJ, EPI_, 16, BF_, false, 128, true>), grid, block, 0, s, a)
#define PATCH_LAUNCH_EPI16(BF_, IMAP_)                                            \
    {                                                                             \
        if (epi == BMI_EPI_PLAIN) PATCH_LAUNCH(BMI_EPI_PLAIN, 16, BF_, IMAP_);    \
        else if (epi == BMI_EPI_LITE) PATCH_LAUNCH(BMI_EPI_LITE, 16, BF_, IMAP_); \
        else PATCH_LAUNCH(BMI_EPI_GENERAL, 16, BF_, IMAP_);                       \
    }
    if (a.imap) {   // dynamic early exit: 16x16 maps, and the 8x8 / 4x4 ones whose grid is too small for conv3x3_pw (batches of a
                    // dozen images or fewer: the full run takes this kernel there too, so the compacted stages get the same bits)
        if constexpr (TW <= 16) {
            if (a.bf16) PATCH_LAUNCH_EPI16(true, true)
            else PATCH_LAUNCH_EPI16(false, true)
        } else {
            return BMI_ERR_UNSUPPORTED;
        }
    } else if (ms == 16 && TH == 16 && TW == 16 && epi == BMI_EPI_LITE && conv_epilogue_kind_launch(a, 16) != BMI_EPI_LITE) {
        // the BasicBlock tails of the 16x16 maps: the lite epilogue with its launch-uniform terms compiled in (conv_epilogue.h; same bits)
        if constexpr (TH == 16 && TW == 16) {
            const bool direct = opt_patch_direct() != 0;      // ("patch_direct" = 0: the lite epilogue through LDS; A/B, tests — the same bits)
            if (conv_epilogue_kind_launch(a, 16) == BMI_EPI_LITE_RES) {
                if (direct) { if (a.bf16) PATCH_LAUNCH_DIRECT(BMI_EPI_LITE_RES, true); else PATCH_LAUNCH_DIRECT(BMI_EPI_LITE_RES, false); }
                else if (a.bf16) PATCH_LAUNCH(BMI_EPI_LITE_RES, 16, true, false);
                else PATCH_LAUNCH(BMI_EPI_LITE_RES, 16, false, false);
            } else if (conv_epilogue_kind_launch(a, 16) == BMI_EPI_LITE_RES_MC && direct) {
                if (a.bf16) PATCH_LAUNCH_DIRECT(BMI_EPI_LITE_RES_MC, true); else PATCH_LAUNCH_DIRECT(BMI_EPI_LITE_RES_MC, false);
            } else if (conv_epilogue_kind_launch(a, 16) == BMI_EPI_LITE_RES_MSK) {
                if (a.bf16) PATCH_LAUNCH(BMI_EPI_LITE_RES_MSK, 16, true, false);
                else PATCH_LAUNCH(BMI_EPI_LITE_RES_MSK, 16, false, false);
            } else {
                if (a.bf16) PATCH_LAUNCH(BMI_EPI_LITE_RES_MC, 16, true, false);
                else PATCH_LAUNCH(BMI_EPI_LITE_RES_MC, 16, false, false);
            }
        }
    } else if (ms == 16 && TH == 16 && TW == 16 && epi == BMI_EPI_PLAIN && opt_patch_direct() == 2) {      // ("patch_direct" = 2: measured +3...5 % on the plain launches in the network: not the default)
        if constexpr (TH == 16 && TW == 16) {     // the plain launches of the 16x16 class: the register epilogue
            if (a.bf16) PATCH_LAUNCH_DIRECT(BMI_EPI_PLAIN, true); else PATCH_LAUNCH_DIRECT(BMI_EPI_PLAIN, false);
        }
    } else if (a.bf16) {
        PATCH_LAUNCH_EPI16(true, false)
    } else if (ms == 16) {     // bmi_set_option / BMI_MFMA_SHAPE; default chosen by measured wall time
        PATCH_LAUNCH_EPI16(false, false)
    } else {
        if (epi == BMI_EPI_PLAIN) PATCH_LAUNCH(BMI_EPI_PLAIN, 32, false, false);
        else PATCH_LAUNCH(BMI_EPI_GENERAL, 32, false, false);
    }
#undef PATCH_LAUNCH_EPI16
#undef PATCH_LAUNCH_DIRECT
#undef PATCH_LAUNCH
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// 64-channel tiles (Cout % 128 == 64 on 32-wide maps): one instantiation per element type, the per-quad epilogue takes every launch form.
static int launch_patch64(const ConvArgs& a_in, hipStream_t s) {
    ConvArgs a = a_in;
    a.xcd_split = 1;
    const long tiles = (long)a.N * (a.Ho / 8) * (a.Cout / 64);
    if (tiles <= 0 || tiles > 0x7fffffffL) return BMI_ERR_INVALID;
    if (a.imap || a.in2 || a.in2_bits || a.pool || a.partial) return BMI_ERR_UNSUPPORTED;
    // operands read through a lazy site: NHWC bit images only; a masked residual needs the register-form epilogue (no site or the 2-bit elementwise one)
    if ((a.in_bits || a.res_bits) && a.lazy_planar) return BMI_ERR_UNSUPPORTED;
    if (a.res_bits && (a.site_inner || !(a.site.kind == BMI_SITE_NONE || (a.site.kind == BMI_SITE_ELEMENTWISE && a.site.log2_bits == 1 && !a.site.drop_all))))
        return BMI_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)tiles), block(256);
    if (a.bf16) hipLaunchKernelGGL((conv3x3_patch_kernel<8, 32, 1, 4, BMI_EPI_GENERAL, 16, true, false, 64>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((conv3x3_patch_kernel<8, 32, 1, 4, BMI_EPI_GENERAL, 16, false, false, 64>), grid, block, 0, s, a);
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
    if ((a.in_bits || a.res_bits) && a.Cout % 128 == 0) return BMI_ERR_UNSUPPORTED;   // the 128-channel tiles apply no keep bits to `in` / `res` (the 64-channel tile does)
    if (a.in2 && (!a.wgt2 || a.Cin2 % 64 != 0 || a.in2_mod <= 0 || a.stride2 < 1)) return BMI_ERR_INVALID;
    if (a.ksize != 3 || a.pad != 1 || a.stride != 1 || a.Cin % 64 != 0 || a.Cout % 64 != 0) return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.in_mod <= 0 || a.B <= 0 || (a.res && a.res_mod <= 0)) return BMI_ERR_INVALID;
    if ((size_t)a.in_mod * a.H * a.W * a.Cin >= 0x7fffffffull) return BMI_ERR_UNSUPPORTED;   // 31-bit DMA source offsets
    if (a.Cout % 128 != 0) {      // 64-channel tiles: the 32-wide maps only ("conv_patch64" = 0: conv_igemm as before; A/B, tests)
        if (!opt_conv_patch64() || a.Ho % 8 != 0 || a.Wo != 32) return BMI_ERR_UNSUPPORTED;
        return launch_patch64(a, s);
    }
    if (a.Ho == 16 && a.Wo == 16) return launch_patch<16, 16, 1, 4>(a, s);
    if (a.Ho == 8 && a.Wo == 8) return launch_patch<8, 8, 2, 2>(a, s);
    if (a.Ho == 4 && a.Wo == 4) return launch_patch<4, 4, 8, 2>(a, s);
    if (a.Ho % 8 == 0 && a.Wo == 32) return launch_patch<8, 32, 1, 4>(a, s);
    return BMI_ERR_UNSUPPORTED;
}
