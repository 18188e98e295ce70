// The seam between two Bottleneck blocks as ONE launch (gfx950): the expand conv of block k — conv3 + BN + residual + ReLU — and the
// reduce conv of block k+1 — conv1 + BN + ReLU — which reads nothing but the tensor the expand conv has just produced
// (SA-style ResNet-50: Bottleneck.forward, the chain of resnet18.py:51-85 at 4x the channels).
//
// Why: both convs are HBM-bound 1x1 launches (conv1x1_stream.hip).  Per block, in units of the narrow tensor, conv3 reads 1 + 4 (residual)
// and writes 4, conv1 of the next block reads those 4 again and writes 1: the wide tensor crosses HBM three times.  Here the workgroup that
// has produced a 128-pixel x 128-channel piece of the wide tensor (fp16, in LDS for its coalesced store anyway) feeds it straight into
// the second GEMM as the B operand: the wide tensor is written once (the residual of block k+1 needs it) and read once (as that residual):
// 16 -> 12 units per block.
//
//   workgroup   = 128 pixels x ALL channels: 256 threads, 4 waves = 2 (channels) x 2 (pixels), v_mfma_f32_16x16x32, two workgroups per CU
//   per 128-channel chunk c of the wide tensor (Cw / 128 chunks):
//     stage A   acc_a[128 ch x 128 px] = W3[chunk rows] . m          K = Cmid in 64-channel K-steps ([weights 16 KB | pixels 16 KB] tiles,
//               LDS-DMA, source-side XOR swizzle: conv1x1_stream's loop); the residual piece is DMA'd into the Y buffer at the top of the chunk
//     epilogue  epilogue_lite (conv_epilogue.h) with the residual already in LDS: BN, + residual, ReLU, one rounding, the fp16 piece
//               written in place into Y, coalesced 16-byte stores to HBM — the code and the bits of the unfused launch
//     stage B   acc_b[Cn x 128 px] += W1[:, chunk] . Y               two 64-deep K-steps; the B fragments are read from Y in the
//               epilogue's own layout (chunk q of pixel p at q ^ (p & 15): 16 distinct 16-byte slots per ds_read_b128 lane group)
//   end         epilogue_plain on acc_b (BN + ReLU), 128 channels at a time through Y
//   K order     stage B accumulates the wide channels in ascending 32-channel MFMA steps, exactly as conv1x1_stream / conv_igemm_wide do
//               for the unfused reduce conv: both tensors are bit-identical to the two-launch chain (tests/test_gpu_kernels.py,
//               tests/test_full_batch.py)
//   two loops   PIPE (Cmid = 128, Cn = 128: layer2 of ResNet-50, what the engine takes by default): 80 KB of LDS — Y, the pixel K-step and
//               THREE 16 KB weight slots that rotate so that every tile but stage A's second K-step is in flight a whole phase before it is
//               read (table at the loop); the folded-BN vectors are applied by the kernel before the tiles of stage B are issued (hipcc puts
//               a vmcnt(0) in front of the first use of any register load: inside epilogue_lite it would wait for those tiles).
//               Generic (any Cmid <= 512, Cn = 128 | 256): 64 KB, single-buffered K-steps; Cn = 256 (NB = 2, the 8x8 maps) needs 192
//               accumulator registers next to the epilogue's temporaries, spills 12 VGPRs and is SLOWER than the two launches (2.19 vs
//               1.86 ms at 16000 image-samples) — kept for the tests ("conv_seam" = 2 | 3), not selected.
//   registers   every per-lane address is re-derived per chunk from an opaque copy of threadIdx (SEAM_LANE_SETUP): hoisted out of the chunk
//               loop they were ~60 VGPRs next to the 128 accumulator registers (the first build spilled 34-58 VGPRs; now 206-218, none).
// Measured (MI355X, 16000 image-samples of 16x16 x 128 -> 512 -> 128, profiles/r05_resnet50_me_*): 2.24-2.29 ms per launch inside the ResNet-50
// step against 1.90-2.01 + 0.99 ms for the two launches; rocprofv3 FETCH + WRITE per launch 5.28 + 5.24 = 10.5 GB against 9.45 + 5.25 = 14.7 GB
// (without the non-temporal hints on the residual DMA and the wide stores 14.1 GB: the narrow input tile, re-fetched for each chunk, missed L2
// behind the streams); the step 55.8 -> 54.1 ms (same box).  What bounds it (same-box ablations, profiles/experiments/r5_seam_ablation.log):
// without the residual DMA and the stores 1.70 ms — the LDS pipe (480 KB of LDS traffic per chunk and workgroup: 64 x 64 wave tiles read 8 KB
// per 16 MFMAs) and the chain of nine barriers per chunk; HBM adds ~0.6 ms on top because two workgroups per CU overlap the two poorly.
// Taken for: expand = ksize 1, stride 1, Cmid % 64 == 0, Cmid <= 512, Cw % 128 == 0, a residual with one row per output row, ReLU,
// no site; reduce = ksize 1, stride 1, plain epilogue, Cn = 128 (256 under "conv_seam" >= 2) output channels (NB = Cn / 128).
#include "conv_epilogue.h"
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define GLDS16(SRC, LDSPTR)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),       \
                                     (__attribute__((address_space(3))) void*)(LDSPTR), 16, 0, 0)

#ifndef SEAM_ABL_NOMFMA
#define SEAM_ABL_NOMFMA 0    // timing probes (wrong results by construction): no MFMAs | no residual DMA | no stage B at all
#endif
#ifndef SEAM_ABL_NORESDMA
#define SEAM_ABL_NORESDMA 0
#endif
#ifndef SEAM_ABL_NOB
#define SEAM_ABL_NOB 0
#endif
#ifndef SEAM_WGS
#define SEAM_WGS 2
#endif
#ifndef SEAM_NT
#define SEAM_NT 1            // the residual DMA and the wide tensor's stores carry the non-temporal hint
#endif
#define GLDS16_NT(SRC, LDSPTR)                                                                  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),       \
                                     (__attribute__((address_space(3))) void*)(LDSPTR), 16, 0, SEAM_NT ? 2 : 0)
struct SeamArgs {
    ConvArgs a;   // expand conv: in = m, wgt = W3 [Cw][Cmid], res, out = y
    ConvArgs b;   // reduce conv: wgt = W1 [Cn][Cw], out = z (its `in` is a.out: never read from memory here)
};

template <bool BF, int NB, bool PIPE>
__global__ __launch_bounds__(256, SEAM_WGS) void conv1x1_seam_kernel(SeamArgs g) {
    const ConvArgs& a = g.a;
    const ConvArgs& b = g.b;
    constexpr int TJ = 2, SBP = 128;
    constexpr int OPB = 32768;
    static_assert(!PIPE || NB == 1, "the pipelined form: 128 narrow channels on both sides");
    __shared__ __attribute__((aligned(16))) char smem[OPB + BMI_EPILOGUE_LDS_BYTES / 2 + (PIPE ? 16384 : 0)];
    char* const wb0 = smem + OPB + BMI_EPILOGUE_LDS_BYTES / 2;      // PIPE: a third 16 KB weight slot
    char* const opd = smem;               // operand buffer
    char* const Y = smem + OPB;           // residual in / wide piece out / B operand of stage B / epilogue_plain's tile at the end
    constexpr int XBASE = 128 * 128;      // stage A: pixel tile behind the weight tile
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int wc = wave >> 1, wp = wave & 1;
    const int pix0 = blockIdx.x * SBP;
    const int Cw = a.Cout;
    const int nKa = a.Cin / 64;
    const int nchunk = Cw / 128;

    typedef float accv __attribute__((ext_vector_type(4)));
    accv accb[NB][4][4];
#pragma unroll
    for (int h = 0; h < NB; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) accb[h][i][j] = accv{0.f, 0.f, 0.f, 0.f};

    auto pixmap = [&](int p, int& n, int& rem) -> bool { n = 0; rem = 0; return pix0 + p < a.M; };   // (no site: never asked)
    auto offmap_a = [&](int p, size_t& off) -> bool { off = (size_t)(pix0 + p) * Cw; return pix0 + p < a.M; };
    auto offmap_b = [&](int p, size_t& off) -> bool { off = (size_t)(pix0 + p) * b.Cout; return pix0 + p < a.M; };

#define SEAM_ISSUE_A(KS)                                                                                         \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
            GLDS16(w3src + (size_t)(ch0 + 32 * i) * a.Cin + (KS) * 64, opd + (i * 256 + wave * 64) * 16);        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) GLDS16(xsrc[i] + (KS) * 64, opd + XBASE + (i * 256 + wave * 64) * 16); \
    }
    // stage-B weight tile of K-step KS' (64 wide channels of the chunk): rows = the NB x 128 output channels, 128-byte rows
#define SEAM_ISSUE_W1(KSB, BASE)                                                                                 \
    {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < 4 * NB; ++i)                                                       \
            GLDS16(w1src + (size_t)(32 * i) * Cw + ch0 + (KSB) * 64, (BASE) + (i * 256 + wave * 64) * 16);       \
    }
#define SEAM_MFMA_B(KSB, BASE)                                                                                   \
    {                                                                                                            \
        _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) {                                                    \
            const int coff = ((4 * sub + kq) ^ sw_r) << 4;                                                       \
            const int q = (KSB) * 8 + 4 * sub + kq;                                                              \
            half8 bf[4];                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
                bf[j] = *(const half8*)(Y + (wp * 64 + 16 * j + r) * 256 + ((q ^ r) << 4));                      \
            _Pragma("unroll") for (int h = 0; h < NB; ++h) {                                                     \
                half8 af[4];                                                                                     \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
                    af[i] = *(const half8*)((BASE) + (h * 128 + wc * 64 + 16 * i + r) * 128 + coff);             \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                    \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) if (!SEAM_ABL_NOMFMA || (i == 0 && j == 0)) accb[h][i][j] = mfma_16x16x32<BF>(af[i], bf[j], accb[h][i][j]); \
                __builtin_amdgcn_sched_barrier(0);      /* (fragments of one step at a time: the tile has 128-192 accumulator registers) */ \
            }                                                                                                    \
        }                                                                                                        \
    }

    // a 16-byte DMA piece q = tid + 256 i -> tile row 32 i + (tid >> 3), slot tid & 7 holds chunk slot ^ ((row >> 1) & 7) (source-side swizzle)
#define SEAM_LANE_SETUP()                                                                                        \
    int tid = tid0;                                                                                              \
    asm volatile("" : "+v"(tid));                                                                                \
    const int lane = tid & 63;                                                                                   \
    const int r = lane & 15, kq = lane >> 4;                                                                     \
    const int rowt = tid >> 3;                                                                                   \
    const int srcchunk = ((tid & 7) ^ ((rowt >> 1) & 7)) * 8;                                                    \
    const _Float16* xsrc[4];                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                              \
        const int m = pix0 + 32 * i + rowt;                                                                      \
        xsrc[i] = a.in + (size_t)(m < a.M ? m : 0) * a.Cin + srcchunk; /* rows beyond the tensor read pixel 0: computed, never stored */ \
    }                                                                                                            \
    const _Float16* const w3src = a.wgt + (size_t)rowt * a.Cin + srcchunk; /* + (ch0 + 32 i) * Cin + 64 ks */    \
    const _Float16* const w1src = b.wgt + (size_t)rowt * Cw + srcchunk;    /* + (32 i) * Cw + ch0 + 64 ks' */    \
    /* residual piece q = tid + 256 i (i < 8): pixel row p = 16 i + (tid >> 4), position tid & 15 holds chunk pos ^ (p & 15) */ \
    const int rp = tid >> 4;                                                                                     \
    const _Float16* const rsrc0 = a.res + (size_t)pix0 * Cw + (((tid & 15) ^ (rp & 15)) << 3);                   \
    const int a_off = (wc * 64 + r) * 128;                                                                       \
    const int b_off = XBASE + (wp * 64 + r) * 128;                                                               \
    const int sw_r = (r >> 1) & 7;
#define SEAM_ISSUE_RES()                                                                                         \
    if (SEAM_ABL_NORESDMA) {                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) GLDS16(rsrc0, Y + (i * 256 + wave * 64) * 16);             \
    } else if (a.M - pix0 >= SBP) {                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) GLDS16_NT(rsrc0 + (size_t)(16 * i + rp) * Cw + ch0, Y + (i * 256 + wave * 64) * 16); \
    } else { /* the last, partial tile: rows beyond the tensor read row 0 of the residual, never stored */       \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
            const int p = 16 * i + rp;                                                                           \
            GLDS16(pix0 + p < a.M ? rsrc0 + (size_t)p * Cw + ch0 : a.res + (((tid & 15) ^ (rp & 15)) << 3), Y + (i * 256 + wave * 64) * 16); \
        }                                                                                                        \
    }
#define SEAM_ISSUE_W3(CH0, KS, BASE)                                                                             \
    { _Pragma("unroll") for (int i = 0; i < 4; ++i) GLDS16(w3src + (size_t)((CH0) + 32 * i) * a.Cin + (KS) * 64, (BASE) + (i * 256 + wave * 64) * 16); }
#define SEAM_ISSUE_X(KS)                                                                                         \
    { _Pragma("unroll") for (int i = 0; i < 4; ++i) GLDS16(xsrc[i] + (KS) * 64, opd + XBASE + (i * 256 + wave * 64) * 16); }
#define SEAM_MFMA_A(WBASE)                                                                                       \
    {                                                                                                            \
        _Pragma("unroll") for (int sub = 0; sub < 2; ++sub) {                                                    \
            const int coff = ((4 * sub + kq) ^ sw_r) << 4;                                                       \
            half8 af[4], bf[4];                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) af[i] = *(const half8*)((WBASE) + a_off + i * 16 * 128 + coff); \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(opd + b_off + j * 16 * 128 + coff);      \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
                _Pragma("unroll") for (int j = 0; j < 4; ++j) if (!SEAM_ABL_NOMFMA || (i == 0 && j == 0)) acc[i][j] = mfma_16x16x32<BF>(af[i], bf[j], acc[i][j]); \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
        }                                                                                                        \
    }
#define SEAM_BARRIER()                                                                                           \
    {                                                                                                            \
        __builtin_amdgcn_s_barrier();                                                                            \
        asm volatile("" ::: "memory");                                                                           \
    }
    if constexpr (PIPE) {
        // Cmid = 128 (two K-steps in stage A), 128 narrow channels out: three 16 KB weight slots (W-half and wb0; the X-half holds the pixel
        // K-steps) rotate so that every tile but the second K-step's is in flight a whole phase before it is read:
        //   phase          reads                      issues at its end (into what the phase has just released)
        //   A, K-step 0    wb0 (W3 k0), X (k0)        W3 k1 -> W-half, X k1 -> X, W1 k'0 -> wb0
        //   A, K-step 1    W-half, X                  W1 k'1 -> W-half, X k0 (the same pixels again, for the next chunk) -> X
        //   epilogue A     Y (residual in, piece out)
        //   B, K-step 0    wb0, Y                     W3 k0 of the next chunk -> wb0
        //   B, K-step 1    W-half, Y
        // and the residual of a chunk is issued at its top.  vmcnt retires in issue order, so each wait names how many YOUNGER pieces may
        // still be in flight (4 per tile, 8 for the residual and for the epilogue's stores of a full tile).
        {
            const int ch0 = 0;
            SEAM_LANE_SETUP();
            (void)ch0; (void)kq; (void)w1src; (void)rsrc0; (void)rp; (void)a_off; (void)b_off; (void)sw_r;
            SEAM_ISSUE_X(0);
            SEAM_ISSUE_W3(0, 0, wb0);
        }
        const bool full = a.M - pix0 >= SBP;
        // The folded-BN vectors are register loads, and hipcc waits for EVERY outstanding vector-memory operation before the first use of one
        // (LDS-DMA and register loads share vmcnt): inside epilogue_lite that wait would sit right behind the tiles issued for stage B.  So
        // they are fetched at the top of the chunk and applied here, after stage A's last MFMA and before those tiles are issued — the same
        // expression as in epilogue_lite, which then runs with scale = bias = null (x * 1 + 0).
        ConvArgs an = a;
        an.scale = nullptr;
        an.bias = nullptr;
        an.out_mul = 1.f;
        for (int c = 0; c < nchunk; ++c) {
            const int ch0 = c * 128;
            SEAM_LANE_SETUP();
            f32x4_e bsc[4], bbi[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c4 = ch0 + wc * 64 + 16 * i + 4 * kq;
                bsc[i] = f32x4_e{1.f, 1.f, 1.f, 1.f};
                bbi[i] = f32x4_e{0.f, 0.f, 0.f, 0.f};
                if (a.scale) bsc[i] = *(const f32x4_e*)(a.scale + c4);
                if (a.bias) bbi[i] = *(const f32x4_e*)(a.bias + c4);
            }
            SEAM_ISSUE_RES();
            accv acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = accv{0.f, 0.f, 0.f, 0.f};
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // all but this chunk's residual
            SEAM_BARRIER();
            SEAM_MFMA_A(wb0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SEAM_BARRIER();
            SEAM_ISSUE_W3(ch0, 1, opd);
            SEAM_ISSUE_X(1);
            SEAM_ISSUE_W1(0, wb0);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // all but W1 k'0 (the residual too)
            SEAM_BARRIER();
            SEAM_MFMA_A(opd);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SEAM_BARRIER();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4_e sc = bsc[i], bi = bbi[i];
                sc *= a.out_mul;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][e] = acc[i][j][e] * sc[e] + bi[e];
            }
            __builtin_amdgcn_sched_barrier(0);
            SEAM_ISSUE_W1(1, opd);
            SEAM_ISSUE_X(0);
            // epilogue A; its wait (8 younger pieces allowed) also covers W1 k'0, published by its barrier
            epilogue_lite<TJ, BF, true, false, BMI_SITE_NONE, 8, SEAM_NT != 0>(an, acc, Y, tid, ch0, pixmap, offmap_a);
            if (!SEAM_ABL_NOB) SEAM_MFMA_B(0, wb0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (full) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // W1 k'1 has landed: younger are the pixels' K-step 0 and the eight stores
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (a partial tile stores fewer rows)
            SEAM_BARRIER();
            SEAM_ISSUE_W3((c + 1 < nchunk ? ch0 + 128 : 0), 0, wb0);        // (after the last chunk: a tile nobody reads, so that the counts stay the same)
            if (!SEAM_ABL_NOB) SEAM_MFMA_B(1, opd);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SEAM_BARRIER();                                                 // Y and the W-half are free
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else
    for (int c = 0; c < nchunk; ++c) {
        const int ch0 = c * 128;
        // Every per-lane address is re-derived per chunk from an opaque copy of the thread index: hoisted out of this loop they are
        // ~60 VGPRs next to the 128 accumulator registers of the two stages (the kernel spilled).
        SEAM_LANE_SETUP();
        // ---- stage A: the operand buffer and Y are free (end of the previous chunk) ----
        SEAM_ISSUE_A(0);
        SEAM_ISSUE_RES();
        accv acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = accv{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < nKa; ++ks) {
            // (vmcnt retires in order: the residual sits BEHIND the first K-step's operands and in front of the later ones)
            if (ks == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int coff = ((4 * sub + kq) ^ sw_r) << 4;
                half8 af[4], bf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const half8*)(opd + a_off + i * 16 * 128 + coff);
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *(const half8*)(opd + b_off + j * 16 * 128 + coff);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = mfma_16x16x32<BF>(af[i], bf[j], acc[i][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();             // every wave has read K-step ks: the buffer may be refilled
            asm volatile("" ::: "memory");
            if (ks + 1 < nKa) SEAM_ISSUE_A(ks + 1);
        }
        // stage B's weights stream in behind the epilogue (NB = 1: both K-steps, 2 x 16 KB; NB = 2: the first, 32 KB)
        SEAM_ISSUE_W1(0, opd);
        if constexpr (NB == 1) SEAM_ISSUE_W1(1, opd + 16384);
        // ---- epilogue A: BN, + residual (in Y), ReLU, fp16 piece in place in Y, coalesced stores (waits for every DMA above) ----
        epilogue_lite<TJ, BF, true, false, BMI_SITE_NONE>(a, acc, Y, tid, ch0, pixmap, offmap_a);
        // ---- stage B ----
        SEAM_MFMA_B(0, opd);
        if constexpr (NB == 1) {
            SEAM_MFMA_B(1, opd + 16384);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            SEAM_ISSUE_W1(1, opd);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            SEAM_MFMA_B(1, opd);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // Y and the operand buffer are free
        asm volatile("" ::: "memory");
    }
#undef SEAM_ISSUE_A
#undef SEAM_LANE_SETUP
#undef SEAM_ISSUE_RES
#undef SEAM_ISSUE_W3
#undef SEAM_ISSUE_X
#undef SEAM_MFMA_A
#undef SEAM_BARRIER
#undef SEAM_ISSUE_W1
#undef SEAM_MFMA_B
#pragma unroll
    for (int h = 0; h < NB; ++h) epilogue_plain<TJ, 16, BF>(b, accb[h], Y, tid0, h * 128, offmap_b);
}

// (256 narrow channels — NB = 2, the 8x8 maps of ResNet-50 — need 192 accumulator registers: the kernel spills and has measured slower than
//  the two launches, 2.19 vs 1.86 ms at 16000 image-samples; the engine takes it under "conv_seam" = 2 | 3 only: tests)
bool conv_takes_seam_kernel(int cmid, int cw, int cn) {
    return cmid % 64 == 0 && cmid <= 512 && cw % 128 == 0 && (cn == 128 || (cn == 256 && opt_conv_seam() >= 2));
}

// BMI_ERR_UNSUPPORTED -> the caller issues the two launches.
int launch_conv1x1_seam(const ConvArgs& a, const ConvArgs& b, hipStream_t s) {
    if (!opt_conv_seam()) return BMI_ERR_UNSUPPORTED;
    auto one_by_one = [](const ConvArgs& c) { return c.ksize == 1 && c.stride == 1 && c.pad == 0 && !c.wgt_b && !c.in2 && !c.imap && !c.in_bits && !c.pool && !c.partial; };
    if (!one_by_one(a) || !one_by_one(b)) return BMI_ERR_UNSUPPORTED;
    if (!conv_takes_seam_kernel(a.Cin, a.Cout, b.Cout) || b.Cin != a.Cout || b.in != a.out || b.M != a.M || a.bf16 != b.bf16) return BMI_ERR_UNSUPPORTED;
    if (!a.res || a.res_mod < a.N || a.in_mod < a.N || !a.relu || a.site.kind != BMI_SITE_NONE || a.site_inner) return BMI_ERR_UNSUPPORTED;
    if (b.res || b.site.kind != BMI_SITE_NONE || b.site_inner || b.out_mul != 1.f) return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.M <= 0) return BMI_ERR_INVALID;
    static const int n_cu = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cu = 0;
        return cu;
    }();
    // the minimum-grid rule of the other kernels (on the engine's full-chunk image count, never this launch's): two workgroups per CU
    const long tiles = ((long)a.M + 127) / 128;
    const long tiles_sel = a.n_ref > 0 ? ((long)a.n_ref * a.Ho * a.Wo + 127) / 128 : tiles;
    if (opt_conv_seam() < 2 && tiles_sel < (n_cu > 0 ? 4 * n_cu : 1024)) return BMI_ERR_UNSUPPORTED;
    SeamArgs g;
    g.a = a;
    g.b = b;
    const dim3 grid((unsigned)tiles), block(256);
#define SEAM_LAUNCH(BF_)                                                                                          \
    {                                                                                                             \
        if (b.Cout == 128 && a.Cin == 128 && opt_conv_seam() != 3) hipLaunchKernelGGL((conv1x1_seam_kernel<BF_, 1, true>), grid, block, 0, s, g); \
        else if (b.Cout == 128) hipLaunchKernelGGL((conv1x1_seam_kernel<BF_, 1, false>), grid, block, 0, s, g);   \
        else hipLaunchKernelGGL((conv1x1_seam_kernel<BF_, 2, false>), grid, block, 0, s, g);                      \
    }
    if (a.bf16) SEAM_LAUNCH(true) else SEAM_LAUNCH(false)
#undef SEAM_LAUNCH
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}
