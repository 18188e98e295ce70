// Exit head, fused: relu -> global average pool -> [site] -> Linear -> [site on the logits] -> softmax -> T-sample moments
// in ONE kernel per exit (ex{1,2,3}linear / linear with their avg_pool2d / exit dropout, SA/models/resnet18/resnet18.py:
// 309-314, :320-325, :331-335, :338-344; the softmax and the np.average over the passes of FullAnalysis._get_output,
// SA/train/results_analyzer.py:242-248).  Round 1 ran this as pool_mask + linear_softmax + a per-sample [E][N][C]
// probability / logit scratch + a moments kernel (9 launches per chunk); here nothing per-sample is materialised.
//
//   workgroup   = one image b of the batch x one group of 32 of the launch's samples, 256 threads (one workgroup per image
//                 walking all groups left 250 workgroups of latency-bound work on 256 CUs: 2x slower than the three kernels
//                 it replaced); the groups of an image leave float64 partial sums in a workspace scratch that head_join_kernel
//                 adds into S1/S2/SL[b][:] in group order (round 3: bit-reproducible; round 2 used hardware float64 atomics,
//                 whose order varied from run to run).
//   phase A     = pooling, coalesced: wave w pools columns (samples) w, w+4, .. of the group, lane = one 8-channel group
//                 (16 B per pixel row: 64 lanes cover a 512-channel row = 1 KB contiguous), ReLU + mean in fp32, the
//                 feature-side site (MC dropout / Masksembles1D on the [B, K] tensor), then fp32 into LDS
//                 feat[32 samples][KC] with the 16-byte chunks XOR-swizzled by the sample index.
//   phase B     = logits[class][sample] on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), classes on the row axis, the 32
//                 samples on the column axis; the 4 waves split K, partial sums meet in LDS.
//   finalize    = wave 0: bias, [dropout on the logits], softmax in registers (one cross-half shuffle) -> LDS [class][sample];
//                 then all waves: the sums over the group's samples of p, p^2 and logit by WAVEFRONT SHUFFLE REDUCTION in
//                 float64 (32 lanes = 32 samples of one class; the reference averages in float64; per-sample values are
//                 bit-identical however the samples are chunked or sharded, so only the float64 summation order depends
//                 on it: <= 1e-13 relative).
#include <cstdlib>

#include "conv_epilogue.h"
#include "kernels.h"

typedef float f32x16_h __attribute__((ext_vector_type(16)));
typedef float f32x4_h __attribute__((ext_vector_type(4)));
typedef _Float16 half8_h __attribute__((ext_vector_type(8)));

#define HEAD_KC 512                       // K chunk held in LDS (floats per sample)
#define HEAD_FEAT_BYTES (32 * HEAD_KC * 4)

// KIND: 0 fp16, 1 fp32, 2 bf16 input tensor; 3 | 4: a pair32 tensor of the split engines (conv_epilogue.h), fp16 | bf16 halves
// CSPLIT (RT >= 3 class tiles, K % 32 == 0; round 2): the four waves split the CLASSES instead of K — wave w owns class tile w
// for the whole K — and the classifier weights go through LDS.  Why: with the K split every wave holds partial sums of all RT
// tiles (64 accumulators at RT = 4) and they meet in a [4][RT][16][64] fp32 LDS array (64 KB) next to the 64 KB feature chunk:
// 128 KB of LDS and 231 + 64 registers = ONE workgroup of four waves per CU, and ablation builds showed where that hurts: of
// the 330 us of a 100-class head at T = 100 the pooling pass alone was 200 (four waves do not keep enough loads in flight).
// Here a wave keeps 16 accumulators, nothing is exchanged (the logits tile goes straight to the [class][sample] array the
// softmax reads), the feature chunk is 256 deep (32 KB) and a wave stages its [32 classes][32 k] weight blocks in 4.5 KB of
// its own (8 lanes per 128-byte row segment instead of one 2 KB row per lane): 66 KB of LDS, two workgroups per CU.
// The kernel body, shared by the one-head launch and the batched one (head_fused_multi_kernel: blockIdx.z = which head of the pack).
template <int RT, int KIND, bool CSPLIT>
__device__ __forceinline__ void head_body(const HeadArgs& a) {
    static_assert(!CSPLIT || (RT >= 3 && RT <= 4), "class split: one wave per class tile");
    constexpr int KC = CSPLIT ? 256 : HEAD_KC;                 // K chunk held in LDS (floats per sample)
    constexpr int FEAT_BYTES = 32 * KC * 4;
    // behind the features: K split -> the partial sums [4 waves][RT][16 regs][64 lanes]; class split -> 4 wave-private weight
    // blocks [32][36] during the K loop, then (both) the [class][33] softmax / logit arrays
    constexpr int PART_BYTES = CSPLIT ? 2 * 32 * RT * 33 * 4 : 4 * RT * 16 * 64 * 4;
    static_assert(!CSPLIT || PART_BYTES >= 4 * 32 * 36 * 4, "weight blocks alias the softmax arrays");
    __shared__ __attribute__((aligned(16))) char smem[FEAT_BYTES + PART_BYTES];
    float* const feat = (float*)smem;
    float* const part = (float*)(smem + FEAT_BYTES);

    const int b = a.imap ? a.imap[blockIdx.x] : (int)blockIdx.x;    // dynamic early exit: only the still-active images
    const int g = blockIdx.y;                                  // this workgroup's group of 32 samples
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int K = a.K, C = a.C;
    const float inv_hw = 1.0f / (float)a.HW;
    {
        f32x16_h acc[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

        f32x16_h accw[2];                                      // CSPLIT: this wave's class tile, two interleaved chains
#pragma unroll
        for (int e = 0; e < 16; ++e) { accw[0][e] = 0.f; accw[1][e] = 0.f; }
        for (int k0 = 0; k0 < K; k0 += KC) {
            const int kc = min(KC, K - k0);                    // multiple of 32
            const int swz = (kc & 63) == 0 ? 15 : 7;          // the XOR must stay inside the row's kc / 4 chunks
            // this wave's slice of the classifier weights, two class tiles (64 classes) at a time: the first pair is requested
            // BEFORE the pooling pass (its L2 latency hides under phase A), 16 float4 per class tile and lane
            const int kq = kc >> 3;                            // k per (wave, half): multiple of 4, <= 64
            const int koff = wave * (kc >> 2) + hh * kq;
            const float* wp = a.w + (size_t)r * K + k0 + koff;
            constexpr int NPAIR = (RT + 1) / 2, W2 = RT < 2 ? RT : 2;
            f32x4_h wpre[CSPLIT ? 1 : W2][CSPLIT ? 1 : 16];
            f32x4_h wst[CSPLIT ? 4 : 1];                       // CSPLIT: the next [32 classes][32 k] block, 4 float4 per lane
#define HEAD_FETCH_W(CK)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                                \
        const int f_ = lane + 64 * i_, row_ = f_ >> 3, kq_ = f_ & 7;                                                  \
        wst[i_] = *(const f32x4_h*)(a.w + (size_t)(32 * wave + row_) * K + k0 + 32 * (CK) + 4 * kq_);                 \
    }
#define HEAD_LOAD_W(PS)                                                                                              \
    _Pragma("unroll") for (int ii = 0; ii < W2; ++ii)                                                                 \
        _Pragma("unroll") for (int q = 0; q < 16; ++q)                                                                \
            wpre[ii][q] = (2 * (PS) + ii < RT && 4 * q < kq) ? *(const f32x4_h*)(wp + (size_t)(32 * (2 * (PS) + ii)) * K + 4 * q) \
                                                             : f32x4_h{0.f, 0.f, 0.f, 0.f};
            if constexpr (CSPLIT) { if (wave < RT) { HEAD_FETCH_W(0) } } else { HEAD_LOAD_W(0) }
            // ---- phase A: pool 8 samples per wave into LDS ----
            // deterministic input (in_mod == B: exit-only dropout, the image's features are the same for every sample): pooled
            // once per wave and chunk, then only the site differs per sample (VGG-19 multi-exit: 0.16 -> 0.0x ms per head)
            const bool det = a.in_mod == a.B;
            float pooled[8];
            bool have = false;
            for (int jj = 0; jj < (CSPLIT ? 4 : 8); ++jj) {
                // interleaved: a launch with few samples (T = 8) still uses all waves.  CSPLIT (256-deep chunk = 32 lanes of 8
                // channels): the two lane halves pool two samples at once
                const int j = CSPLIT ? jj * 8 + wave * 2 + (lane >> 5) : jj * 4 + wave;
                const int tl = g * 32 + j;
                const int c8 = CSPLIT ? (lane & 31) : lane;
                if (c8 * 8 < kc) {
                    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    if (tl < a.tc) {
                        const int n = tl * a.B + b;
                        const size_t row0 = (size_t)(n % a.in_mod) * a.HW * K + k0 + c8 * 8;
                        if (det && have) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = pooled[e];
                        } else {
#pragma unroll 8
                        for (int p = 0; p < a.HW; ++p) {
                            if constexpr (KIND >= 3) {
                                float x[8];
                                pair_decode<KIND == 4, 8>((const _Float16*)a.in + pair32_off((size_t)(n % a.in_mod) * a.HW + p, K, k0 + c8 * 8), x);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += fmaxf(x[e], 0.f);
                            } else if constexpr (KIND == 1) {
                                const float* src = (const float*)a.in + row0 + (size_t)p * K;
                                const f32x4_h x0 = *(const f32x4_h*)src, x1 = *(const f32x4_h*)(src + 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v[e] += fmaxf(x0[e], 0.f); v[4 + e] += fmaxf(x1[e], 0.f); }
                            } else {
                                const half8_h x = *(const half8_h*)((const _Float16*)a.in + row0 + (size_t)p * K);
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += fmaxf(a16_to_f32<KIND == 2>(x[e]), 0.f);   // F.relu before the pool
                            }
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) { v[e] *= inv_hw; pooled[e] = v[e]; }
                        have = true;
                        }
                        const int t = a.t0 + tl;
                        const int kk = k0 + c8 * 8;
                        if (a.site.kind == BMI_SITE_ELEMENTWISE || a.site.kind == BMI_SITE_CHANNEL) {
                            // [B, K] tensor: element = b*K + k (a per-(image, channel) draw is the same thing here)
                            const uint32_t keep = site_keep8(a.site, (uint64_t)b * K + kk, (uint32_t)t);
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = ((keep >> e) & 1u) ? v[e] * a.site.scale : 0.f;
                        } else if (a.site.kind == BMI_SITE_MASKSEMBLE) {
                            const float* mrow = a.site.masks + (size_t)((a.site.cnt0 + t) % a.site.num_masks) * K + kk;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] *= mrow[e];
                        }
                    }
                    float* dst = feat + j * kc;
                    *(f32x4_h*)(dst + (((2 * c8) ^ (j & swz)) << 2)) = f32x4_h{v[0], v[1], v[2], v[3]};
                    *(f32x4_h*)(dst + (((2 * c8 + 1) ^ (j & swz)) << 2)) = f32x4_h{v[4], v[5], v[6], v[7]};
                }
            }
            __syncthreads();
            // ---- phase B: this wave's quarter of the chunk's K, lane half hh takes half of that ----
            if constexpr (CSPLIT) {
                if (wave < RT) {                               // (wave-uniform)
                    const float* fr = feat + r * kc;
                    float* const Wt = part + wave * (32 * 36); // wave-private [32 classes][36]; the LDS operations of a wave are in order
                    const int nck = kc >> 5;
                    for (int ck = 0; ck < nck; ++ck) {
#pragma unroll
                        for (int i_ = 0; i_ < 4; ++i_) {
                            const int f_ = lane + 64 * i_;
                            *(f32x4_h*)(Wt + (f_ >> 3) * 36 + 4 * (f_ & 7)) = wst[i_];
                        }
                        if (ck + 1 < nck) { HEAD_FETCH_W(ck + 1) }
                        f32x4_h aq[4], bq[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bq[q] = *(const f32x4_h*)(fr + ((((32 * ck + 16 * hh + 4 * q) >> 2) ^ (r & swz)) << 2));
                            aq[q] = *(const f32x4_h*)(Wt + r * 36 + 16 * hh + 4 * q);
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) accw[e & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[q][e], bq[q][e], accw[e & 1], 0, 0, 0);
                    }
                }
            } else {
                const float* fr = feat + r * kc;
#pragma unroll
                for (int ps = 0; ps < NPAIR; ++ps) {
                    if (ps > 0) { HEAD_LOAD_W(ps) }            // later pairs (C > 64): loaded when their turn comes
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        if (4 * q < kq) {
                            const f32x4_h b4 = *(const f32x4_h*)(fr + ((((koff + 4 * q) >> 2) ^ (r & swz)) << 2));
#pragma unroll
                            for (int ii = 0; ii < W2; ++ii) {
                                if (2 * ps + ii < RT) {
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        acc[2 * ps + ii] = __builtin_amdgcn_mfma_f32_32x32x2f32(wpre[ii][q][e], b4[e], acc[2 * ps + ii], 0, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
#undef HEAD_LOAD_W
#undef HEAD_FETCH_W
            __syncthreads();                                   // feat is free for the next chunk
        }
        float* const pb_p = part;                              // [class][33]: softmax of the group's 32 samples (aliases `part`:
        float* const pb_l = part + 32 * RT * 33;               //  wave 0 has read all of it before it writes) and their logits
        if constexpr (CSPLIT) {
            // every wave's raw logits tile -> pb_l (the weight blocks it aliases are dead: barrier first)
            __syncthreads();
            if (wave < RT) {
#pragma unroll
                for (int e = 0; e < 16; ++e) pb_l[(32 * wave + (e & 3) + 8 * (e >> 2) + 4 * hh) * 33 + r] = accw[0][e] + accw[1][e];
            }
        } else {
        // ---- the four K-quarters meet in LDS ----
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) part[((wave * RT + i) * 16 + e) * 64 + lane] = acc[i][e];
        }
        __syncthreads();
        if (wave == 0) {
            const int tl = g * 32 + r;
            const uint32_t t = (uint32_t)(a.t0 + tl);
            // (three plain passes: with the partial sums, the bias / logits-site code and the running max in ONE loop body
            //  hipcc gave up unrolling it for 4 class tiles and put the accumulators in scratch)
            if constexpr (CSPLIT) {
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][e] = pb_l[(32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh) * 33 + r];
            } else {
#pragma unroll
            for (int w = 1; w < 4; ++w)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][e] += part[((w * RT + i) * 16 + e) * 64 + lane];
            }
            const bool drop_logits = a.site_logits.kind == BMI_SITE_ELEMENTWISE;
            // bias (registers e of class tile i = classes 32*i + (e & 3) + 8*(e >> 2) + 4*hh)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (c < C) acc[i][e] += a.bias[c];
                }
            if (drop_logits) {
                // dropout on the logits (converter/pytorch wraps the last Linear too, nn2bnn.py:33-45): [B, C] tensor, element =
                // b*C + c.  Through LDS in a ROLLED loop over this lane's class quads: unrolled over 4 class tiles the two Philox
                // calls per quad pushed the loop past hipcc's unroll budget and the accumulators into scratch (320 B per lane,
                // the C = 100 head ran 107 us instead of 30).
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int c = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        if (c < C) pb_l[c * 33 + r] = acc[i][e];
                    }
#pragma unroll 1
                for (int c4 = 4 * hh; c4 < C; c4 += 8) {
                    const uint64_t elem = (uint64_t)(b + a.b0) * C + c4;  // any alignment: b * C need not be a multiple of 4 (b0: first image of an image-partitioned launch)
                    const uint32_t sh = (uint32_t)(elem & 7);
                    uint32_t keep = site_keep8(a.site_logits, elem & ~(uint64_t)7, t) >> sh;
                    if (sh > 4) keep |= site_keep8(a.site_logits, (elem & ~(uint64_t)7) + 8, t) << (8 - sh);   // the quad straddles two calls
                    for (int e = 0; e < 4; ++e) {
                        if (c4 + e < C) {
                            const float v = pb_l[(c4 + e) * 33 + r];
                            pb_l[(c4 + e) * 33 + r] = ((keep >> e) & 1u) ? v * a.site_logits.scale : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int c = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        if (c < C) acc[i][e] = pb_l[c * 33 + r];
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (c < C) mx = fmaxf(mx, acc[i][e]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    const float ex = c < C ? expf(acc[i][e] - mx) : 0.f;
                    sum += ex;
                    if (c < C) { pb_l[c * 33 + r] = acc[i][e]; pb_p[c * 33 + r] = ex; }
                }
            sum += __shfl_xor(sum, 32);
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int c = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hh;
                    if (c < C) pb_p[c * 33 + r] = pb_p[c * 33 + r] / sum;      // same quotient the per-sample path stored
                }
        }
        __syncthreads();
        // ---- moments: 32 lanes = the group's 32 samples of one class; float64 wavefront-shuffle butterfly; two classes per
        //      wave and step; one hardware f64 atomic add per (class, quantity) joins the image's groups ----
        {
            const int ts = lane & 31;
            const bool live = g * 32 + ts < a.tc;
            for (int c0 = 0; c0 < C; c0 += 8) {
                const int c = c0 + wave * 2 + (lane >> 5);
                const bool ok = live && c < C;
                const double p = ok ? (double)pb_p[c * 33 + ts] : 0.0;
                double s1 = p, s2 = p * p, sl = ok ? (double)pb_l[c * 33 + ts] : 0.0;
                // per-sample logits out (bmi_forward_mcd_samples: what the reference's evaluate() consumes pass by pass)
                if (a.logits && ok) a.logits[(size_t)(g * 32 + ts) * a.logits_tstride + (size_t)b * C + c] = pb_l[c * 33 + ts];
                if (!a.S1) continue;
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) {
                    s1 += __shfl_xor(s1, m);
                    s2 += __shfl_xor(s2, m);
                    sl += __shfl_xor(sl, m);
                }
                if (ts == 0 && c < C) {
                    const size_t o = (size_t)b * C + c;
                    if (a.part) {                           // several groups per image: partial sums, joined in group order by head_join_kernel
                        const size_t plane = (size_t)a.B * C;
                        double* const pp = a.part + (size_t)g * 3 * plane + o;
                        pp[0] = s1; pp[plane] = s2; pp[2 * plane] = sl;
                    } else if (gridDim.y == 1) {            // the only writer of this address in the launch
                        a.S1[o] += s1; a.S2[o] += s2; a.SL[o] += sl;
                    } else {                                // (single-kernel entry point without a scratch: order varies from run to run)
                        unsafeAtomicAdd(a.S1 + o, s1);
                        unsafeAtomicAdd(a.S2 + o, s2);
                        unsafeAtomicAdd(a.SL + o, sl);
                    }
                }
            }
        }
    }
}

template <int RT, int KIND, bool CSPLIT = false>
__global__ __launch_bounds__(256, CSPLIT ? 2 : 1) void head_fused_kernel(HeadArgs a) {
    head_body<RT, KIND, CSPLIT>(a);
}

// Several exit heads in ONE launch (round 6): with exit-only dropout — the configuration every run of the paper uses,
// Software_Artifact/script_figs/journal_script.sh:10-63 — the whole network is the once-per-batch prefix and the sample-folded suffix is
// NOTHING BUT the four (VGG-19: five) heads, each a ~60 us launch of mostly fixed latency at T = 10: grid.z walks the pack, every workgroup
// runs the one-head body on its head's arguments (same arithmetic, same bits; tests/test_full_batch.py).  The heads of a pack agree in
// everything the template parameters and the grid depend on (class tiles, input kind, images, samples); launch_head_fused_multi checks.
struct HeadArgsPack { HeadArgs a[BMI_HEAD_PACK_MAX]; };
template <int RT, int KIND, bool CSPLIT = false>
__global__ __launch_bounds__(256, CSPLIT ? 2 : 1) void head_fused_multi_kernel(HeadArgsPack p) {
    head_body<RT, KIND, CSPLIT>(p.a[blockIdx.z]);
}

// Joins the per-group partial sums of an image in GROUP ORDER into the caller's accumulators: with hardware float64 atomics the
// groups met in whatever order the workgroups finished, and the last bit of the sums of more than 64 samples changed from run to
// run (round-2 verdict); an ordered "last arriver adds all" reduction inside the head kernel needed agent-scope fences that doubled
// it.  This is one more launch of B x C threads per exit and chunk (~3 us), only when a launch carries more than 32 samples.
__global__ __launch_bounds__(256) void head_join_kernel(const double* __restrict__ part, int groups, int B, int C, const int* imap, int Bc,
                                                        double* S1, double* S2, double* SL) {
    const int rows = imap ? Bc : B;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const int row = i / C, c = i - row * C;
    const int b = imap ? imap[row] : row;
    const size_t o = (size_t)b * C + c, plane = (size_t)B * C;
    double s1 = 0.0, s2 = 0.0, sl = 0.0;
    for (int g = 0; g < groups; ++g) {
        const double* pp = part + (size_t)g * 3 * plane + o;
        s1 += pp[0]; s2 += pp[plane]; sl += pp[2 * plane];
    }
    S1[o] += s1; S2[o] += s2; SL[o] += sl;
}

struct HeadJoinPack { const double* part[BMI_HEAD_PACK_MAX]; double *S1[BMI_HEAD_PACK_MAX], *S2[BMI_HEAD_PACK_MAX], *SL[BMI_HEAD_PACK_MAX]; };
__global__ __launch_bounds__(256) void head_join_multi_kernel(HeadJoinPack p, int groups, int B, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int z = blockIdx.y;
    const size_t o = (size_t)i, plane = (size_t)B * C;
    double s1 = 0.0, s2 = 0.0, sl = 0.0;
    for (int g = 0; g < groups; ++g) {
        const double* pp = p.part[z] + (size_t)g * 3 * plane + o;
        s1 += pp[0]; s2 += pp[plane]; sl += pp[2 * plane];
    }
    p.S1[z][o] += s1; p.S2[z][o] += s2; p.SL[z][o] += sl;
}

template <int RT>
static void launch_rt_multi(const HeadArgsPack& p, int n, hipStream_t s) {
    const HeadArgs& a = p.a[0];
    const dim3 grid((unsigned)a.B, (unsigned)((a.tc + 31) / 32), (unsigned)n), block(256);
#define HEAD_LAUNCH_M(CS)                                                                                                 \
    switch (a.in_kind) {                                                                                                  \
        case 1: hipLaunchKernelGGL((head_fused_multi_kernel<RT, 1, CS>), grid, block, 0, s, p); break;                    \
        case 2: hipLaunchKernelGGL((head_fused_multi_kernel<RT, 2, CS>), grid, block, 0, s, p); break;                    \
        case 3: hipLaunchKernelGGL((head_fused_multi_kernel<RT, 3, CS>), grid, block, 0, s, p); break;                    \
        case 4: hipLaunchKernelGGL((head_fused_multi_kernel<RT, 4, CS>), grid, block, 0, s, p); break;                    \
        default: hipLaunchKernelGGL((head_fused_multi_kernel<RT, 0, CS>), grid, block, 0, s, p); break;                   \
    }
    if constexpr (RT >= 3) {
        static const int csplit = [] { const char* v = std::getenv("BMI_HEAD_CSPLIT"); return v ? std::atoi(v) : 1; }();
        if (csplit) {
            HEAD_LAUNCH_M(true)
            return;
        }
    }
    HEAD_LAUNCH_M(false)
#undef HEAD_LAUNCH_M
}

template <int RT>
static void launch_rt(const HeadArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)(a.imap ? a.Bc : a.B), (unsigned)((a.tc + 31) / 32)), block(256);
#define HEAD_LAUNCH(CS)                                                                                                   \
    switch (a.in_kind) {                                                                                                  \
        case 1: hipLaunchKernelGGL((head_fused_kernel<RT, 1, CS>), grid, block, 0, s, a); break;                          \
        case 2: hipLaunchKernelGGL((head_fused_kernel<RT, 2, CS>), grid, block, 0, s, a); break;                          \
        case 3: hipLaunchKernelGGL((head_fused_kernel<RT, 3, CS>), grid, block, 0, s, a); break;                          \
        case 4: hipLaunchKernelGGL((head_fused_kernel<RT, 4, CS>), grid, block, 0, s, a); break;                          \
        default: hipLaunchKernelGGL((head_fused_kernel<RT, 0, CS>), grid, block, 0, s, a); break;                         \
    }
    if constexpr (RT >= 3) {
        static const int csplit = [] { const char* v = std::getenv("BMI_HEAD_CSPLIT"); return v ? std::atoi(v) : 1; }();
        if (csplit) {
            HEAD_LAUNCH(true)
            return;
        }
    }
    HEAD_LAUNCH(false)
#undef HEAD_LAUNCH
}

static int head_prepare(HeadArgs& a) {
    const int groups = (a.tc + 31) / 32;
    if (groups <= 1) a.part = nullptr;             // one group per image: the workgroup adds into S1 / S2 / SL itself
    if (!a.in || !a.w || !a.bias) return BMI_ERR_INVALID;
    if (!a.S1 || !a.S2 || !a.SL) { if (!a.logits) return BMI_ERR_INVALID; a.S1 = a.S2 = a.SL = nullptr; a.part = nullptr; }     // logits only
    if (a.B <= 0 || a.tc <= 0 || a.in_mod <= 0 || a.HW <= 0 || a.C <= 0 || a.in_kind < 0 || a.in_kind > 4) return BMI_ERR_INVALID;
    if (a.in_mod != a.B && a.in_mod != a.B * a.tc) return BMI_ERR_INVALID;
    if (a.imap && (a.Bc <= 0 || a.Bc > a.B)) return BMI_ERR_INVALID;
    if (a.K % 32 != 0 || a.C > 128) return BMI_ERR_UNSUPPORTED;
    if (a.site_logits.kind != BMI_SITE_NONE && a.site_logits.kind != BMI_SITE_ELEMENTWISE) return BMI_ERR_UNSUPPORTED;
    return BMI_OK;
}

int launch_head_fused(const HeadArgs& a_in, hipStream_t s) {
    HeadArgs a = a_in;
    const int groups = (a.tc + 31) / 32;
    const int rcp = head_prepare(a);
    if (rcp != BMI_OK) return rcp;
    switch ((a.C + 31) / 32) {
        case 1: launch_rt<1>(a, s); break;
        case 2: launch_rt<2>(a, s); break;
        case 3: launch_rt<3>(a, s); break;
        default: launch_rt<4>(a, s); break;
    }
    BMI_CHECK_LAUNCH();
    if (a.part) {
        const int n = (a.imap ? a.Bc : a.B) * a.C;
        hipLaunchKernelGGL(head_join_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.part, groups, a.B, a.C, a.imap, a.Bc, a.S1,
                           a.S2, a.SL);
        BMI_CHECK_LAUNCH();
    }
    return BMI_OK;
}

// n heads in one launch (n <= BMI_HEAD_PACK_MAX); BMI_ERR_UNSUPPORTED when the pack is not uniform (the caller launches them one by one).
// Every head needs its OWN partial-sum scratch (`part`): they run concurrently.
int launch_head_fused_multi(const HeadArgs* list, int n, hipStream_t s) {
    if (!list || n < 1 || n > BMI_HEAD_PACK_MAX) return BMI_ERR_INVALID;
    if (n == 1) return launch_head_fused(list[0], s);
    HeadArgsPack p;
    for (int i = 0; i < n; ++i) {
        p.a[i] = list[i];
        const int rcp = head_prepare(p.a[i]);
        if (rcp != BMI_OK) return rcp;
        const HeadArgs &x = p.a[i], &y = p.a[0];
        if (x.imap || x.C != y.C || x.in_kind != y.in_kind || x.B != y.B || x.tc != y.tc || (x.part != nullptr) != (y.part != nullptr) ||
            (x.S1 != nullptr) != (y.S1 != nullptr))
            return BMI_ERR_UNSUPPORTED;
        for (int j = 0; j < i; ++j)
            if (x.part && x.part == p.a[j].part) return BMI_ERR_INVALID;
    }
    for (int i = n; i < BMI_HEAD_PACK_MAX; ++i) p.a[i] = p.a[0];
    const HeadArgs& a = p.a[0];
    switch ((a.C + 31) / 32) {
        case 1: launch_rt_multi<1>(p, n, s); break;
        case 2: launch_rt_multi<2>(p, n, s); break;
        case 3: launch_rt_multi<3>(p, n, s); break;
        default: launch_rt_multi<4>(p, n, s); break;
    }
    BMI_CHECK_LAUNCH();
    if (a.part) {
        HeadJoinPack j;
        for (int i = 0; i < BMI_HEAD_PACK_MAX; ++i) {
            const HeadArgs& x = p.a[i < n ? i : 0];
            j.part[i] = x.part; j.S1[i] = x.S1; j.S2[i] = x.S2; j.SL[i] = x.SL;
        }
        const int groups = (a.tc + 31) / 32, m = a.B * a.C;
        hipLaunchKernelGGL(head_join_multi_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)n), dim3(256), 0, s, j, groups, a.B, a.C);
        BMI_CHECK_LAUNCH();
    }
    return BMI_OK;
}
