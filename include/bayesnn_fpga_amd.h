/*
 * bayesnn_fpga_amd.h — C ABI of libbayesnn_fpga_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for ONE path of os-hxfan/BayesNN_FPGA: the Monte-Carlo-dropout /
 * Masksembles multi-exit inference forward pass and its T-sample per-exit moment
 * reduction.  The reference has no FFI of its own: its boundary is the
 * Python API of Software_Artifact/software (abbreviated SA/), so every entry point below
 * cites the reference code it replaces:
 *
 *   bmi_create / bmi_plan      the layer graph that SA/models/resnet18/resnet18.py:260-300
 *                              (ResNet18MCEarlyExit.__init__), :212-244 (ResNet18MC) and
 *                              SA/models/vgg19/vgg19.py:327-382 build as nn.Modules
 *   bmi_forward_mcd            the T-pass loop of FullAnalysis._get_output,
 *                              SA/train/results_analyzer.py:236-246: `for i in range(mc_passes):
 *                              output = self.model(b_x)` + per-exit softmax, i.e. T x
 *                              ResNet18MCEarlyExit.forward (resnet18.py:302-346) with
 *                              MCDropout.forward (:207-210) / Masksembles*.forward
 *                              (SA/utils.py:156-169, :218-231) at every stochastic site
 *   bmi_finalize               np.average over the T passes, results_analyzer.py:247-248, plus
 *                              the build-defined T-sample variance (ddof = 0)
 *   bmi_philox_mask            the RNG primitive under F.dropout (resnet18.py:210), replaced by
 *                              the counter-based Philox convention of csrc/philox.h
 *   bmi_stem_conv_fwd, bmi_conv_igemm_fwd   conv+BN(+residual)(+ReLU) of BasicBlock.forward
 *                              (resnet18.py:32-48) and of the exit heads (:306-308,:318-319,:329)
 *   bmi_mask_apply, bmi_mask_bits   MCDropout / Masksembles2D on a stage output (:278-280)
 *   bmi_head_fused             one exit head end to end: F.avg_pool2d(F.relu(.),4) + flatten + exit dropout (:309-313),
 *                              ex{1,2,3}linear / linear (:314,:325,:335,:344), softmax (results_analyzer.py:242) and
 *                              the accumulation behind np.average over the passes (:247-248)
 *   bmi_dense_f32              hidden Dense layers of the VGG-11 classifier stack (Keras definition, see below)
 *
 * Conventions: plain pointers and sizes only; every function returns 0 or a negative
 * errno-style code (no exceptions cross the ABI); all device buffers are owned by the
 * caller; every launch is asynchronous on the caller's hipStream_t (passed as void*);
 * no allocation or synchronisation happens inside a launch function, so a caller may
 * capture bmi_forward_mcd into a hipGraph.
 *
 * Threading: one host thread per engine handle at a time (a handle carries the state of the call in flight); different handles —
 * one per GPU, or the two / three "batches in flight" of one GPU — may be driven from different threads concurrently.  The
 * kernel-selection switches of bmi_set_option are process DEFAULTS that bmi_create copies into the handle: a live engine is not
 * affected by later bmi_set_option calls from any thread (bmi_engine_set_option edits one engine's copy).
 *
 * Data layout in HBM: activations are NHWC fp16 ([image][y][x][channel]); conv weights
 * fp16 [Cout][ky][kx][Cin]; folded-BN scale/bias fp32 [Cout]; classifier weights fp32
 * [ceil32(C)][K] (rows >= C zero); with bmi_model_desc.dtype = BMI_DTYPE_BF16 "fp16" reads
 * bfloat16 throughout; the network input is fp32 NCHW exactly as the reference
 * receives it; moment accumulators are float64 [E][B][C].
 */
#ifndef BAYESNN_FPGA_AMD_H
#define BAYESNN_FPGA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMI_VERSION 600

#define BMI_OK 0
#define BMI_ERR_INVALID (-22)      /* EINVAL: bad descriptor / argument            */
#define BMI_ERR_NOMEM (-12)        /* ENOMEM: workspace too small                  */
#define BMI_ERR_HIP (-5)           /* EIO: a HIP runtime call or launch failed     */
#define BMI_ERR_UNSUPPORTED (-95)  /* EOPNOTSUPP: shape outside the kernels' range */

typedef struct bmi_engine_s* bmi_handle;
typedef void* bmi_stream; /* hipStream_t */

/* stochastic-site kinds */
#define BMI_SITE_NONE 0
#define BMI_SITE_ELEMENTWISE 1 /* MCDropout: x * keep / (1-p), one Bernoulli per element      */
#define BMI_SITE_CHANNEL 2     /* dropout2d semantics: one Bernoulli per (image, channel)      */
#define BMI_SITE_MASKSEMBLE 3  /* x * masks[(cnt0 + t) mod M][channel], no rescale             */

#define BMI_SITE_POS_OUTER 0 /* after scale/bias/residual/ReLU (HEAD: on the pooled features) */
#define BMI_SITE_POS_INNER 1 /* between the layer and what follows it (see bmi_op_desc)       */

typedef struct bmi_site {
    int32_t kind;        /* BMI_SITE_*                                                  */
    int32_t site_id;     /* call-order index of the stochastic layer inside one forward */
    float p;             /* drop probability (ELEMENTWISE / CHANNEL)                    */
    int32_t num_masks;   /* MASKSEMBLE: M                                               */
    const float* masks;  /* MASKSEMBLE: device fp32 [M][C] of 0/1                       */
} bmi_site;

/* op kinds */
#define BMI_OP_STEM 1  /* direct conv on the fp32 NCHW network input (Cin <= 4)        */
#define BMI_OP_CONV 2  /* implicit-GEMM conv (Cin % 64 == 0, Cout % 64 == 0)           */
#define BMI_OP_MASK 3  /* stand-alone stochastic site on a tensor                      */
#define BMI_OP_HEAD 4  /* global avg-pool + site + Linear + softmax + moment sums -> exit `out` (Cin % 32 == 0) */
#define BMI_OP_MAXPOOL 5 /* 2x2 stride-2 max-pool                                      */
#define BMI_OP_DENSE 6 /* hidden fully-connected layer on a flattened [1][1][K] tensor, fp32 weights,
                          accumulation and OUTPUT (the tensor `out` is then fp32 in the workspace and may
                          only feed another DENSE or a HEAD): out = relu?(in . weight^T + bias) (site)   */

typedef struct bmi_tensor_desc {
    int32_t h, w, c; /* per-image NHWC extent; tensor 0 is the network input */
} bmi_tensor_desc;

typedef struct bmi_op_desc {
    int32_t kind;
    int32_t in;        /* input tensor id                                              */
    int32_t out;       /* output tensor id; HEAD: exit index                           */
    int32_t residual;  /* CONV: tensor added before the ReLU, or -1                    */
    int32_t in2;       /* CONV 3x3 stride-1: input of a fused 1x1 strided shortcut conv (the BasicBlock
                          downsample path, resnet18.py:42-45) whose result is added before the ReLU, or -1.
                          With in2 both BN scales must be folded into the fp16 weights (scale = NULL)
                          and `bias` is the sum of the two BN biases.  Split engines only: `scale` may carry
                          one per-channel factor common to both weight sets (the host's power-of-two lift). */
    int32_t ksize, stride, pad;
    int32_t relu;      /* apply ReLU after scale/bias(+residual)                       */
    const void* weight;  /* device; CONV fp16 [Cout][k][k][Cin]; STEM fp32 [Cout][k][k][Cin];
                            HEAD fp32 [ceil32(out_dim)][Cin]; DENSE fp32 [Cout][Cin]   */
    const void* weight2;       /* CONV with in2: device fp16 [Cout][Cin2] (BN scale folded in)          */
    const float* scale;  /* device fp32 [Cout] folded BN scale (NULL = 1)              */
    const float* bias;   /* device fp32 [Cout] folded BN bias / Linear bias            */
    bmi_site site;       /* CONV/STEM/MASK: applied to the op's output; HEAD: applied to
                            the pooled features before the Linear                      */
    /* "inner" sites — the converter/pytorch insertion rule (Hardware_Artifact/converter/pytorch/nn2bnn.py:32-45,
     * Dropouts.py:25-56) wraps the layer itself, so the mask lands BEFORE a following BatchNorm / on the logits:
     *   CONV/STEM/MASK, site_pos = BMI_SITE_POS_INNER:
     *       out = relu?( (conv * scale + bias) * mask + bias_post (+ residual) )
     *       (scale/bias: the BN scale and scale * conv.bias; bias_post: the BN shift; no outer site then)
     *   HEAD, site_pos = BMI_SITE_POS_INNER: logits = (Linear(pool(x))) * mask  (dropout after the last layer) */
    const float* bias_post; /* device fp32 [Cout] or NULL (only read with BMI_SITE_POS_INNER)      */
    int32_t site_pos;       /* BMI_SITE_POS_*                                                      */
} bmi_op_desc;

/* element type of the 16-bit activations and conv weights of an engine */
#define BMI_DTYPE_F16 0  /* IEEE half: v_mfma_f32_*_f16 (default; meets the 1e-3 parity bar)            */
#define BMI_DTYPE_BF16 1 /* bfloat16:  v_mfma_f32_16x16x32_bf16 (8 mantissa bits: measured error in DESIGN.md) */
#define BMI_DTYPE_F32 2  /* the EXACT engine, for parity: fp32 activations in the workspace, fp32 conv weights (`weight`
                            fp32 [Cout][k][k][Cin]), every conv on v_mfma_f32_32x32x2_f32 in one generic per-tap kernel
                            (csrc/conv_exact.hip) — the arithmetic of the reference's fp32 CPU path, 1/16 of the fp16 MFMA
                            rate.  Graph features that exist for speed only are not built (in2 is BMI_ERR_UNSUPPORTED; no
                            pair / pooling / lazy-site fusion); bmi_forward_mcd_exit is BMI_ERR_UNSUPPORTED.              */

#define BMI_DTYPE_F16X2 3  /* the SPLIT engines, parity at speed (csrc/conv_split.hip): every conv operand a 16-bit head + tail pair, v = hi + lo
                             (hi = rn16(v), lo = rn16(v - hi)) — `weight` is 16-bit [2][Cout][k][k][Cin], plane 0 = the heads, plane 1 = the
                             tails, split ONCE by the host; the activations live in the workspace in the same form ("pair32": per pixel,
                             32-channel blocks [hi x 32 | lo x 32], 4 bytes per element: csrc/conv_epilogue.h), encoded once by the
                             kernel that produces a tensor — and
                             w.x = w_lo.x_hi + w_hi.x_lo + w_hi.x_hi on v_mfma_f32_32x32x16_f16 (fp32 accumulate; lo.lo dropped): three MFMAs
                             per K-step instead of the exact engine's sixteen.  Precision of an operand: 22 significant bits while its tail is a
                             NORMAL fp16 number, i.e. |v| >= 2^-3; below that the tail is an fp16 subnormal (ulp 2^-24) and the operand carries
                             an ABSOLUTE error floor of ~2^-25 = 3e-8 (|v| = 1e-3: ~15 bits; |v| < 6e-8 is lost).  For WEIGHTS the host removes
                             the floor: bayesnn_fpga_amd/engine.py splits each output channel's weights after an exact power-of-two scale that
                             brings max|w| of the channel to [2^7, 2^8) and folds 2^-k into the channel's BN scale (exact); activations are
                             O(1) behind BatchNorm and keep the floor.  The reference's fp32 arithmetic to ~1e-6 where plain fp16 is at
                             1e-4..2e-3 (peaky logits of trained / converted nets, Hardware_Artifact/converter/pytorch/nn2bnn.py:32-45 on
                             SA/models/vgg19/vgg19.py:256-324).  |values| < 65504.
                             Graph support: what the 16-bit engines take, bmi_forward_mcd_exit included (row tables in conv_split) — in2 (the fused 1x1 shortcut) of any conv geometry with Cin2 % 32 == 0 (extra
                             K-steps of the same kernel), pair launches and split-K are taken; channel counts Cin % 32 == 0, Cout % 64 == 0;
                             no lazy first site, no pooled epilogue (the tensors are materialised).                                     */
#define BMI_DTYPE_BF16X3 4 /* the same on v_mfma_f32_32x32x16_bf16: bf16 head + tail (16 significant bits, fp32's exponent range), three
                             bf16 MFMAs per K-step — BASELINE configs[1] ("bf16") inside north_star's 1e-3 on the bf16 matrix pipe      */

typedef struct bmi_model_desc {
    int32_t n_tensors;
    const bmi_tensor_desc* tensors;
    int32_t n_ops;
    const bmi_op_desc* ops;
    int32_t n_exits;
    int32_t out_dim;
    int32_t dtype; /* BMI_DTYPE_*: conv weights (`weight`, `weight2`) must be in this type (F32: fp32; F16X2 / BF16X3: 16-bit head
                      and tail planes [2][Cout][k][k][Cin]) */
} bmi_model_desc;

/* per-op-kind device time, filled by bmi_profile_read */
#define BMI_PROFILE_SLOTS 8 /* index = BMI_OP_* (the exit heads' slot includes the moment sums) */

int bmi_version(void);
const char* bmi_error_string(int code);

/* Process-wide switches: kernel selection (used for same-process A/B measurement and by the tests to cover both code
 * paths) and the element type of the unit-test entry points.  Results are equal TO ROUNDING across them, not always bit for
 * bit: "mfma_shape_*", "epilogue_lite", "conv_seam", "split_tile" and "split_shx" leave every bit alone; "conv_pw", "conv_s2", "conv_stream" and "conv_wide" move a
 * conv to a kernel family that sums its K dimension in another order, "splitk" adds nine fp32 partial sums separately,
 * "dense_exact" swaps the split-fp16 product for the exact-f32 MFMA.  (Kernel selection itself looks at the conv's shape
 * and the engine's planned batch x chunk only, so two runs of one engine — whole, t-sharded, image-sharded, partial chunks —
 * agree bit for bit: the float64 moment sums of an image's 32-sample groups are joined in group order.)  Returns BMI_ERR_INVALID for an unknown
 * name / value.  Names:
 *   "mfma_shape_patch", "mfma_shape_wide"   16 | 32: MFMA instruction shape of conv3x3_patch / conv_igemm_wide
 *                                           (v_mfma_f32_16x16x32_f16 | v_mfma_f32_32x32x16_f16); 0 = built-in default
 *   "xcd_split"                             0 | 1 | 2 | 4: channel-tile classes of the XCD-aware tile order (0 = chosen from
 *                                           the conv's weight bytes so that one XCD's weights stay L2-resident)
 *   "conv_pw"                               0 | 1 | 2 (= 1 without the minimum-grid rule: tests): 3x3 stride-1 convs on 8x8 / 4x4 maps with Cout % 256 == 0 run in conv3x3_pw (256 x 256
 *                                           tile, 8 waves) instead of conv3x3_patch (128 x 128, 2 workgroups per CU); 3 | 4 (= 1 | 2 with the four-wave,
 *                                           software-pipelined conv3x3_pw4 where it applies: measurement reference, the same bits)
 *   "conv_stream"                           0 | 1 | 2 (= 1 without the minimum-grid rule and for plain launches too: tests) | 3 (= 1 with the
 *                                           256-pixel tile: A/B): HBM-bound 1x1 convs (Cin <= 512; with a residual, or Cout % 256 != 0) run
 *                                           in conv1x1_stream (128 x 128 tile, three or four workgroups per CU) instead of conv_igemm_wide
 *   "split_shx"                             0 | 1: the split engines' 3x3 stride-1 convs fetch the pixel tile of a tap row once for its three taps (the same
 *                                           bits; 0: once per tap)
 *   "split_tile"                            0 | 1: the split engines' conv kernel narrows its channel tile (256 -> 128 -> 64) while a launch of the planned
 *                                           batch x chunk would be fewer than two workgroups per CU (the same bits; 0: always the widest tile)
 *   "conv_seam"                             0 | 1 | 2 | 3, read by bmi_create and per launch: conv3 + BN + shortcut add + ReLU of one Bottleneck and conv1 + BN +
 *                                           ReLU of the next (128 narrow channels; 2 | 3: 256 too, no minimum grid — tests; 3: the unpipelined loop) run
 *                                           as one conv1x1_seam launch, the wide tensor fed to the second conv from LDS; 0: the two launches
 *   "conv_s2"                               0 | 1 | 2 (= 1 without the minimum-grid rule: tests): 3x3 stride-2 convs with a BN + ReLU epilogue on
 *                                           32x32 / 16x16 / 8x8 maps with Cout % 256 == 0 (a pair's channels together) run in conv3x3_s2 (input
 *                                           patch resident in LDS as four parity planes, persistent) instead of conv_igemm_wide
 *   "conv_pool"                             0 | 1 | 2: a 3x3 conv whose 4x4 output map feeds one exit head and nothing else writes fp32 means over the
 *                                           map (ReLU + avg_pool2d(4) fused into the epilogue) instead of the map: the plain stride-2 convs in
 *                                           conv3x3_s2 (1, 2) and the stride-1 conv in front of the final head in conv3x3_pw (1)
 *   "mask_lazy"                             0 | 1: the elementwise site that expands the once-per-batch prefix (32x32 maps) to the folded batch writes
 *                                           keep bits + one scaled copy of the B images; conv3x3_s2 / conv3x3_patch (fused shortcut input) clear the
 *                                           dropped elements in LDS, any other consumer makes the masked tensor appear first (1, default), or the
 *                                           masked tensor is always written (0); the same bits either way
 *   "pw_persist"                            0 | 1: plain-epilogue launches of conv3x3_pw (BN + ReLU, with or without the fused shortcut) run in its
 *                                           persistent form — one workgroup per CU walks the tiles, the last chunk of a tile prefetches the next tile's
 *                                           first weight stages and sub-patch, the epilogue stages through 64 KB beside them (1, default: -3..-5 % per
 *                                           launch) — or one workgroup per tile (0); the same bits either way
 *   "lazy_planar"                           0 | 1: a lazy site whose readers are all stride-2 consumers (conv3x3_s2 on 32x32 maps, the fused 1x1
 *                                           stride-2 shortcut of conv3x3_patch) stores its scaled copy and keep bits as 32-channel planes with
 *                                           the even columns of a row in front of the odd ones — what such a reader DMAs is then contiguous,
 *                                           whole 128-byte lines (1, default) — or in NHWC (0); the same bits either way
 *   "conv_wide"                             0 | 1: 0 skips conv_igemm_wide (A/B against the per-tap kernel)
 *   "splitk"                                0 | 1, read by bmi_plan: 3x3 convs of the once-per-batch prefix whose grid is <= 64 tiles (VGG's convs on
 *                                           2x2 maps) run split-K: one workgroup per (tile, tap), fp32 partial sums, a finishing pass
 *   "dense_exact"                           0 | 1: hidden dense layers (BMI_OP_DENSE / bmi_dense_f32) on the exact-f32 MFMA (1) or as
 *                                           fp16 head + tail products on the fp16 MFMA with fp32 accumulation (0, default: fp32-equivalent
 *                                           to a few 1e-7, 2.5x faster)
 *   "lazy_order"                            0 | 1: the kernels that read a deterministic tensor through a lazy site's keep bits walk their
 *                                           tiles sample-minor — the samples of one activation tile back to back on one XCD, which finds it
 *                                           in its L2 (1, default) — or in the plain order (0).  Placement only: the same bits
 *   "epilogue_lite"                         0 | 1 | 2: BN + residual + ReLU + 2-bit elementwise-site launches finish on the accumulator
 *                                           registers with one fp16 trip through LDS (1, default; 2: without the forms that have the
 *                                           site kind and the residual compiled in) or in the general two-round fp32 epilogue (0);
 *                                           the same bits every way
 *   "wide_persist_min_x10"                  10..1000: conv_igemm_wide runs persistent (one workgroup per CU walking the tiles)
 *                                           when tiles * 10 > value * CUs
 *   "unit_entry_dtype"                      BMI_DTYPE_*: how the single-kernel entry points below (unit tests) interpret
 *                                           their 16-bit buffers; engines carry their own dtype in bmi_model_desc.  BMI_DTYPE_F32:
 *                                           bmi_conv_igemm_fwd / bmi_stem_conv_fwd (output) / bmi_mask_apply / bmi_maxpool2 take fp32
 *                                           buffers (and fp32 conv weights) and run the exact engine's kernels; BMI_DTYPE_F16X2 / BF16X3: the
 *                                           activation buffers of those entry points (and of bmi_head_fused / bmi_dense_f32 with in_is_f32 = 0)
 *                                           are pair32 tensors, bmi_conv_igemm_fwd takes the 16-bit head / tail weight planes and runs conv_split
 *   "ws_no_reuse"                           0 | 1, read by bmi_plan: every suffix tensor keeps its own workspace range (per-layer
 *                                           traces through bmi_tensor_info; the workspace grows to the sum of the activations)
 *   "conv_patch64"                          0 | 1: 3x3 stride-1 convs with Cout % 128 == 64 on 32-wide maps (the 64 -> 64 BasicBlocks behind the stem) run in
 *                                           conv3x3_patch's 64-channel tile (1, default) or in the per-tap conv_igemm (0); another K order (64- vs
 *                                           32-channel chunks): equal to rounding
 *   "splitk_tiles"                          0..1024, read by bmi_plan: the largest grid (128 x 128 tiles at the planned batch) of a deterministic 3x3 conv
 *                                           with Cin >= 256 that still runs split-K (default 64: a quarter of the CUs)
 *   "pair_prefix"                           0 | 1, read by bmi_create: pair fusion (two plain convs on one input as one launch) also in the once-per-batch
 *                                           prefix (1, default: the exit-only step +3 %, profiles/experiments/r6_exit_only_variants.txt) or in the suffix only (0)
 *   "patch_direct"                          0 | 1 | 2: conv3x3_patch's BasicBlock tails on 16x16 maps (residual, residual + 2-bit site) finish on the accumulator
 *                                           registers and store straight to HBM (1, default; 2: its plain launches too — measured slower in the network) or
 *                                           take the epilogues through LDS (0).  The same bits every way
 *   "head_batch"                            0 | 1: consecutive exit heads of the sample-folded suffix run as ONE launch (1, default) — with exit-only
 *                                           dropout, the configuration of every run of the paper (journal_script.sh:10-63), the suffix is nothing but
 *                                           the four / five heads — or one launch per head (0).  The same bits either way
 * Initial values come from the environment (BMI_MFMA_SHAPE, BMI_MFMA_SHAPE_WIDE, BMI_XCD_SPLIT).
 *
 * SCOPE (C-ABI 600).  bmi_set_option edits the PROCESS DEFAULTS: what the single-kernel entry points read, and what bmi_create COPIES into
 * the handle it returns.  An engine runs every later call (bmi_plan, bmi_forward_*) under its own copy, so a live engine never changes
 * kernels because another host thread (see "Threading" at the top) changed a default; bmi_engine_set_option edits the copy
 * of ONE engine (same names and ranges; switches read by bmi_create / bmi_plan take effect at the next bmi_plan at the latest, the
 * graph-merging one — "conv_seam" — only at launch time: an op merged at bmi_create falls back to its two launches). */
int bmi_set_option(const char* name, int32_t value);
int bmi_engine_set_option(bmi_handle h, const char* name, int32_t value);

/* Host-only: validates and copies the graph, marks which tensors are stochastic, splits a
 * conv that carries a site but has only deterministic inputs into conv + MASK (so the
 * deterministic prefix runs once per batch).  No HIP call. */
int bmi_create(const bmi_model_desc* desc, bmi_handle* out);
int bmi_destroy(bmi_handle h);

/* Host-only: lays the activation buffers out for batches of up to `max_batch` images and
 * chunks of up to `chunk_samples` Monte-Carlo samples folded into the GEMM M dimension. */
int bmi_plan(bmi_handle h, int32_t max_batch, int32_t chunk_samples, size_t* workspace_bytes);

/* MACs (conv + linear) per image of the deterministic prefix and per (image, sample) of the
 * stochastic suffix; and the op counts after the split. */
int bmi_query(bmi_handle h, int64_t* prefix_macs, int64_t* suffix_macs, int32_t* n_prefix_ops, int32_t* n_suffix_ops);

/* Traces (tools/layer_trace.py): where tensor `id` (1 .. n_tensors-1 of the descriptor) of a PLANNED engine lives in the caller's
 * workspace: byte offset, bytes per element (2: the engine's 16-bit type, 4: fp32 or pair32), per_sample bit 0: it holds one image set per
 * Monte-Carlo sample of the chunk ([chunk*B][h][w][c], image = t_local*B + b) rather than the once-per-batch B images; bit 1: the tensor
 * is in the split engines' pair32 layout (per pixel, 32-channel blocks of 32 heads + 32 tails, 16-bit each); and its extent.
 * Suffix tensors share workspace ranges by live range unless the engine was planned under bmi_set_option("ws_no_reuse", 1);
 * a fused launch may leave a tensor unwritten (set "mask_lazy" = 0, "conv_pool" = 0 for a full trace).  Keep-bit tensors and
 * tensors nothing reads are BMI_ERR_UNSUPPORTED.  No reference counterpart (a forward hook on an nn.Module). */
int bmi_tensor_info(bmi_handle h, int32_t id, int64_t* offset, int32_t* elem_bytes, int32_t* per_sample, int32_t* th, int32_t* tw,
                    int32_t* tc);

/* Runs samples t_begin .. t_begin+t_count-1 for one batch and ADDS, per exit e, image b and
 * class c:  S1 += softmax_p, S2 += softmax_p^2, SL += logit   (float64 [E][B][C]). */
int bmi_forward_mcd(bmi_handle h, const float* x_nchw, int32_t batch, int32_t t_begin, int32_t t_count,
                    uint64_t seed, int32_t mask_cnt0, double* S1, double* S2, double* SL, void* workspace,
                    size_t workspace_bytes, bmi_stream stream);

/* The same for images image_offset .. image_offset+batch-1 of a LARGER batch: `x_nchw` and S1/S2/SL hold this share only
 * ([batch] rows), while every dropout mask is drawn at the image's index in the whole batch — so the shares of a batch
 * partitioned by IMAGES over several GPUs (the fallback of SURVEY.md §8.5 when there are fewer Monte-Carlo samples than
 * ranks: every rank runs all T samples on its images, nobody idles) reproduce the rows of the one-GPU run bit for bit when
 * the engine is planned for the whole batch (kernel selection looks at the planned batch, not at the call's).
 * A site's index offset must be a whole number of Philox calls (image_offset x elements per image % 64 == 0: true for every
 * tensor of the CNN families here), else BMI_ERR_UNSUPPORTED — bmi_image_offset_ok (host-only) tells beforehand, so that the
 * ranks of a group can refuse a partition TOGETHER instead of one rank failing while the others wait in the all-reduce.
 * Masksembles masks do not depend on the image.  No reference counterpart (single device: SA/train/train_utils.py:10-11). */
int bmi_image_offset_ok(bmi_handle h, int32_t image_offset);
int bmi_forward_mcd_images(bmi_handle h, const float* x_nchw, int32_t batch, int32_t image_offset, int32_t t_begin,
                           int32_t t_count, uint64_t seed, int32_t mask_cnt0, double* S1, double* S2, double* SL,
                           void* workspace, size_t workspace_bytes, bmi_stream stream);

/* Per-sample outputs of the folded path — what the reference's evaluate() consumes pass by pass (SA/train/evaluate.py:8-22 ->
 * SA/train/train_utils.py:32-38 -> _MultiExitAccuracy._metrics, SA/train/loss/base_classes.py:39-66: the logits of every exit of every
 * stochastic forward): samples t_begin .. t_begin+t_count-1 of the batch as in bmi_forward_mcd, and
 *     logits[(t - t_begin)][e][b][c]   fp32 [t_count][E][batch][C]
 * written by the fused head kernel beside the moment sums (S1 / S2 / SL as in bmi_forward_mcd), or instead of them (all three NULL).
 * mask_stride: the Masksembles mask of sample t is (mask_cnt0 + (t - t_begin) * mask_stride) mod M — the reference's layers count
 * their forward calls (SA/utils.py:165-169), so when its evaluate() walks a loader of n batches T times, pass i of batch k is call
 * i * n + k: the T passes of batch k folded into ONE call here take mask_cnt0 = cnt + k, mask_stride = n.  mask_cnt0 is the mask of the
 * call's FIRST sample whatever t_begin (mask_stride = 1 and t_begin = 0: the masks of bmi_forward_mcd, which indexes (mask_cnt0 + t) mod M
 * with the global sample index t).  MC-dropout masks depend on the sample index t alone, as everywhere.  Captures into a hipGraph like
 * bmi_forward_mcd. */
int bmi_forward_mcd_samples(bmi_handle h, const float* x_nchw, int32_t batch, int32_t t_begin, int32_t t_count, uint64_t seed,
                            int32_t mask_cnt0, int32_t mask_stride, float* logits, double* S1, double* S2, double* SL, void* workspace,
                            size_t workspace_bytes, bmi_stream stream);

/* Confidence-threshold early exiting ON the device — what the reference only models after the fact
 * (FullAnalysis.confidence_exiting / is_confident / flop_saver, SA/train/results_analyzer.py:606-630, :638-677, :725-733):
 * runs samples 0 .. t_count-1 of the batch stage by stage; after the head of exit e (first_exit <= e < n_exits-1; the
 * reference's loop starts at exit 1) an image whose confidence max_c mean_t softmax[e][b][c] exceeds `threshold` is
 * assigned exit_of_image[b] = e and takes no part in the later stages: the following launches cover only the still-active
 * images (compact tile grid, tensors keep their original rows, masks keep their original element indices, so every value
 * that IS computed equals the full run's bit for bit).  Images that never pass get n_exits-1.
 * S1/S2/SL [E][batch][C] must be ZERO on entry; on return the rows of exits an image did not reach hold no samples.
 * active_after[e] (host, [n_exits]) = images still active after exit e's test.  Needs t_count <= the planned chunk;
 * synchronises the stream once per tested exit (the host sizes the next stage's grids), so it cannot be graph-captured.
 * Graphs with MASK / MAXPOOL / DENSE ops behind the first tested exit are BMI_ERR_UNSUPPORTED. */
int bmi_forward_mcd_exit(bmi_handle h, const float* x_nchw, int32_t batch, int32_t t_count, uint64_t seed, int32_t mask_cnt0,
                         double threshold, int32_t first_exit, double* S1, double* S2, double* SL, int32_t* exit_of_image,
                         int32_t* active_after, void* workspace, size_t workspace_bytes, bmi_stream stream);

/* mean = S1/T, var = S2/T - mean^2 (clamped at 0), logit_mean = SL/T; n = E*B*C. */
int bmi_finalize(int64_t n, int32_t t_total, const double* S1, const double* S2, const double* SL, double* mean,
                 double* var, double* logit_mean, bmi_stream stream);
/* The same, and ADDS to *nonfinite (device int32, caller-zeroed) the number of elements whose sums are not finite: the 16-bit engines
 * saturate at 65 504 (fp16) — an overflowing activation becomes inf, then NaN in the softmax, and would otherwise travel into the
 * reference's np.average (SA/train/results_analyzer.py:247-248) unnoticed.  No synchronisation: read the counter with the results. */
int bmi_finalize_checked(int64_t n, int32_t t_total, const double* S1, const double* S2, const double* SL, double* mean,
                         double* var, double* logit_mean, int32_t* nonfinite, bmi_stream stream);

/* Per-op-kind HIP-event timing of bmi_forward_mcd (off by default; adds two event records per
 * launch).  bmi_profile_read synchronises the recorded events and resets the accumulators. */
int bmi_profile_enable(bmi_handle h, int32_t enable);
int bmi_profile_read(bmi_handle h, double ms[BMI_PROFILE_SLOTS], int64_t launches[BMI_PROFILE_SLOTS]);
/* The BMI_OP_CONV slot of the LAST bmi_profile_read, split by the kernel that took each launch, with the
 * algorithmic FLOPs (2 * MACs) and algorithmic HBM bytes (every operand tensor and the weights read once, the output
 * written once) of those launches: what bench.py prices against the MFMA and the HBM roofline. */
#define BMI_CONV_FAMILY_PATCH 0 /* conv3x3_patch_kernel  */
#define BMI_CONV_FAMILY_WIDE 1  /* conv_igemm_wide_kernel */
#define BMI_CONV_FAMILY_IGEMM 2 /* conv_igemm_kernel */
#define BMI_CONV_FAMILY_PW 3    /* conv3x3_pw_kernel */
#define BMI_CONV_FAMILY_STREAM 4 /* conv1x1_stream_kernel */
#define BMI_CONV_FAMILY_S2 5    /* conv3x3_s2_kernel */
#define BMI_CONV_FAMILY_SPLIT 6 /* conv_split_kernel (the split engines; FLOPs = algorithmic, i.e. one third of the MFMA work) */
#define BMI_CONV_FAMILY_SEAM 7  /* conv1x1_seam_kernel (two 1x1 convs of neighbouring Bottlenecks in one launch; FLOPs and bytes of both) */
#define BMI_CONV_FAMILIES 8
int bmi_profile_conv_families(bmi_handle h, double ms[BMI_CONV_FAMILIES], int64_t launches[BMI_CONV_FAMILIES],
                              double flops[BMI_CONV_FAMILIES], double bytes[BMI_CONV_FAMILIES]);

/* The individual launches behind the most recent bmi_profile_read, in launch order: op kind (BMI_OP_*), conv family (-1 for
 * non-conv ops), the op's output tensor id (identifies the op in the graph), images carried, device milliseconds, algorithmic
 * FLOPs and HBM bytes (conv ops).  *count = number of launches recorded; at most `capacity` entries are written; any array
 * may be NULL.  (tools/per_launch.py prints the table: which launch of a model sits where against its roofline.) */
int bmi_profile_launches(bmi_handle h, int32_t capacity, int32_t* count, int32_t* kind, int32_t* family, int32_t* out_tensor,
                         int32_t* images, double* ms, double* flops, double* bytes);

/* ---- single-kernel entry points (unit parity tests) -------------------------------------- */

/* keep bits (0/1 bytes) of elements 0..n-1 of stream (seed, site, t). */
int bmi_philox_mask(uint8_t* keep, int64_t n, uint64_t seed, int32_t site, int32_t t, float p, bmi_stream stream);

int bmi_stem_conv_fwd(const float* x_nchw, const float* weight, const float* scale, const float* bias, void* out_nhwc,
                      int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t ksize, int32_t stride,
                      int32_t pad, int32_t relu, bmi_stream stream);

/* out[n] = conv(in[n % in_mod]) * scale + bias (+ res[n % res_mod]) (ReLU) (site); `batch` is
 * the per-sample image count B used by the site's element index (n = t_local*B + b). */
/* keep bits (1 bit per element, byte g = elements 8g..8g+7) of an elementwise site for the folded batch
 * n = samples*batch images of hw pixels x c channels */
int bmi_mask_bits(void* bits, int32_t n, int32_t hw, int32_t c, const bmi_site* site, int32_t batch, int32_t t0,
                  uint64_t seed, bmi_stream stream);

/* in_keep_bits (or NULL): input-side MC-dropout — image n reads in[n % in_mod] with the dropped elements of
 * folded image n zeroed while staging; out_mul multiplies the BN scale (pass 1/(1-p) then, else 1). */
int bmi_conv_igemm_fwd(const void* in, const void* in_keep_bits, float out_mul, const void* weight,
                       const float* scale, const float* bias, const void* res, void* out,
                       int32_t n, int32_t in_mod, int32_t res_mod, int32_t h, int32_t w, int32_t cin, int32_t cout,
                       int32_t ksize, int32_t stride, int32_t pad, int32_t relu, const bmi_site* site,
                       int32_t batch, int32_t t0, uint64_t seed, int32_t mask_cnt0, bmi_stream stream);

/* out = relu?( conv3x3_s1_p1(in; weight) + conv1x1_stride2(in2; weight2) + bias ), the fused BasicBlock tail
 * with downsample (both BN scales folded into the weights); in2 is [n][2h][2w][cin2]. */
int bmi_conv3x3_shortcut_fwd(const void* in, const void* weight, const void* in2, const void* weight2, const float* bias,
                             void* out, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t cin2,
                             int32_t relu, bmi_stream stream);

/* The seam between two Bottleneck blocks as one launch (csrc/conv1x1_seam.hip): out_wide = relu(bn3(conv1x1(in; weight3)) + res) [n][h][w][cw] and
 * out_narrow = relu1?(bn1(conv1x1(out_wide; weight1))) [n][h][w][cn], the second conv fed from the tile the first has just produced (out_wide
 * is written, not read back).  cmid % 64 == 0, cmid <= 512, cw % 128 == 0, cn = 128 (256 too under "conv_seam" >= 2: slower than the
 * two launches, kept for the tests); anything else, and launches under the kernel's minimum grid, run as the two launches.  Bit-identical
 * to bmi_conv_igemm_fwd twice.  Replaces conv3 + bn3 + shortcut add + ReLU of one Bottleneck and conv1 + bn1 + ReLU of the next (the
 * Bottleneck form of SA/models/resnet18/resnet18.py:51-85; BASELINE configs[4]). */
int bmi_conv1x1_seam_fwd(const void* in, const void* weight3, const float* scale3, const float* bias3, const void* res, void* out_wide,
                         const void* weight1, const float* scale1, const float* bias1, void* out_narrow, int32_t n, int32_t h, int32_t w,
                         int32_t cmid, int32_t cw, int32_t cn, int32_t relu1, bmi_stream stream);

/* Two convolutions that read the SAME input with the same geometry, as one launch of the 256 x 256-tile kernel:
 * out_a = relu?(bn_a(conv(in; weight_a))) [..][cout_a], out_b likewise [..][cout_b].  cout_a % 128 == 0,
 * (cout_a + cout_b) % 256 == 0, cin % 64 == 0.  Replaces the pair layerN[0].conv1 / ex{N-1}conv1 of
 * ResNet18MCEarlyExit.forward (SA/models/resnet18/resnet18.py:306, :318, :329 next to :280-299). */
int bmi_conv_pair_fwd(const void* in, const void* weight_a, const float* scale_a, const float* bias_a, void* out_a,
                      const void* weight_b, const float* scale_b, const float* bias_b, void* out_b, int32_t n,
                      int32_t in_mod, int32_t h, int32_t w, int32_t cin, int32_t cout_a, int32_t cout_b, int32_t ksize,
                      int32_t stride, int32_t pad, int32_t relu, bmi_stream stream);

int bmi_mask_apply(const void* in, void* out, int32_t n, int32_t in_mod, int32_t hw, int32_t c, const bmi_site* site,
                   int32_t batch, int32_t t0, uint64_t seed, int32_t mask_cnt0, bmi_stream stream);

int bmi_maxpool2(const void* in, void* out, int32_t n, int32_t h, int32_t w, int32_t c, bmi_stream stream);

/* One exit head for samples t0 .. t0+tc-1 of `batch` images, fused (csrc/head_fused.hip):
 *   feat[n][k]   = mean_hw(relu(in[n % in_mod][hw][k])) (site)          n = t_local*batch + b, in_mod = batch or batch*tc
 *   logits[n][c] = feat[n] . weight_pad[c] + bias[c]  (site_logits: ELEMENTWISE dropout on the [batch, out_dim] logits, the
 *                  converter/pytorch rule wraps the last Linear too, nn2bnn.py:33-45; NULL = none)
 *   S1[b][c] += sum_t softmax(logits)[c], S2 += sum_t softmax^2, SL += sum_t logits        (float64 [batch][out_dim])
 * `in` is fp16 / bf16 (in_is_f32 = 0, per unit_entry_dtype) or fp32; weight_pad fp32 [ceil32(out_dim)][k], rows >= out_dim
 * zero; k % 32 == 0, out_dim <= 128.  No per-sample probabilities are materialised. */
int bmi_head_fused(const void* in, int32_t in_is_f32, int32_t in_mod, int32_t hw, int32_t k, const float* weight_pad,
                   const float* bias, int32_t out_dim, const bmi_site* site, const bmi_site* site_logits, int32_t batch, int32_t t0,
                   int32_t tc, uint64_t seed, int32_t mask_cnt0, double* S1, double* S2, double* SL, bmi_stream stream);

/* Hidden dense layer in fp32 (BMI_OP_DENSE): out[n][c] = relu?(in[n % in_mod] . weight[c] + bias[c]) (site on the
 * [batch, cout] tensor, sample index n / batch + t0).  `in` is fp16 (in_is_f32 = 0) or fp32 [.][k]; weight fp32 [cout][k];
 * k % 32 == 0, cout % 64 == 0.  Replaces the Dense 512 layers of the VGG-11 classifier stack
 * (Hardware_Artifact/bayes_hw/models/models.py:262-281) with their dropout (:268-281). */
int bmi_dense_f32(const void* in, int32_t in_is_f32 /* 0: 16-bit (unit_entry_dtype), 1: fp32 */, const float* weight, const float* bias, float* out, int32_t n,
                  int32_t in_mod, int32_t k, int32_t cout, int32_t relu, const bmi_site* site, int32_t batch, int32_t t0,
                  uint64_t seed, int32_t mask_cnt0, bmi_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* BAYESNN_FPGA_AMD_H */
