// Internal launch interfaces shared by the kernels and the engine (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bayesnn_fpga_amd.h"
#include "philox.h"

struct ConvArgs {
    const _Float16* in;
    const _Float16* wgt;  // [Cout][k*k*Cin]
    const float* scale;   // may be null
    const float* bias;    // may be null
    const _Float16* res;  // may be null
    const uint8_t* res_bits; // keep bits (as in_bits) for `res`: the residual is a lazy site's tensor — `res` then holds the B scaled images (res_mod = B) and
                             // the dropped elements are cleared where the residual is added (conv3x3_patch's 64-channel tile), or null
    const uint8_t* in_bits;  // keep bits of an elementwise MC-dropout site on the INPUT (1 bit per element,
                             // byte g = elements 8g..8g+7 of the folded tensor), or null; conv_igemm (zeroes while staging,
                             // 1/(1-p) in out_mul) and conv3x3_s2 on 32x32 maps (clears the elements in LDS; `in` pre-scaled)
    float out_mul;           // multiplies the folded-BN scale (the site's 1/(1-p) when conv_igemm reads in_bits), else 1
    // fused 1x1 strided shortcut (conv3x3_patch only): out += conv1x1(in2; wgt2), both BN scales folded
    // into the fp16 weights by the host, the biases summed (BasicBlock.forward :42-45 downsample path)
    const _Float16* in2;     // [N or in2_mod][H2][W2][Cin2], or null
    const _Float16* wgt2;    // [Cout][Cin2]
    int in2_mod, H2, W2, Cin2, stride2;
    // pair mode (conv_igemm_wide only): a second conv on the same input with the same geometry; channel tiles
    // >= split of the launch's Cout (= both convs' channels) use these and write out_b [.][Cout - split]
    const _Float16* wgt_b;   // or null
    const float* scale_b;
    const float* bias_b;
    _Float16* out_b;
    int split;
    // fused ReLU + global average pool of a 4x4 output map (conv3x3_s2 only): when set, the conv (pool: the launch's conv / the
    // first of a pair; pool_b: the second) writes fp32 means over its 16 output pixels, [row][Cout of that conv], INSTEAD of its
    // fp16 activation tensor — the conv feeds nothing but an exit head (relu -> avg_pool2d(4) -> Linear, resnet18.py:309-314)
    float* pool;
    float* pool_b;
    // inner site (converter/pytorch semantics): out = relu?((acc*scale + bias) * mask + bias_post (+ res))
    const float* bias_post;  // or null
    int site_inner;          // 1: a.site multiplies before bias_post / residual / ReLU
    _Float16* out;
    int N;        // output images in this launch (= samples_in_chunk * B in the suffix)
    int n_ref;    // images of a FULL chunk of this engine (B x planned chunk; 0 = N): kernel SELECTION looks at this, not at N,
                  // so a t-shard (fewer samples per launch, e.g. one rank of eight) runs the same kernels — and gets the
                  // same bits — as the single-rank run
    int in_mod;   // input image = n % in_mod  (B when the input is deterministic)
    int res_mod;
    int H, W, Cin;
    int Ho, Wo, Cout;
    int ksize, stride, pad;
    int relu;
    int M;         // N * Ho * Wo
    int B;         // images per Monte-Carlo sample
    int t0;        // first sample index of this launch
    int bf16;      // 1: activations / weights are bfloat16 bits (v_mfma_*_bf16), 0: fp16
    // dynamic early exit (bmi_forward_mcd_exit): the launch covers N = samples * Bc COMPACT images; compact image n is row
    // imap[n] = (n / Bc) * B + active[n % Bc] of every tensor (which keep their original folded layout) and of the Philox index
    const int* imap;   // device [N] row of each compact image, or null (identity)
    int Bc;            // still-active images per sample (informational)
    // split-K (small grids: a 3x3 conv on a 2x2 map of a 250-image batch is 32 tiles): nsplit workgroups per tile each take a
    // contiguous range of the K-steps and store raw fp32 partial sums [nsplit][M][Cout]; launch_splitk_finish adds them in a
    // fixed order and applies BN / ReLU.  Plain epilogue only; null = no split.
    float* partial;
    int nsplit;
    int xcd_split; // channel-tile classes of the XCD-aware tile order (xcd_tile_map's cs); set by the launcher
    SiteArgs site;
    const uint8_t* in2_bits; // keep bits (as in_bits) for in2: conv3x3_patch on 16x16 maps clears the dropped elements of the shortcut's
                             // pixels in LDS; in2 is then the deterministic tensor, pre-scaled by 1/(1-p) in fp16.  Or null.
                             // (Last: the kernels that never read it keep their argument layout — a field in the middle moved
                             // conv3x3_pw's 4x4 bf16 instantiations from 0 to 212 bytes of scratch.)
    int lazy_planar;         // the operand that comes with keep bits (in + in_bits: conv3x3_s2; in2 + in2_bits: conv3x3_patch) and its bits are
                             // stored in the lazy site's PLANAR layout (lazy_planar_off below) instead of NHWC
    int lazy_order;          // (set by the launcher) a launch that reads `in` through keep bits walks its tiles sample-minor: lazy_tile_map (conv_epilogue.h)
};

// Layout of a lazy site's scaled copy and keep bits when every reader is a stride-2 consumer (conv3x3_s2 on 32x32 maps, the fused 1x1
// stride-2 shortcut of conv3x3_patch): element (c, y, x) of an image of hw = H x w pixels (w even, C % 32 == 0) sits at
//     (c >> 5) * hw * 32 + (y * w + (x & 1) * (w >> 1) + (x >> 1)) * 32 + (c & 31)
// — 32-channel planes, and inside a row the even columns in front of the odd ones.  A stride-2 reader takes every other column of a
// row: its 64-byte cells (32 channels) are then CONTIGUOUS, whole 128-byte lines that nobody asks for twice; in NHWC a 32-channel
// chunk is half of a pixel's line and the other half is requested nine K-steps later (rocprofv3 FETCH_SIZE of the headline's pair launch:
// 6.68 GB in NHWC, 3.67 GB planar = one fetch of each tile's image + the bits).  The keep bits follow the same permutation (bit index = element index), so a piece's 16 bytes and its bits
// byte stay at "activation byte offset / 16" of each other.
__host__ __device__ inline long lazy_planar_off(int c, int y, int x, int hw, int w) {
    return (long)(c >> 5) * hw * 32 + ((long)y * w + (x & 1) * (w >> 1) + (x >> 1)) * 32 + (c & 31);
}

struct EltArgs {  // MASK op
    int pair;     // 0, or 1 | 2: the tensors are in the split engines' pair32 layout (conv_epilogue.h), fp16 | bf16 halves
    int bf16;     // 16-bit tensors hold bfloat16 bits
    const _Float16* in;
    void* out;    // fp16 [N][HW][C]
    int N, in_mod, HW, C;
    int B, t0;
    SiteArgs site;
    const float* bias_post;  // MASK with an inner site: out = relu?(x * mask + bias_post[c]); or null
    int relu;
    int tchunk;              // (set by the launcher) samples per work item of mask_apply_lb1_kernel
};

struct HeadArgs {   // fused exit head (head_fused.hip)
    const void* in;      // [N or in_mod][HW][K]; in_kind 0: fp16, 1: fp32, 2: bf16
    int in_kind;
    int in_mod;          // B (deterministic input: every sample reads the same image) or B * tc
    int HW, K;
    int B, t0, tc;       // images per sample, first sample index, samples in this launch
    const float* w;      // fp32 [ceil32(C)][K], rows >= C zero
    const float* bias;   // fp32 [C]
    int C;               // out_dim
    const int* imap;      // dynamic early exit: grid.x = Bc workgroups, image = imap[blockIdx.x]; or null
    int Bc;
    SiteArgs site;        // on the pooled [B, K] features
    SiteArgs site_logits; // ELEMENTWISE dropout on the [B, C] logits, or NONE
    int b0;               // batch index of this launch's image 0 (bmi_forward_mcd_images), for the logits site: b0 * C is not a
                          // multiple of a Philox call, so it cannot ride in site_logits.elem_off
    double *S1, *S2, *SL; // this exit's [B][C] moment accumulators
    float* logits;        // per-sample logits out: sample tl of this launch, class c of image b -> logits[tl * logits_tstride + b * C + c]; or null
    size_t logits_tstride;
    double* part;         // scratch [ceil(tc / 32)][3][B][C] for the per-group partial sums of a launch with more than 32 samples, or
                          // null (hardware atomics then: the single-kernel entry point)
};
int launch_head_fused(const HeadArgs& a, hipStream_t s);
#define BMI_HEAD_PACK_MAX 8
int launch_head_fused_multi(const HeadArgs* list, int n, hipStream_t s);   // n uniform heads in one launch (grid.z); BMI_ERR_UNSUPPORTED -> one by one

int launch_conv_igemm(const ConvArgs& a, hipStream_t s);
int launch_conv_igemm_wide(const ConvArgs& a, hipStream_t s);  // 256x256 tiles; BMI_ERR_UNSUPPORTED -> conv_igemm
bool conv_takes_wide_kernel(int cin, int cout);
int launch_conv3x3_patch(const ConvArgs& a, hipStream_t s);   // BMI_ERR_UNSUPPORTED -> use conv_igemm
int launch_conv3x3_pw(const ConvArgs& a, hipStream_t s);      // 8x8 / 4x4 maps, Cout % 256 == 0; BMI_ERR_UNSUPPORTED -> conv3x3_patch
int launch_conv3x3_s2(const ConvArgs& a, hipStream_t s);      // 3x3 stride-2 convs, plain epilogue (pair mode too); BMI_ERR_UNSUPPORTED -> conv_igemm_wide
bool conv_takes_s2_kernel(int ksize, int stride, int pad, int cin, int cout, int h, int w, int ho, int wo);
int launch_conv(const ConvArgs& a, hipStream_t s, int* family = nullptr);   // picks the kernel; *family = BMI_CONV_FAMILY_*
int launch_stem_conv(const float* x, const float* w, const float* scale, const float* bias, _Float16* out, int n,
                     int cin, int h, int wdt, int cout, int ksize, int stride, int pad, int relu, int dt /* BMI_DTYPE_* of out */, hipStream_t s);
// the exact engine (BMI_DTYPE_F32, conv_exact.hip): the 16-bit pointer fields of ConvArgs / EltArgs hold fp32 tensors and fp32 weights
int launch_conv_exact(const ConvArgs& a, hipStream_t s);
// the split engines (BMI_DTYPE_F16X2 / BF16X3, conv_split.hip): fp32 tensors as above, a.wgt = 16-bit [2][Cout][k*k*Cin] head / tail planes
int launch_conv_split(const ConvArgs& a, int bf16, hipStream_t s);
bool conv_takes_split_kernel(int cin, int cout);
int launch_mask_apply_f32(const EltArgs& a, hipStream_t s);
int launch_maxpool2_f32(const float* in, float* out, int n, int h, int w, int c, hipStream_t s, int pair = 0);   // pair: 1 | 2 = pair32 tensors (fp16 | bf16)
int launch_mask_apply(const EltArgs& a, hipStream_t s);
int launch_maxpool2(const _Float16* in, _Float16* out, int n, int h, int w, int c, int bf16, hipStream_t s);
// hidden dense layer, fp32 weights [cout][k] / accumulate / output; `in` 16-bit (in_kind 0: fp16, 2: bf16) or fp32 (1) [n or in_mod][k]
int launch_dense_f32(const void* in, int in_kind, const float* w, const float* bias, float* out, int n, int in_mod, int k,
                     int cout, int relu, const SiteArgs& site, int batch, int t0, hipStream_t s);
int launch_exit_decide(const double* S1e, int C, int t_total, double thr, const int* in, int bc, int* out, int* count,
                       int* exit_of, int e, hipStream_t s);
int launch_fill_int(int* p, int n, int v, hipStream_t s);
// dst[r][c] = src[(cnt0 + r * stride) % m][c], r < m (a Masksembles table in the order a strided walk visits it)
int launch_mask_permute(const float* src, float* dst, int m, int c, int cnt0, int stride, hipStream_t s);
int launch_expand_rows(const int* active, int bc, int batch, int tc, int* rows, hipStream_t s);   // rows[tl*bc + i] = tl*batch + active[i]
int launch_finalize(int64_t n, int t_total, const double* S1, const double* S2, const double* SL, double* mean,
                    double* var, double* lm, int* nonfinite, hipStream_t s);
int launch_philox_mask(uint8_t* keep, int64_t n, uint64_t seed, int site, int t, float p, hipStream_t s);
// planar_w > 0: the bits in the lazy site's planar layout (rows of planar_w pixels; 2-bit sites, c % 64 == 0)
int launch_mask_bits(uint8_t* bits, int n, int hw, int c, const SiteArgs& site, int batch, int t0, hipStream_t s, int planar_w = 0);
// out = round16(in * scale); planar_w > 0: images of planar_hw pixels x planar_c channels stored in the planar layout
int launch_scale_copy(const _Float16* in, _Float16* out, long n, float scale, int bf16, hipStream_t s, int planar_hw = 0, int planar_w = 0, int planar_c = 0);
int launch_splitk_finish(const ConvArgs& a, hipStream_t s);   // after a split-K conv_igemm launch
int launch_conv1x1_seam(const ConvArgs& a, const ConvArgs& b, hipStream_t s);   // conv1x1_seam.hip: a = expand conv (+ residual), b = the reduce conv that reads a.out
bool conv_takes_seam_kernel(int cmid, int cw, int cn);
int launch_conv1x1_stream(const ConvArgs& a, hipStream_t s);
bool conv_takes_patch_kernel(int ksize, int stride, int pad, int cin, int cout, int ho, int wo);

// Kernel-selection switches.  bmi_set_option edits the PROCESS DEFAULTS; bmi_create snapshots them into the engine handle, and every entry point
// that takes a handle runs under that snapshot (a thread-local pointer set for the duration of the call: BmiOptionScope), so a live engine
// never changes kernels because another thread — or a test — called bmi_set_option; bmi_engine_set_option edits ONE engine's snapshot.
// The single-kernel entry points (no handle) read the process defaults.
struct BmiOptions {
    int mfma_shape_patch = 0, mfma_shape_wide = 0;   // 16 or 32 (0 until first use: the BMI_MFMA_SHAPE environment default)
    int unit_dtype = 0;          // BMI_DTYPE_* of the single-kernel entry points
    int wide_persist_min = 10;   // persistent wide kernel when blocks * 10 > value * n_cu (> one tile per CU; same-process A/B at T = 13, 25, 50: neutral vs 2 tiles per CU)
    int conv_pw = 1;             // 1: conv3x3_pw takes the shapes it supports, 0: conv3x3_patch everywhere
    int conv_wide = 1;           // 0: conv_igemm_wide is skipped (A/B against the per-tap kernel)
    int mask_lazy = 1;           // 1: a lazy site (engine.hip, bmi_create) writes keep bits + one scaled copy and its consumers mask in LDS, 0: always materialised
    int conv_pool = 1;           // 1: a conv whose 4x4 map feeds one exit head only writes the pooled means (conv3x3_s2), 0: never
    int conv_s2 = 1;             // 1: plain 3x3 stride-2 convs run in conv3x3_s2 (2 = without its minimum-grid rule: tests), 0: conv_igemm_wide
    int split_shx = 1;           // 1: conv_split's 3x3 stride-1 launches fetch a tap row's pixel tile once for its three taps (0: once per tap: A/B, tests)
    int split_tile = 1;          // 1: conv_split narrows its channel tile on small grids (0: always the widest that divides Cout: A/B, tests)
    int conv_seam = 1;           // 1: expand conv + residual of Bottleneck k and the reduce conv of Bottleneck k+1 run as one conv1x1_seam launch (2 = without its minimum-grid rule: tests), 0: two launches
    int conv_stream = 1;         // 1: HBM-bound 1x1 convs run in conv1x1_stream (2 = without its minimum-grid rule: tests), 0: never
    int splitk = 1;              // 1: bmi_plan gives skinny deterministic 3x3 convs (<= 64 tiles, Cin >= 256) a split-K launch
    int dense_exact = 0;         // 1: hidden dense layers on the exact-f32 MFMA instead of the split-fp16 form
    int lazy_order = 1;          // 1: the readers of a lazy site walk their tiles sample-minor (lazy_tile_map): speed only, never results
    int epilogue_lite = 1;       // 1: BN + residual + ReLU + 2-bit elementwise-site launches finish in epilogue_lite
    int xcd_split = 0;           // 0 = by weight bytes, else 1 | 2 | 4
    int pw_persist = 1;          // 1: plain-epilogue launches of conv3x3_pw run in its persistent form (conv3x3_pwp_kernel), 0: never
    int lazy_planar = 1;         // 1: lazy sites whose readers are all stride-2 consumers store their scaled copy + bits in the planar layout
    int ws_no_reuse = 0;         // bmi_plan: every suffix tensor keeps its own workspace range (per-layer traces)
    int conv_patch64 = 1;        // 1: 64 -> 64-class 3x3 stride-1 convs on 32-wide maps run in conv3x3_patch's 64-channel tile, 0: conv_igemm
    int splitk_tiles = 64;       // bmi_plan: a deterministic 3x3 conv (Cin >= 256, no residual / shortcut / site) of at most this many 128 x 128 tiles runs split-K
    int pair_prefix = 1;         // bmi_create: pair fusion (two plain convs on one input in one launch) in the once-per-batch prefix too (0: suffix only)
    int patch_direct = 1;        // 1: conv3x3_patch's BasicBlock tails on 16x16 maps (residual, residual + 2-bit site) finish on the accumulator registers (2: the plain launches too), 0: through LDS
    int head_batch = 1;          // 1: consecutive exit heads of the suffix (exit-only dropout: the suffix is nothing but the heads) run as ONE launch, 0: one launch per head
};
BmiOptions& bmi_default_options();                 // the process defaults (engine.hip)
extern thread_local const BmiOptions* bmi_tl_options;   // the snapshot of the engine whose entry point is running on this thread, else null
inline const BmiOptions& bmi_options() { return bmi_tl_options ? *bmi_tl_options : bmi_default_options(); }
struct BmiOptionScope {
    const BmiOptions* prev;
    explicit BmiOptionScope(const BmiOptions* o) : prev(bmi_tl_options) { bmi_tl_options = o; }
    ~BmiOptionScope() { bmi_tl_options = prev; }
    BmiOptionScope(const BmiOptionScope&) = delete;
    BmiOptionScope& operator=(const BmiOptionScope&) = delete;
};
#define BMI_OPT(name) inline int opt_##name() { return bmi_options().name; }
BMI_OPT(mfma_shape_patch) BMI_OPT(mfma_shape_wide) BMI_OPT(unit_dtype) BMI_OPT(wide_persist_min) BMI_OPT(conv_pw) BMI_OPT(conv_wide)
BMI_OPT(mask_lazy) BMI_OPT(conv_pool) BMI_OPT(conv_s2) BMI_OPT(split_shx) BMI_OPT(split_tile) BMI_OPT(conv_seam) BMI_OPT(conv_stream)
BMI_OPT(splitk) BMI_OPT(dense_exact) BMI_OPT(lazy_order) BMI_OPT(epilogue_lite) BMI_OPT(xcd_split) BMI_OPT(pw_persist) BMI_OPT(lazy_planar)
BMI_OPT(ws_no_reuse) BMI_OPT(head_batch) BMI_OPT(conv_patch64) BMI_OPT(splitk_tiles) BMI_OPT(pair_prefix) BMI_OPT(patch_direct)
#undef BMI_OPT
int xcd_split_for(int n_ctiles, size_t weight_bytes);

SiteArgs resolve_site(const bmi_site* site, uint64_t seed, int mask_cnt0, uint64_t elem_off = 0);

#define BMI_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return BMI_ERR_HIP;          \
    } while (0)
