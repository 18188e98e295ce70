// Graph executor behind the C ABI (include/bayesnn_fpga_amd.h).
//
// What it replaces in the reference: the Python T-loop of FullAnalysis._get_output
// (SA/train/results_analyzer.py:236-248) around ResNet18MCEarlyExit.forward
// (SA/models/resnet18/resnet18.py:302-346).  Instead of T sequential full-model forwards it
//   1. marks every tensor that does not depend on a stochastic site as DETERMINISTIC and runs
//      that prefix once per batch (the reference's own cost model assumes exactly this split,
//      results_analyzer.py:632-637);
//   2. folds `chunk` Monte-Carlo samples into the GEMM M dimension of every suffix op
//      (image index n = t_local * B + b), so weights are read once per chunk;
//   3. accumulates softmax moments per exit in float64 on the device.
// Activation buffers live in ONE caller-owned workspace; the suffix tensors are packed by live range (first-fit): 7 GB of
// the 288 GB for the bench's 25 500-image-sample chunk (large chunks won every A/B, so activations do round-trip HBM).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "conv_epilogue.h"
#include "kernels.h"

namespace {

struct TensorInfo {
    int h = 0, w = 0, c = 0;
    bool stoch = false;
    bool bits = false;          // holds keep bits (1 bit per element) instead of fp16 activations
    bool f32 = false;           // 4-byte elements: fp32 (output of a DENSE op; every tensor of the exact engine) or pair32 (the split engines' conv / stem / site / pool tensors)
    bool dense_out = false;     // written by a DENSE op: plain fp32 in every engine
    bool pooled_now = false;    // (run time) the producing conv of this chunk wrote fp32 means over its 4x4 map instead of the tensor
    // Lazy site (the output of an elementwise MASK op on a deterministic tensor, see bmi_create): tensors that hold the keep bits of
    // the folded batch and the deterministic input times 1/(1-p) (fp16), or -1
    int lazy_bits = -1, lazy_scaled = -1;
    bool lazy_planar_plan = false;   // (bmi_plan) ... and every one of those readers' launches passes its kernel's minimum-grid rule at the planned chunk
    bool lazy_planar = false;   // every reader is a stride-2 consumer (conv3x3_s2 on 32x32 maps / conv3x3_patch's fused shortcut) and the site draws 2 bits
                                // per element: bits + scaled copy are stored in the planar layout (kernels.h lazy_planar_off)
    bool lazy_pending = false;  // (run time) bits + scaled copy are written, the masked tensor itself is not (yet)
    bool lazy_planar_now = false;   // (run time) ... in the planar layout
    EltArgs lazy_call;          // (run time) the mask launch that materialises it on demand
    int first = -1, last = -1;  // suffix op indices (stochastic tensors only)
    size_t offset = 0;          // byte offset in the workspace
};

#define OP_MASKBITS 100   // internal: a MASK op rewritten to emit keep bits for its conv_igemm consumers

struct OpInfo {
    bmi_op_desc d;
    bool stoch = false;
    int ho = 0, wo = 0, cout = 0;
    int bits_tensor = -1;   // CONV: keep bits applied to the input while staging, or -1
    float out_mul = 1.f;    // CONV: multiplies the folded-BN scale (1/(1-p) of the input-side site)
    int nsplit = 0;         // CONV (prefix): split-K workgroups per tile (bmi_plan; 0 = none)
    bool pool_ok = false;   // CONV: its 4x4 output feeds ONE exit head and nothing else: conv3x3_s2 may write the pooled means instead
    bool pair_pool_ok = false;   // ... the same for the second conv of a pair
    bool pool_pw_ok = false;     // CONV (3x3 stride 1, 4x4 map, feeds ONE exit head only): conv3x3_pw may write the pooled means instead
    bool has_pair = false;  // CONV: a second conv on the same input rides in this launch (conv_igemm_wide pair mode)
    bmi_op_desc pair_d;
    int pair_cout = 0;
    bool has_seam = false;  // CONV (1x1 expand + residual + ReLU of a Bottleneck): the NEXT op, a plain 1x1 conv that reads this conv's output (conv1 of the
    bmi_op_desc seam_d;     // following Bottleneck), rides in this launch when conv1x1_seam takes it (else the two launches, in order)
    int seam_cout = 0;
};

struct ProfRec {
    int slot;
    hipEvent_t a, b;
    int family = -1;     // BMI_CONV_FAMILY_* of a conv launch
    double flops = 0;    // its algorithmic FLOPs
    double bytes = 0;    // its algorithmic HBM bytes (inputs + residual + weights once, output once)
    int out = -1;        // output tensor of the op (identifies the op in the graph)
    int images = 0;      // images (samples x batch) the launch carried
    float ms = 0.f;      // filled by bmi_profile_read
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct bmi_engine_s {
    BmiOptions opts;            // the process defaults at bmi_create (bmi_engine_set_option edits this copy): every entry point below runs under it
    std::vector<TensorInfo> tensors;
    std::vector<OpInfo> prefix, suffix;
    int n_exits = 0, out_dim = 0;
    int bf16 = 0;               // BMI_DTYPE_BF16
    int f32 = 0;                // fp32 activations in the workspace: BMI_DTYPE_F32 (the exact engine: fp32 conv weights, conv_exact.hip) and the split engines
    int split = 0;              // BMI_DTYPE_F16X2 (1) / BMI_DTYPE_BF16X3 (2): 16-bit head + tail weight planes, conv_split.hip
    int dtype = 0;              // BMI_DTYPE_*
    bool no_reuse = false;      // bmi_plan: every suffix tensor keeps its own workspace range ("ws_no_reuse": per-layer traces)
    int64_t prefix_macs = 0, suffix_macs = 0;
    // plan
    int max_batch = 0, chunk = 0;
    size_t ws_bytes = 0, exit_off = 0;   // exit_off: 2 active-image lists + a counter (dynamic early exit)
    size_t splitk_off = 0;               // fp32 partial sums of the split-K prefix convs
    int image_offset = 0;                // batch index of the current call's image 0 (bmi_forward_mcd_images), else 0
    size_t head_off = 0, head_part_bytes = 0;   // float64 partial sums of a head launch's 32-sample groups (joined in group order), one region per exit
    // bmi_forward_mcd_samples: per-sample logits out, Masksembles masks walked with a stride
    float* logits_out = nullptr;         // (run time) [t_count][E][batch][C] of the current call, or null
    bool no_moments = false;             // (run time) the heads write per-sample logits only
    int logits_t_begin = 0, logits_batch = 0;
    int mask_stride = 1, mask_t_begin = 0;    // (run time) mask_permuted: a site's mask of sample t is row (t - mask_t_begin) % M of its PERMUTED table
    bool mask_permuted = false;
    std::vector<std::pair<const float*, size_t>> perm;   // (bmi_plan) Masksembles tables (device pointer of the site) -> workspace offset of the permuted copy
    // profiling
    bool profiling = false;
    double fam_ms[BMI_CONV_FAMILIES] = {0}, fam_flops[BMI_CONV_FAMILIES] = {0}, fam_bytes[BMI_CONV_FAMILIES] = {0};
    int64_t fam_launches[BMI_CONV_FAMILIES] = {0};
    std::vector<ProfRec> recs, last;   // last: the launches of the most recent bmi_profile_read (bmi_profile_launches)
    std::vector<hipEvent_t> pool;
};

SiteArgs resolve_site(const bmi_site* site, uint64_t seed, int mask_cnt0, uint64_t elem_off) {
    SiteArgs s;
    std::memset(&s, 0, sizeof(s));
    s.scale = 1.f;
    if (!site) return s;
    s.elem_off = elem_off;
    s.kind = site->kind;
    s.site_id = site->site_id;
    s.seed_lo = (uint32_t)seed;
    s.seed_hi = (uint32_t)(seed >> 32);
    if (site->kind == BMI_SITE_ELEMENTWISE || site->kind == BMI_SITE_CHANNEL) {
        s.log2_bits = bmi_site_log2_bits(site->p);
        s.thresh = bmi_drop_threshold(site->p, s.log2_bits, &s.drop_all);
        s.scale = bmi_drop_scale(site->p);
    } else if (site->kind == BMI_SITE_MASKSEMBLE) {
        s.masks = site->masks;
        s.num_masks = site->num_masks;
        s.cnt0 = mask_cnt0;
    }
    return s;
}

// Kernel selection for one conv launch: LDS-tile 3x3 patch kernel -> wide-tile per-tap implicit GEMM (Cout % 256 == 0)
// -> per-tap implicit GEMM.
// BMI_CONV_IMPL=igemm skips the patch kernel (A/B, tests).
int launch_conv(const ConvArgs& a, hipStream_t s, int* family) {
    int fam_dummy;
    if (!family) family = &fam_dummy;
    static const int mode = [] {
        const char* v = std::getenv("BMI_CONV_IMPL");
        return (v && std::strcmp(v, "igemm") == 0) ? 2 : 1;
    }();
    if (mode <= 1) {
        *family = BMI_CONV_FAMILY_PW;
        const int rcw = launch_conv3x3_pw(a, s);
        if (rcw != BMI_ERR_UNSUPPORTED) return rcw;
        *family = BMI_CONV_FAMILY_PATCH;
        const int rc = launch_conv3x3_patch(a, s);
        if (rc != BMI_ERR_UNSUPPORTED) return rc;
    }
    {
        *family = BMI_CONV_FAMILY_STREAM;
        const int rc = launch_conv1x1_stream(a, s);
        if (rc != BMI_ERR_UNSUPPORTED) return rc;
    }
    {
        *family = BMI_CONV_FAMILY_S2;
        const int rc = launch_conv3x3_s2(a, s);
        if (rc != BMI_ERR_UNSUPPORTED) return rc;
    }
    if (opt_conv_wide()) {
        *family = BMI_CONV_FAMILY_WIDE;
        const int rc = launch_conv_igemm_wide(a, s);
        if (rc != BMI_ERR_UNSUPPORTED) return rc;
    }
    *family = BMI_CONV_FAMILY_IGEMM;
    return launch_conv_igemm(a, s);
}

static bool site_ok(const bmi_site& s) {
    switch (s.kind) {
        case BMI_SITE_NONE: return true;
        case BMI_SITE_ELEMENTWISE:
        case BMI_SITE_CHANNEL: return s.p >= 0.f && s.p <= 1.f && s.site_id >= 0;
        case BMI_SITE_MASKSEMBLE: return s.num_masks > 0 && s.masks != nullptr && s.site_id >= 0;
        default: return false;
    }
}

static int shape_from_env(const char* primary, const char* fallback) {
    const char* v = std::getenv(primary);
    if (!v && fallback) v = std::getenv(fallback);
    const int x = v ? std::atoi(v) : 0;
    return x == 16 || x == 32 ? x : BMI_DEFAULT_MFMA_SHAPE;
}
static BmiOptions initial_options() {
    BmiOptions o;
    o.mfma_shape_patch = shape_from_env("BMI_MFMA_SHAPE", nullptr);
    o.mfma_shape_wide = shape_from_env("BMI_MFMA_SHAPE_WIDE", "BMI_MFMA_SHAPE");
    o.unit_dtype = BMI_DTYPE_F16;
    const char* e = std::getenv("BMI_XCD_SPLIT");
    const int x = e ? std::atoi(e) : 0;
    o.xcd_split = x == 1 || x == 2 || x == 4 ? x : 0;
    return o;
}
BmiOptions& bmi_default_options() { static BmiOptions o = initial_options(); return o; }
thread_local const BmiOptions* bmi_tl_options = nullptr;
static int unit_pair() { const int d = opt_unit_dtype(); return d == BMI_DTYPE_F16X2 ? 1 : (d == BMI_DTYPE_BF16X3 ? 2 : 0); }
static bool unit_f32act() { const int d = opt_unit_dtype(); return d == BMI_DTYPE_F32 || d == BMI_DTYPE_F16X2 || d == BMI_DTYPE_BF16X3; }

// One named switch of an option set (bmi_set_option: the process defaults; bmi_engine_set_option: one engine's snapshot).
static int set_named_option(BmiOptions& o, const char* name, int32_t value) {
    if (!name) return BMI_ERR_INVALID;
    struct Row { const char* name; int BmiOptions::*field; int lo, hi; };
    static const Row rows[] = {
        {"unit_entry_dtype", &BmiOptions::unit_dtype, BMI_DTYPE_F16, BMI_DTYPE_BF16X3},
        {"pw_persist", &BmiOptions::pw_persist, 0, 1},
        {"lazy_planar", &BmiOptions::lazy_planar, 0, 1},
        {"ws_no_reuse", &BmiOptions::ws_no_reuse, 0, 1},                 // read by bmi_plan
        {"wide_persist_min_x10", &BmiOptions::wide_persist_min, 10, 1000},
        {"conv_pw", &BmiOptions::conv_pw, 0, 4},
        {"conv_s2", &BmiOptions::conv_s2, 0, 2},
        {"conv_pool", &BmiOptions::conv_pool, 0, 2},
        {"mask_lazy", &BmiOptions::mask_lazy, 0, 1},
        {"conv_wide", &BmiOptions::conv_wide, 0, 1},
        {"split_shx", &BmiOptions::split_shx, 0, 2},
        {"split_tile", &BmiOptions::split_tile, 0, 1},
        {"conv_seam", &BmiOptions::conv_seam, 0, 3},                      // read by bmi_create (which ops merge) and at launch
        {"conv_stream", &BmiOptions::conv_stream, 0, 3},
        {"splitk", &BmiOptions::splitk, 0, 1},                            // read by bmi_plan
        {"dense_exact", &BmiOptions::dense_exact, 0, 1},
        {"lazy_order", &BmiOptions::lazy_order, 0, 1},
        {"epilogue_lite", &BmiOptions::epilogue_lite, 0, 2},
        {"head_batch", &BmiOptions::head_batch, 0, 1},
        {"conv_patch64", &BmiOptions::conv_patch64, 0, 1},
        {"splitk_tiles", &BmiOptions::splitk_tiles, 0, 1024},            // read by bmi_plan
        {"pair_prefix", &BmiOptions::pair_prefix, 0, 1},                  // read by bmi_create
        {"patch_direct", &BmiOptions::patch_direct, 0, 2},
    };
    for (const Row& r : rows)
        if (std::strcmp(name, r.name) == 0) {
            if (value < r.lo || value > r.hi) return BMI_ERR_INVALID;
            o.*(r.field) = value;
            return BMI_OK;
        }
    if (std::strcmp(name, "xcd_split") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return BMI_ERR_INVALID;
        o.xcd_split = value;
        return BMI_OK;
    }
    const bool patch = std::strcmp(name, "mfma_shape_patch") == 0, wide = std::strcmp(name, "mfma_shape_wide") == 0;
    if (!patch && !wide) return BMI_ERR_INVALID;
    if (value != 0 && value != 16 && value != 32) return BMI_ERR_INVALID;
    (patch ? o.mfma_shape_patch : o.mfma_shape_wide) = value ? value : BMI_DEFAULT_MFMA_SHAPE;
    return BMI_OK;
}
// Channel-tile classes for xcd_tile_map: keep the weights one XCD streams under ~2.5 MB of its 4 MB L2.
int xcd_split_for(int n_ctiles, size_t weight_bytes) {
    int cs = opt_xcd_split();
    if (cs == 0) cs = weight_bytes > (8u << 20) ? 4 : (weight_bytes > (3u << 20) ? 2 : 1);
    while (cs > 1 && n_ctiles % cs != 0) cs >>= 1;
    return cs;
}

extern "C" {

int bmi_version(void) { return BMI_VERSION; }

int bmi_set_option(const char* name, int32_t value) { return set_named_option(bmi_default_options(), name, value); }

const char* bmi_error_string(int code) {
    switch (code) {
        case BMI_OK: return "ok";
        case BMI_ERR_INVALID: return "invalid argument or descriptor";
        case BMI_ERR_NOMEM: return "workspace too small";
        case BMI_ERR_HIP: return "HIP runtime / launch failure";
        case BMI_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        default: return "unknown error";
    }
}

int bmi_create(const bmi_model_desc* desc, bmi_handle* out) {
    if (!desc || !out || desc->n_tensors < 2 || desc->n_ops < 1 || !desc->tensors || !desc->ops) return BMI_ERR_INVALID;
    if (desc->n_exits < 1 || desc->out_dim < 1) return BMI_ERR_INVALID;
    if (desc->out_dim > 128) return BMI_ERR_UNSUPPORTED;
    if (desc->dtype < BMI_DTYPE_F16 || desc->dtype > BMI_DTYPE_BF16X3) return BMI_ERR_INVALID;
    bmi_engine_s* e = new (std::nothrow) bmi_engine_s();
    if (!e) return BMI_ERR_NOMEM;
    e->opts = bmi_default_options();
    BmiOptionScope opt_scope(&e->opts);
    e->n_exits = desc->n_exits;
    e->out_dim = desc->out_dim;
    e->bf16 = desc->dtype == BMI_DTYPE_BF16;
    e->split = desc->dtype == BMI_DTYPE_F16X2 ? 1 : (desc->dtype == BMI_DTYPE_BF16X3 ? 2 : 0);
    e->f32 = desc->dtype == BMI_DTYPE_F32 || e->split;
    e->dtype = desc->dtype;
    e->tensors.resize(desc->n_tensors);
    for (int i = 0; i < desc->n_tensors; ++i) {
        const bmi_tensor_desc& t = desc->tensors[i];
        if (t.h < 1 || t.w < 1 || t.c < 1) { delete e; return BMI_ERR_INVALID; }
        e->tensors[i].h = t.h; e->tensors[i].w = t.w; e->tensors[i].c = t.c;
        e->tensors[i].f32 = e->f32 && i > 0;   // the exact engine keeps every activation in fp32
    }
    std::vector<char> written(desc->n_tensors, 0);
    written[0] = 1;  // network input
    std::vector<char> exit_seen(desc->n_exits, 0);
    int rc = BMI_OK;
    for (int k = 0; k < desc->n_ops && rc == BMI_OK; ++k) {
        OpInfo op;
        op.d = desc->ops[k];
        const bmi_op_desc& d = op.d;
        auto tensor_ok = [&](int id) { return id >= 0 && id < desc->n_tensors; };
        if (!tensor_ok(d.in) || !written[d.in] || !site_ok(d.site)) { rc = BMI_ERR_INVALID; break; }
        const TensorInfo tin = e->tensors[d.in];
        bool in_st = tin.stoch;
        if (tin.f32 && !e->f32 && d.kind != BMI_OP_DENSE && d.kind != BMI_OP_HEAD) { rc = BMI_ERR_UNSUPPORTED; break; }
        switch (d.kind) {
            case BMI_OP_STEM:
            case BMI_OP_CONV: {
                if (!tensor_ok(d.out) || d.out == 0 || written[d.out] || !d.weight || d.ksize < 1 || d.stride < 1 || d.pad < 0) {
                    rc = BMI_ERR_INVALID; break;
                }
                if ((d.kind == BMI_OP_STEM) != (d.in == 0)) { rc = BMI_ERR_INVALID; break; }
                if (d.site_pos != BMI_SITE_POS_OUTER && d.site_pos != BMI_SITE_POS_INNER) { rc = BMI_ERR_INVALID; break; }
                const bool inner = d.site_pos == BMI_SITE_POS_INNER && d.site.kind != BMI_SITE_NONE;
                if (inner && (d.in2 >= 0 || d.site.kind == BMI_SITE_MASKSEMBLE)) { rc = BMI_ERR_UNSUPPORTED; break; }
                const TensorInfo to = e->tensors[d.out];  // by value: the split below grows the vector
                op.ho = (tin.h + 2 * d.pad - d.ksize) / d.stride + 1;
                op.wo = (tin.w + 2 * d.pad - d.ksize) / d.stride + 1;
                op.cout = to.c;
                if (op.ho != to.h || op.wo != to.w) { rc = BMI_ERR_INVALID; break; }
                if (d.kind == BMI_OP_CONV && !e->f32 && (tin.c % 64 != 0 || to.c % 64 != 0)) { rc = BMI_ERR_UNSUPPORTED; break; }
                // the engines with fp32 activations: one generic kernel each (conv_exact / conv_split), 32-deep K-steps, 64-channel tiles
                if (d.kind == BMI_OP_CONV && e->f32 && (tin.c % 32 != 0 || to.c % 64 != 0)) { rc = BMI_ERR_UNSUPPORTED; break; }
                if (d.kind == BMI_OP_STEM && (to.c % 8 != 0 || to.c * d.ksize * d.ksize * tin.c > 4096 || d.residual >= 0)) {
                    rc = BMI_ERR_UNSUPPORTED; break;
                }
                if (d.residual >= 0) {
                    if (!tensor_ok(d.residual) || !written[d.residual] || d.residual == 0) { rc = BMI_ERR_INVALID; break; }
                    const TensorInfo& tr = e->tensors[d.residual];
                    if (tr.h != to.h || tr.w != to.w || tr.c != to.c) { rc = BMI_ERR_INVALID; break; }
                    in_st = in_st || tr.stoch;
                }
                int64_t macs = (int64_t)op.ho * op.wo * op.cout * d.ksize * d.ksize * tin.c;
                if (d.kind == BMI_OP_CONV && d.in2 >= 0 && e->f32 && !e->split) { rc = BMI_ERR_UNSUPPORTED; break; }   // a speed feature (BN scales folded into the weights): never in the exact engine
                if (d.kind == BMI_OP_CONV && d.in2 >= 0) {
                    if (!tensor_ok(d.in2) || !written[d.in2] || d.in2 == 0 || !d.weight2 || (d.scale && !e->split)) { rc = BMI_ERR_INVALID; break; }   // (split engines: the host's per-channel power-of-two lift comes back as `scale`)
                    const TensorInfo& t2 = e->tensors[d.in2];
                    if (t2.h % op.ho != 0 || t2.h / op.ho != t2.w / op.wo || t2.w % op.wo != 0) { rc = BMI_ERR_UNSUPPORTED; break; }
                    // fp16 / bf16: conv3x3_patch / conv3x3_pw carry the shortcut; the split engines: extra K-steps of conv_split (any conv geometry)
                    if (e->split ? t2.c % 32 != 0
                                 : (d.ksize != 3 || d.stride != 1 || d.pad != 1 || t2.c % 64 != 0 || !conv_takes_patch_kernel(3, 1, 1, tin.c, op.cout, op.ho, op.wo))) {
                        rc = BMI_ERR_UNSUPPORTED; break;
                    }
                    in_st = in_st || t2.stoch;
                    macs += (int64_t)op.ho * op.wo * op.cout * t2.c;
                }
                if (!in_st && d.site.kind != BMI_SITE_NONE) {
                    // deterministic conv feeding a site: keep the conv in the once-per-batch prefix
                    // and apply the site while expanding to the folded sample batch.
                    if (inner && d.residual >= 0) { rc = BMI_ERR_UNSUPPORTED; break; }
                    TensorInfo tmp = to;
                    tmp.stoch = false;
                    e->tensors.push_back(tmp);
                    const int tmp_id = (int)e->tensors.size() - 1;
                    OpInfo conv = op;
                    conv.d.out = tmp_id;
                    conv.d.site.kind = BMI_SITE_NONE;
                    conv.d.site_pos = BMI_SITE_POS_OUTER;
                    conv.d.bias_post = nullptr;
                    if (inner) conv.d.relu = 0;   // the prefix keeps conv*scale+bias; mask, BN shift and ReLU follow in the MASK op
                    conv.stoch = false;
                    e->prefix.push_back(conv);
                    e->prefix_macs += macs;
                    OpInfo m;
                    std::memset(&m.d, 0, sizeof(m.d));
                    m.d.kind = BMI_OP_MASK;
                    m.d.in = tmp_id;
                    m.d.out = d.out;
                    m.d.residual = -1;
                    m.d.in2 = -1;
                    m.d.site = d.site;
                    if (inner) { m.d.bias_post = d.bias_post; m.d.relu = d.relu; m.d.site_pos = BMI_SITE_POS_INNER; }
                    m.stoch = true;
                    m.ho = to.h; m.wo = to.w; m.cout = to.c;
                    e->suffix.push_back(m);
                    e->tensors[d.out].stoch = true;
                } else {
                    op.stoch = in_st || d.site.kind != BMI_SITE_NONE;
                    e->tensors[d.out].stoch = op.stoch;
                    (op.stoch ? e->suffix : e->prefix).push_back(op);
                    (op.stoch ? e->suffix_macs : e->prefix_macs) += macs;
                }
                written[d.out] = 1;
                break;
            }
            case BMI_OP_MASK: {
                if (!tensor_ok(d.out) || d.out == 0 || written[d.out] || d.in == 0 || d.site.kind == BMI_SITE_NONE ||
                    d.site_pos != BMI_SITE_POS_OUTER) {
                    rc = BMI_ERR_INVALID; break;
                }
                op.d.bias_post = nullptr;
                op.d.relu = 0;
                const TensorInfo& to = e->tensors[d.out];
                if (to.h != tin.h || to.w != tin.w || to.c != tin.c) { rc = BMI_ERR_INVALID; break; }
                if (tin.c % 8 != 0) { rc = BMI_ERR_UNSUPPORTED; break; }
                op.stoch = true;
                op.ho = to.h; op.wo = to.w; op.cout = to.c;
                e->tensors[d.out].stoch = true;
                e->suffix.push_back(op);
                written[d.out] = 1;
                break;
            }
            case BMI_OP_MAXPOOL: {
                if (!tensor_ok(d.out) || d.out == 0 || written[d.out] || d.in == 0) { rc = BMI_ERR_INVALID; break; }
                const TensorInfo& to = e->tensors[d.out];
                if (to.h * 2 != tin.h || to.w * 2 != tin.w || to.c != tin.c) { rc = BMI_ERR_INVALID; break; }
                if (tin.c % 8 != 0) { rc = BMI_ERR_UNSUPPORTED; break; }
                op.stoch = in_st;
                op.ho = to.h; op.wo = to.w; op.cout = to.c;
                e->tensors[d.out].stoch = in_st;
                (in_st ? e->suffix : e->prefix).push_back(op);
                written[d.out] = 1;
                break;
            }
            case BMI_OP_DENSE: {
                if (!tensor_ok(d.out) || d.out == 0 || written[d.out] || d.in == 0 || !d.weight || !d.bias ||
                    d.site_pos != BMI_SITE_POS_OUTER) {
                    rc = BMI_ERR_INVALID; break;
                }
                const TensorInfo& to = e->tensors[d.out];
                if (tin.h != 1 || tin.w != 1 || to.h != 1 || to.w != 1) { rc = BMI_ERR_INVALID; break; }
                if (tin.c % 32 != 0 || to.c % 64 != 0) { rc = BMI_ERR_UNSUPPORTED; break; }
                op.d.residual = -1; op.d.in2 = -1;
                // a site makes the layer per-sample even on a deterministic input (the layer is tiny: no conv + MASK split)
                op.stoch = in_st || d.site.kind != BMI_SITE_NONE;
                op.ho = 1; op.wo = 1; op.cout = to.c;
                e->tensors[d.out].stoch = op.stoch;
                e->tensors[d.out].f32 = true;
                e->tensors[d.out].dense_out = true;
                (op.stoch ? e->suffix : e->prefix).push_back(op);
                (op.stoch ? e->suffix_macs : e->prefix_macs) += (int64_t)tin.c * to.c;
                written[d.out] = 1;
                break;
            }
            case BMI_OP_HEAD: {
                if (d.in == 0 || d.out < 0 || d.out >= desc->n_exits || exit_seen[d.out] || !d.weight || !d.bias) {
                    rc = BMI_ERR_INVALID; break;
                }
                if (tin.c % 32 != 0) { rc = BMI_ERR_UNSUPPORTED; break; }   // head_fused splits K over 4 waves x 2 lane halves x float4
                if (d.site_pos == BMI_SITE_POS_INNER && d.site.kind != BMI_SITE_NONE && d.site.kind != BMI_SITE_ELEMENTWISE) {
                    rc = BMI_ERR_UNSUPPORTED; break;   // dropout on the logits is elementwise (F.dropout after nn.Linear)
                }
                exit_seen[d.out] = 1;
                op.stoch = true;  // heads always run per sample (they emit per-sample softmax)
                op.cout = desc->out_dim;
                e->suffix.push_back(op);
                e->suffix_macs += (int64_t)tin.c * desc->out_dim;
                break;
            }
            default: rc = BMI_ERR_INVALID;
        }
    }
    if (rc == BMI_OK)
        for (int x = 0; x < desc->n_exits; ++x)
            if (!exit_seen[x]) rc = BMI_ERR_INVALID;
    if (rc != BMI_OK) { delete e; return rc; }
    // Input-side dropout: an elementwise site that expands a deterministic tensor and is consumed only
    // as the input of convs that stage their input through registers (conv_igemm) is not materialised:
    // the MASK op emits keep bits (16x fewer bytes), the consumers read the deterministic tensor (L2 /
    // Infinity Cache resident, B images) and zero the dropped elements while staging; 1/(1-p) is
    // folded into their BN scale.  Same-box A/B on the headline config: HBM traffic of the site drops
    // 16x but the step time is unchanged (-0.47 ms in the mask kernel, +0.5 ms in the three consumers),
    // so it is OFF by default; BMI_MASK_BITS=1 enables it (covered by the parity tests).
    {
        const char* env = std::getenv("BMI_MASK_BITS");
        const bool enable = env && std::atoi(env) == 1 && !e->f32;
        for (size_t mi = 0; enable && mi < e->suffix.size(); ++mi) {
            OpInfo& m = e->suffix[mi];
            if (m.d.kind != BMI_OP_MASK || m.d.site.kind != BMI_SITE_ELEMENTWISE || e->tensors[m.d.in].stoch) continue;
            if (m.d.site_pos == BMI_SITE_POS_INNER) continue;   // carries a BN shift / ReLU: must be materialised
            if (m.d.site.p >= 1.f || e->tensors[m.d.in].c % 8 != 0) continue;
            bool ok = true;
            int uses = 0;
            for (const OpInfo& c : e->suffix) {
                if (&c == &m) continue;
                const bool as_in = c.d.in == m.d.out, as_res = c.d.kind == BMI_OP_CONV && c.d.residual == m.d.out;
                if (!as_in && !as_res) continue;
                ++uses;
                const TensorInfo& ti = e->tensors[m.d.out];
                if (as_res || c.d.kind != BMI_OP_CONV ||
                    conv_takes_patch_kernel(c.d.ksize, c.d.stride, c.d.pad, ti.c, c.cout, c.ho, c.wo))
                    ok = false;
            }
            if (!ok || uses == 0) continue;
            m.d.kind = OP_MASKBITS;
            e->tensors[m.d.out].bits = true;
            for (OpInfo& c : e->suffix)
                if (&c != &m && c.d.kind == BMI_OP_CONV && c.d.in == m.d.out) {
                    c.bits_tensor = m.d.out;
                    c.d.in = m.d.in;
                    c.out_mul = bmi_drop_scale(m.d.site.p);
                }
        }
    }
    // Pair fusion: two suffix convs that read the same tensor with the same geometry and a plain BN(+ReLU) epilogue
    // (layerN.0.conv1 and the first conv of the exit head in front of it, resnet18.py:306/:318/:329 vs :280-299)
    // run as ONE conv_igemm_wide launch: the input tile is fetched once for both and a 128-channel conv still fills
    // the kernel's 256-channel tile.  The later conv moves up to the earlier one's position (it depends on nothing
    // in between).  BMI_CONV_PAIR=0 keeps them separate (A/B, tests).
    {
        const char* env = std::getenv("BMI_CONV_PAIR");
        const bool enable = (!env || std::atoi(env) != 0) && (!e->f32 || e->split);      // (the exact engine's kernel has no pair mode)
        auto plain = [&](const OpInfo& c) {
            return c.d.kind == BMI_OP_CONV && !c.has_pair && c.d.residual < 0 && c.d.in2 < 0 && c.d.site.kind == BMI_SITE_NONE &&
                   c.bits_tensor < 0 && c.out_mul == 1.f && c.d.scale && c.d.bias;
        };
        auto merge = [&](std::vector<OpInfo>& ops, bool prefix) {
            for (size_t i = 0; i < ops.size(); ++i) {
                if (!plain(ops[i])) continue;
                const OpInfo A = ops[i];
                const TensorInfo& ti = e->tensors[A.d.in];
                // (prefix: a conv with 256+ input channels may get a split-K launch from bmi_plan on a small batch — VGG's exit convs, 37 -> 22 us —
                //  which a pair launch does not have: those stay single.  Measured: VGG-11 on f16x2 15.0 -> 14.4 M with every prefix pair merged)
                if (prefix && ti.c >= 256) continue;
                if (conv_takes_patch_kernel(A.d.ksize, A.d.stride, A.d.pad, ti.c, A.cout, A.ho, A.wo)) continue;
                for (size_t j = i + 1; j < ops.size(); ++j) {
                    const OpInfo& Bo = ops[j];
                    if (!plain(Bo) || Bo.d.in != A.d.in || Bo.d.ksize != A.d.ksize || Bo.d.stride != A.d.stride ||
                        Bo.d.pad != A.d.pad || Bo.d.relu != A.d.relu)
                        continue;
                    if (A.cout % 128 != 0 || !conv_takes_wide_kernel(ti.c, A.cout + Bo.cout)) continue;
                    if (e->split && Bo.cout % 128 != 0) continue;
                    ops[i].has_pair = true;
                    ops[i].pair_d = Bo.d;
                    ops[i].pair_cout = Bo.cout;
                    ops.erase(ops.begin() + (long)j);
                    break;
                }
            }
        };
        if (enable) merge(e->suffix, false);
        // "pair_prefix" (round 6): the same for the once-per-batch prefix — with exit-only dropout the whole network is prefix and the pairs are
        // there: the paper's configuration 3.60-3.63 M -> 3.70-3.77 M MCD-samples/s, same box (profiles/experiments/r6_exit_only_variants.txt).
        if (enable && opt_pair_prefix() && !e->f32) merge(e->prefix, true);
    }
    // Seam fusion (Bottleneck nets): conv3 + BN + residual + ReLU of block k followed at once by conv1 + BN + ReLU of block k+1 on its output:
    // one conv1x1_seam launch produces both tensors and the wide one is not read back (conv1x1_seam.hip).  Decided per launch in run_op.
    for (size_t i = 0; !e->f32 && opt_conv_seam() && i + 1 < e->suffix.size(); ++i) {
        OpInfo& A = e->suffix[i];
        const OpInfo& Bo = e->suffix[i + 1];
        auto one = [&](const OpInfo& c) {
            return c.d.kind == BMI_OP_CONV && c.stoch && !c.has_pair && c.d.in2 < 0 && c.d.site.kind == BMI_SITE_NONE && c.bits_tensor < 0 && c.out_mul == 1.f &&
                   c.d.ksize == 1 && c.d.stride == 1 && c.d.pad == 0 && c.d.scale && c.d.bias;
        };
        if (!one(A) || !one(Bo) || A.d.residual < 0 || !A.d.relu || Bo.d.residual >= 0 || Bo.d.in != A.d.out) continue;
        if (!e->tensors[A.d.in].stoch || !e->tensors[A.d.residual].stoch) continue;
        if (!conv_takes_seam_kernel(e->tensors[A.d.in].c, A.cout, Bo.cout)) continue;
        A.has_seam = true;
        A.seam_d = Bo.d;
        A.seam_cout = Bo.cout;
        e->suffix.erase(e->suffix.begin() + (long)i + 1);
    }
    // ReLU + global average pool fused into the producing conv: a plain 3x3 stride-2 conv whose 4x4 output map feeds ONE exit head and
    // nothing else (ex1conv3 / ex2conv2 / ex3conv1 of the ResNets: relu -> avg_pool2d(4) -> Linear, resnet18.py:309-314, :320-325,
    // :331-335) may write fp32 means [row][Cout] instead of the map when conv3x3_s2 takes the launch (decided per launch: run_op).
    for (std::vector<OpInfo>* ops : {&e->prefix, &e->suffix}) {
        auto readers = [&](int id, int* heads) {
            int n = 0;
            *heads = 0;
            for (const std::vector<OpInfo>* o2 : {&e->prefix, &e->suffix})
                for (const OpInfo& c : *o2) {
                    const bmi_op_desc& d = c.d;
                    const bool conv = d.kind == BMI_OP_CONV || d.kind == BMI_OP_STEM;
                    if (d.in == id) { ++n; if (d.kind == BMI_OP_HEAD) ++*heads; }
                    if (conv && d.residual == id) ++n;
                    if (conv && d.in2 == id) ++n;
                    if (c.bits_tensor == id) ++n;
                }
            return n;
        };
        for (OpInfo& c : *ops) {
            if (c.d.kind != BMI_OP_CONV || e->f32) continue;
            auto eligible = [&](const bmi_op_desc& d, int cout) {
                const TensorInfo& to = e->tensors[d.out];
                int heads = 0;
                return d.ksize == 3 && d.stride == 2 && d.pad == 1 && to.h == 4 && to.w == 4 && d.residual < 0 && d.in2 < 0 &&
                       d.site.kind == BMI_SITE_NONE && cout % 128 == 0 && readers(d.out, &heads) == 1 && heads == 1;
            };
            // ... and the last conv of the net (layer4[1].conv2: stride 1, with its residual) in front of the final head: conv3x3_pw's
            // lite epilogue does the same on its registers
            auto eligible_pw = [&](const bmi_op_desc& d, int cout) {
                const TensorInfo& to = e->tensors[d.out];
                int heads = 0;
                return d.ksize == 3 && d.stride == 1 && d.pad == 1 && to.h == 4 && to.w == 4 && d.in2 < 0 && d.relu &&
                       (d.site.kind == BMI_SITE_NONE || d.site_pos != BMI_SITE_POS_INNER) && cout % 256 == 0 && !c.has_pair &&
                       readers(d.out, &heads) == 1 && heads == 1;
            };
            c.pool_pw_ok = c.bits_tensor < 0 && eligible_pw(c.d, c.cout);
            c.pool_ok = c.bits_tensor < 0 && eligible(c.d, c.cout);
            c.pair_pool_ok = c.has_pair && eligible(c.pair_d, c.pair_cout);
        }
    }
    // live ranges of the stochastic tensors over the suffix
    for (int k = 0; k < (int)e->suffix.size(); ++k) {
        const bmi_op_desc& d = e->suffix[k].d;
        auto touch = [&](int id) {
            if (id < 0) return;
            TensorInfo& t = e->tensors[id];
            if (!t.stoch) return;
            if (t.first < 0) t.first = k;
            t.last = k;
        };
        touch(d.in);
        if (d.kind == BMI_OP_CONV) touch(d.in2);
        touch(e->suffix[k].bits_tensor);
        if (d.kind == BMI_OP_CONV) touch(d.residual);
        if (d.kind != BMI_OP_HEAD) touch(d.out);
        if (e->suffix[k].has_pair) touch(e->suffix[k].pair_d.out);
        if (e->suffix[k].has_seam) touch(e->suffix[k].seam_d.out);
    }
    // Lazy sites.  The first elementwise site of a "block"-dropout ResNet expands the once-per-batch prefix (B images) to the folded
    // batch: 3.3 GB written by the MASK op and read back by its consumers on the headline config.  Where a consumer can apply the
    // mask itself — conv3x3_s2 on 32x32 maps (clears the dropped elements of its patch pieces in LDS), conv3x3_patch for the input of
    // a fused shortcut on 16x16 maps — the op writes the keep bits (1/16 of the bytes) and ONE scaled copy of the B images
    // instead; kept x 1/(1-p) rounded to fp16 and ANDed with the bits is what the MASK op itself stores, so the result is bit for
    // bit the materialised one.  Decided per launch (run_op): a consumer whose kernel does not take the launch makes the
    // MASK op's own launch happen first ("mask_lazy" = 0: always).
    for (size_t mi = 0; mi < e->suffix.size() && !e->f32; ++mi) {
        const bmi_op_desc md = e->suffix[mi].d;
        if (md.kind != BMI_OP_MASK || md.site.kind != BMI_SITE_ELEMENTWISE || md.site_pos == BMI_SITE_POS_INNER || md.site.p >= 1.f) continue;
        const TensorInfo ti = e->tensors[md.in];
        if (ti.stoch || ti.c % 32 != 0) continue;
        // every reader must be able to apply the bits itself (else the tensor is written anyway and the bits are extra work):
        //   conv3x3_s2 on 32x32 maps (also as a pair launch); conv3x3_patch for the input of a fused shortcut on 16x16 maps;
        //   1x1 convs (conv1x1_stream clears the elements in LDS, conv_igemm while staging) and the 3x3 stride-2 convs that run in
        //   conv_igemm anyway (ResNet-50's first site: 256 -> 128 k3s2, 256 -> 128 k1, 256 -> 512 k1s2).  A 3x3 stride-1 reader would
        //   lose its patch kernel to the per-tap one: not lazy.
        int readers = 0;
        bool all = true, all_s2 = true;
        for (const OpInfo& c : e->suffix) {
            const bool reads = c.d.in == md.out || (c.d.kind == BMI_OP_CONV && (c.d.residual == md.out || c.d.in2 == md.out)) || c.bits_tensor == md.out;
            if (!reads || &c == &e->suffix[mi]) continue;
            ++readers;
            bool ok = false;
            // (round 6) conv3x3_patch's 64-channel tile — 3x3 stride-1 convs with Cout % 128 == 64 on 32-wide maps: the BasicBlocks behind the stem, which is
            // where the first "layer" site sits — clears the dropped elements of its input patch in LDS and of a residual where it is added
            auto p64 = [&](const OpInfo& q) {
                const TensorInfo& qi = e->tensors[q.d.in];
                return opt_conv_patch64() && q.d.kind == BMI_OP_CONV && !q.has_pair && q.d.in2 < 0 && q.d.ksize == 3 && q.d.stride == 1 && q.d.pad == 1 &&
                       qi.c % 64 == 0 && q.cout % 64 == 0 && q.cout % 128 != 0 && q.wo == 32 && q.ho % 8 == 0 && q.bits_tensor < 0;
            };
            if (p64(c) && c.d.in2 != md.out) {
                // as the input (any epilogue), and / or as the residual (the register-form epilogue: no site or the 2-bit elementwise one, outer)
                const bool res_ok = c.d.residual != md.out ||
                                    (c.d.site_pos != BMI_SITE_POS_INNER && (c.d.site.kind == BMI_SITE_NONE ||
                                                                            (c.d.site.kind == BMI_SITE_ELEMENTWISE && bmi_site_log2_bits(c.d.site.p) == 1 && c.d.site.p < 1.f)));
                if (res_ok) { all = all && true; all_s2 = false; continue; }
            }
            if (c.d.kind == BMI_OP_CONV && c.d.residual != md.out && c.bits_tensor < 0) {
                if (c.d.in2 == md.out) { ok = c.d.in != md.out && c.ho == 16 && c.wo == 16; all_s2 = all_s2 && ti.h == 2 * c.ho && ti.w == 2 * c.wo; }
                else if (c.d.in2 < 0) {
                    const bool s2 = ti.h == 32 && ti.w == 32 && c.d.residual < 0 && c.d.site.kind == BMI_SITE_NONE &&
                                    conv_takes_s2_kernel(c.d.ksize, c.d.stride, c.d.pad, ti.c, c.cout + (c.has_pair ? c.pair_cout : 0), ti.h, ti.w, c.ho, c.wo);
                    const bool igemm = !c.has_pair && ti.c % 64 == 0 && c.cout % 64 == 0 &&
                                       (c.d.ksize == 1 || (c.d.ksize == 3 && c.d.stride == 2));
                    ok = s2 || igemm;
                    all_s2 = all_s2 && s2;
                }
            }
            all = all && ok;
        }
        if (!all || readers == 0) continue;
        TensorInfo tb = ti, tsc = ti;
        tb.stoch = true; tb.bits = true; tb.first = e->tensors[md.out].first; tb.last = e->tensors[md.out].last;
        tsc.stoch = false; tsc.first = tsc.last = -1;
        e->tensors.push_back(tb);
        e->tensors.push_back(tsc);
        e->tensors[md.out].lazy_bits = (int)e->tensors.size() - 2;
        e->tensors[md.out].lazy_scaled = (int)e->tensors.size() - 1;
        e->tensors[md.out].lazy_planar = all_s2 && ti.c % 64 == 0 && ti.w >= 2 && (ti.w & (ti.w - 1)) == 0 && ((ti.h * ti.w) & (ti.h * ti.w - 1)) == 0 &&
                                         bmi_site_log2_bits(md.site.p) == 1;
    }
    *out = e;
    return BMI_OK;
}

int bmi_engine_set_option(bmi_handle h, const char* name, int32_t value) {
    if (!h) return BMI_ERR_INVALID;
    return set_named_option(h->opts, name, value);
}

int bmi_destroy(bmi_handle h) {
    if (!h) return BMI_ERR_INVALID;
    for (auto& r : h->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto& ev : h->pool) (void)hipEventDestroy(ev);
    delete h;
    return BMI_OK;
}

int bmi_plan(bmi_handle h, int32_t max_batch, int32_t chunk_samples, size_t* workspace_bytes) {
    if (!h || max_batch < 1 || chunk_samples < 1 || !workspace_bytes) return BMI_ERR_INVALID;
    BmiOptionScope opt_scope(&h->opts);
    const size_t B = (size_t)max_batch, NS = (size_t)max_batch * chunk_samples;
    for (const TensorInfo& t : h->tensors)  // pixel indices (N * H * W) stay inside int32
        if (NS * t.h * t.w >= 0x7fffffffull) return BMI_ERR_UNSUPPORTED;
    size_t off = 0;
    for (size_t i = 1; i < h->tensors.size(); ++i) {
        TensorInfo& t = h->tensors[i];
        if (t.stoch) continue;
        t.offset = off;
        off += align_up(B * t.h * t.w * t.c * (t.f32 ? 4 : 2), 256);
    }
    // first-fit packing of the suffix tensors by live range
    struct Blk { size_t off, size; int last; };
    std::vector<Blk> live;
    std::vector<int> order;
    for (size_t i = 1; i < h->tensors.size(); ++i)
        if (h->tensors[i].stoch && h->tensors[i].first >= 0) order.push_back((int)i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return h->tensors[a].first < h->tensors[b].first; });
    const size_t st_base = off;
    size_t st_peak = 0;
    h->no_reuse = opt_ws_no_reuse() != 0;
    for (int id : order) {
        TensorInfo& t = h->tensors[id];
        if (!h->no_reuse) live.erase(std::remove_if(live.begin(), live.end(), [&](const Blk& b) { return b.last < t.first; }), live.end());
        std::sort(live.begin(), live.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
        const size_t size = align_up(t.bits ? NS * t.h * t.w * t.c / 8 : NS * t.h * t.w * t.c * (t.f32 ? 4 : 2), 256);
        size_t pos = 0;
        for (const Blk& b : live) {
            if (pos + size <= b.off) break;
            pos = std::max(pos, b.off + b.size);
        }
        t.offset = st_base + pos;
        live.push_back({pos, size, t.last});
        st_peak = std::max(st_peak, pos + size);
    }
    off = st_base + st_peak;
    h->exit_off = off;
    off += align_up((2 * (size_t)max_batch + 64 + NS) * sizeof(int), 256);   // two image lists, a counter, the row table
    // Split-K for the skinny deterministic 3x3 convs (VGG's 512 -> 512 convs on 2x2 maps: 250 images are 1000 pixels = 32 tiles
    // of 128 x 128 on 256 CUs, 65 us at 73 TFLOP/s): one workgroup per (tile, tap), fp32 partial sums, a finishing pass.
    // Decided here from the shape and the planned batch only.
    h->splitk_off = off;
    size_t sk_bytes = 0;
    for (OpInfo& op : h->prefix) {
        op.nsplit = 0;
        const bmi_op_desc& d = op.d;
        if (d.kind != BMI_OP_CONV || (h->f32 && !h->split) || !opt_splitk() || d.ksize != 3 || d.residual >= 0 || d.in2 >= 0 || d.site.kind != BMI_SITE_NONE ||
            op.has_pair || op.bits_tensor >= 0 || op.cout % 128 != 0)
            continue;
        const TensorInfo& ti = h->tensors[d.in];
        const size_t M = B * op.ho * op.wo;
        if (h->split) {
            // the split engines (conv_split: 256-pixel tiles, 64-channel tiles on small grids): enough contiguous K ranges per tile for two
            // workgroups per CU (256 CUs), nine at most, four at least (three ranges of a 252-workgroup launch measured slower: 85 -> 110 us);
            // 72 K-steps (Cin = 256) or more
            const size_t blocks = (M + 255) / 256 * (op.cout / 64);
            const int ns = (int)std::min<size_t>(9, (512 + blocks - 1) / blocks);
            if (ti.c < 256 || ns < 4) continue;
            op.nsplit = ns;
            sk_bytes = std::max(sk_bytes, align_up((size_t)op.nsplit * M * op.cout * sizeof(float), 256));
            continue;
        }
        const size_t tiles = (M + 127) / 128 * (op.cout / 128);
        if (ti.c % 64 != 0 || ti.c < 256 || tiles > (size_t)opt_splitk_tiles()) continue;      // stride 1 or 2 (VGG-19's 256 -> 512 exit convs: 37 -> 22 us); at
                                                                       // Cin = 128 (18 K-steps) the split measured slower: 23 -> 28 us
        op.nsplit = 9;
        sk_bytes = std::max(sk_bytes, align_up((size_t)op.nsplit * M * op.cout * sizeof(float), 256));
    }
    off += sk_bytes;
    h->head_off = off;
    h->head_part_bytes = align_up((size_t)((chunk_samples + 31) / 32) * 3 * max_batch * h->out_dim * sizeof(double), 256);
    off += h->head_part_bytes * (size_t)h->n_exits;      // one region per exit: the heads of a batched launch (launch_head_fused_multi) run concurrently
    h->perm.clear();
    for (const std::vector<OpInfo>* ops : {&h->prefix, &h->suffix})
        for (const OpInfo& op : *ops) {
            const bmi_site& st = op.d.site;
            if (st.kind != BMI_SITE_MASKSEMBLE) continue;
            bool seen = false;
            for (const auto& pr : h->perm) seen = seen || pr.first == st.masks;
            if (seen) continue;
            const int width = op.d.kind == BMI_OP_HEAD ? h->tensors[op.d.in].c : op.cout;       // a site's table is [M][channels of its tensor]
            h->perm.push_back({st.masks, off});
            off += align_up((size_t)st.num_masks * width * sizeof(float), 256);
        }
    h->ws_bytes = off;
    h->max_batch = max_batch;
    h->chunk = chunk_samples;
    // A lazy site keeps its PLANAR layout only while every stride-2 reader's launch passes conv3x3_s2's minimum-grid rule at the planned
    // full chunk (n_ref = max_batch x chunk: what the launcher looks at; 256 CUs): a reader that declines makes run_op materialise the
    // tensor — correct either way, but the planar bits + copy would then have been written for nothing (and conv_igemm / conv1x1_stream
    // refuse a planar operand).  Small engines (tests, T = 1 mirrors) therefore plan NHWC lazy sites.
    for (TensorInfo& t : h->tensors) {
        t.lazy_planar_plan = t.lazy_planar;
        if (!t.lazy_planar) continue;
        const int id = (int)(&t - h->tensors.data());
        for (const OpInfo& c : h->suffix) {
            if (c.d.kind != BMI_OP_CONV || c.d.in != id || c.d.in2 >= 0) continue;       // (the fused-shortcut reader is conv3x3_patch: no grid rule on the operand)
            const int cout = c.cout + (c.has_pair ? c.pair_cout : 0);
            const long imgs = std::max(1, 256 / (c.ho * c.wo));
            const long tiles = ((long)NS + imgs - 1) / imgs * (cout / (cout % 256 ? 128 : 256));
            if (opt_conv_s2() != 2 && tiles < 3 * 256 / 4) t.lazy_planar_plan = false;
            if (opt_conv_s2() == 0) t.lazy_planar_plan = false;
        }
    }
    *workspace_bytes = off;
    return BMI_OK;
}

int bmi_tensor_info(bmi_handle h, int32_t id, int64_t* offset, int32_t* elem_bytes, int32_t* per_sample, int32_t* th, int32_t* tw, int32_t* tc) {
    if (!h || id < 1 || id >= (int32_t)h->tensors.size() || h->max_batch == 0) return BMI_ERR_INVALID;
    const TensorInfo& t = h->tensors[id];
    if (t.bits || (t.stoch && t.first < 0)) return BMI_ERR_UNSUPPORTED;   // keep bits / a tensor nothing reads have no activation layout
    if (offset) *offset = (int64_t)t.offset;
    if (elem_bytes) *elem_bytes = t.f32 ? 4 : 2;
    if (per_sample) *per_sample = (t.stoch ? 1 : 0) | ((h->split && !t.dense_out) ? 2 : 0);
    if (th) *th = t.h;
    if (tw) *tw = t.w;
    if (tc) *tc = t.c;
    return BMI_OK;
}

int bmi_query(bmi_handle h, int64_t* prefix_macs, int64_t* suffix_macs, int32_t* n_prefix_ops, int32_t* n_suffix_ops) {
    if (!h) return BMI_ERR_INVALID;
    if (prefix_macs) *prefix_macs = h->prefix_macs;
    if (suffix_macs) *suffix_macs = h->suffix_macs;
    if (n_prefix_ops) *n_prefix_ops = (int32_t)h->prefix.size();
    if (n_suffix_ops) *n_suffix_ops = (int32_t)h->suffix.size();
    return BMI_OK;
}

}  // extern "C"

namespace {

struct ProfScope {
    bmi_engine_s* e;
    hipStream_t s;
    ProfRec r;
    bool on;
    ProfScope(bmi_engine_s* e_, int slot, hipStream_t s_) : e(e_), s(s_), on(e_->profiling) {
        if (!on) return;
        auto get = [&]() {
            hipEvent_t ev;
            if (!e->pool.empty()) { ev = e->pool.back(); e->pool.pop_back(); }
            else (void)hipEventCreate(&ev);
            return ev;
        };
        r.slot = slot; r.a = get(); r.b = get();
        (void)hipEventRecord(r.a, s);
    }
    void tag(int family, double flops, double bytes) { r.family = family; r.flops = flops; r.bytes = bytes; }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, s);
        e->recs.push_back(r);
    }
};

// The arguments of one exit head's launch (head_fused.hip).
HeadArgs make_head_args(bmi_engine_s* e, const OpInfo& op, char* ws, int N, int B, int t0, uint64_t seed, int cnt0, double* S1, double* S2, double* SL,
                        const int* imap, int Bc) {
    const bmi_op_desc& d = op.d;
    const TensorInfo& tin = e->tensors[d.in];
    const int b0 = e->image_offset;
    const int n_rows = imap ? (N / Bc) * B : N;
    HeadArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = ws + tin.offset;
    a.in_kind = (e->split && !tin.dense_out) ? 2 + e->split        // the split engines' pair32 tensors
                : (tin.f32 || tin.pooled_now) ? 1 : (e->bf16 ? 2 : 0);      // pooled_now: fp32 means [row][K] written by the conv
    a.in_mod = tin.stoch ? n_rows : B;
    a.imap = imap; a.Bc = Bc;
    a.HW = tin.pooled_now ? 1 : tin.h * tin.w; a.K = tin.c; a.B = B; a.t0 = t0; a.tc = imap ? N / Bc : N / B;
    a.w = (const float*)d.weight; a.bias = d.bias; a.C = e->out_dim;
    const bool on_logits = d.site_pos == BMI_SITE_POS_INNER;
    const uint64_t soff = (uint64_t)b0 * (uint64_t)tin.c;          // (site_off of run_op: [B, K] features, elementwise and per-channel draws alike)
    SiteArgs sf = resolve_site(on_logits ? nullptr : &d.site, seed, cnt0, soff);
    SiteArgs sl = resolve_site(on_logits ? &d.site : nullptr, seed, cnt0);
    for (SiteArgs* sa : {&sf, &sl})                                 // Masksembles tables walked with a stride (bmi_forward_mcd_samples): see run_op
        if (sa->kind == BMI_SITE_MASKSEMBLE && e->mask_permuted)
            for (const auto& pr : e->perm)
                if (pr.first == sa->masks) {
                    sa->masks = (const float*)(ws + pr.second);
                    sa->cnt0 = (sa->num_masks - e->mask_t_begin % sa->num_masks) % sa->num_masks;
                    break;
                }
    a.site = sf;
    a.site_logits = sl;
    a.b0 = b0;
    const size_t eo = (size_t)d.out * B * e->out_dim;
    a.S1 = S1 + eo; a.S2 = S2 + eo; a.SL = SL + eo;
    if (e->no_moments) a.S1 = a.S2 = a.SL = nullptr;
    a.part = (double*)(ws + e->head_off + (size_t)d.out * e->head_part_bytes);     // this exit's own region
    if (e->logits_out) {            // per-sample logits of this exit: [t - t_begin][E][batch][C]
        const size_t plane = (size_t)e->logits_batch * e->out_dim;
        a.logits = e->logits_out + ((size_t)(t0 - e->logits_t_begin) * e->n_exits + d.out) * plane;
        a.logits_tstride = (size_t)e->n_exits * plane;
    }
    return a;
}

// imap / rows / Bc: dynamic early exit: N = samples * Bc compact images of the B-image batch; imap = the Bc active images
// (heads), rows = the N-entry row table (ConvArgs::imap); null = all images
int run_op(bmi_engine_s* e, const OpInfo& op, const float* x, char* ws, int N, int B, int t0, uint64_t seed, int cnt0,
           double* S1, double* S2, double* SL, hipStream_t s, const int* imap = nullptr, int Bc = 0, const int* rows = nullptr) {
    const bmi_op_desc& d = op.d;
    const TensorInfo& tin = e->tensors[d.in];
    // image-partitioned launch (bmi_forward_mcd_images): element index of image 0 in the index space of a site on a tensor with
    // `per_image` elements per image (elementwise) / `channels` (per-(image, channel) draws)
    const int b0 = e->image_offset;
    auto site_off = [&](const bmi_site& st, size_t per_image, size_t channels) -> uint64_t {
        return (uint64_t)b0 * (st.kind == BMI_SITE_CHANNEL ? channels : per_image);
    };
    const int n_rows = imap ? (N / Bc) * B : N;     // rows of a stochastic tensor (original folded layout)
    // Masksembles masks walked with a stride (bmi_forward_mcd_samples): the kernels index row (cnt0 + t) % M of a site's table; the
    // call has gathered table'[r] = table[(mask_cnt0 + r stride) % M] into the workspace, and cnt0' = -t_begin mod M makes that row
    // (t - t_begin) % M — no kernel knows about the stride
    auto resolve_site = [&](const bmi_site* site, uint64_t sd, int c0, uint64_t elem_off = 0) {
        SiteArgs sa = ::resolve_site(site, sd, c0, elem_off);
        if (sa.kind == BMI_SITE_MASKSEMBLE && e->mask_permuted)
            for (const auto& pr : e->perm)
                if (pr.first == sa.masks) {
                    sa.masks = (const float*)(ws + pr.second);
                    sa.cnt0 = (sa.num_masks - e->mask_t_begin % sa.num_masks) % sa.num_masks;
                    break;
                }
        return sa;
    };
    if (imap && d.kind != BMI_OP_CONV && d.kind != BMI_OP_HEAD) return BMI_ERR_UNSUPPORTED;
    if (imap && e->f32 && !e->split) return BMI_ERR_UNSUPPORTED;      // (the exact engine: parity only)
    // a lazy site's tensor (see bmi_create) is written now if this op cannot apply the mask itself
    auto pending = [&](int id) { return id >= 0 && e->tensors[id].lazy_pending; };
    auto materialise = [&](int id) -> int {
        if (!pending(id)) return BMI_OK;
        e->tensors[id].lazy_pending = false;
        return launch_mask_apply(e->tensors[id].lazy_call, s);
    };
    // (a pending RESIDUAL stays pending only for conv3x3_patch's 64-channel tile, which masks it where it is added: tried first in the CONV case below)
    const bool res_lazy_ok = d.kind == BMI_OP_CONV && pending(d.residual) && !op.has_pair && d.in2 < 0 && d.ksize == 3 && d.stride == 1 && d.pad == 1 &&
                             op.cout % 64 == 0 && op.cout % 128 != 0 && op.wo == 32 && op.ho % 8 == 0 && opt_conv_patch64() && !imap;
    if (d.kind != BMI_OP_CONV || (pending(d.residual) && !res_lazy_ok) || (pending(d.in) && pending(d.in2))) {
        int rcm = materialise(d.in);
        if (rcm == BMI_OK && d.kind == BMI_OP_CONV) { rcm = materialise(d.residual); if (rcm == BMI_OK) rcm = materialise(d.in2); }
        if (rcm != BMI_OK) return rcm;
    }
    ProfScope prof(e, d.kind == OP_MASKBITS ? BMI_OP_MASK : d.kind, s);
    prof.r.out = d.out; prof.r.images = N;
    switch (d.kind) {
        case BMI_OP_STEM:
            return launch_stem_conv(x, (const float*)d.weight, d.scale, d.bias, (_Float16*)(ws + e->tensors[d.out].offset), N,
                                    tin.c, tin.h, tin.w, op.cout, d.ksize, d.stride, d.pad, d.relu, e->dtype, s);     // (F32: fp32 out; F16X2 / BF16X3: pair32 out)
        case BMI_OP_CONV: {
            ConvArgs a;
            std::memset(&a, 0, sizeof(a));
            a.bf16 = e->bf16;
            a.in = (const _Float16*)(ws + tin.offset);
            a.wgt = (const _Float16*)d.weight;
            a.scale = d.scale; a.bias = d.bias;
            a.out = (_Float16*)(ws + e->tensors[d.out].offset);
            a.N = N;
            // kernel selection looks at the PLANNED batch (x the planned chunk for suffix launches), never at this call's batch or sample
            // count: a t-shard, a partial chunk, an image share (bmi_forward_mcd_images) and a loader's smaller last batch all run the
            // kernels — and get the bits — of the full batch
            a.n_ref = (tin.stoch || (d.residual >= 0 && e->tensors[d.residual].stoch) || op.stoch) ? e->max_batch * e->chunk : e->max_batch;
            a.imap = rows; a.Bc = Bc;
            a.in_mod = tin.stoch ? n_rows : B;
            if (d.residual >= 0) {
                a.res = (const _Float16*)(ws + e->tensors[d.residual].offset);
                a.res_mod = e->tensors[d.residual].stoch ? n_rows : B;
            }
            a.H = tin.h; a.W = tin.w; a.Cin = tin.c;
            a.Ho = op.ho; a.Wo = op.wo; a.Cout = op.cout;
            a.ksize = d.ksize; a.stride = d.stride; a.pad = d.pad; a.relu = d.relu;
            a.M = N * op.ho * op.wo;
            a.B = B; a.t0 = t0;
            a.site = resolve_site(&d.site, seed, cnt0, site_off(d.site, (size_t)op.ho * op.wo * op.cout, (size_t)op.cout));
            if (d.site_pos == BMI_SITE_POS_INNER && d.site.kind != BMI_SITE_NONE) { a.site_inner = 1; a.bias_post = d.bias_post; }
            a.out_mul = op.out_mul;
            if (d.in2 >= 0) {
                const TensorInfo& t2 = e->tensors[d.in2];
                a.in2 = (const _Float16*)(ws + t2.offset);
                a.wgt2 = (const _Float16*)d.weight2;
                a.in2_mod = t2.stoch ? n_rows : B;
                a.H2 = t2.h; a.W2 = t2.w; a.Cin2 = t2.c; a.stride2 = t2.h / op.ho;
            }
            if (op.bits_tensor >= 0) a.in_bits = (const uint8_t*)(ws + e->tensors[op.bits_tensor].offset);
            auto lazy_in = [&](ConvArgs& m) {   // the input as (scaled deterministic tensor, keep bits)
                m.in = (const _Float16*)(ws + e->tensors[tin.lazy_scaled].offset);
                m.in_mod = B;
                m.in_bits = (const uint8_t*)(ws + e->tensors[tin.lazy_bits].offset);
                m.lazy_planar = tin.lazy_planar_now;
            };
            if (e->f32) {   // the exact / split engines: one generic kernel each (a.in / a.res / a.out hold fp32)
                if (!e->split) { prof.tag(-1, 0, 0); return launch_conv_exact(a, s); }
                const int cout_l = op.cout + (op.has_pair ? op.pair_cout : 0);      // pair mode: two convs on this input in one launch (the 256-channel tile)
                const int cin2_l = d.in2 >= 0 ? e->tensors[d.in2].c : 0;            // fused 1x1 shortcut: extra K-steps
                const double fl = 2.0 * N * op.ho * op.wo * (double)cout_l * (d.ksize * d.ksize * tin.c + cin2_l);
                auto tb = [&](int id) { const TensorInfo& t = e->tensors[id]; return 4.0 * (t.stoch ? N : B) * t.h * t.w * t.c; };
                prof.tag(BMI_CONV_FAMILY_SPLIT, fl, tb(d.in) + 4.0 * N * op.ho * op.wo * (double)cout_l + 4.0 * (double)cout_l * (d.ksize * d.ksize * tin.c + cin2_l) +
                                                    (d.residual >= 0 ? tb(d.residual) : 0.0) + (d.in2 >= 0 ? tb(d.in2) : 0.0));
                if (op.nsplit > 1 && !op.stoch) {       // split-K (bmi_plan): raw fp32 partial sums per K range, then the finishing pass
                    a.partial = (float*)(ws + e->splitk_off);
                    a.nsplit = op.nsplit;
                }
                if (op.has_pair) {
                    ConvArgs p = a;
                    p.wgt_b = (const _Float16*)op.pair_d.weight;
                    p.scale_b = op.pair_d.scale; p.bias_b = op.pair_d.bias;
                    p.out_b = (_Float16*)(ws + e->tensors[op.pair_d.out].offset);
                    p.split = op.cout;
                    p.Cout = cout_l;
                    const int rcp = launch_conv_split(p, e->split == 2, s);
                    if (rcp != BMI_ERR_UNSUPPORTED) return rcp;
                    ConvArgs q = a;          // not taken: two launches
                    q.wgt = p.wgt_b; q.scale = p.scale_b; q.bias = p.bias_b; q.out = p.out_b; q.Cout = op.pair_cout;
                    const int rc1 = launch_conv_split(a, e->split == 2, s);
                    return rc1 != BMI_OK ? rc1 : launch_conv_split(q, e->split == 2, s);
                }
                return launch_conv_split(a, e->split == 2, s);
            }
            double flops = 2.0 * N * op.ho * op.wo * (double)(op.cout + (op.has_pair ? op.pair_cout : 0)) * d.ksize * d.ksize * tin.c;
            if (d.in2 >= 0) flops += 2.0 * N * op.ho * op.wo * (double)op.cout * e->tensors[d.in2].c;
            // algorithmic bytes: every operand tensor once (a deterministic operand counts its B images), weights once; an operand read
            // through a lazy site (the B scaled images + the folded batch's keep bits) counts what such a launch actually has to move
            auto tbytes = [&](int id) { const TensorInfo& t = e->tensors[id]; return 2.0 * (t.stoch ? N : B) * t.h * t.w * t.c; };
            auto lazy_saving = [&](int id) { const TensorInfo& t = e->tensors[id]; return tbytes(id) - (2.0 * B + N / 8.0) * t.h * t.w * t.c; };
            double bytes = tbytes(d.in) + 2.0 * N * op.ho * op.wo * (double)(op.cout + (op.has_pair ? op.pair_cout : 0)) +
                           2.0 * (double)(op.cout + (op.has_pair ? op.pair_cout : 0)) * d.ksize * d.ksize * tin.c;
            if (d.residual >= 0) bytes += tbytes(d.residual);
            if (d.in2 >= 0) bytes += tbytes(d.in2) + 2.0 * op.cout * e->tensors[d.in2].c;
            if (op.has_seam) {
                // the next Bottleneck's reduce conv on this conv's output: one conv1x1_seam launch, or the two launches in order
                ConvArgs b;
                std::memset(&b, 0, sizeof(b));
                b.bf16 = e->bf16;
                b.in = a.out;
                b.wgt = (const _Float16*)op.seam_d.weight;
                b.scale = op.seam_d.scale; b.bias = op.seam_d.bias;
                b.out = (_Float16*)(ws + e->tensors[op.seam_d.out].offset);
                b.N = N; b.n_ref = a.n_ref; b.imap = rows; b.Bc = Bc; b.in_mod = n_rows;
                b.H = op.ho; b.W = op.wo; b.Cin = op.cout; b.Ho = op.ho; b.Wo = op.wo; b.Cout = op.seam_cout;
                b.ksize = 1; b.stride = 1; b.pad = 0; b.relu = op.seam_d.relu;
                b.M = a.M; b.B = B; b.t0 = t0;
                b.site = resolve_site(nullptr, seed, cnt0, 0);
                b.out_mul = 1.f;
                const double flops_b = 2.0 * N * op.ho * op.wo * (double)op.seam_cout * op.cout;
                const double bytes_b = 2.0 * N * op.ho * op.wo * (double)op.seam_cout + 2.0 * (double)op.seam_cout * op.cout;   // (its input never leaves the chip)
                const int rcs = launch_conv1x1_seam(a, b, s);
                if (rcs != BMI_ERR_UNSUPPORTED) {
                    prof.tag(BMI_CONV_FAMILY_SEAM, flops + flops_b, bytes + bytes_b);
                    return rcs;
                }
                int fam = -1;
                const int rc1 = launch_conv(a, s, &fam);
                prof.tag(fam, flops + flops_b, bytes + bytes_b + 2.0 * N * op.ho * op.wo * (double)op.cout);
                return rc1 != BMI_OK ? rc1 : launch_conv(b, s);
            }
            if (op.has_pair) {
                ConvArgs p = a;
                p.wgt_b = (const _Float16*)op.pair_d.weight;
                p.scale_b = op.pair_d.scale; p.bias_b = op.pair_d.bias;
                p.out_b = (_Float16*)(ws + e->tensors[op.pair_d.out].offset);
                p.split = op.cout;
                p.Cout = op.cout + op.pair_cout;
                e->tensors[d.out].pooled_now = e->tensors[op.pair_d.out].pooled_now = false;
                if (pending(d.in)) {         // lazy site on the input: conv3x3_s2 clears the dropped elements in LDS, or the tensor is written now
                    ConvArgs m = p;
                    lazy_in(m);
                    const int rcl = launch_conv3x3_s2(m, s);
                    prof.tag(BMI_CONV_FAMILY_S2, flops, bytes - lazy_saving(d.in));
                    if (rcl != BMI_ERR_UNSUPPORTED) return rcl;
                    const int rcm = materialise(d.in);
                    if (rcm != BMI_OK) return rcm;
                }
                if (opt_conv_pool() && (op.pool_ok || op.pair_pool_ok)) {
                    ConvArgs q = p;          // the pooled means take the place of the map in the workspace (16 x 4 B <= 16 x 16 x 2 B per channel)
                    if (op.pool_ok) q.pool = (float*)q.out;
                    if (op.pair_pool_ok) q.pool_b = (float*)q.out_b;
                    const int rcp = launch_conv3x3_s2(q, s);
                    prof.tag(BMI_CONV_FAMILY_S2, flops, bytes);
                    if (rcp != BMI_ERR_UNSUPPORTED) {
                        e->tensors[d.out].pooled_now = op.pool_ok;
                        e->tensors[op.pair_d.out].pooled_now = op.pair_pool_ok;
                        return rcp;
                    }
                }
                int rc = launch_conv3x3_s2(p, s);
                prof.tag(BMI_CONV_FAMILY_S2, flops, bytes);
                if (rc != BMI_ERR_UNSUPPORTED) return rc;
                rc = launch_conv_igemm_wide(p, s);
                prof.tag(BMI_CONV_FAMILY_WIDE, flops, bytes);
                if (rc != BMI_ERR_UNSUPPORTED) return rc;
                ConvArgs q = a;          // not taken after all: two plain launches
                q.wgt = p.wgt_b; q.scale = p.scale_b; q.bias = p.bias_b; q.out = p.out_b; q.Cout = op.pair_cout;
                int fam = -1;
                const int rc2 = launch_conv(a, s, &fam);
                prof.tag(fam, flops, bytes);
                return rc2 != BMI_OK ? rc2 : launch_conv(q, s);
            }
            if (pending(d.residual)) {       // (res_lazy_ok) the residual through its keep bits: conv3x3_patch's 64-channel tile — with the input too when it is the
                ConvArgs m = a;              //  same pending tensor or another one
                const TensorInfo& tr = e->tensors[d.residual];
                m.res = (const _Float16*)(ws + e->tensors[tr.lazy_scaled].offset);
                m.res_mod = B;
                m.res_bits = (const uint8_t*)(ws + e->tensors[tr.lazy_bits].offset);
                m.lazy_planar = tr.lazy_planar_now;
                if (pending(d.in)) { lazy_in(m); m.lazy_planar = m.lazy_planar || tr.lazy_planar_now; }
                const int rcl = launch_conv3x3_patch(m, s);
                prof.tag(BMI_CONV_FAMILY_PATCH, flops, bytes - lazy_saving(d.residual) - (pending(d.in) ? lazy_saving(d.in) : 0.0));
                if (rcl != BMI_ERR_UNSUPPORTED) return rcl;
                const int rcm = materialise(d.residual);
                if (rcm != BMI_OK) return rcm;
            }
            if (pending(d.in)) {             // whichever kernel of the chain applies keep bits: conv1x1_stream, conv3x3_s2, conv_igemm
                ConvArgs m = a;
                lazy_in(m);
                int faml = -1;
                const int rcl = launch_conv(m, s, &faml);
                prof.tag(faml, flops, bytes - lazy_saving(d.in));
                if (rcl != BMI_ERR_UNSUPPORTED) return rcl;
                const int rcm = materialise(d.in);
                if (rcm != BMI_OK) return rcm;
            }
            if (pending(d.in2)) {            // ... on the input of a fused shortcut: conv3x3_patch on 16x16 maps
                int rcl = BMI_ERR_UNSUPPORTED;
                if (op.ho == 16 && op.wo == 16) {
                    const TensorInfo& t2 = e->tensors[d.in2];
                    ConvArgs m = a;
                    m.in2 = (const _Float16*)(ws + e->tensors[t2.lazy_scaled].offset);
                    m.in2_mod = B;
                    m.in2_bits = (const uint8_t*)(ws + e->tensors[t2.lazy_bits].offset);
                    m.lazy_planar = t2.lazy_planar_now;
                    rcl = launch_conv3x3_patch(m, s);
                    prof.tag(BMI_CONV_FAMILY_PATCH, flops, bytes - lazy_saving(d.in2));
                }
                if (rcl != BMI_ERR_UNSUPPORTED) return rcl;
                const int rcm = materialise(d.in2);
                if (rcm != BMI_OK) return rcm;
            }
            e->tensors[d.out].pooled_now = false;
            if (op.nsplit > 1 && !op.stoch && !rows) {
                a.partial = (float*)(ws + e->splitk_off);
                a.nsplit = op.nsplit;
                prof.tag(BMI_CONV_FAMILY_IGEMM, flops, bytes);
                return launch_conv_igemm(a, s);
            }
            e->tensors[d.out].pooled_now = false;
            if (opt_conv_pool() == 1 && op.pool_pw_ok) {      // ("conv_pool" = 2: conv3x3_s2's only)
                ConvArgs q = a;
                q.pool = (float*)q.out;
                const int rcp = launch_conv3x3_pw(q, s);
                if (rcp != BMI_ERR_UNSUPPORTED) {
                    prof.tag(BMI_CONV_FAMILY_PW, flops, bytes);
                    e->tensors[d.out].pooled_now = true;
                    return rcp;
                }
            }
            if (opt_conv_pool() && op.pool_ok) {
                ConvArgs q = a;
                q.pool = (float*)q.out;
                const int rcp = launch_conv3x3_s2(q, s);
                if (rcp != BMI_ERR_UNSUPPORTED) {
                    prof.tag(BMI_CONV_FAMILY_S2, flops, bytes);
                    e->tensors[d.out].pooled_now = true;
                    return rcp;
                }
            }
            int fam = -1;
            const int rcc = launch_conv(a, s, &fam);
            prof.tag(fam, flops, bytes);
            return rcc;
        }
        case OP_MASKBITS:
            return launch_mask_bits((uint8_t*)(ws + e->tensors[d.out].offset), N, tin.h * tin.w, tin.c,
                                    resolve_site(&d.site, seed, cnt0, site_off(d.site, (size_t)tin.h * tin.w * tin.c, (size_t)tin.c)), B, t0, s);
        case BMI_OP_MASK: {
            EltArgs a;
            std::memset(&a, 0, sizeof(a));
            a.bf16 = e->bf16;
            a.in = (const _Float16*)(ws + tin.offset);
            a.out = ws + e->tensors[d.out].offset;
            a.N = N; a.in_mod = tin.stoch ? N : B; a.HW = tin.h * tin.w; a.C = tin.c; a.B = B; a.t0 = t0;
            a.site = resolve_site(&d.site, seed, cnt0, site_off(d.site, (size_t)tin.h * tin.w * tin.c, (size_t)tin.c));
            if (d.site_pos == BMI_SITE_POS_INNER) { a.bias_post = d.bias_post; a.relu = d.relu; }
            a.pair = e->split;                                     // the split engines: pair32 tensors in and out
            if (e->f32) return launch_mask_apply_f32(a, s);
            TensorInfo& to = e->tensors[d.out];
            to.lazy_pending = false;
            if (to.lazy_bits >= 0 && opt_mask_lazy() && !tin.stoch && N % B == 0) {
                const bool planar = to.lazy_planar_plan && opt_lazy_planar();
                to.lazy_planar_now = planar;
                const int rcb = launch_mask_bits((uint8_t*)(ws + e->tensors[to.lazy_bits].offset), N, tin.h * tin.w, tin.c, a.site, B, t0, s, planar ? tin.w : 0);
                if (rcb == BMI_OK) {
                    const int rcs = launch_scale_copy(a.in, (_Float16*)(ws + e->tensors[to.lazy_scaled].offset), (long)B * tin.h * tin.w * tin.c,
                                                      a.site.scale, e->bf16, s, planar ? tin.h * tin.w : 0, planar ? tin.w : 0, planar ? tin.c : 0);
                    if (rcs != BMI_OK) return rcs;
                    to.lazy_call = a;
                    to.lazy_pending = true;
                    return BMI_OK;
                }
                if (rcb != BMI_ERR_UNSUPPORTED) return rcb;
            }
            return launch_mask_apply(a, s);
        }
        case BMI_OP_MAXPOOL:
            if (e->f32) return launch_maxpool2_f32((const float*)(ws + tin.offset), (float*)(ws + e->tensors[d.out].offset), N, tin.h, tin.w, tin.c, s, e->split);
            return launch_maxpool2((const _Float16*)(ws + tin.offset), (_Float16*)(ws + e->tensors[d.out].offset), N, tin.h,
                                   tin.w, tin.c, e->bf16, s);
        case BMI_OP_DENSE:
            // input: fp32 (a dense layer's output; any tensor of the exact engine), the engine's 16-bit type, or pair32 (kinds 3 | 4)
            return launch_dense_f32(ws + tin.offset, (e->split && !tin.dense_out) ? 2 + e->split : (tin.f32 ? 1 : (e->bf16 ? 2 : 0)), (const float*)d.weight, d.bias,
                                    (float*)(ws + e->tensors[d.out].offset), N, tin.stoch ? N : B, tin.c, op.cout, d.relu,
                                    resolve_site(&d.site, seed, cnt0, site_off(d.site, (size_t)op.cout, (size_t)op.cout)), B, t0, s);
        case BMI_OP_HEAD:
            // pool + site + Linear + softmax + the chunk's moment sums in one launch (head_fused.hip)
            return launch_head_fused(make_head_args(e, op, ws, N, B, t0, seed, cnt0, S1, S2, SL, imap, Bc), s);
    }
    return BMI_ERR_INVALID;
}

// One chunk of the suffix.  Consecutive exit heads run as ONE launch ("head_batch"; launch_head_fused_multi): with exit-only dropout — what
// every run of the paper uses, Software_Artifact/script_figs/journal_script.sh:10-63 — the suffix is nothing but the heads, each a launch of
// mostly fixed latency; the same arithmetic per head, the same bits.
int run_suffix(bmi_engine_s* e, const float* x, char* ws, int N, int B, int t0, uint64_t seed, int cnt0, double* S1, double* S2, double* SL,
               hipStream_t s) {
    const std::vector<OpInfo>& ops = e->suffix;
    for (size_t i = 0; i < ops.size();) {
        size_t j = i;
        if (opt_head_batch())
            while (j < ops.size() && j - i < BMI_HEAD_PACK_MAX && ops[j].d.kind == BMI_OP_HEAD && !e->tensors[ops[j].d.in].lazy_pending) ++j;
        if (j - i >= 2) {
            HeadArgs list[BMI_HEAD_PACK_MAX];
            for (size_t k = i; k < j; ++k) list[k - i] = make_head_args(e, ops[k], ws, N, B, t0, seed, cnt0, S1, S2, SL, nullptr, 0);
            int rc;
            {
                ProfScope prof(e, BMI_OP_HEAD, s);
                prof.r.out = ops[i].d.out; prof.r.images = N;
                rc = launch_head_fused_multi(list, (int)(j - i), s);
            }
            if (rc == BMI_OK) { i = j; continue; }
            if (rc != BMI_ERR_UNSUPPORTED) return rc;
            if (e->profiling && !e->recs.empty()) {          // not taken: drop the empty record, the heads follow one by one
                e->pool.push_back(e->recs.back().a); e->pool.push_back(e->recs.back().b);
                e->recs.pop_back();
            }
        }
        const int rc = run_op(e, ops[i], x, ws, N, B, t0, seed, cnt0, S1, S2, SL, s);
        if (rc != BMI_OK) return rc;
        ++i;
    }
    return BMI_OK;
}

}  // namespace

extern "C" {

int bmi_image_offset_ok(bmi_handle h, int32_t image_offset) {
    if (!h || image_offset < 0) return BMI_ERR_INVALID;
    if (image_offset == 0) return BMI_OK;
    // a site's index offset must be a whole number of Philox calls at every bit width (64 elements): per-image element counts
    // of every site tensor — conv / mask outputs, pooled features, dense outputs
    for (const std::vector<OpInfo>* ops : {&h->prefix, &h->suffix})
        for (const OpInfo& op : *ops) {
            const bmi_op_desc& d = op.d;
            if (d.site.kind != BMI_SITE_ELEMENTWISE && d.site.kind != BMI_SITE_CHANNEL) continue;
            if (d.kind == BMI_OP_HEAD && d.site_pos == BMI_SITE_POS_INNER) continue;      // logits site: the kernel adds b0 itself
            const TensorInfo& tin = h->tensors[d.in];
            const size_t unit = d.kind == BMI_OP_HEAD ? (size_t)tin.c
                                : d.kind == BMI_OP_DENSE || d.site.kind == BMI_SITE_CHANNEL ? (size_t)op.cout
                                : (size_t)op.ho * op.wo * op.cout;
            if (((size_t)image_offset * unit) % 64 != 0) return BMI_ERR_UNSUPPORTED;
        }
    return BMI_OK;
}

int bmi_forward_mcd_images(bmi_handle h, const float* x_nchw, int32_t batch, int32_t image_offset, int32_t t_begin,
                           int32_t t_count, uint64_t seed, int32_t mask_cnt0, double* S1, double* S2, double* SL,
                           void* workspace, size_t workspace_bytes, bmi_stream stream) {
    const int rco = bmi_image_offset_ok(h, image_offset);
    if (rco != BMI_OK) return rco;
    h->image_offset = image_offset;
    const int rc = bmi_forward_mcd(h, x_nchw, batch, t_begin, t_count, seed, mask_cnt0, S1, S2, SL, workspace, workspace_bytes, stream);
    h->image_offset = 0;
    return rc;
}

int bmi_forward_mcd(bmi_handle h, const float* x_nchw, int32_t batch, int32_t t_begin, int32_t t_count, uint64_t seed,
                    int32_t mask_cnt0, double* S1, double* S2, double* SL, void* workspace, size_t workspace_bytes,
                    bmi_stream stream) {
    if (!h || !x_nchw || !S1 || !S2 || !SL || !workspace) return BMI_ERR_INVALID;
    BmiOptionScope opt_scope(&h->opts);
    if (batch < 1 || t_count < 1 || t_begin < 0 || mask_cnt0 < 0) return BMI_ERR_INVALID;
    if (h->max_batch == 0 || batch > h->max_batch) return BMI_ERR_INVALID;
    if (workspace_bytes < h->ws_bytes) return BMI_ERR_NOMEM;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    for (const OpInfo& op : h->prefix) {
        const int rc = run_op(h, op, x_nchw, ws, batch, batch, 0, seed, mask_cnt0, S1, S2, SL, s);
        if (rc != BMI_OK) return rc;
    }
    for (int t0 = t_begin; t0 < t_begin + t_count; t0 += h->chunk) {
        const int tc = std::min(h->chunk, t_begin + t_count - t0);
        const int N = tc * batch;
        const int rc = run_suffix(h, x_nchw, ws, N, batch, t0, seed, mask_cnt0, S1, S2, SL, s);
        if (rc != BMI_OK) return rc;
    }
    return BMI_OK;
}

int bmi_forward_mcd_samples(bmi_handle h, const float* x_nchw, int32_t batch, int32_t t_begin, int32_t t_count, uint64_t seed,
                            int32_t mask_cnt0, int32_t mask_stride, float* logits, double* S1, double* S2, double* SL, void* workspace,
                            size_t workspace_bytes, bmi_stream stream) {
    if (!h || !logits || !workspace || mask_stride < 1 || t_begin < 0 || mask_cnt0 < 0) return BMI_ERR_INVALID;
    BmiOptionScope opt_scope(&h->opts);
    if ((S1 || S2 || SL) && !(S1 && S2 && SL)) return BMI_ERR_INVALID;
    if (h->max_batch == 0 || batch < 1 || batch > h->max_batch) return BMI_ERR_INVALID;
    if (workspace_bytes < h->ws_bytes) return BMI_ERR_NOMEM;
    int cnt0 = mask_cnt0;
    if ((mask_stride != 1 || t_begin != 0) && !h->perm.empty()) {
        // gather every Masksembles table in the order this call walks it: table'[r] = table[(mask_cnt0 + r * stride) % M]
        // (stride 1 from t_begin > 0 too: the kernels index (cnt0 + t) mod M with the GLOBAL sample index t, which would rotate the walk by t_begin —
        //  this call's contract is mask_cnt0 for its FIRST sample, whatever t_begin)
        for (const std::vector<OpInfo>* ops : {&h->prefix, &h->suffix})
            for (const OpInfo& op : *ops) {
                const bmi_site& st = op.d.site;
                if (st.kind != BMI_SITE_MASKSEMBLE) continue;
                for (auto& pr : h->perm)
                    if (pr.first == st.masks) {
                        const int width = op.d.kind == BMI_OP_HEAD ? h->tensors[op.d.in].c : op.cout;
                        const int rc = launch_mask_permute(st.masks, (float*)((char*)workspace + pr.second), st.num_masks, width, mask_cnt0, mask_stride,
                                                           (hipStream_t)stream);
                        if (rc != BMI_OK) return rc;
                        break;
                    }
            }
        h->mask_stride = mask_stride;
        h->mask_permuted = true;
        h->mask_t_begin = t_begin;
        cnt0 = 0;
    }
    h->logits_out = logits; h->logits_t_begin = t_begin; h->logits_batch = batch;
    static double dummy;          // (bmi_forward_mcd's argument check; the head never dereferences S* when S1 is the dummy: see run_op)
    const bool moments = S1 != nullptr;
    h->no_moments = !moments;
    const int rc = bmi_forward_mcd(h, x_nchw, batch, t_begin, t_count, seed, cnt0, moments ? S1 : &dummy, moments ? S2 : &dummy, moments ? SL : &dummy,
                                   workspace, workspace_bytes, stream);
    h->logits_out = nullptr; h->mask_stride = 1; h->mask_t_begin = 0; h->mask_permuted = false; h->no_moments = false;
    return rc;
}

int bmi_forward_mcd_exit(bmi_handle h, const float* x_nchw, int32_t batch, int32_t t_count, uint64_t seed, int32_t mask_cnt0,
                         double threshold, int32_t first_exit, double* S1, double* S2, double* SL, int32_t* exit_of_image,
                         int32_t* active_after, void* workspace, size_t workspace_bytes, bmi_stream stream) {
    if (!h || !x_nchw || !S1 || !S2 || !SL || !workspace || !exit_of_image || !active_after) return BMI_ERR_INVALID;
    BmiOptionScope opt_scope(&h->opts);
    if (batch < 1 || t_count < 1 || mask_cnt0 < 0 || first_exit < 0) return BMI_ERR_INVALID;
    if (h->max_batch == 0 || batch > h->max_batch) return BMI_ERR_INVALID;
    if (t_count > h->chunk || (h->f32 && !h->split)) return BMI_ERR_UNSUPPORTED;     // an exit's decision needs ALL samples of the stage in the workspace
    if (workspace_bytes < h->ws_bytes) return BMI_ERR_NOMEM;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    int* lists[2] = {(int*)(ws + h->exit_off), (int*)(ws + h->exit_off) + h->max_batch};
    int* count_dev = (int*)(ws + h->exit_off) + 2 * h->max_batch;
    int* rows_dev = count_dev + 64;
    const int last = h->n_exits - 1;
    int rc = launch_fill_int(exit_of_image, batch, last, s);
    if (rc != BMI_OK) return rc;
    for (int x = 0; x < h->n_exits; ++x) active_after[x] = 0;
    for (const OpInfo& op : h->prefix) {
        rc = run_op(h, op, x_nchw, ws, batch, batch, 0, seed, mask_cnt0, S1, S2, SL, s);
        if (rc != BMI_OK) return rc;
    }
    const int* imap = nullptr;     // null: every image is still active
    const int* rows = nullptr;
    int bc = batch, cur = 0;
    for (const OpInfo& op : h->suffix) {
        rc = run_op(h, op, x_nchw, ws, t_count * bc, batch, 0, seed, mask_cnt0, S1, S2, SL, s, imap, bc, rows);
        if (rc != BMI_OK) return rc;
        if (op.d.kind != BMI_OP_HEAD) continue;
        const int e = op.d.out;
        if (e < first_exit || e >= last) { active_after[e] = bc; continue; }
        // confidence test of exit e over the still-active images, on the device; the host only learns how many go on
        rc = launch_exit_decide(S1 + (size_t)e * batch * h->out_dim, h->out_dim, t_count, threshold, imap, bc, lists[cur], count_dev,
                                exit_of_image, e, s);
        if (rc != BMI_OK) return rc;
        int n_active = 0;
        if (hipMemcpyAsync(&n_active, count_dev, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            return BMI_ERR_HIP;
        active_after[e] = n_active;
        if (n_active == 0) return BMI_OK;                    // every image has left: the later stages do not run at all
        imap = lists[cur];
        bc = n_active;
        cur ^= 1;
        rc = launch_expand_rows(imap, bc, batch, t_count, rows_dev, s);     // compact image -> tensor row, for the conv kernels
        if (rc != BMI_OK) return rc;
        rows = rows_dev;
    }
    return BMI_OK;
}

int bmi_finalize(int64_t n, int32_t t_total, const double* S1, const double* S2, const double* SL, double* mean,
                 double* var, double* logit_mean, bmi_stream stream) {
    if (!S1 || !S2 || !SL || !mean || !var || !logit_mean) return BMI_ERR_INVALID;
    return launch_finalize(n, t_total, S1, S2, SL, mean, var, logit_mean, nullptr, (hipStream_t)stream);
}

int bmi_finalize_checked(int64_t n, int32_t t_total, const double* S1, const double* S2, const double* SL, double* mean,
                         double* var, double* logit_mean, int32_t* nonfinite, bmi_stream stream) {
    if (!S1 || !S2 || !SL || !mean || !var || !logit_mean || !nonfinite) return BMI_ERR_INVALID;
    return launch_finalize(n, t_total, S1, S2, SL, mean, var, logit_mean, nonfinite, (hipStream_t)stream);
}

int bmi_profile_enable(bmi_handle h, int32_t enable) {
    if (!h) return BMI_ERR_INVALID;
    h->profiling = enable != 0;
    return BMI_OK;
}

int bmi_profile_read(bmi_handle h, double ms[BMI_PROFILE_SLOTS], int64_t launches[BMI_PROFILE_SLOTS]) {
    if (!h || !ms || !launches) return BMI_ERR_INVALID;
    for (int i = 0; i < BMI_PROFILE_SLOTS; ++i) { ms[i] = 0; launches[i] = 0; }
    for (int i = 0; i < BMI_CONV_FAMILIES; ++i) { h->fam_ms[i] = 0; h->fam_flops[i] = 0; h->fam_bytes[i] = 0; h->fam_launches[i] = 0; }
    int rc = BMI_OK;
    h->last.clear();
    for (auto& r : h->recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) rc = BMI_ERR_HIP;
        r.ms = t;
        h->last.push_back(r);
        if (r.slot >= 0 && r.slot < BMI_PROFILE_SLOTS) { ms[r.slot] += t; launches[r.slot] += 1; }
        if (r.slot == BMI_OP_CONV && r.family >= 0 && r.family < BMI_CONV_FAMILIES) {
            h->fam_ms[r.family] += t; h->fam_flops[r.family] += r.flops; h->fam_bytes[r.family] += r.bytes; h->fam_launches[r.family] += 1;
        }
        h->pool.push_back(r.a);
        h->pool.push_back(r.b);
    }
    h->recs.clear();
    return rc;
}

int bmi_profile_conv_families(bmi_handle h, double ms[BMI_CONV_FAMILIES], int64_t launches[BMI_CONV_FAMILIES],
                              double flops[BMI_CONV_FAMILIES], double bytes[BMI_CONV_FAMILIES]) {
    if (!h || !ms || !launches || !flops || !bytes) return BMI_ERR_INVALID;
    for (int i = 0; i < BMI_CONV_FAMILIES; ++i) { ms[i] = h->fam_ms[i]; launches[i] = h->fam_launches[i]; flops[i] = h->fam_flops[i]; bytes[i] = h->fam_bytes[i]; }
    return BMI_OK;
}

int bmi_profile_launches(bmi_handle h, int32_t capacity, int32_t* count, int32_t* kind, int32_t* family, int32_t* out_tensor,
                         int32_t* images, double* ms, double* flops, double* bytes) {
    if (!h || !count || capacity < 0) return BMI_ERR_INVALID;
    *count = (int32_t)h->last.size();
    for (int i = 0; i < *count && i < capacity; ++i) {
        const ProfRec& r = h->last[i];
        if (kind) kind[i] = r.slot;
        if (family) family[i] = r.family;
        if (out_tensor) out_tensor[i] = r.out;
        if (images) images[i] = r.images;
        if (ms) ms[i] = r.ms;
        if (flops) flops[i] = r.flops;
        if (bytes) bytes[i] = r.bytes;
    }
    return BMI_OK;
}

// ---- single-kernel entry points -------------------------------------------------------------

int bmi_philox_mask(uint8_t* keep, int64_t n, uint64_t seed, int32_t site, int32_t t, float p, bmi_stream stream) {
    if (!keep) return BMI_ERR_INVALID;
    return launch_philox_mask(keep, n, seed, site, t, p, (hipStream_t)stream);
}

int bmi_stem_conv_fwd(const float* x_nchw, const float* weight, const float* scale, const float* bias, void* out_nhwc,
                      int32_t n, int32_t cin, int32_t h, int32_t w, int32_t cout, int32_t ksize, int32_t stride,
                      int32_t pad, int32_t relu, bmi_stream stream) {
    if (!x_nchw || !weight || !out_nhwc) return BMI_ERR_INVALID;
    return launch_stem_conv(x_nchw, weight, scale, bias, (_Float16*)out_nhwc, n, cin, h, w, cout, ksize, stride, pad, relu,
                            opt_unit_dtype(), (hipStream_t)stream);
}

int bmi_mask_bits(void* bits, int32_t n, int32_t hw, int32_t c, const bmi_site* site, int32_t batch, int32_t t0,
                  uint64_t seed, bmi_stream stream) {
    if (!bits || !site || !site_ok(*site)) return BMI_ERR_INVALID;
    return launch_mask_bits((uint8_t*)bits, n, hw, c, resolve_site(site, seed, 0), batch, t0, (hipStream_t)stream);
}

int bmi_conv_igemm_fwd(const void* in, const void* in_keep_bits, float out_mul, const void* weight,
                       const float* scale, const float* bias, const void* res, void* out,
                       int32_t n, int32_t in_mod, int32_t res_mod, int32_t h, int32_t w, int32_t cin, int32_t cout,
                       int32_t ksize, int32_t stride, int32_t pad, int32_t relu, const bmi_site* site,
                       int32_t batch, int32_t t0, uint64_t seed, int32_t mask_cnt0, bmi_stream stream) {
    if (!in || !weight || !out || ksize < 1 || stride < 1 || pad < 0) return BMI_ERR_INVALID;
    if (site && !site_ok(*site)) return BMI_ERR_INVALID;
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.bf16 = opt_unit_dtype() == BMI_DTYPE_BF16;
    a.in = (const _Float16*)in; a.wgt = (const _Float16*)weight; a.scale = scale; a.bias = bias;
    a.res = (const _Float16*)res; a.out = (_Float16*)out;
    a.N = n; a.in_mod = in_mod; a.res_mod = res_mod;
    a.H = h; a.W = w; a.Cin = cin; a.Cout = cout;
    a.Ho = (h + 2 * pad - ksize) / stride + 1;
    a.Wo = (w + 2 * pad - ksize) / stride + 1;
    a.ksize = ksize; a.stride = stride; a.pad = pad; a.relu = relu;
    a.M = n * a.Ho * a.Wo;
    a.B = batch; a.t0 = t0;
    a.site = resolve_site(site, seed, mask_cnt0);
    a.in_bits = (const uint8_t*)in_keep_bits;
    a.out_mul = out_mul;
    if (opt_unit_dtype() == BMI_DTYPE_F32) return launch_conv_exact(a, (hipStream_t)stream);
    if (unit_f32act()) return launch_conv_split(a, opt_unit_dtype() == BMI_DTYPE_BF16X3, (hipStream_t)stream);
    return launch_conv(a, (hipStream_t)stream);
}

int bmi_conv1x1_seam_fwd(const void* in, const void* weight3, const float* scale3, const float* bias3, const void* res, void* out_wide,
                         const void* weight1, const float* scale1, const float* bias1, void* out_narrow, int32_t n, int32_t h, int32_t w,
                         int32_t cmid, int32_t cw, int32_t cn, int32_t relu1, bmi_stream stream) {
    if (!in || !weight3 || !res || !out_wide || !weight1 || !out_narrow || n <= 0) return BMI_ERR_INVALID;
    if (opt_unit_dtype() != BMI_DTYPE_F16 && opt_unit_dtype() != BMI_DTYPE_BF16) return BMI_ERR_UNSUPPORTED;
    ConvArgs a, b;
    std::memset(&a, 0, sizeof(a));
    a.bf16 = opt_unit_dtype() == BMI_DTYPE_BF16;
    a.in = (const _Float16*)in; a.wgt = (const _Float16*)weight3; a.scale = scale3; a.bias = bias3;
    a.res = (const _Float16*)res; a.out = (_Float16*)out_wide;
    a.N = n; a.in_mod = n; a.res_mod = n;
    a.H = a.Ho = h; a.W = a.Wo = w; a.Cin = cmid; a.Cout = cw;
    a.ksize = 1; a.stride = 1; a.pad = 0; a.relu = 1;
    a.M = n * h * w; a.B = n; a.out_mul = 1.f;
    a.site = resolve_site(nullptr, 0, 0);
    b = a;
    b.in = a.out; b.wgt = (const _Float16*)weight1; b.scale = scale1; b.bias = bias1; b.res = nullptr; b.res_mod = 0;
    b.out = (_Float16*)out_narrow; b.Cin = cw; b.Cout = cn; b.relu = relu1;
    const int rc = launch_conv1x1_seam(a, b, (hipStream_t)stream);
    if (rc != BMI_ERR_UNSUPPORTED) return rc;
    const int rc1 = launch_conv(a, (hipStream_t)stream);      // the engine's fallback: the two launches
    return rc1 != BMI_OK ? rc1 : launch_conv(b, (hipStream_t)stream);
}

int bmi_conv_pair_fwd(const void* in, const void* weight_a, const float* scale_a, const float* bias_a, void* out_a,
                      const void* weight_b, const float* scale_b, const float* bias_b, void* out_b, int32_t n,
                      int32_t in_mod, int32_t h, int32_t w, int32_t cin, int32_t cout_a, int32_t cout_b, int32_t ksize,
                      int32_t stride, int32_t pad, int32_t relu, bmi_stream stream) {
    if (!in || !weight_a || !weight_b || !out_a || !out_b || ksize < 1 || stride < 1 || pad < 0) return BMI_ERR_INVALID;
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.bf16 = opt_unit_dtype() == BMI_DTYPE_BF16;
    a.in = (const _Float16*)in; a.wgt = (const _Float16*)weight_a; a.scale = scale_a; a.bias = bias_a; a.out = (_Float16*)out_a;
    a.wgt_b = (const _Float16*)weight_b; a.scale_b = scale_b; a.bias_b = bias_b; a.out_b = (_Float16*)out_b;
    a.split = cout_a;
    a.N = n; a.in_mod = in_mod; a.H = h; a.W = w; a.Cin = cin; a.Cout = cout_a + cout_b;
    a.Ho = (h + 2 * pad - ksize) / stride + 1;
    a.Wo = (w + 2 * pad - ksize) / stride + 1;
    a.ksize = ksize; a.stride = stride; a.pad = pad; a.relu = relu;
    a.M = n * a.Ho * a.Wo;
    a.B = n; a.out_mul = 1.f;
    a.site = resolve_site(nullptr, 0, 0);
    if (opt_unit_dtype() == BMI_DTYPE_F32) return BMI_ERR_UNSUPPORTED;
    if (unit_f32act()) return launch_conv_split(a, opt_unit_dtype() == BMI_DTYPE_BF16X3, (hipStream_t)stream);     // (pair32 tensors, head / tail weight planes)
    const int rc = launch_conv3x3_s2(a, (hipStream_t)stream);      // the engine's order: conv3x3_s2 where it applies, else conv_igemm_wide
    return rc != BMI_ERR_UNSUPPORTED ? rc : launch_conv_igemm_wide(a, (hipStream_t)stream);
}

int bmi_conv3x3_shortcut_fwd(const void* in, const void* weight, const void* in2, const void* weight2, const float* bias,
                             void* out, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t cin2,
                             int32_t relu, bmi_stream stream) {
    if (!in || !weight || !in2 || !weight2 || !out) return BMI_ERR_INVALID;
    ConvArgs a;
    std::memset(&a, 0, sizeof(a));
    a.bf16 = opt_unit_dtype() == BMI_DTYPE_BF16;
    a.in = (const _Float16*)in; a.wgt = (const _Float16*)weight; a.bias = bias; a.out = (_Float16*)out;
    a.N = n; a.in_mod = n; a.H = h; a.W = w; a.Cin = cin; a.Cout = cout; a.Ho = h; a.Wo = w;
    a.ksize = 3; a.stride = 1; a.pad = 1; a.relu = relu; a.M = n * h * w; a.B = n; a.out_mul = 1.f;
    a.in2 = (const _Float16*)in2; a.wgt2 = (const _Float16*)weight2; a.in2_mod = n; a.H2 = 2 * h; a.W2 = 2 * w;
    a.Cin2 = cin2; a.stride2 = 2;
    a.site = resolve_site(nullptr, 0, 0);
    if (opt_unit_dtype() == BMI_DTYPE_F32) return BMI_ERR_UNSUPPORTED;
    if (unit_f32act()) return launch_conv_split(a, opt_unit_dtype() == BMI_DTYPE_BF16X3, (hipStream_t)stream);     // (pair32 tensors, head / tail weight planes)
    return launch_conv(a, (hipStream_t)stream);     // conv3x3_pw where it applies, else conv3x3_patch
}

static int elt_args(EltArgs& a, const void* in, void* out, int n, int in_mod, int hw, int c, const bmi_site* site, int batch,
                    int t0, uint64_t seed, int cnt0) {
    if (!in || !out) return BMI_ERR_INVALID;
    if (site && !site_ok(*site)) return BMI_ERR_INVALID;
    std::memset(&a, 0, sizeof(a));
    a.bf16 = opt_unit_dtype() == BMI_DTYPE_BF16;
    a.in = (const _Float16*)in; a.out = out; a.N = n; a.in_mod = in_mod; a.HW = hw; a.C = c; a.B = batch; a.t0 = t0;
    a.site = resolve_site(site, seed, cnt0);
    return BMI_OK;
}

int bmi_mask_apply(const void* in, void* out, int32_t n, int32_t in_mod, int32_t hw, int32_t c, const bmi_site* site,
                   int32_t batch, int32_t t0, uint64_t seed, int32_t mask_cnt0, bmi_stream stream) {
    EltArgs a;
    const int rc = elt_args(a, in, out, n, in_mod, hw, c, site, batch, t0, seed, mask_cnt0);
    if (rc == BMI_OK && unit_f32act()) { a.pair = unit_pair(); return launch_mask_apply_f32(a, (hipStream_t)stream); }
    return rc != BMI_OK ? rc : launch_mask_apply(a, (hipStream_t)stream);
}

int bmi_maxpool2(const void* in, void* out, int32_t n, int32_t h, int32_t w, int32_t c, bmi_stream stream) {
    if (!in || !out) return BMI_ERR_INVALID;
    if (unit_f32act()) return launch_maxpool2_f32((const float*)in, (float*)out, n, h, w, c, (hipStream_t)stream, unit_pair());
    return launch_maxpool2((const _Float16*)in, (_Float16*)out, n, h, w, c, opt_unit_dtype() == BMI_DTYPE_BF16, (hipStream_t)stream);
}

int bmi_dense_f32(const void* in, int32_t in_is_f32, const float* weight, const float* bias, float* out, int32_t n,
                  int32_t in_mod, int32_t k, int32_t cout, int32_t relu, const bmi_site* site, int32_t batch, int32_t t0,
                  uint64_t seed, int32_t mask_cnt0, bmi_stream stream) {
    if (!in || !weight || !bias || !out) return BMI_ERR_INVALID;
    if (site && !site_ok(*site)) return BMI_ERR_INVALID;
    return launch_dense_f32(in, in_is_f32 ? 1 : (unit_pair() ? 2 + unit_pair() : (opt_unit_dtype() == BMI_DTYPE_BF16 ? 2 : 0)), weight, bias, out, n, in_mod, k, cout, relu, resolve_site(site, seed, mask_cnt0), batch,
                            t0, (hipStream_t)stream);
}

int bmi_head_fused(const void* in, int32_t in_is_f32, int32_t in_mod, int32_t hw, int32_t k, const float* weight_pad,
                   const float* bias, int32_t out_dim, const bmi_site* site, const bmi_site* site_logits, int32_t batch, int32_t t0,
                   int32_t tc, uint64_t seed, int32_t mask_cnt0, double* S1, double* S2, double* SL, bmi_stream stream) {
    if ((site && !site_ok(*site)) || (site_logits && !site_ok(*site_logits))) return BMI_ERR_INVALID;
    HeadArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = in;
    a.in_kind = in_is_f32 ? 1 : (unit_pair() ? 2 + unit_pair() : (opt_unit_dtype() == BMI_DTYPE_BF16 ? 2 : 0));
    a.in_mod = in_mod; a.HW = hw; a.K = k; a.B = batch; a.t0 = t0; a.tc = tc;
    a.w = weight_pad; a.bias = bias; a.C = out_dim;
    a.site = resolve_site(site, seed, mask_cnt0);
    a.site_logits = resolve_site(site_logits, seed, mask_cnt0);
    a.S1 = S1; a.S2 = S2; a.SL = SL;
    return launch_head_fused(a, (hipStream_t)stream);
}

}  // extern "C"
