// 3x3 / pad-1 convolution, input patch in LDS, WEIGHTS STRAIGHT INTO REGISTERS (gfx950).
//
// Successor of conv3x3_patch.hip.  Per-phase stamps of that kernel (tools/stamps.py) showed the
// weight-tile DMA itself is cheap (~70 cycles of wait per K-step) but sharing the tile between the
// four waves costs a workgroup barrier per K-step whose skew is ~700 cycles, as much as 2/3 of the
// K-step's own MFMA time.  Here no weight tile is shared: the host packs the weights once in MFMA
// A-fragment order ([cout tile][chunk][tap][wave row][fragment][lane][8 halfs], see
// pack_conv3x3_weights_kernel), so each wave fetches the 8 KB it needs for a K-step with eight fully
// coalesced 1 KB global_load_dwordx4, one K-step ahead, into one of two register sets.  The only
// workgroup barriers left are the two around each 64-channel patch reload (2-8 per tile instead
// of 18-72), the LDS holds only the patch, and the A operand no longer crosses the LDS at all.
//
//   work-group  = IMGS images x (TH x TW) output pixels (64*TJ pixels) x 128 output channels, 256 threads
//   wave tile   = 64 channels x 32*TJ pixels of v_mfma_f32_32x32x16_f16 (channels on the row axis)
//   patch       = LDS-DMA from global (zero page for padding/halo), swizzled by patch coordinates
//   epilogue    = coalesced through LDS (conv_epilogue.h)
// Reference semantics: BasicBlock.forward SA/models/resnet18/resnet18.py:32-48.
#include "conv_epilogue.h"
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef BMI_PATCH_STAMPS
extern "C" __device__ unsigned long long g_wstamps[8192 * 8];
__device__ unsigned long long g_wstamps[8192 * 8];
#define WSTAMP(SLOT) if (tid == 0 && blockIdx.x < 8192) g_wstamps[blockIdx.x * 8 + (SLOT)] = __builtin_readcyclecounter();
#define WSTAMP_ADD(SLOT, T0) if (tid == 0 && blockIdx.x < 8192) g_wstamps[blockIdx.x * 8 + (SLOT)] += __builtin_readcyclecounter() - (T0);
extern "C" int bmi_debug_wstamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wstamps), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}
extern "C" int bmi_debug_wstamps_clear() {
    static unsigned long long z[8192 * 8];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wstamps), z, sizeof(z)) == hipSuccess ? 0 : -5;
}
#else
#define WSTAMP(SLOT)
#define WSTAMP_ADD(SLOT, T0)
#endif

static __device__ unsigned int g_zero_page[64];   // 256 B of zeros: DMA source for padding / halo / tail cells

#define GLDS16(SRC, LDSPTR)                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(SRC),       \
                                     (__attribute__((address_space(3))) void*)(LDSPTR), 16, 0, 0)

template <int S, int TH, int TW, int IMGS, int TJ>
struct WregGeom {
    static constexpr int BC = 128, TI = 2;
    static constexpr int BP = IMGS * TH * TW;
    static_assert(BP == 64 * TJ, "pixel tile must be 2 waves x TJ x 32");
    static constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
    static constexpr int HALF = S == 1 ? 0 : (((PW + 1) / 2 + 1) & ~1);
    static constexpr int PWP = S == 1 ? ((PW + 1) & ~1) : 2 * HALF;
    static constexpr int KA = TW >= 16 ? 0 : (TW == 8 ? 8 : 4);
    static constexpr int CELLS = IMGS * PH * PWP;
    static constexpr int ITER_P = (CELLS * 8 + 255) / 256;
    static constexpr int PATCH_BYTES = ITER_P * 256 * 16;
    static constexpr int LDS_BYTES = PATCH_BYTES > BMI_EPILOGUE_LDS_BYTES ? PATCH_BYTES : BMI_EPILOGUE_LDS_BYTES;
};

template <int S, int TH, int TW, int IMGS, int TJ>
__global__ __launch_bounds__(256, 2) void conv3x3_wreg_kernel(ConvArgs a) {
    using G = WregGeom<S, TH, TW, IMGS, TJ>;
    constexpr int BC = G::BC, PH = G::PH, PW = G::PW, PWP = G::PWP, HALF = G::HALF, KA = G::KA;
    __shared__ __attribute__((aligned(16))) char smem[G::LDS_BYTES];
    char* const patch = smem;

    const int tid = threadIdx.x;
    WSTAMP(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int wc = wave >> 1, wp = wave & 1;

    const int n_ctiles = a.Cout / BC;
    const int tiles_x = a.Wo / TW, tiles_y = a.Ho / TH;
    int bid, ctile;
    xcd_tile_map(blockIdx.x, (int)(gridDim.x / n_ctiles), n_ctiles, bid, ctile);
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int n0 = (bid / tiles_y) * IMGS;
    const int ch0 = ctile * BC;
    const int y0 = ty * TH, x0 = tx * TW;
    const int nchunks = a.Cin / 64;

    // ---- patch DMA: piece q = tid + 256*i -> LDS slot (cell = q >> 3, physical chunk = q & 7).
    // The source addresses are recomputed at every (rare) patch load instead of being kept in
    // registers across the K loop; `t_` is made opaque so the computation is not hoisted back out.
#define ISSUE_PATCH(C0)                                                                                     \
    {                                                                                                       \
        int t_ = tid;                                                                                       \
        asm volatile("" : "+v"(t_));                                                                        \
        _Pragma("unroll") for (int i = 0; i < G::ITER_P; ++i) {                                             \
            const int q = t_ + 256 * i;                                                                     \
            const int cell = q >> 3, cp = q & 7;                                                            \
            const int rowc = cell / PWP, col = cell - rowc * PWP;                                           \
            const int img = rowc / PH, py = rowc - img * PH;                                                \
            int px, key;                                                                                    \
            if (S == 1) { px = col; key = px + KA * py; }                                                   \
            else { const int par = col / HALF, hx = col - par * HALF; px = 2 * hx + par; key = hx + KA * py; } \
            const int c = cp ^ ((key >> 1) & 7);                                                            \
            const int n = n0 + img;                                                                         \
            const int iy = y0 * S - 1 + py, ix = x0 * S - 1 + px;                                           \
            const bool ok = cell < G::CELLS && px < PW && n < a.N && (unsigned)iy < (unsigned)a.H &&        \
                            (unsigned)ix < (unsigned)a.W;                                                   \
            const int nm = n < a.in_mod ? n : n % a.in_mod; /* division only for broadcast inputs */        \
            const _Float16* src = ok ? a.in + (((size_t)nm * a.H + iy) * a.W + ix) * a.Cin + c * 8 + (C0)   \
                                     : (const _Float16*)g_zero_page;                                        \
            GLDS16(src, patch + (i * 256 + wave * 64) * 16);                                                \
        }                                                                                                   \
    }

    // ---- packed weights: block (ctile, chunk, tap, wc) = 8 fragments x 1 KB; fragment f = i*4 + kk.
    // Wave-uniform base (SGPRs) + one per-lane byte offset, so the loads use the saddr form.
    const char* wtile = (const char*)a.wpk + ((size_t)ctile * nchunks * 9 * 2 + wc) * 8192;
    const unsigned wlane = lane * 16;
    half8 wa[8], wb[8];
#define LOAD_SET(SET, CHUNK, TAP)                                                                           \
    {                                                                                                       \
        const char* p_ = wtile + (size_t)((CHUNK) * 9 + (TAP)) * 16384;                                     \
        _Pragma("unroll") for (int f = 0; f < 8; ++f) SET[f] = *(const half8*)(p_ + f * 1024 + wlane);     \
    }

    // ---- per-lane patch geometry of the B fragments ------------------------------------------------------
    int bcell[TJ], bkey[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int p = wp * (32 * TJ) + 32 * j + r;
        const int img = p / (TH * TW), rem = p - img * (TH * TW);
        const int oy = rem / TW, ox = rem - oy * TW;
        bcell[j] = (img * PH + oy * S) * PWP + ox;
        bkey[j] = ox + KA * (oy * S);
    }

    f32x16 acc[2][TJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // one K-step = one tap x 64 channels = 4 k-substeps of 2*TJ MFMAs.  ALL fragments of a K-step
    // are register-resident before its first MFMA: the weight set (8 x half8 from global) and the
    // pixel set (4*TJ x half8 from the LDS patch) of step s+1 are requested during step s, so
    // neither the L2 nor the LDS latency sits between MFMAs.
    half8 ba[4 * TJ], bb[4 * TJ];
#define LOAD_B(SET, TAP)                                                                                    \
    {                                                                                                       \
        constexpr int ky_ = (TAP) / 3, kx_ = (TAP) % 3;                                                     \
        constexpr int coff_ = S == 1 ? ky_ * PWP + kx_ : ky_ * PWP + (kx_ & 1) * HALF + (kx_ >> 1);         \
        constexpr int koff_ = S == 1 ? kx_ + KA * ky_ : (kx_ >> 1) + KA * ky_;                              \
        _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                    \
            /* opaque to the optimiser: otherwise the 9 taps' addresses are hoisted out of the chunk */     \
            /* loop (loop-invariant) and cost 18*TJ live VGPRs */                                           \
            asm volatile("" : "+v"(bcell[j]), "+v"(bkey[j]));                                               \
            const int boff_ = (bcell[j] + coff_) * 128;                                                     \
            const int bsw_ = ((bkey[j] + koff_) >> 1) & 7;                                                  \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                \
                SET[kk * TJ + j] = *(const half8*)(patch + boff_ + (((2 * kk + hh) ^ bsw_) << 4));          \
        }                                                                                                   \
    }
#ifdef BMI_PATCH_STAMPS
#define WAIT_SET_STAMP()                                                                                    \
    {                                                                                                       \
        const unsigned long long t0_ = __builtin_readcyclecounter();                                        \
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                    \
        WSTAMP_ADD(4, t0_);                                                                                 \
    }
#else
#define WAIT_SET_STAMP()
#endif
#define COMPUTE(WSET, BSET)                                                                                 \
    {                                                                                                       \
        /* nothing moves across: the requests for step s+1 stay AHEAD of the MFMAs of step s (hipcc  */    \
        /* otherwise sinks the loads next to their first use, i.e. back into the step that needs them) */   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        WAIT_SET_STAMP();                                                                                   \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                    \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                   \
                _Pragma("unroll") for (int j = 0; j < TJ; ++j)                                              \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(WSET[i * 4 + kk], BSET[kk * TJ + j], acc[i][j], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }

    // One 64-channel chunk: 9 K-steps alternating between the two register sets; the set for the
    // NEXT chunk's first tap is fetched during tap 8, so consecutive chunks swap the roles of A and
    // B (no register copy): the chunk loop is unrolled by two with static names.
#ifdef BMI_PATCH_STAMPS
#define WSTAMP_T0() const unsigned long long tq0_ = __builtin_readcyclecounter()
#else
#define WSTAMP_T0()
#endif
// timing-only ablations (outputs are wrong): -DABL_NOW skips the in-loop weight loads, -DABL_NOB the
// in-loop patch-fragment reads
#ifdef ABL_NOW
#define LW_(SET, CHUNK, TAP)
#else
#define LW_(SET, CHUNK, TAP) LOAD_SET(SET, CHUNK, TAP)
#endif
#ifdef ABL_NOB
#define LB_(SET, TAP)
#else
#define LB_(SET, TAP) LOAD_B(SET, TAP)
#endif
#define CHUNK_BODY(WA, BA, WB, BB, CHUNK)                                                                   \
    {                                                                                                       \
        WSTAMP_T0();                                                                                        \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* this wave's patch pieces (and set A) landed */  \
        WSTAMP_ADD(5, tq0_);                                                                                \
        lds_barrier();                                   /* ... and everybody else's */                     \
        WSTAMP_ADD(6, tq0_);                                                                                \
        if ((CHUNK) == 0) { WSTAMP(1); }                                                                    \
        LOAD_B(BA, 0);                                                                                      \
        LOAD_SET(WB, CHUNK, 1); LOAD_B(BB, 1); COMPUTE(WA, BA);                                             \
        LW_(WA, CHUNK, 2); LB_(BA, 2); COMPUTE(WB, BB);                                                     \
        LW_(WB, CHUNK, 3); LB_(BB, 3); COMPUTE(WA, BA);                                                     \
        LW_(WA, CHUNK, 4); LB_(BA, 4); COMPUTE(WB, BB);                                                     \
        LW_(WB, CHUNK, 5); LB_(BB, 5); COMPUTE(WA, BA);                                                     \
        LW_(WA, CHUNK, 6); LB_(BA, 6); COMPUTE(WB, BB);                                                     \
        LW_(WB, CHUNK, 7); LB_(BB, 7); COMPUTE(WA, BA);                                                     \
        LW_(WA, CHUNK, 8); LB_(BA, 8); COMPUTE(WB, BB);                                                     \
        if ((CHUNK) + 1 < nchunks) {                                                                        \
            LOAD_SET(WB, (CHUNK) + 1, 0);                                                                   \
            COMPUTE(WA, BA);                                                                                \
            lds_barrier(); /* every wave is done reading this chunk's patch */                              \
            ISSUE_PATCH(((CHUNK) + 1) * 64);                                                                \
        } else {                                                                                            \
            COMPUTE(WA, BA);                                                                                \
        }                                                                                                   \
    }

    ISSUE_PATCH(0);
    LOAD_SET(wa, 0, 0);
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
        CHUNK_BODY(wa, ba, wb, bb, chunk);
        if (chunk + 1 < nchunks) CHUNK_BODY(wb, bb, wa, ba, chunk + 1);
    }
#undef CHUNK_BODY
#undef ISSUE_PATCH
#undef LOAD_SET
#undef COMPUTE
#undef LOAD_B

    WSTAMP(2);
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
    epilogue_coalesced<TJ, false>(a, acc, smem, tid, ch0, pixmap, offmap);
    WSTAMP(3);
}

// ---------------------------------------------------------------------------------------------------------
// [Cout][9][Cin] fp16  ->  fragment order.  One thread per 16-byte piece.
__global__ void pack_conv3x3_weights_kernel(const _Float16* __restrict__ w, _Float16* __restrict__ out, int Cout, int Cin) {
    const long total = (long)Cout * 9 * Cin / 8;
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int nchunks = Cin / 64;
    long q = id;
    const int lane = (int)(q % 64); q /= 64;
    const int f = (int)(q % 8); q /= 8;
    const int wc = (int)(q % 2); q /= 2;
    const int tap = (int)(q % 9); q /= 9;
    const int chunk = (int)(q % nchunks);
    const int ct = (int)(q / nchunks);
    const int i = f >> 2, kk = f & 3;
    const int row = ct * 128 + wc * 64 + i * 32 + (lane & 31);
    const int k = tap * Cin + chunk * 64 + kk * 16 + (lane >> 5) * 8;
    *(uint4*)(out + id * 8) = *(const uint4*)(w + (size_t)row * 9 * Cin + k);
}

int launch_pack_conv3x3_weights(const _Float16* w, _Float16* out, int cout, int cin, hipStream_t s) {
    if (cout % 128 != 0 || cin % 64 != 0) return BMI_ERR_UNSUPPORTED;
    const long total = (long)cout * 9 * cin / 8;
    hipLaunchKernelGGL(pack_conv3x3_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, out, cout, cin);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

template <int S, int TH, int TW, int IMGS, int TJ>
static int launch_wreg(const ConvArgs& a, hipStream_t s) {
    const long tiles = (long)((a.N + IMGS - 1) / IMGS) * (a.Ho / TH) * (a.Wo / TW) * (a.Cout / 128);
    if (tiles <= 0 || tiles > 0x7fffffffL) return BMI_ERR_INVALID;
    hipLaunchKernelGGL((conv3x3_wreg_kernel<S, TH, TW, IMGS, TJ>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    BMI_CHECK_LAUNCH();
    return BMI_OK;
}

// Returns BMI_ERR_UNSUPPORTED when the shape has no configuration or no packed weights were given.
int launch_conv3x3_wreg(const ConvArgs& a, hipStream_t s) {
    if (a.in_bits || a.in2) return BMI_ERR_UNSUPPORTED;
    if (!a.wpk || a.ksize != 3 || a.pad != 1 || a.Cin % 64 != 0 || a.Cout % 128 != 0) return BMI_ERR_UNSUPPORTED;
    if (a.N <= 0 || a.in_mod <= 0 || a.B <= 0 || (a.res && a.res_mod <= 0)) return BMI_ERR_INVALID;
    if ((size_t)a.in_mod * a.H * a.W * a.Cin >= 0x7fffffffull) return BMI_ERR_UNSUPPORTED;   // 31-bit DMA source offsets
    if (a.stride == 1) {
        if (a.Ho % 8 == 0 && a.Wo == 16) return launch_wreg<1, 8, 16, 1, 2>(a, s);   // TJ = 4 does not fit 256 VGPRs
        if (a.Ho == 8 && a.Wo == 8) return launch_wreg<1, 8, 8, 2, 2>(a, s);
        if (a.Ho == 4 && a.Wo == 4) return launch_wreg<1, 4, 4, 8, 2>(a, s);
        if (a.Ho % 4 == 0 && a.Wo == 32) return launch_wreg<1, 4, 32, 1, 2>(a, s);
    }
    return BMI_ERR_UNSUPPORTED;
}
