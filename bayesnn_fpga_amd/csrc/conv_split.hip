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
// One barrier per K-step.  The fp16-engine launch forms that exist for speed only (fused shortcut, pair, pooling, lazy sites,
// split-K, dynamic-exit row tables) are not built for this dtype (bmi_create keeps them out, as for the exact engine).
#include <type_traits>

#include "conv_epilogue.h"
#include "kernels.h"

typedef float f32x16_s __attribute__((ext_vector_type(16)));
typedef float f32x4_s __attribute__((ext_vector_type(4)));

#define SP_PT 256
// LDS-DMA, 16 B per lane to LDSPTR + 16 lane.  Inline asm, not __builtin_amdgcn_global_load_lds: to hipcc's waitcnt pass the builtin is a
// load AND a store ("mixed events": no in-order counting), and while one is in flight every register dependency on a plain global load
// becomes s_waitcnt vmcnt(0) — which would also wait for the input fetches issued for the K-step after next.  The DMA writes no register,
// so hiding it is safe; the K loop's own counted waits cover it (it is older than what they leave in flight).
#define SP_GLDS16(SRC, LDSPTR)                                                                                               \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"                                            \
                 :: "s"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(LDSPTR)), "v"(SRC) : "m0", "memory")

// v[0..7] -> heads and tails (zeros when !ok: an out-of-image tap / a tile row beyond M)
template <bool BF>
__device__ __forceinline__ void split8(const f32x4_s& x0, const f32x4_s& x1, bool ok, half8_t& hi, half8_t& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = ok ? (e < 4 ? x0[e] : x1[e - 4]) : 0.f;
        const _Float16 h = a16_from_f32<BF>(v);
        hi[e] = h;
        lo[e] = a16_from_f32<BF>(v - a16_to_f32<BF>(h));
    }
}

template <bool BF, int TI>
__global__ __launch_bounds__(512) void conv_split_kernel(ConvArgs a) {
    constexpr int CT = 64 * TI;
    constexpr int WPL = CT * 64;                       // bytes of one weight plane of a stage
    constexpr int XOFF = 2 * WPL;                      // X hi plane
    constexpr int XPL = SP_PT * 64;
    constexpr int STAGE = 2 * WPL + 2 * XPL;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

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

    // ---- weight DMA: piece q = tid + 512 i of the stage's [hi | lo] image: plane q / (4 CT), row (q >> 2) % CT, slot q & 3 ----
    const _Float16* wsrc[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int q = tid + 512 * i, plane = q / (4 * CT), qq = q - plane * 4 * CT, row = qq >> 2;
        wsrc[i] = a.wgt + ((size_t)plane * a.Cout + ch0 + row) * Ktot + (((qq & 3) ^ ((row >> 2) & 3)) << 3);
    }
#define SP_ISSUE_W(KOFF, ST)                                                                             \
    _Pragma("unroll") for (int i = 0; i < TI; ++i) SP_GLDS16(wsrc[i] + (KOFF), (ST) + (i * 512 + wave * 64) * 16);

    // ---- input staging: item f = tid + 512 i: tile row (pixel) (tid >> 2) + 128 i, channels 8 (tid & 3) .. of the K-step ----
    const float* xbase[2];
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
        xbase[i] = in + (size_t)(n % a.in_mod) * a.H * a.W * a.Cin + 8 * (tid & 3);
        xdst[i] = XOFF + row * 64 + (((tid & 3) ^ ((row >> 2) & 3)) << 4);
    }
    // X registers of one K-step: two items of 8 channels (two float4) + their bounds flags.  Two sets: while one is split and written
    // (it was loaded a whole K-step ago: no exposed wait), the other receives the loads of the K-step after next.
    struct XRegs { f32x4_s v[2][2]; bool ok[2]; };
    XRegs xa, xb;
    auto load_x = [&](XRegs& R, int ky, int kx, int c0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int iy_ = iy0[i] + ky, ix_ = ix0[i] + kx;
            R.ok[i] = (unsigned)iy_ < (unsigned)a.H && (unsigned)ix_ < (unsigned)a.W;
            // an out-of-image tap loads from the clamped position and is zeroed when it is split: no branch, no select on the address
            // (a branch in the K loop splits it into basic blocks, and hipcc then waits vmcnt(0) where a counted wait would do)
            const int iyc = min(max(iy_, 0), a.H - 1), ixc = min(max(ix_, 0), a.W - 1);
            const float* p_ = xbase[i] + ((iyc * a.W + ixc) * a.Cin + c0);
            R.v[i][0] = *(const f32x4_s*)p_;
            R.v[i][1] = *(const f32x4_s*)(p_ + 4);
        }
    };
    auto write_x = [&](const XRegs& R, char* st) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            half8_t hi_, lo_;
            split8<BF>(R.v[i][0], R.v[i][1], R.ok[i], hi_, lo_);
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
    const int b_off = XOFF + (wp * 64 + r) * 64;
    // one 16-deep sub-step SS of the K-step in stage ST: (TI + 2) x 2 fragment reads, TI x 2 x 3 MFMAs; the small products first
    // (lo . hi, hi . lo), then hi . hi
#define SP_SUBSTEP(ST, SS)                                                                               \
    {                                                                                                    \
        const int coff = ((2 * (SS) + kq) ^ sw) << 4;                                                    \
        half8_t ah[TI], al[TI], bh[2], bl[2];                                                            \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) {                                                 \
            ah[i] = *(const half8_t*)((ST) + a_off + i * 32 * 64 + coff);                                \
            al[i] = *(const half8_t*)((ST) + WPL + a_off + i * 32 * 64 + coff);                          \
        }                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                  \
            bh[j] = *(const half8_t*)((ST) + b_off + j * 32 * 64 + coff);                                \
            bl[j] = *(const half8_t*)((ST) + XPL + b_off + j * 32 * 64 + coff);                          \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                   \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16<BF>(al[i], bh[j], acc[i][j]);  \
        _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                   \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16<BF>(ah[i], bl[j], acc[i][j]);  \
        _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                   \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16<BF>(ah[i], bh[j], acc[i][j]);  \
    }

    const int nK = a.ksize * a.ksize * (a.Cin / 32);
    // (ky, kx, c0) of the K-step whose input is fetched next.  Every step fetches (at the end: clamped, unused), so the number of
    // loads in flight — what the counted vmcnt waits below rely on — never changes
    int ky = 0, kx = 0, c0 = 0;
    auto advance = [&]() {            // scalar selects, no branch; runs past the last K-step (ky = ksize) at the end: the fetches of
        const int c1 = c0 + 32;       // those steps are clamped to valid addresses and never used
        const bool wrapc = c1 == a.Cin;
        c0 = wrapc ? 0 : c1;
        const int kx1 = kx + (wrapc ? 1 : 0);
        const bool wrapx = kx1 == a.ksize;
        kx = wrapx ? 0 : kx1;
        ky += wrapx ? 1 : 0;
    };
    // K-step 0 -> stage 0; the loads of K-step 1 -> xa
    SP_ISSUE_W(0, smem);
    load_x(xa, 0, 0, 0);
    write_x(xa, smem);
    advance();
    load_x(xa, ky, kx, c0);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");         // all but the four loads just issued: this wave's weight DMA has landed
    lds_barrier();
    // K-step ks: stage ST is complete; P holds the input of K-step ks + 1 (loaded during K-step ks - 1), Q takes that of ks + 2
    auto step = [&](XRegs& P, XRegs& Q, int ks, char* st, char* nst) {
        SP_ISSUE_W(min((ky * a.ksize + kx) * a.Cin + c0, Ktot - 32), nst);      // (ky, kx, c0) = K-step ks + 1 here
        advance();
        load_x(Q, ky, kx, c0);
        __builtin_amdgcn_sched_barrier(0);      // the fetches stay at the top of the step (hipcc sinks them to its end otherwise)
        SP_SUBSTEP(st, 0);
        SP_SUBSTEP(st, 1);
        __builtin_amdgcn_sched_barrier(0);
        // P's loads are a whole K-step old; hipcc's wait for them (vmcnt(4): everything but Q's four loads) covers the weight DMA,
        // which is older than Q.  The split runs on the VALU under the tail of this wave's MFMAs and beside its SIMD partner's.
        write_x(P, nst);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // (explicit for the DMA: hipcc does not know it)
        lds_barrier();                     // (raw s_barrier: __syncthreads() would drain Q too) stage `st` is free again, `nst` is complete
    };
    for (int ks = 0; ks < nK; ks += 2) {
        step(xa, xb, ks, smem, smem + STAGE);
        if (ks + 1 < nK) step(xb, xa, ks + 1, smem + STAGE, smem);
    }
#undef SP_ISSUE_W
#undef SP_SUBSTEP

    // ---- epilogue: this lane's pixel of each pixel tile, the quads of each channel tile (fp32 in and out) ----
    float* const out = (float*)a.out;
    const float* const res = (const float*)a.res;
    // (one instantiation per pixel tile j, not a loop: hipcc gives up unrolling 2 x TI x 4 copies of the site arithmetic, and a loop
    //  that stays rolled indexes the accumulators dynamically — they then live in scratch for the whole kernel)
    auto finish = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int m_o = pix0 + wp * 64 + 32 * j + r;
        const bool vo = m_o < a.M;
        const int m_c = vo ? m_o : 0;
        const int n = m_c / HoWo, rem = m_c - n * HoWo;
        PixelCtx p;
        p.out_off = ((size_t)n * HoWo + rem) * a.Cout;
        p.resp = nullptr;
        const int tl = n / a.B;
        p.b = n - tl * a.B;
        p.t = a.t0 + tl;
        p.e_pix = p.b * HoWo + rem;
        p.mrow = a.site.kind == BMI_SITE_MASKSEMBLE ? a.site.masks + (size_t)((a.site.cnt0 + p.t) % a.site.num_masks) * a.Cout : nullptr;
        const float* resp = res ? res + ((size_t)(n % a.res_mod) * HoWo + rem) * a.Cout : nullptr;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c4 = ch0 + wc * 32 * TI + 32 * i + 8 * q + 4 * kq;
                float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                epilogue_quad_f32(a, p, resp, v, c4);
                if (vo) *(f32x4_s*)(out + p.out_off + c4) = f32x4_s{v[0], v[1], v[2], v[3]};
            }
    };
    finish(std::integral_constant<int, 0>{});
    finish(std::integral_constant<int, 1>{});
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
