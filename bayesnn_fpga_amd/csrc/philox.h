// Philox4x32-10 (Salmon et al., SC'11; Random123 philox4x32_R(10,...)) and the dropout-mask
// convention shared with the CPU oracle (oracle/philox.py):
//   key = (seed_lo, seed_hi); counter = (g_lo, g_hi, t, site), g = element_index / 8;
//   element e uses the 16-bit half (e & 1) of word r[(e & 7) >> 1]  (even e: low half);
//   keep iff half >= thresh16, thresh16 = floor(fl32(p) * 65536 + 0.5)  (65536: drop everything).
// element_index is the NHWC-linear index inside ONE Monte-Carlo sample's activation, so ONE call
// covers eight consecutive channels of one pixel — the 16 bytes a lane moves in the coalesced conv
// epilogue, in mask_apply and in pool_mask.  (Round 1 started with 32 bits per element; the
// stand-alone mask kernel was Philox-bound, so the convention went to 16 bits: P(drop) is then
// quantised to 1/65536, a relative error below 8e-6 for every p the reference sweeps.)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BMI_HD __host__ __device__ __forceinline__
#else
#define BMI_HD inline
#endif

struct philox4 {
    uint32_t w[4];
};

BMI_HD philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    philox4 out;
    out.w[0] = c0; out.w[1] = c1; out.w[2] = c2; out.w[3] = c3;
    return out;
}

// Host-side resolved form of a stochastic site, passed by value to kernels.
struct SiteArgs {
    int kind;            // BMI_SITE_*
    int site_id;
    uint32_t thresh;     // keep iff 16-bit half-word >= thresh
    int drop_all;        // p >= 1
    float scale;         // fl32(1 / fl32(1 - p)); 1 for MASKSEMBLE
    const float* masks;  // MASKSEMBLE [M][C]
    int num_masks;
    int cnt0;
    uint32_t seed_lo, seed_hi;
};

static inline uint32_t bmi_drop_threshold(float p, int* drop_all) {
    double v = (double)p * 65536.0 + 0.5;
    uint64_t t = v <= 0 ? 0 : (uint64_t)v;  // floor
    *drop_all = t >= 65536;
    return *drop_all ? 0xFFFFu : (uint32_t)t;
}

// keep flag of element e (0..7) of the call's group
BMI_HD bool philox_keep(const philox4& r, int e, uint32_t thresh16) {
    return ((r.w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu) >= thresh16;
}

static inline float bmi_drop_scale(float p) {
    const float q = 1.0f - p;
    return q <= 0.f ? 0.f : 1.0f / q;
}
