// Philox4x32-10 (Salmon et al., SC'11; Random123 philox4x32_R(10,...)) and the dropout-mask
// convention shared with the CPU oracle (oracle/philox.py):
//   k    = bits per element: the fewest of {2, 4, 8, 16} with fl32(p) * 2^k an integer, else 16
//          (p = 0.25, 0.5 -> 2; 0.125, 0.375 -> 4; 0.1 -> 16, P(drop) quantised to 1/65536);
//   key  = (seed_lo, seed_hi); counter = (g_lo, g_hi, t, site), g = element_index / (128 / k);
//   element e uses field f = e % (128 / k): bits [f*k, (f+1)*k) of the 128 output bits r[0] | r[1] << 32 | ...;
//   keep iff field >= thresh, thresh = floor(fl32(p) * 2^k + 0.5)  (2^k: drop everything).
// element_index is the NHWC-linear index inside ONE Monte-Carlo sample's activation, so the eight consecutive
// channels a lane moves (16 bytes) always sit in ONE call — and at k = 2 one call covers 64 of them, which the
// kernels share between lanes.  History: 32 bits per element, then 16 (the stand-alone mask kernel was
// Philox-bound), then p-dependent: v_mad_u64_u32 issues at a quarter rate, one call costs a wave ~480 cycles.
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
    uint32_t thresh;     // keep iff the element's k-bit field >= thresh
    int log2_bits;       // log2(k): 1..4 (k = 2, 4, 8, 16 bits per element)
    int drop_all;        // p >= 1
    float scale;         // fl32(1 / fl32(1 - p)); 1 for MASKSEMBLE
    const float* masks;  // MASKSEMBLE [M][C]
    int num_masks;
    int cnt0;
    uint32_t seed_lo, seed_hi;
    // Element index of this LAUNCH's image 0 in the site's index space: 0 for a whole batch; b0 * (elements per image of the
    // site's tensor) when a launch covers images b0.. of the batch only (bmi_forward_mcd_images: a rank's share of an
    // image-partitioned batch).  A multiple of 64 elements (one Philox call at 2 bits per element): the fields inside a call do
    // not move.  Added where the call index is formed — philox_site_call below and the two mask kernels that form it themselves.
    uint64_t elem_off;
};

// log2 of the bits drawn per element for drop probability p
static inline int bmi_site_log2_bits(float p) {
    for (int lb = 1; lb <= 3; ++lb) {
        const double v = (double)p * (double)(1u << (1 << lb));
        if (v == (double)(uint64_t)v) return lb;
    }
    return 4;
}

static inline uint32_t bmi_drop_threshold(float p, int log2_bits, int* drop_all) {
    const uint32_t full = 1u << (1 << log2_bits);   // 2^k
    double v = (double)p * (double)full + 0.5;
    uint64_t t = v <= 0 ? 0 : (uint64_t)v;  // floor
    *drop_all = t >= full;
    return *drop_all ? full - 1 : (uint32_t)t;
}

// Keep flags of the 8 consecutive elements elem0 .. elem0+7 (elem0 % 8 == 0) inside the Philox output `r` of their
// call: bit e = element elem0 + e.  The 8 fields are 8k contiguous bits: one 16-bit half (k = 2), one word (k = 4),
// two words (k = 8) or all four (k = 16).
BMI_HD uint32_t philox_keep8(const philox4& r, uint32_t elem0, int log2_bits, uint32_t thresh) {
    const uint32_t r0 = r.w[0], r1 = r.w[1], r2 = r.w[2], r3 = r.w[3];   // scalars: a dynamically indexed r.w[] goes to scratch
    const uint32_t f0 = elem0 & ((1u << (7 - log2_bits)) - 1);   // first field
    const uint32_t bitpos = f0 << log2_bits;                     // multiple of 8k
    uint32_t keep = 0;
    if (log2_bits == 4) {
        const uint32_t w[4] = {r0, r1, r2, r3};
#pragma unroll
        for (int e = 0; e < 8; ++e) keep |= (((w[e >> 1] >> ((e & 1) * 16)) & 0xFFFFu) >= thresh) ? (1u << e) : 0u;
    } else if (log2_bits == 3) {
        const uint32_t w0 = bitpos ? r2 : r0, w1 = bitpos ? r3 : r1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            keep |= (((w0 >> (8 * e)) & 0xFFu) >= thresh) ? (1u << e) : 0u;
            keep |= (((w1 >> (8 * e)) & 0xFFu) >= thresh) ? (16u << e) : 0u;
        }
    } else {
        const uint32_t lo = (bitpos & 32) ? r1 : r0, hi = (bitpos & 32) ? r3 : r2;
        uint32_t w = (bitpos & 64) ? hi : lo;
        if (log2_bits == 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) keep |= (((w >> (4 * e)) & 0xFu) >= thresh) ? (1u << e) : 0u;
        } else {
            w >>= (bitpos & 16);   // 0 or 16
#pragma unroll
            for (int e = 0; e < 8; ++e) keep |= (((w >> (2 * e)) & 0x3u) >= thresh) ? (1u << e) : 0u;
        }
    }
    return keep;
}

// Philox call index of element elem0 and the call itself
BMI_HD philox4 philox_site_call(const SiteArgs& s, uint64_t elem0, uint32_t t) {
    // g = elem0 >> sh as two 32-bit halves (a variable 64-bit shift made hipcc park elem0 in scratch)
    elem0 += s.elem_off;
    const uint32_t sh = 7u - (uint32_t)s.log2_bits;                  // 3 .. 6
    const uint32_t lo = (uint32_t)elem0, hi = (uint32_t)(elem0 >> 32);
    return philox4x32_10((lo >> sh) | (hi << (32u - sh)), hi >> sh, t, (uint32_t)s.site_id, s.seed_lo, s.seed_hi);
}

// keep flags (bit e) of elements elem0 .. elem0+7 (elem0 % 8 == 0) of site stream (seed, site, t); 0 when p >= 1
BMI_HD uint32_t site_keep8(const SiteArgs& s, uint64_t elem0, uint32_t t) {
    if (s.drop_all) return 0u;
    return philox_keep8(philox_site_call(s, elem0, t), (uint32_t)elem0, s.log2_bits, s.thresh);
}

static inline float bmi_drop_scale(float p) {
    const float q = 1.0f - p;
    return q <= 0.f ? 0.f : 1.0f / q;
}
