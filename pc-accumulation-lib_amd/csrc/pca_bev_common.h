// pca_bev_common.h -- wave-level building blocks of the BEV rasteriser (histogram medians, exact sums, closed forms).
#pragma once
#include "pca_common.h"

// Intensity sums are accumulated as exact integers so that the result does not depend on the order in
// which points arrive: value = hi * 2^-20 + lo * 2^-60 with hi = floor(v * 2^20), lo = rint(frac * 2^40).
// Resolution 2^-60 (8.7e-19) per point; both partial sums fit an int64 for any realistic cell.
#define FX_HI 1048576.0                     /* 2^20 */
#define FX_LO 1099511627776.0               /* 2^40 */
#define FX_HI_INV (1.0 / 1048576.0)
#define FX_LO_INV (1.0 / 1152921504606846976.0)   /* 2^-60 */

// order-preserving u64 image of a double (for integer atomic min)
__device__ __forceinline__ uint64_t f64_order_key(double d)
{
    const uint64_t b = (uint64_t)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_from_order_key(uint64_t k)
{
    const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// 256-bin histogram spread 4 bins per lane: value at 0-based rank k
__device__ __forceinline__ uint32_t hist_rank(const uint4 h, uint32_t excl, uint32_t s, uint32_t k)
{
    const int lane = threadIdx.x & 63;
    const bool own = (k >= excl) && (k < excl + s);
    uint32_t val = 0;
    if (own) {
        const uint32_t r = k - excl;
        val = 4 * lane + (r < h.x ? 0 : (r < h.x + h.y ? 1 : (r < h.x + h.y + h.z ? 2 : 3)));
    }
    const uint64_t m = __ballot(own);
    const int src = m ? (int)__ffsll((unsigned long long)m) - 1 : 0;
    return __builtin_amdgcn_readlane(val, src);
}

// 2*median (lower + upper middle value) given the lane's 4 bins, their sum s and its exclusive prefix over the wave
__device__ __forceinline__ uint32_t hist_med2_scanned(const uint4 h, uint32_t excl, uint32_t s, uint32_t n)
{
    if (n == 0) return 0;
    const uint32_t lo = hist_rank(h, excl, s, (n - 1) >> 1);
    return (n & 1u) ? 2u * lo : lo + hist_rank(h, excl, s, n >> 1);      // odd n: both middles are the same entry
}
// 2*median of a 256-bin histogram with n entries, 4 bins per lane, one whole wave
__device__ __forceinline__ uint32_t hist_med2_regs(const uint4 h, uint32_t n)
{
    if (n == 0) return 0;
    const uint32_t s = h.x + h.y + h.z + h.w;
    return hist_med2_scanned(h, wave_incl_scan_add(s) - s, s, n);
}
__device__ __forceinline__ uint32_t hist_med2(const uint32_t *hist256, uint32_t n)
{
    return hist_med2_regs(*reinterpret_cast<const uint4 *>(hist256 + 4 * (threadIdx.x & 63)), n);
}

// closed-form maps of one (cell, set): writes 7 values {road, intensity, r, g, b, dynamic, elevation}
__device__ __forceinline__ void finalize_cell(const pca_bev_params &q, uint32_t n, uint32_t n_r, uint32_t n_d, long long ihi,
                                              long long ilo, double zmin, const uint32_t m2[3], double out[7])
{
    const double a_all = (double)n, a_r = (double)n_r, a_d = (double)n_d;
    out[0] = (a_r + 1.0) / ((a_r + 1.0) + ((a_all - a_r) + 1.0));          // dirichlet expectation, road
    const double isum = (double)ihi * FX_HI_INV + (double)ilo * FX_LO_INV;
    const double iraw = isum / (a_r + 1.0);
    const double zarg = q.int_sep_scaler * (iraw - q.int_mid_threshold);
    double inten = q.int_scaler * (1.0 / (1.0 + exp(-zarg)));
    if (inten > 1.0) inten = 1.0;
    out[1] = inten;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) out[2 + ch] = (n ? (double)m2[ch] / 2.0 : q.rgb_fill) / 255.0;
    out[5] = (a_d + 1.0) / ((a_d + 1.0) + ((a_all - a_d) + 1.0));          // dirichlet expectation, vehicles
    out[6] = n ? zmin : 0.0;
}
