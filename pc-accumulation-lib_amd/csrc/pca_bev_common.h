// pca_bev_common.h -- wave-level building blocks of the BEV rasteriser (medians, exact sums).
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

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}

// Value of lane (l ^ J), VALU only (no LDS traffic, no bpermute latency):
//   J = 1, 2  DPP quad_perm;  J = 8  DPP row_ror:8 (rotation by half a 16-lane row);
//   J = 4     DPP row_shl:4 / row_shr:4 selected by lane bit 2;
//   J = 16,32 gfx950 v_permlane16_swap / v_permlane32_swap of the value with itself.
template <int J>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int lane)
{
    if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);
    else if constexpr (J == 4) {
        const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0xF, true);   // from l+4
        const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);   // from l-4
        return (lane & 4) ? dn : up;
    } else if constexpr (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane & 32) ? r[0] : r[1];
    }
}

template <int K, int J>
__device__ __forceinline__ void sort_stage(uint32_t &key, int lane)
{
    const uint32_t o = lane_xor<J>(key, lane);
    const bool up = (lane & K) == 0, lower = (lane & J) == 0;
    key = (up == lower) ? pk_min(key, o) : pk_max(key, o);
}

// ascending bitonic sort of two independent u16 keys per lane across the 64 lanes of the wave (21 stages)
__device__ __forceinline__ uint32_t wave_sort_pk16(uint32_t key)
{
    const int lane = threadIdx.x & 63;
    sort_stage<2, 1>(key, lane);
    sort_stage<4, 2>(key, lane); sort_stage<4, 1>(key, lane);
    sort_stage<8, 4>(key, lane); sort_stage<8, 2>(key, lane); sort_stage<8, 1>(key, lane);
    sort_stage<16, 8>(key, lane); sort_stage<16, 4>(key, lane); sort_stage<16, 2>(key, lane); sort_stage<16, 1>(key, lane);
    sort_stage<32, 16>(key, lane); sort_stage<32, 8>(key, lane); sort_stage<32, 4>(key, lane); sort_stage<32, 2>(key, lane);
    sort_stage<32, 1>(key, lane);
    sort_stage<64, 32>(key, lane); sort_stage<64, 16>(key, lane); sort_stage<64, 8>(key, lane); sort_stage<64, 4>(key, lane);
    sort_stage<64, 2>(key, lane); sort_stage<64, 1>(key, lane);
    return key;
}

__device__ __forceinline__ uint32_t lane_rank_in(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// value (key>>1) of the lane that is member number `k` (0-based, in lane order) of `members`
__device__ __forceinline__ uint32_t pick_member(uint32_t key16, uint64_t members, uint32_t rank, uint32_t k)
{
    const uint64_t hit = __ballot(((members >> (threadIdx.x & 63)) & 1ull) && rank == k);
    const int src = hit ? (int)__ffsll((unsigned long long)hit) - 1 : 0;
    return __builtin_amdgcn_readlane(key16, src) >> 1;
}

// sorted 16-bit keys (value<<1 | set, 0xffff = empty) -> 2*median of the present / future members
__device__ __forceinline__ void medians_from_sorted(uint32_t key16, uint32_t n_p, uint32_t n_f, uint32_t &m2_p,
                                                    uint32_t &m2_f)
{
    const bool valid = key16 != 0xffffu;
    const uint64_t mp = __ballot(valid && (key16 & 1u) == 0u), mf = __ballot(valid && (key16 & 1u) == 1u);
    const uint32_t rp = lane_rank_in(mp), rf = lane_rank_in(mf);
    m2_p = m2_f = 0;
    if (n_p) m2_p = pick_member(key16, mp, rp, (n_p - 1) >> 1) + pick_member(key16, mp, rp, n_p >> 1);
    if (n_f) m2_f = pick_member(key16, mf, rf, (n_f - 1) >> 1) + pick_member(key16, mf, rf, n_f >> 1);
}

// One cell with n = n_p + n_f <= 64 colour values, one per lane (present lanes first): 2*median of r, g, b for
// present, future and full.  rgb = r | g<<8 | b<<16.  out[set][ch], set 2 = full.
__device__ __forceinline__ void cell_medians_64(uint32_t rgb, uint32_t n_p, uint32_t n_f, uint32_t out[3][3])
{
    const int lane = threadIdx.x & 63;
    const uint32_t n = n_p + n_f;
    const bool act = (uint32_t)lane < n;
    const uint32_t set = (act && (uint32_t)lane >= n_p) ? 1u : 0u;
    const uint32_t kr = act ? (((rgb & 255u) << 1) | set) : 0xffffu;
    const uint32_t kg = act ? ((((rgb >> 8) & 255u) << 1) | set) : 0xffffu;
    const uint32_t kb = act ? ((((rgb >> 16) & 255u) << 1) | set) : 0xffffu;
    const uint32_t sa = wave_sort_pk16(kr | (kg << 16));
    const uint32_t sb = wave_sort_pk16(kb | 0xffff0000u);
    const uint32_t ch[3] = {sa & 0xffffu, sa >> 16, sb & 0xffffu};
    const uint32_t lo_l = (n - 1) >> 1, hi_l = n >> 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        medians_from_sorted(ch[c], n_p, n_f, out[0][c], out[1][c]);
        out[2][c] = (__builtin_amdgcn_readlane(ch[c], lo_l) >> 1) + (__builtin_amdgcn_readlane(ch[c], hi_l) >> 1);
    }
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

// 2*median (lower + upper middle value) of a 256-bin histogram with n entries, called by one whole wave
__device__ __forceinline__ uint32_t hist_med2(const uint32_t *hist256, uint32_t n)
{
    if (n == 0) return 0;
    const int lane = threadIdx.x & 63;
    const uint4 h = *reinterpret_cast<const uint4 *>(hist256 + 4 * lane);
    const uint32_t s = h.x + h.y + h.z + h.w;
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const uint32_t excl = inc - s;
    return hist_rank(h, excl, s, (n - 1) >> 1) + hist_rank(h, excl, s, n >> 1);
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
