// pca_wave.h -- wave64 scans and reductions on the VALU (DPP), not through the LDS crossbar.
// __shfl_up / __shfl_xor compile to ds_bpermute_b32: every step is an LDS-pipe round trip (~100+ cycles of latency
// in a dependent chain) and competes with the LDS atomics the BEV kernels live on.  The DPP forms below are one
// VALU instruction per step.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// v from the DPP-selected lane; lanes without a valid source (or masked off) read `old`
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_or(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}

#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_scan_add(uint32_t v)
{
    uint32_t t = v;
    t += dpp_or<DPP_ROW_SHR(1)>(0u, v);
    t += dpp_or<DPP_ROW_SHR(2)>(0u, v);
    t += dpp_or<DPP_ROW_SHR(3)>(0u, v);
    t += dpp_or<DPP_ROW_SHR(4), 0xf, 0xe>(0u, t);
    t += dpp_or<DPP_ROW_SHR(8), 0xf, 0xc>(0u, t);
    t += dpp_or<DPP_ROW_BCAST15, 0xa>(0u, t);
    t += dpp_or<DPP_ROW_BCAST31, 0xc>(0u, t);
    return t;
}

// full-wave reductions (result uniform, via lane 63)
#define PCA_WAVE_REDUCE(NAME, IDENT, OP)                                                   \
    __device__ __forceinline__ uint32_t NAME(uint32_t v)                                   \
    {                                                                                      \
        uint32_t t = v, o;                                                                 \
        o = dpp_or<DPP_ROW_SHR(1)>(IDENT, t); t = OP(t, o);                                \
        o = dpp_or<DPP_ROW_SHR(2)>(IDENT, t); t = OP(t, o);                                \
        o = dpp_or<DPP_ROW_SHR(4)>(IDENT, t); t = OP(t, o);                                \
        o = dpp_or<DPP_ROW_SHR(8)>(IDENT, t); t = OP(t, o);                                \
        o = dpp_or<DPP_ROW_BCAST15, 0xa>(IDENT, t); t = OP(t, o);                          \
        o = dpp_or<DPP_ROW_BCAST31, 0xc>(IDENT, t); t = OP(t, o);                          \
        return (uint32_t)__builtin_amdgcn_readlane((int)t, 63);                            \
    }
#define PCA_OP_ADD(a, b) ((a) + (b))
#define PCA_OP_MIN(a, b) ((a) < (b) ? (a) : (b))
#define PCA_OP_MAX(a, b) ((a) > (b) ? (a) : (b))
PCA_WAVE_REDUCE(wave_reduce_add, 0u, PCA_OP_ADD)
PCA_WAVE_REDUCE(wave_reduce_min, 0xffffffffu, PCA_OP_MIN)
PCA_WAVE_REDUCE(wave_reduce_max, 0u, PCA_OP_MAX)

// ---------------------------------------------------------------------------------------------
// 32 x 32 bit-matrix transpose across lanes, done at once in both halves of the wave: on entry lane l holds row
// (l & 31) of its half; on return bit j of lane l is bit (l & 31) of the entry value of lane (l & 32) + j.
// Five butterfly stages; partners come over DPP / v_permlane16_swap, nothing goes through LDS.
// ---------------------------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ uint32_t lane_xor_fetch(uint32_t x)
{
    if (S == 1) return dpp_or<0xb1>(0u, x);                                   // quad_perm [1,0,3,2]
    if (S == 2) return dpp_or<0x4e>(0u, x);                                   // quad_perm [2,3,0,1]
    if (S == 4) return dpp_or<0x1b>(0u, dpp_or<0x141>(0u, x));                // row_half_mirror (^7) then [3,2,1,0] (^3)
    if (S == 8) return dpp_or<0x128>(0u, x);                                  // row_ror:8
    const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);      // r[0] = rows x0 x0 x2 x2, r[1] = x1 x1 x3 x3
    return (threadIdx.x & 16) ? r[0] : r[1];
}
template <int S, uint32_t M>
__device__ __forceinline__ uint32_t transpose_stage(uint32_t x)
{
    const bool lo = !(threadIdx.x & S);
    const uint32_t p = lane_xor_fetch<S>(x);
    const uint32_t moved = lo ? (p << S) : (p >> S);
    const uint32_t keep = lo ? M : ~M;                      // M = bit positions with bit S clear
    return (x & keep) | (moved & ~keep);
}
__device__ __forceinline__ uint32_t wave_bit_transpose32(uint32_t x)
{
    x = transpose_stage<16, 0x0000ffffu>(x);
    x = transpose_stage<8, 0x00ff00ffu>(x);
    x = transpose_stage<4, 0x0f0f0f0fu>(x);
    x = transpose_stage<2, 0x33333333u>(x);
    x = transpose_stage<1, 0x55555555u>(x);
    return x;
}
