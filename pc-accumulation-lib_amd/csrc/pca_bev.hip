// pca_bev.hip -- BEV rasteriser for gfx950 (K4 bin, scan, K4b scatter, K5-K7 per-cell reduce+finalize).
//
// Pipeline (all HBM-bound; no global atomics except ONE returning u32 add per in-view point):
//   bev_bin      window points -> rotate/translate/crop/height/floor -> key = cell*2+set,
//                rank = atomicAdd(count[key], 1)                      (writes key, rank per point)
//   bev_scan     exclusive scan of count[] (decoupled look-back)     -> segment offsets per (cell,set)
//   bev_scatter  record {z, intensity, rgbs} -> recs[offs[key] + rank]   (counting sort by cell)
//   bev_cells    one wave per cell: LDS histograms (256 bins x 3 channels x {present,future}) for the
//                exact medians, ballot counts, integer intensity sum, min z; closed-form maps; fp16.
// 'full' = present + future is formed per cell from the two sets (histograms add, counts add, min of mins).
#include "pca_common.h"

#define BLK 256
#define KEY_INVALID 0xffffffffu
#define CELLS_PER_BLOCK 64
// Intensity sums are accumulated as exact integers so that the result does not depend on the order in
// which points arrive: value = hi * 2^-20 + lo * 2^-60 with hi = floor(v * 2^20), lo = rint(frac * 2^40).
// Resolution 2^-60 (8.7e-19) per point; both partial sums fit an int64 for any realistic cell.
#define FX_HI 1048576.0                     /* 2^20 */
#define FX_LO 1099511627776.0               /* 2^40 */
#define FX_HI_INV (1.0 / 1048576.0)
#define FX_LO_INV (1.0 / 1152921504606846976.0)   /* 2^-60 */

struct Rec16 { double z; float inten; uint32_t rgbs; };
struct Rec24 { double z; double inten; uint32_t rgbs; uint32_t pad; };

struct BevArgs {
    pca_store st;
    const double *intensity64;
    const int64_t *frame_off;
    int slot_begin, slot_split, slot_end;
    int64_t max_points;
    pca_bev_params prm;
    uint32_t *cnt;     // [2*ncell]
    uint32_t *offs;    // [2*ncell+1]
    uint32_t *key;     // [max_points]
    uint32_t *rank;    // [max_points]
    void *recs;        // Rec16/Rec24 [max_points]
    double *planes;
    uint16_t *planes_f16;
    uint64_t *state;
    uint32_t *ticket;
    uint32_t epoch;
    int scan_tiles;
};

// ---------------------------------------------------------------------------------------------
// K4  bin
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLK) void bev_bin(const BevArgs a)
{
    const int64_t lo = a.frame_off[a.slot_begin], hi0 = a.frame_off[a.slot_end], sp = a.frame_off[a.slot_split];
    const int64_t hi = (hi0 - lo > a.max_points) ? lo + a.max_points : hi0;
    if (hi0 - lo > a.max_points && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.ticket + 1, PCA_STATUS_STORE_OVERFLOW);
    const pca_bev_params &q = a.prm;
    const double v = q.view, vlo = -0.5 * v, vhi = 0.5 * v, pxd = (double)q.px, half_px = 0.5 * pxd;
    const bool use_h = !(q.height_filter != q.height_filter);
    // UNR points per thread and iteration: all loads first, then the (returning) atomics, then the stores,
    // so that several atomics per lane are in flight instead of one.
    constexpr int UNR = 4;
    const int64_t gsz = (int64_t)gridDim.x * BLK;
    for (int64_t base = lo + (int64_t)blockIdx.x * BLK + threadIdx.x; base < hi; base += UNR * gsz) {
        double X[UNR], Y[UNR], Z[UNR];
        uint8_t D[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t p = base + u * gsz;
            const bool in = p < hi;
            X[u] = in ? a.st.x[p] : 0.0;
            Y[u] = in ? a.st.y[p] : 0.0;
            Z[u] = in ? a.st.z[p] : 0.0;
            D[u] = in ? a.st.dyn[p] : (uint8_t)1;
        }
        uint32_t key[UNR], rk[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t p = base + u * gsz;
            const double x = X[u] - q.origin[0];
            const double y = Y[u] - q.origin[1];
            const double z = Z[u] - q.origin[2];
            double ax = q.R[0] * x; ax = fma(q.R[1], y, ax); ax = fma(q.R[2], z, ax);
            double ay = q.R[3] * x; ay = fma(q.R[4], y, ay); ay = fma(q.R[5], z, ay);
            double az = q.R[6] * x; az = fma(q.R[7], y, az); az = fma(q.R[8], z, az);
            ax += q.dx;
            ay += q.dy;
            bool keep = (ax > vlo) && (ax < vhi) && (ay > vlo) && (ay < vhi);
            if (use_h) keep = keep && (az < q.height_filter);
            keep = keep && (D[u] != 1) && (p < hi);
            key[u] = KEY_INVALID;
            if (keep) {
                int i = (int)floor(ax / v * pxd + half_px);
                int j = (int)floor(ay / v * pxd + half_px);
                i = i > q.px - 1 ? q.px - 1 : (i < 0 ? 0 : i);
                j = j > q.px - 1 ? q.px - 1 : (j < 0 ? 0 : j);
                const uint32_t cell = (uint32_t)((q.px - 1 - j) * q.px + i);
                key[u] = cell * 2u + (p >= sp ? 1u : 0u);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) rk[u] = key[u] != KEY_INVALID ? atomicAdd(&a.cnt[key[u]], 1u) : 0u;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t p = base + u * gsz;
            if (p < hi) { a.key[p - lo] = key[u]; a.rank[p - lo] = rk[u]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// exclusive scan of cnt[0..n) -> offs[0..n], offs[n] = total      (1024 entries per tile)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLK) void bev_scan(const BevArgs a)
{
    __shared__ int s_tile;
    __shared__ uint32_t s_w[BLK / PCA_WAVE];
    __shared__ uint64_t s_excl;
    const int n = 2 * a.prm.px * a.prm.px;
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(a.ticket, 1u);
        if ((int)t == a.scan_tiles - 1) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tile = (int)t;
    }
    __syncthreads();
    const int tile = s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = tile * 1024 + threadIdx.x * 4;
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = (base + k < n) ? a.cnt[base + k] : 0u;
    const uint32_t tsum = c[0] + c[1] + c[2] + c[3];
    uint32_t inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BLK / PCA_WAVE; ++w) {
        if (w < wave) wbase += s_w[w];
        total += s_w[w];
    }
    if (wave == 0) {
        const uint64_t e = lb_exclusive_prefix(a.state, tile, (uint64_t)total, a.epoch);
        if (lane == 0) s_excl = e;
    }
    __syncthreads();
    uint32_t run = (uint32_t)s_excl + wbase + (inc - tsum);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) a.offs[base + k] = run;
        run += c[k];
    }
    if (tile == a.scan_tiles - 1 && threadIdx.x == BLK - 1) a.offs[n] = (uint32_t)s_excl + total;
}

// ---------------------------------------------------------------------------------------------
// K4b scatter records into cell order
// ---------------------------------------------------------------------------------------------
template <bool I64>
__global__ __launch_bounds__(BLK) void bev_scatter(const BevArgs a)
{
    const int64_t lo = a.frame_off[a.slot_begin], hi0 = a.frame_off[a.slot_end];
    const int64_t hi = (hi0 - lo > a.max_points) ? lo + a.max_points : hi0;
    const double oz = a.prm.origin[2];
    for (int64_t p = lo + (int64_t)blockIdx.x * BLK + threadIdx.x; p < hi; p += (int64_t)gridDim.x * BLK) {
        const uint32_t key = a.key[p - lo];
        if (key == KEY_INVALID) continue;
        const uint32_t pos = a.offs[key] + a.rank[p - lo];
        const double z = a.st.z[p] - oz;        // rotation about z: row 3 of R is (0,0,1) -> z unchanged
        if (I64) {
            Rec24 r; r.z = z; r.inten = a.intensity64[p]; r.rgbs = a.st.rgbs[p]; r.pad = 0;
            reinterpret_cast<Rec24 *>(a.recs)[pos] = r;
        } else {
            Rec16 r; r.z = z; r.inten = a.st.intensity[p]; r.rgbs = a.st.rgbs[p];
            reinterpret_cast<Rec16 *>(a.recs)[pos] = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K5-K7 per-cell reduce + finalize
//
// A workgroup (4 waves) owns 64 consecutive cells, a wave 16 of them, one after the other.  Per cell the
// wave produces raw statistics for the sets {present, future}:
//     n, n_road, n_dynobj, intensity sum (two exact integer limbs), min z, 2*median of r, g, b
// and parks them in LDS; after the cell loop 192 threads finish 3 sets x 64 cells in parallel
// (dirichlet ratios, sigmoid, /255, 'full' = present (+) future) and the block writes 21 x 64 values
// with coalesced stores.
//   n <= 64 (the common case): one record per lane, registers only.  Medians come from two packed
//            16-bit bitonic sorts (keys value<<1|set; r,g in one register, b in the other): the full-set
//            median is read at lanes (n-1)/2, n/2, the per-set medians at the lane whose rank among the
//            set's lanes (mbcnt of a ballot) is the wanted one.  Counts are ballots; intensity sums and
//            min z are LDS atomics on a 6-word per-wave scratch (integer adds / order-preserving u64 min).
//   n  > 64: per-wave LDS histograms, 256 bins x 3 channels x 2 sets (the previous general path).
// ---------------------------------------------------------------------------------------------
struct CellStat {           // per (cell, set in {present, future})
    uint32_t n, n_road, n_dyn;
    uint32_t med2[3];       // lower + upper median of r, g, b (0..510); only valid if n > 0
    long long ihi, ilo;     // intensity sum limbs
    double zmin;
};

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

// sorted 16-bit keys (value<<1 | set, 0xffff = empty) -> 2*median for present, future (per-set ranks)
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

__device__ __forceinline__ uint32_t hist_med2(const uint4 h, uint32_t n)
{
    if (n == 0) return 0;
    const int lane = threadIdx.x & 63;
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

template <bool I64>
__device__ __forceinline__ void load_rec(const BevArgs &a, uint32_t r, bool act, double &z, double &iv, uint32_t &rgbs)
{
    z = 0; iv = 0; rgbs = 0;
    if (!act) return;
    if (I64) {
        const Rec24 rec = reinterpret_cast<const Rec24 *>(a.recs)[r];
        z = rec.z; iv = rec.inten; rgbs = rec.rgbs;
    } else {
        const Rec16 rec = reinterpret_cast<const Rec16 *>(a.recs)[r];
        z = rec.z; rgbs = rec.rgbs;
        iv = a.prm.intensity_div255 ? (double)rec.inten / 255.0 : (double)rec.inten;
    }
}

template <bool I64>
__global__ __launch_bounds__(BLK) void bev_cells(const BevArgs a)
{
    constexpr int NW = BLK / PCA_WAVE;
    constexpr int CPW = CELLS_PER_BLOCK / NW;                   // cells per wave
    __shared__ uint32_t s_hist[NW][2][3][256];                  // 24 KB, only touched by cells with n > 64
    __shared__ CellStat s_stat[2][CELLS_PER_BLOCK];             // 6 KB
    __shared__ unsigned long long s_acc[NW][8];                 // [set*3 + {hi, lo, zkey}]
    __shared__ double s_out[21][CELLS_PER_BLOCK];               // 10.5 KB
    __shared__ uint32_t s_full[NW][CPW][3];                     // 2*median of the full set
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const pca_bev_params &q = a.prm;
    const int ncell = q.px * q.px;
    const int cell0 = blockIdx.x * CELLS_PER_BLOCK;
    const int wcell0 = cell0 + wave * CPW;
    uint32_t *hflat = &s_hist[wave][0][0][0];
    unsigned long long *acc = s_acc[wave];
    bool hist_clean = false;

    // segment offsets of this wave's cells: 2*CPW+1 consecutive words, one per lane
    uint32_t myoff = 0;
    {
        const int64_t idx = 2ll * wcell0 + lane;
        if (lane <= 2 * CPW && idx <= 2ll * ncell) myoff = a.offs[idx];
        // cells past the grid: give them the last valid offset so that they look empty
        const uint32_t last = a.offs[2ll * ncell];
        if (idx > 2ll * ncell) myoff = last;
    }

    // software pipeline: the first 64 records of the next cell are loaded while this one is reduced
    double zn, ivn;
    uint32_t rgbn;
    {
        const uint32_t o_p = __builtin_amdgcn_readlane(myoff, 0), o_e = __builtin_amdgcn_readlane(myoff, 2);
        load_rec<I64>(a, o_p + lane, o_p + lane < o_e, zn, ivn, rgbn);
    }

    for (int ci = 0; ci < CPW; ++ci) {
        const int lc = wave * CPW + ci;
        const uint32_t o_p = __builtin_amdgcn_readlane(myoff, 2 * ci), o_f = __builtin_amdgcn_readlane(myoff, 2 * ci + 1),
                       o_e = __builtin_amdgcn_readlane(myoff, 2 * ci + 2);
        const uint32_t n_p = o_f - o_p, n_f = o_e - o_f, n = o_e - o_p;
        double z = zn, iv = ivn;
        uint32_t rgbs = rgbn;
        if (ci + 1 < CPW) {
            const uint32_t p2 = o_e, e2 = __builtin_amdgcn_readlane(myoff, 2 * ci + 4);
            load_rec<I64>(a, p2 + lane, p2 + lane < e2, zn, ivn, rgbn);
        }
        uint32_t nr[2] = {0, 0}, nd[2] = {0, 0}, med2[2][3] = {{0, 0, 0}, {0, 0, 0}};
        long long shi[2] = {0, 0}, slo[2] = {0, 0};
        double zm[2] = {0.0, 0.0};

        if (n > 0 && n <= 64) {
            // ------------------------------------------------------------ register path
            const bool act = lane < n;
            const uint32_t set = (act && lane >= n_p) ? 1u : 0u;
            const unsigned sem = rgbs >> 24;
            const bool road = act && ((int)sem == q.road_class);
            const bool dynobj = act && ((q.dynobj_mask[sem >> 6] >> (sem & 63)) & 1ull);
            const uint64_t m_p = n_p >= 64 ? ~0ull : ((1ull << n_p) - 1ull);
            const uint64_t m_all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
            const uint64_t m_f = m_all & ~m_p;
            const uint64_t b_road = __ballot(road), b_dyn = __ballot(dynobj);
            nr[0] = (uint32_t)__popcll(b_road & m_p); nr[1] = (uint32_t)__popcll(b_road & m_f);
            nd[0] = (uint32_t)__popcll(b_dyn & m_p); nd[1] = (uint32_t)__popcll(b_dyn & m_f);
            if (lane < 6) acc[lane] = (lane % 3 == 2) ? ~0ull : 0ull;
            if (road) {
                const double sc = iv * FX_HI, fl = floor(sc);
                atomicAdd(&acc[set * 3 + 0], (unsigned long long)(long long)fl);
                atomicAdd(&acc[set * 3 + 1], (unsigned long long)(long long)rint((sc - fl) * FX_LO));
            }
            if (act) atomicMin(&acc[set * 3 + 2], (unsigned long long)f64_order_key(z));
            // medians
            const uint32_t kr = act ? (((rgbs & 255u) << 1) | set) : 0xffffu;
            const uint32_t kg = act ? ((((rgbs >> 8) & 255u) << 1) | set) : 0xffffu;
            const uint32_t kb = act ? ((((rgbs >> 16) & 255u) << 1) | set) : 0xffffu;
            const uint32_t sa = wave_sort_pk16(kr | (kg << 16));
            const uint32_t sb = wave_sort_pk16(kb | 0xffff0000u);
            medians_from_sorted(sa & 0xffffu, n_p, n_f, med2[0][0], med2[1][0]);
            medians_from_sorted(sa >> 16, n_p, n_f, med2[0][1], med2[1][1]);
            medians_from_sorted(sb & 0xffffu, n_p, n_f, med2[0][2], med2[1][2]);
            // 'full' medians straight from the sorted order (stored in the future slot's spare: see below)
            const uint32_t lo_l = (n - 1) >> 1, hi_l = n >> 1;
            const uint32_t fr = (__builtin_amdgcn_readlane(sa & 0xffffu, lo_l) >> 1) + (__builtin_amdgcn_readlane(sa & 0xffffu, hi_l) >> 1);
            const uint32_t fg = (__builtin_amdgcn_readlane(sa >> 16, lo_l) >> 1) + (__builtin_amdgcn_readlane(sa >> 16, hi_l) >> 1);
            const uint32_t fb = (__builtin_amdgcn_readlane(sb & 0xffffu, lo_l) >> 1) + (__builtin_amdgcn_readlane(sb & 0xffffu, hi_l) >> 1);
            if (lane == 0) { s_full[wave][ci][0] = fr; s_full[wave][ci][1] = fg; s_full[wave][ci][2] = fb; }
            shi[0] = (long long)acc[0]; slo[0] = (long long)acc[1]; zm[0] = f64_from_order_key(acc[2]);
            shi[1] = (long long)acc[3]; slo[1] = (long long)acc[4]; zm[1] = f64_from_order_key(acc[5]);
        } else if (n > 64) {
            // ------------------------------------------------------------ histogram path
            if (!hist_clean) {
                for (int i = lane; i < 2 * 3 * 256; i += 64) hflat[i] = 0;
                hist_clean = true;
            }
            if (lane < 6) acc[lane] = (lane % 3 == 2) ? ~0ull : 0ull;
            uint32_t c_road[2] = {0, 0}, c_dyn[2] = {0, 0};
            for (uint32_t r0 = o_p; r0 < o_e; r0 += 64) {
                const uint32_t r = r0 + lane;
                const bool act = r < o_e;
                if (r0 != o_p) load_rec<I64>(a, r, act, z, iv, rgbs);
                const uint32_t set = (act && r >= o_f) ? 1u : 0u;
                const unsigned sem = rgbs >> 24;
                const bool road = act && ((int)sem == q.road_class);
                const bool dynobj = act && ((q.dynobj_mask[sem >> 6] >> (sem & 63)) & 1ull);
                if (road) {
                    c_road[set]++;
                    const double sc = iv * FX_HI, fl = floor(sc);
                    atomicAdd(&acc[set * 3 + 0], (unsigned long long)(long long)fl);
                    atomicAdd(&acc[set * 3 + 1], (unsigned long long)(long long)rint((sc - fl) * FX_LO));
                }
                if (dynobj) c_dyn[set]++;
                if (act) atomicMin(&acc[set * 3 + 2], (unsigned long long)f64_order_key(z));
                // an all-equal wave (e.g. rgb == 0 with GT semantics) adds once instead of 64-way conflicting
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const unsigned val = (rgbs >> (8 * ch)) & 255u;
                    const unsigned key = act ? (unsigned)(set * 3 * 256 + ch * 256) + val : 0xffffffffu;
                    const unsigned first = __builtin_amdgcn_readfirstlane(key);
                    const uint64_t same = __ballot(key == first), actm = __ballot(act);
                    if (first != 0xffffffffu && same == actm) {
                        if (lane == 0) atomicAdd(&hflat[first], (uint32_t)__popcll(actm));
                    } else if (act) {
                        atomicAdd(&hflat[key], 1u);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint32_t v = c_road[s], w = c_dyn[s];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); w += __shfl_xor(w, o, 64); }
                nr[s] = v; nd[s] = w;
            }
            uint32_t full[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                uint4 *hp4 = reinterpret_cast<uint4 *>(&s_hist[wave][0][ch][4 * lane]);
                uint4 *hf4 = reinterpret_cast<uint4 *>(&s_hist[wave][1][ch][4 * lane]);
                const uint4 hp = *hp4, hf = *hf4;
                const uint4 hu = make_uint4(hp.x + hf.x, hp.y + hf.y, hp.z + hf.z, hp.w + hf.w);
                med2[0][ch] = hist_med2(hp, n_p);
                med2[1][ch] = hist_med2(hf, n_f);
                full[ch] = hist_med2(hu, n);
                *hp4 = make_uint4(0, 0, 0, 0);
                *hf4 = make_uint4(0, 0, 0, 0);
            }
            if (lane == 0) { s_full[wave][ci][0] = full[0]; s_full[wave][ci][1] = full[1]; s_full[wave][ci][2] = full[2]; }
            shi[0] = (long long)acc[0]; slo[0] = (long long)acc[1]; zm[0] = f64_from_order_key(acc[2]);
            shi[1] = (long long)acc[3]; slo[1] = (long long)acc[4]; zm[1] = f64_from_order_key(acc[5]);
        }
        if (lane < 2) {
            const int s = lane;
            CellStat st;
            st.n = s ? n_f : n_p;
            st.n_road = s ? nr[1] : nr[0];
            st.n_dyn = s ? nd[1] : nd[0];
            st.med2[0] = s ? med2[1][0] : med2[0][0];
            st.med2[1] = s ? med2[1][1] : med2[0][1];
            st.med2[2] = s ? med2[1][2] : med2[0][2];
            st.ihi = s ? shi[1] : shi[0];
            st.ilo = s ? slo[1] : slo[0];
            st.zmin = s ? zm[1] : zm[0];
            s_stat[s][lc] = st;
        }
    }
    __syncthreads();

    // ---- finalize: thread -> (set, cell); 'full' combines the two stored sets ----
    if (threadIdx.x < 3 * CELLS_PER_BLOCK) {
        const int s = threadIdx.x / CELLS_PER_BLOCK, lc = threadIdx.x % CELLS_PER_BLOCK;
        const CellStat p = s_stat[0][lc], f = s_stat[1][lc];
        uint32_t n, n_r, n_d, m2[3];
        long long ihi, ilo;
        double zmin;
        if (s < 2) {
            const CellStat &c = s ? f : p;
            n = c.n; n_r = c.n_road; n_d = c.n_dyn; ihi = c.ihi; ilo = c.ilo; zmin = c.zmin;
            m2[0] = c.med2[0]; m2[1] = c.med2[1]; m2[2] = c.med2[2];
        } else {
            n = p.n + f.n; n_r = p.n_road + f.n_road; n_d = p.n_dyn + f.n_dyn; ihi = p.ihi + f.ihi; ilo = p.ilo + f.ilo;
            zmin = (p.n && f.n) ? (p.zmin < f.zmin ? p.zmin : f.zmin) : (p.n ? p.zmin : f.zmin);
            const int w = lc / CPW, ci = lc % CPW;
            m2[0] = s_full[w][ci][0]; m2[1] = s_full[w][ci][1]; m2[2] = s_full[w][ci][2];
        }
        const double a_all = (double)n, a_r = (double)n_r, a_d = (double)n_d;
        const double road = (a_r + 1.0) / ((a_r + 1.0) + ((a_all - a_r) + 1.0));
        const double dynp = (a_d + 1.0) / ((a_d + 1.0) + ((a_all - a_d) + 1.0));
        const double isum = (double)ihi * FX_HI_INV + (double)ilo * FX_LO_INV;
        const double iraw = isum / (a_r + 1.0);
        const double zarg = q.int_sep_scaler * (iraw - q.int_mid_threshold);
        double inten = q.int_scaler * (1.0 / (1.0 + exp(-zarg)));
        if (inten > 1.0) inten = 1.0;
        s_out[7 * s + 0][lc] = road;
        s_out[7 * s + 1][lc] = inten;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            s_out[7 * s + 2 + ch][lc] = (n ? (double)m2[ch] / 2.0 : q.rgb_fill) / 255.0;
        s_out[7 * s + 5][lc] = dynp;
        s_out[7 * s + 6][lc] = n ? zmin : 0.0;
    }
    __syncthreads();
    // coalesced plane writes: 64 consecutive cells per plane
    for (int idx = threadIdx.x; idx < 21 * CELLS_PER_BLOCK; idx += BLK) {
        const int plane = idx / CELLS_PER_BLOCK, lc = idx % CELLS_PER_BLOCK;
        const int cell = cell0 + lc;
        if (cell >= ncell) continue;
        const double v = s_out[plane][lc];
        if (a.planes) a.planes[(int64_t)plane * ncell + cell] = v;
        if (a.planes_f16) a.planes_f16[(int64_t)plane * ncell + cell] = f64_to_f16_bits(v);
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static inline int64_t align256(int64_t v) { return (v + 255) & ~255ll; }

extern "C" {

int64_t pca_bev_workspace_bytes(int64_t max_points, int px)
{
    if (max_points < 1) max_points = 1;
    const int64_t n = 2ll * px * px;
    return align256(n * 4) + align256((n + 1) * 4) + 2 * align256(max_points * 4) + align256(max_points * 24) + 256;
}

int pca_bev_generate(pca_ctx *ctx, const pca_store *store, const double *intensity64, const int64_t *frame_off,
                     int slot_begin, int slot_split, int slot_end, int64_t max_points, const pca_bev_params *prm,
                     void *workspace, int64_t workspace_bytes, double *planes, uint16_t *planes_f16, void *stream)
{
    if (!ctx) return -1;
    if (!store || !frame_off || !prm || !workspace || (!planes && !planes_f16)) { ctx->err = "bev: bad arguments"; return -1; }
    if (prm->px < 1 || prm->px > 4096) { ctx->err = "bev: px out of range"; return -1; }
    if (!(slot_begin <= slot_split && slot_split <= slot_end)) { ctx->err = "bev: need slot_begin <= slot_split <= slot_end"; return -1; }
    if (max_points < 1) max_points = 1;
    if (max_points >= (1ll << 32)) { ctx->err = "bev: window too large for 32-bit ranks"; return -1; }
    if (workspace_bytes < pca_bev_workspace_bytes(max_points, prm->px)) { ctx->err = "bev: workspace too small"; return -1; }
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    const int64_t n = 2ll * prm->px * prm->px;
    BevArgs a;
    a.st = *store;
    a.intensity64 = intensity64;
    a.frame_off = frame_off;
    a.slot_begin = slot_begin; a.slot_split = slot_split; a.slot_end = slot_end;
    a.max_points = max_points;
    a.prm = *prm;
    char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    a.cnt = reinterpret_cast<uint32_t *>(w); w += align256(n * 4);
    a.offs = reinterpret_cast<uint32_t *>(w); w += align256((n + 1) * 4);
    a.key = reinterpret_cast<uint32_t *>(w); w += align256(max_points * 4);
    a.rank = reinterpret_cast<uint32_t *>(w); w += align256(max_points * 4);
    a.recs = w;
    a.planes = planes;
    a.planes_f16 = planes_f16;
    a.scan_tiles = (int)((n + 1023) / 1024);
    if (pca_ctx_reserve_tiles(ctx, a.scan_tiles, s)) return -1;
    a.state = ctx->tile_state;
    a.ticket = ctx->ticket;
    a.epoch = pca_ctx_next_epoch(ctx, s);
    PCA_CHECK(ctx, hipMemsetAsync(a.cnt, 0, n * 4, s));
    const int64_t want = (max_points + BLK - 1) / BLK;
    const int grid = (int)(want < 2048 ? want : 2048);
    PCA_LAUNCH(ctx, PCA_K_BEV_BIN, bev_bin, dim3(grid), dim3(BLK), s, a);
    PCA_LAUNCH(ctx, PCA_K_BEV_SCAN, bev_scan, dim3(a.scan_tiles), dim3(BLK), s, a);
    const int cgrid = (prm->px * prm->px + CELLS_PER_BLOCK - 1) / CELLS_PER_BLOCK;
    if (intensity64) {
        PCA_LAUNCH(ctx, PCA_K_BEV_SCATTER, bev_scatter<true>, dim3(grid), dim3(BLK), s, a);
        PCA_LAUNCH(ctx, PCA_K_BEV_CELLS, bev_cells<true>, dim3(cgrid), dim3(BLK), s, a);
    } else {
        PCA_LAUNCH(ctx, PCA_K_BEV_SCATTER, bev_scatter<false>, dim3(grid), dim3(BLK), s, a);
        PCA_LAUNCH(ctx, PCA_K_BEV_CELLS, bev_cells<false>, dim3(cgrid), dim3(BLK), s, a);
    }
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
