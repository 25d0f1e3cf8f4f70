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
    for (int64_t p = lo + (int64_t)blockIdx.x * BLK + threadIdx.x; p < hi; p += (int64_t)gridDim.x * BLK) {
        const double x = a.st.x[p] - q.origin[0];
        const double y = a.st.y[p] - q.origin[1];
        const double z = a.st.z[p] - q.origin[2];
        double ax = q.R[0] * x; ax = fma(q.R[1], y, ax); ax = fma(q.R[2], z, ax);
        double ay = q.R[3] * x; ay = fma(q.R[4], y, ay); ay = fma(q.R[5], z, ay);
        double az = q.R[6] * x; az = fma(q.R[7], y, az); az = fma(q.R[8], z, az);
        ax += q.dx;
        ay += q.dy;
        bool keep = (ax > vlo) && (ax < vhi) && (ay > vlo) && (ay < vhi);
        if (use_h) keep = keep && (az < q.height_filter);
        keep = keep && (a.st.dyn[p] != 1);
        uint32_t key = KEY_INVALID, rk = 0;
        if (keep) {
            int i = (int)floor(ax / v * pxd + half_px);
            int j = (int)floor(ay / v * pxd + half_px);
            i = i > q.px - 1 ? q.px - 1 : (i < 0 ? 0 : i);
            j = j > q.px - 1 ? q.px - 1 : (j < 0 ? 0 : j);
            const uint32_t cell = (uint32_t)((q.px - 1 - j) * q.px + i);
            key = cell * 2u + (p >= sp ? 1u : 0u);
            rk = atomicAdd(&a.cnt[key], 1u);
        }
        a.key[p - lo] = key;
        a.rank[p - lo] = rk;
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
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o, 64); v = t < v ? t : v; }
    return v;
}

// value (0..255) holding 0-based rank k of a 256-bin histogram spread 4 bins per lane
__device__ __forceinline__ int hist_rank(const uint4 h, uint32_t excl, uint32_t s, uint32_t k)
{
    const int lane = threadIdx.x & 63;
    const bool own = (k >= excl) && (k < excl + s);
    int val = 0;
    if (own) {
        const uint32_t r = k - excl;
        val = 4 * lane + (r < h.x ? 0 : (r < h.x + h.y ? 1 : (r < h.x + h.y + h.z ? 2 : 3)));
    }
    const uint64_t m = __ballot(own);
    const int src = m ? (int)__ffsll((unsigned long long)m) - 1 : 0;
    return __shfl(val, src, 64);
}

// median of the histogram h (4 bins per lane) with n entries; np.median semantics
__device__ __forceinline__ double hist_median(const uint4 h, uint32_t n, double fill)
{
    if (n == 0) return fill;
    const int lane = threadIdx.x & 63;
    const uint32_t s = h.x + h.y + h.z + h.w;
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    const uint32_t excl = inc - s;
    const int lo = hist_rank(h, excl, s, (n - 1) >> 1);
    const int hi = hist_rank(h, excl, s, n >> 1);
    return ((double)lo + (double)hi) / 2.0;
}

template <bool I64>
__global__ __launch_bounds__(BLK) void bev_cells(const BevArgs a)
{
    __shared__ uint32_t s_hist[BLK / PCA_WAVE][2][3][256];      // 24 KB
    __shared__ double s_out[21][CELLS_PER_BLOCK];               // 10.5 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const pca_bev_params &q = a.prm;
    const int ncell = q.px * q.px;
    const int cell0 = blockIdx.x * CELLS_PER_BLOCK;
    uint32_t(*hist)[3][256] = s_hist[wave];
    uint32_t *hflat = &hist[0][0][0];

    // clear this wave's histograms once; they are restored to zero after every non-empty cell
    for (int i = lane; i < 2 * 3 * 256; i += 64) hflat[i] = 0;

    constexpr int CPW = CELLS_PER_BLOCK / (BLK / PCA_WAVE);     // cells per wave
    for (int ci = 0; ci < CPW; ++ci) {
        const int lc = wave * CPW + ci;
        const int cell = cell0 + lc;
        if (cell >= ncell) break;
        const uint32_t o_p = a.offs[2 * cell], o_f = a.offs[2 * cell + 1], o_e = a.offs[2 * cell + 2];
        const uint32_t n_set[2] = {o_f - o_p, o_e - o_f};
        uint32_t c_road[2] = {0, 0}, c_dyn[2] = {0, 0};
        long long ihi[2] = {0, 0}, ilo[2] = {0, 0};
        double zmin[2] = {__builtin_huge_val(), __builtin_huge_val()};

        for (uint32_t r0 = o_p; r0 < o_e; r0 += 64) {
            const uint32_t r = r0 + lane;
            const bool act = r < o_e;
            double z = 0, iv = 0;
            uint32_t rgbs = 0;
            if (act) {
                if (I64) { const Rec24 rec = reinterpret_cast<const Rec24 *>(a.recs)[r]; z = rec.z; iv = rec.inten; rgbs = rec.rgbs; }
                else {
                    const Rec16 rec = reinterpret_cast<const Rec16 *>(a.recs)[r];
                    z = rec.z; rgbs = rec.rgbs;
                    iv = q.intensity_div255 ? (double)rec.inten / 255.0 : (double)rec.inten;
                }
            }
            const int set = (act && r >= o_f) ? 1 : 0;
            const unsigned sem = rgbs >> 24;
            const bool road = act && ((int)sem == q.road_class);
            const bool dynobj = act && ((q.dynobj_mask[sem >> 6] >> (sem & 63)) & 1ull);
            if (act) {
                if (road) {
                    c_road[set]++;
                    const double sc = iv * FX_HI, fl = floor(sc);
                    ihi[set] += (long long)fl;
                    ilo[set] += (long long)rint((sc - fl) * FX_LO);
                }
                if (dynobj) c_dyn[set]++;
                zmin[set] = z < zmin[set] ? z : zmin[set];
            }
            // histograms: LDS atomics; an all-equal wave (e.g. rgb == 0 with GT semantics) adds once
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const unsigned val = (rgbs >> (8 * ch)) & 255u;
                const unsigned key = act ? (unsigned)(set * 3 * 256 + ch * 256) + val : 0xffffffffu;
                const unsigned first = __builtin_amdgcn_readfirstlane(key);
                const uint64_t same = __ballot(key == first);
                const uint64_t actm = __ballot(act);
                if (first != 0xffffffffu && same == actm) {
                    if (lane == 0) atomicAdd(&hflat[first], (uint32_t)__popcll(actm));
                } else if (act) {
                    atomicAdd(&hflat[key], 1u);
                }
            }
        }
        // wave reductions of the per-lane partials
        uint32_t nr[3], nd[3], na[3];
        long long ish[3], isl[3];
        double zm[3];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            nr[s] = wave_sum_u32(c_road[s]);
            nd[s] = wave_sum_u32(c_dyn[s]);
            ish[s] = wave_sum_i64(ihi[s]);
            isl[s] = wave_sum_i64(ilo[s]);
            zm[s] = wave_min_f64(zmin[s]);
            na[s] = n_set[s];
        }
        nr[2] = nr[0] + nr[1]; nd[2] = nd[0] + nd[1]; na[2] = na[0] + na[1];
        ish[2] = ish[0] + ish[1]; isl[2] = isl[0] + isl[1];
        zm[2] = zm[0] < zm[1] ? zm[0] : zm[1];

        // medians (every lane participates), then lane 0 finishes the closed-form maps
        double med[3][3];
        const bool any = (o_e > o_p);
        uint4 hp[3], hf[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            if (any) {
                hp[ch] = *reinterpret_cast<const uint4 *>(&hist[0][ch][4 * lane]);
                hf[ch] = *reinterpret_cast<const uint4 *>(&hist[1][ch][4 * lane]);
            } else {
                hp[ch] = make_uint4(0, 0, 0, 0);
                hf[ch] = make_uint4(0, 0, 0, 0);
            }
            const uint4 hu = make_uint4(hp[ch].x + hf[ch].x, hp[ch].y + hf[ch].y, hp[ch].z + hf[ch].z, hp[ch].w + hf[ch].w);
            med[0][ch] = hist_median(hp[ch], na[0], q.rgb_fill);
            med[1][ch] = hist_median(hf[ch], na[1], q.rgb_fill);
            med[2][ch] = hist_median(hu, na[2], q.rgb_fill);
        }
        if (any) {   // restore zeros
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                *reinterpret_cast<uint4 *>(&hist[0][ch][4 * lane]) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4 *>(&hist[1][ch][4 * lane]) = make_uint4(0, 0, 0, 0);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const double a_all = (double)na[s], a_r = (double)nr[s], a_d = (double)nd[s];
                const double road = (a_r + 1.0) / ((a_r + 1.0) + ((a_all - a_r) + 1.0));
                const double dynp = (a_d + 1.0) / ((a_d + 1.0) + ((a_all - a_d) + 1.0));
                const double isum = (double)ish[s] * FX_HI_INV + (double)isl[s] * FX_LO_INV;
                const double iraw = isum / (a_r + 1.0);
                const double zarg = q.int_sep_scaler * (iraw - q.int_mid_threshold);
                double inten = q.int_scaler * (1.0 / (1.0 + exp(-zarg)));
                if (inten > 1.0) inten = 1.0;
                s_out[7 * s + 0][lc] = road;
                s_out[7 * s + 1][lc] = inten;
                s_out[7 * s + 2][lc] = med[s][0] / 255.0;
                s_out[7 * s + 3][lc] = med[s][1] / 255.0;
                s_out[7 * s + 4][lc] = med[s][2] / 255.0;
                s_out[7 * s + 5][lc] = dynp;
                s_out[7 * s + 6][lc] = na[s] ? zm[s] : 0.0;
            }
        }
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
