// pca_bev.hip -- BEV rasteriser for gfx950: two-level counting sort with LDS-staged atomics, no global atomics.
//
//   level 1 (tiles of 8x8 cells, T tiles)
//     bev_tile_hist     every workgroup bins a contiguous chunk of the window: rotate / translate / crop /
//                       height / floor -> key = tile<<7 | cell_in_tile<<1 | set; per-workgroup LDS histogram
//                       over tiles -> bh[workgroup][tile]           (reads 25 B/pt, writes 4 B/pt)
//     bev_tile_scan     exclusive scan of bh in (tile, workgroup) order (decoupled look-back)
//     bev_tile_scatter  records {z, intensity, rgb|flags, fine key} -> tile-ordered SoA streams; the position
//                       comes from an LDS cursor per tile (returning LDS atomics)
//   level 2 (one workgroup per tile)
//     bev_tile_cells    pass 1: per (cell,set) counts / exact integer intensity sums / min z by LDS atomics;
//                       pass 2: LDS counting sort of the colours by (cell,set); per-cell exact medians
//                       (n <= 64: bit-sliced radix select, one lane per target; else 256-bin LDS histogram);
//                       closed-form maps, fp16, tile written back.
// 'full' = present (+) future is formed per cell (counts add, min of mins, median of the union).
#include "pca_bev_common.h"
#include <cstdlib>

#define KEY_INVALID 0xffffffffu
#define TS 8                      // tile side [cells]
#define TCELLS (TS * TS)
#define NFK (2 * TCELLS)          // fine keys per tile: cell_in_tile*2 + set
#define AB_THREADS 1024           // workgroup size of the hist / scatter kernels
#define MAX_G 512                 // workgroups of the hist / scatter kernels
#define C_THREADS 256             // workgroup size of the tile kernel
#define RGB_CAP 4096              // colour records resident in LDS per batch of cells
#define FLAG_ROAD (1u << 24)
#define FLAG_DYNOBJ (1u << 25)

// tile-ordered records: one packed stream (5 or 6 dwords) so that a (workgroup, tile) run is one contiguous write
struct __attribute__((packed, aligned(4))) RecF { double z; float inten; uint32_t c; uint32_t fk; };   // 20 B
struct RecD { double z; double inten; uint32_t c; uint32_t fk; };                                      // 24 B

struct BevArgs {
    pca_store st;
    const double *intensity64;
    const int64_t *frame_off;
    int slot_begin, slot_split, slot_end;
    int64_t max_points;
    pca_bev_params prm;
    Mat34 pend_T;         // owed re-transform of slots [slot_begin, pend_slot_end), fused into the hist kernel
    int pend_slot_end;    // <= slot_begin: none
    int tx, T, G;
    uint32_t *key;        // [max_points]
    uint32_t *bh;         // [T][G] kept records per (tile, workgroup)
    uint32_t *boff;       // [T][G] exclusive scan of bh in that order
    uint32_t *tile_off;   // [T+1]
    void *recs;           // RecF / RecD [max_points], tile-ordered; c = r | g<<8 | b<<16 | FLAG_*
    double *planes;
    uint16_t *planes_f16;
    uint64_t *state;
    uint32_t *ticket;
    uint32_t epoch;
    int scan_tiles;
};

struct Window { int64_t lo, hi, sp, c_lo, c_hi; };

__device__ __forceinline__ Window chunk_of(const BevArgs &a)
{
    Window w;
    w.lo = a.frame_off[a.slot_begin];
    const int64_t hi0 = a.frame_off[a.slot_end];
    w.sp = a.frame_off[a.slot_split];
    w.hi = (hi0 - w.lo > a.max_points) ? w.lo + a.max_points : hi0;
    const int64_t chunk = (w.hi - w.lo + a.G - 1) / a.G;
    w.c_lo = w.lo + (int64_t)blockIdx.x * chunk;
    w.c_hi = w.c_lo + chunk < w.hi ? w.c_lo + chunk : w.hi;
    return w;
}

// ---------------------------------------------------------------------------------------------
// level 1a: keys + per-workgroup tile histogram
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AB_THREADS) void bev_tile_hist(const BevArgs a)
{
    extern __shared__ uint32_t s_h[];                       // [T]
    const Window w = chunk_of(a);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.frame_off[a.slot_end] - w.lo > a.max_points)
        atomicOr(a.ticket + 1, PCA_STATUS_STORE_OVERFLOW);
    for (int t = threadIdx.x; t < a.T; t += AB_THREADS) s_h[t] = 0;
    __syncthreads();
    const pca_bev_params &q = a.prm;
    const double v = q.view, vlo = -0.5 * v, vhi = 0.5 * v, pxd = (double)q.px, half_px = 0.5 * pxd;
    const bool use_h = !(q.height_filter != q.height_filter);
    const int64_t pend_hi = a.pend_slot_end > a.slot_begin ? a.frame_off[a.pend_slot_end] : w.lo;
    constexpr int UNR = 4;          // independent points per thread and iteration (memory-level parallelism)
    for (int64_t base = w.c_lo + threadIdx.x; base < w.c_hi; base += UNR * AB_THREADS) {
        double X[UNR], Y[UNR], Z[UNR];
        uint8_t D[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t p = base + u * AB_THREADS;
            const bool in = p < w.c_hi;
            X[u] = in ? a.st.x[p] : 0.0;
            Y[u] = in ? a.st.y[p] : 0.0;
            Z[u] = in ? a.st.z[p] : 0.0;
            D[u] = in ? a.st.dyn[p] : (uint8_t)1;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {                     // owed re-transform: apply and write back
            const int64_t p = base + u * AB_THREADS;
            if (p < w.c_hi && p < pend_hi) {
                const double nx = row4(a.pend_T.m + 0, X[u], Y[u], Z[u]), ny = row4(a.pend_T.m + 4, X[u], Y[u], Z[u]),
                             nz = row4(a.pend_T.m + 8, X[u], Y[u], Z[u]);
                X[u] = nx; Y[u] = ny; Z[u] = nz;
                a.st.x[p] = nx; a.st.y[p] = ny; a.st.z[p] = nz;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t p = base + u * AB_THREADS;
            if (p >= w.c_hi) continue;
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
            keep = keep && (D[u] != 1);
            uint32_t key = KEY_INVALID;
            if (keep) {
                int i = (int)floor(ax / v * pxd + half_px);
                int j = (int)floor(ay / v * pxd + half_px);
                i = i > q.px - 1 ? q.px - 1 : (i < 0 ? 0 : i);
                j = j > q.px - 1 ? q.px - 1 : (j < 0 ? 0 : j);
                const int row = q.px - 1 - j, col = i;
                const uint32_t tile = (uint32_t)((row / TS) * a.tx + (col / TS));
                const uint32_t fk = (uint32_t)(((row % TS) * TS + (col % TS)) * 2) + (p >= w.sp ? 1u : 0u);
                key = (tile << 7) | fk;
                atomicAdd(&s_h[tile], 1u);
            }
            a.key[p - w.lo] = key;
        }
    }
    __syncthreads();
    // tile-major layout [tile][workgroup]: the scan then streams, the strided accesses ride along here and in the scatter
    for (int t = threadIdx.x; t < a.T; t += AB_THREADS) a.bh[(int64_t)t * a.G + blockIdx.x] = s_h[t];
}

// ---------------------------------------------------------------------------------------------
// level 1b: exclusive scan of bh in (tile-major, workgroup-minor) order -> boff, tile_off
// ---------------------------------------------------------------------------------------------
#define SCAN_THREADS 1024
#define SCAN_TILE (4 * SCAN_THREADS)       // few large scan tiles: the look-back then spans at most one window
__global__ __launch_bounds__(SCAN_THREADS) void bev_tile_scan(const BevArgs a)
{
    __shared__ int s_tile;
    __shared__ uint32_t s_w[SCAN_THREADS / 64];
    __shared__ uint64_t s_excl;
    const int n = a.T * a.G;
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(a.ticket, 1u);
        if ((int)t == a.scan_tiles - 1) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tile = (int)t;
    }
    __syncthreads();
    const int tile = s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = tile * SCAN_TILE + threadIdx.x * 4;    // bh and boff are both [tile][workgroup]: contiguous
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = (base + k < n) ? a.bh[base + k] : 0u;
    const uint32_t tsum = c[0] + c[1] + c[2] + c[3];
    uint32_t inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const uint32_t v = lane < SCAN_THREADS / 64 ? s_w[lane] : 0u;
        uint32_t winc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(winc, o, 64);
            if (lane >= o) winc += t;
        }
        if (lane < SCAN_THREADS / 64) s_w[lane] = winc - v;                // exclusive wave offsets
        const uint32_t total = __shfl(winc, 63, 64);
        const uint64_t e = lb_exclusive_prefix(a.state, tile, (uint64_t)total, a.epoch);
        if (lane == 0) s_excl = (e << 32) | total;                          // both fit 32 bits (checked by the host)
    }
    __syncthreads();
    const uint32_t excl = (uint32_t)(s_excl >> 32), total = (uint32_t)s_excl;
    uint32_t run = excl + s_w[wave] + (inc - tsum);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = base + k;
        if (i < n) {
            a.boff[i] = run;
            if (i % a.G == 0) a.tile_off[i / a.G] = run;
        }
        run += c[k];
    }
    if (tile == a.scan_tiles - 1 && threadIdx.x == SCAN_THREADS - 1) a.tile_off[a.T] = excl + total;
}

// ---------------------------------------------------------------------------------------------
// level 1c: scatter records into tile order
// ---------------------------------------------------------------------------------------------
template <bool I64>
__global__ __launch_bounds__(AB_THREADS) void bev_tile_scatter(const BevArgs a)
{
    extern __shared__ uint32_t s_cur[];                     // [T]
    const Window w = chunk_of(a);
    for (int t = threadIdx.x; t < a.T; t += AB_THREADS) s_cur[t] = a.boff[(int64_t)t * a.G + blockIdx.x];
    __syncthreads();
    const pca_bev_params &q = a.prm;
    const double oz = q.origin[2];
    constexpr int UNR = 4;
    for (int64_t base = w.c_lo + threadIdx.x; base < w.c_hi; base += UNR * AB_THREADS) {
        uint32_t key[UNR], pos[UNR], rgbs[UNR];
        double zz[UNR], iv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t p = base + u * AB_THREADS;
            key[u] = p < w.c_hi ? a.key[p - w.lo] : KEY_INVALID;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t p = base + u * AB_THREADS;
            const bool ok = key[u] != KEY_INVALID;
            rgbs[u] = ok ? a.st.rgbs[p] : 0u;
            zz[u] = ok ? a.st.z[p] : 0.0;
            if (I64) iv[u] = ok ? a.intensity64[p] : 0.0;
            else iv[u] = ok ? (double)a.st.intensity[p] : 0.0;
            pos[u] = ok ? atomicAdd(&s_cur[key[u] >> 7], 1u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (key[u] == KEY_INVALID) continue;
            const unsigned sem = rgbs[u] >> 24;
            uint32_t c = rgbs[u] & 0xffffffu;
            if ((int)sem == q.road_class) c |= FLAG_ROAD;
            if ((q.dynobj_mask[sem >> 6] >> (sem & 63)) & 1ull) c |= FLAG_DYNOBJ;
            const double z = zz[u] - oz;                    // rotation about z: row 3 of R is (0,0,1)
            if (I64) {
                RecD r; r.z = z; r.inten = iv[u]; r.c = c; r.fk = key[u] & 127u;
                reinterpret_cast<RecD *>(a.recs)[pos[u]] = r;
            } else {
                RecF r; r.z = z; r.inten = (float)iv[u]; r.c = c; r.fk = key[u] & 127u;
                reinterpret_cast<RecF *>(a.recs)[pos[u]] = r;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// level 2: one workgroup per tile
// ---------------------------------------------------------------------------------------------
struct TileLds {
    uint32_t cnt[NFK], road[NFK], dyn[NFK], off[NFK + 1], cur[NFK];
    unsigned long long ihi[NFK], ilo[NFK], zk[NFK];
    uint32_t med2[3][TCELLS][3];                            // [present, future, full][cell][channel]
    uint32_t hist[3][256];                                  // cells with more than 64 values
    unsigned long long bits[TCELLS][24];                    // cells with <= 64 values: one 64-lane mask per colour bit
};

// ---- medians of cells with at most 64 values: bit-sliced radix select -------------------------------------
// Phase 1 (per cell, whole wave): the cell's values sit one per lane (present lanes first); 24 ballots give
// the 64-bit membership mask of every colour bit, parked in LDS.
// Phase 2 (per wave): one LANE per (cell, set, channel, lower|upper middle) target -- 18 per cell -- walks the 8
// bit planes from the top: zeros = cand & ~plane; rank < popcount(zeros) ? keep zeros : (rank -= ..., keep ones,
// set the bit).  That is ~12 VALU per plane for 64 targets at once instead of a 21-stage sort per channel.
__device__ __forceinline__ void small_cells_bitplanes(TileLds &L, const uint32_t *s_rgb, uint32_t base, int c_begin, int c_end)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = 0; i < TCELLS / 4; ++i) {
        const int cell = 4 * i + wave;
        if (cell < c_begin || cell >= c_end) continue;
        const uint32_t n = L.cnt[2 * cell] + L.cnt[2 * cell + 1];
        if (n == 0 || n > 64) continue;
        const uint32_t v = (uint32_t)lane < n ? s_rgb[L.off[2 * cell] - base + lane] : 0u;
        unsigned long long mine = 0;
#pragma unroll
        for (int b = 0; b < 24; ++b) {
            const unsigned long long m = __ballot((v >> b) & 1u);
            mine = (lane == b) ? m : mine;
        }
        if (lane < 24) L.bits[cell][lane] = mine;
    }
}

__device__ __forceinline__ void small_cells_select(TileLds &L, int c_begin, int c_end)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NT = (TCELLS / 4) * 18;                   // targets of one wave
    for (int t0 = 0; t0 < NT; t0 += 64) {
        const int t = t0 + lane;
        const int i = t / 18, qq = t - 18 * i;
        const int cell = 4 * i + wave;
        const int set = qq / 6, ch = (qq % 6) >> 1, upper = qq & 1;
        bool ok = t < NT && cell >= c_begin && cell < c_end;
        uint32_t n_p = 0, n_f = 0;
        if (ok) { n_p = L.cnt[2 * cell]; n_f = L.cnt[2 * cell + 1]; }
        const uint32_t n = n_p + n_f;
        ok = ok && n > 0 && n <= 64;
        const unsigned long long m_p = n_p >= 64 ? ~0ull : ((1ull << n_p) - 1ull);
        const unsigned long long m_all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
        unsigned long long cand = set == 0 ? m_p : (set == 1 ? (m_all & ~m_p) : m_all);
        const uint32_t n_s = (uint32_t)__popcll(cand);
        ok = ok && n_s > 0;
        uint32_t k = upper ? (n_s >> 1) : ((n_s - 1) >> 1);
        uint32_t val = 0;
        if (ok) {
            const unsigned long long *planes = &L.bits[cell][8 * ch];
#pragma unroll
            for (int b = 7; b >= 0; --b) {
                const unsigned long long B = planes[b];
                const unsigned long long zeros = cand & ~B;
                const uint32_t cz = (uint32_t)__popcll(zeros);
                const bool take0 = k < cz;
                cand = take0 ? zeros : (cand & B);
                k = take0 ? k : k - cz;
                val |= take0 ? 0u : (1u << b);
            }
        }
        const uint32_t other = __shfl_xor(val, 1, 64);      // the partner target (lower <-> upper middle) is the adjacent lane
        if (ok && !upper) L.med2[set][cell][ch] = val + other;
    }
}

// adds the colours of one (cell,set) to the block histogram; values come from the LDS-sorted batch or,
// for a cell too large for LDS, straight from the tile's record streams
template <bool I64> __device__ __forceinline__ constexpr int REC_C() { return I64 ? 4 : 3; }
template <bool I64> __device__ __forceinline__ constexpr int REC_FK() { return I64 ? 5 : 4; }
template <bool I64>
__device__ __forceinline__ const uint32_t *rec_words(const BevArgs &a, uint32_t r)
{
    return reinterpret_cast<const uint32_t *>(a.recs) + (size_t)r * (I64 ? 6 : 5);
}

template <bool I64>
__device__ __forceinline__ void hist_add(TileLds &L, const BevArgs &a, const uint32_t *s_rgb, bool from_lds, uint32_t fk,
                                         uint32_t lds_base, uint32_t r_lo, uint32_t r_hi)
{
    const int lane = threadIdx.x & 63;
    const uint32_t n_iter = from_lds ? L.cnt[fk] : (r_hi - r_lo);
    for (uint32_t i0 = 0; i0 < n_iter; i0 += C_THREADS) {
        const uint32_t i = i0 + threadIdx.x;
        bool act = i < n_iter;
        uint32_t v = 0;
        if (act) {
            if (from_lds) v = s_rgb[lds_base + i];
            else {
                const uint32_t *w = rec_words<I64>(a, r_lo + i);
                act = w[REC_FK<I64>()] == fk;
                if (act) v = w[REC_C<I64>()];
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const unsigned val = (v >> (8 * ch)) & 255u;
            const unsigned key = act ? val : 0xffffffffu;
            // an all-equal wave (e.g. rgb == 0 with GT semantics) adds once instead of conflicting 64 ways
            const unsigned first = __builtin_amdgcn_readfirstlane(key);
            const uint64_t same = __ballot(key == first), actm = __ballot(act);
            if (actm && first != 0xffffffffu && same == actm) {
                if (lane == (int)__ffsll((unsigned long long)actm) - 1) atomicAdd(&L.hist[ch][first], (uint32_t)__popcll(actm));
            } else if (act) {
                atomicAdd(&L.hist[ch][val], 1u);
            }
        }
    }
}

__device__ __forceinline__ void hist_zero(TileLds &L)
{
    for (int i = threadIdx.x; i < 3 * 256; i += C_THREADS) (&L.hist[0][0])[i] = 0;
}

// medians of one cell with more than 64 values, whole workgroup
template <bool I64>
__device__ __forceinline__ void big_cell_medians(TileLds &L, const BevArgs &a, const uint32_t *s_rgb, bool from_lds, int cell,
                                                 uint32_t batch_base, uint32_t r_lo, uint32_t r_hi)
{
    const uint32_t n_p = L.cnt[2 * cell], n_f = L.cnt[2 * cell + 1];
    const uint32_t base_p = L.off[2 * cell] - batch_base, base_f = L.off[2 * cell + 1] - batch_base;
    const int wave = threadIdx.x >> 6;
    hist_zero(L);
    __syncthreads();
    hist_add<I64>(L, a, s_rgb, from_lds, 2 * cell, base_p, r_lo, r_hi);
    __syncthreads();
    if (wave < 3) { const uint32_t m = hist_med2(L.hist[wave], n_p); if ((threadIdx.x & 63) == 0) L.med2[0][cell][wave] = m; }
    __syncthreads();
    hist_add<I64>(L, a, s_rgb, from_lds, 2 * cell + 1, base_f, r_lo, r_hi);
    __syncthreads();
    if (wave < 3) { const uint32_t m = hist_med2(L.hist[wave], n_p + n_f); if ((threadIdx.x & 63) == 0) L.med2[2][cell][wave] = m; }
    __syncthreads();
    hist_zero(L);
    __syncthreads();
    hist_add<I64>(L, a, s_rgb, from_lds, 2 * cell + 1, base_f, r_lo, r_hi);
    __syncthreads();
    if (wave < 3) { const uint32_t m = hist_med2(L.hist[wave], n_f); if ((threadIdx.x & 63) == 0) L.med2[1][cell][wave] = m; }
    __syncthreads();
}

template <bool I64>
__global__ __launch_bounds__(C_THREADS) void bev_tile_cells(const BevArgs a)
{
    __shared__ TileLds L;
    __shared__ __align__(16) unsigned char s_buf[RGB_CAP * 4 > 21 * TCELLS * 8 ? RGB_CAP * 4 : 21 * TCELLS * 8];
    uint32_t *s_rgb = reinterpret_cast<uint32_t *>(s_buf);
    double(*s_out)[TCELLS] = reinterpret_cast<double(*)[TCELLS]>(s_buf);       // reused after the medians
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const pca_bev_params &q = a.prm;
    const int tile = blockIdx.x;
    const uint32_t r_lo = a.tile_off[tile], r_hi = a.tile_off[tile + 1];

    for (int k = threadIdx.x; k < NFK; k += C_THREADS) {
        L.cnt[k] = 0; L.road[k] = 0; L.dyn[k] = 0; L.ihi[k] = 0; L.ilo[k] = 0; L.zk[k] = ~0ull;
    }
    for (int k = threadIdx.x; k < 3 * TCELLS * 3; k += C_THREADS) (&L.med2[0][0][0])[k] = 0;
    __syncthreads();

    // ---- pass 1: per (cell,set) statistics with LDS atomics ----
    // If the tile's records fit the LDS colour buffer (the normal case) every thread keeps its <= RPT records'
    // (key, rank, colour) in registers -- the rank comes back from the counting atomic -- and pass 2 is a pure
    // LDS scatter without touching memory again.
    constexpr int RPT = RGB_CAP / C_THREADS;
    const bool fits = (r_hi - r_lo) <= RGB_CAP;
    uint32_t kr[RPT], cc[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) { kr[u] = 0xffffffffu; cc[u] = 0; }
    auto account = [&](uint32_t k, uint32_t c, double z, double iv, bool want_rank) -> uint32_t {
        uint32_t rank = 0;
        if (want_rank) rank = atomicAdd(&L.cnt[k], 1u); else atomicAdd(&L.cnt[k], 1u);
        atomicMin(&L.zk[k], (unsigned long long)f64_order_key(z));
        if (c & FLAG_DYNOBJ) atomicAdd(&L.dyn[k], 1u);
        if (c & FLAG_ROAD) {
            const double sc = iv * FX_HI, fl = floor(sc);
            atomicAdd(&L.road[k], 1u);
            atomicAdd(&L.ihi[k], (unsigned long long)(long long)fl);
            atomicAdd(&L.ilo[k], (unsigned long long)(long long)rint((sc - fl) * FX_LO));
        }
        return rank;
    };
    auto load = [&](uint32_t r, uint32_t &k, uint32_t &c, double &z, double &iv) {
        if (I64) {
            const RecD rec = reinterpret_cast<const RecD *>(a.recs)[r];
            k = rec.fk; c = rec.c; z = rec.z; iv = rec.inten;
        } else {
            const RecF rec = reinterpret_cast<const RecF *>(a.recs)[r];
            k = rec.fk; c = rec.c; z = rec.z;
            iv = q.intensity_div255 ? (double)rec.inten / 255.0 : (double)rec.inten;
        }
    };
    if (fits) {
        constexpr int HALF = RPT / 2;                       // two rounds of loads: bounds the registers in flight
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t k[HALF], c[HALF];
            double z[HALF], iv[HALF];
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                const uint32_t r = r_lo + (h * HALF + u) * C_THREADS + threadIdx.x;
                k[u] = 0xffffffffu; c[u] = 0; z[u] = 0; iv[u] = 0;
                if (r < r_hi) load(r, k[u], c[u], z[u], iv[u]);
            }
#pragma unroll
            for (int u = 0; u < HALF; ++u) {
                if (k[u] == 0xffffffffu) continue;
                const uint32_t rank = account(k[u], c[u], z[u], iv[u], true);
                kr[h * HALF + u] = k[u] | (rank << 8);
                cc[h * HALF + u] = c[u] & 0xffffffu;
            }
        }
    } else {
        for (uint32_t r = r_lo + threadIdx.x; r < r_hi; r += C_THREADS) {
            uint32_t k, c;
            double z, iv;
            load(r, k, c, z, iv);
            account(k, c, z, iv, false);
        }
    }
    __syncthreads();
    // ---- offsets of the (cell,set) segments inside the tile ----
    if (wave == 0) {
        const uint32_t c0 = L.cnt[2 * lane], c1 = L.cnt[2 * lane + 1];
        uint32_t inc = c0 + c1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        const uint32_t excl = inc - (c0 + c1);
        L.off[2 * lane] = excl;
        L.off[2 * lane + 1] = excl + c0;
        if (lane == 63) L.off[NFK] = inc;
    }
    __syncthreads();

    // ---- medians ----
    if (fits) {
#pragma unroll
        for (int u = 0; u < RPT; ++u)
            if (kr[u] != 0xffffffffu) s_rgb[L.off[kr[u] & 127u] + (kr[u] >> 8)] = cc[u];
        __syncthreads();
        small_cells_bitplanes(L, s_rgb, 0, 0, TCELLS);
        __syncthreads();
        small_cells_select(L, 0, TCELLS);
        for (int cell = 0; cell < TCELLS; ++cell)
            if (L.cnt[2 * cell] + L.cnt[2 * cell + 1] > 64) big_cell_medians<I64>(L, a, s_rgb, true, cell, 0, r_lo, r_hi);
        __syncthreads();
    } else {
        // tiles too large for the LDS buffer: batches of cells whose colours fit, re-reading the tile's records
        int c_begin = 0;
        while (c_begin < TCELLS) {
            const uint32_t base = L.off[2 * c_begin];
            int c_end = c_begin + 1;
            const bool huge = L.off[2 * c_begin + 2] - base > RGB_CAP;
            if (!huge)
                while (c_end < TCELLS && L.off[2 * c_end + 2] - base <= RGB_CAP) ++c_end;
            if (huge) {
                big_cell_medians<I64>(L, a, s_rgb, false, c_begin, base, r_lo, r_hi);
            } else {
                const uint32_t n_batch = L.off[2 * c_end] - base;
                if (n_batch) {
                    for (int k = 2 * c_begin + threadIdx.x; k < 2 * c_end; k += C_THREADS) L.cur[k] = L.off[k] - base;
                    __syncthreads();
                    for (uint32_t r = r_lo + threadIdx.x; r < r_hi; r += C_THREADS) {
                        const uint32_t *w = rec_words<I64>(a, r);
                        const uint32_t k = w[REC_FK<I64>()];
                        if ((int)(k >> 1) >= c_begin && (int)(k >> 1) < c_end)
                            s_rgb[atomicAdd(&L.cur[k], 1u)] = w[REC_C<I64>()] & 0xffffffu;
                    }
                    __syncthreads();
                    small_cells_bitplanes(L, s_rgb, base, c_begin, c_end);
                    __syncthreads();
                    small_cells_select(L, c_begin, c_end);
                    for (int cell = c_begin; cell < c_end; ++cell)
                        if (L.cnt[2 * cell] + L.cnt[2 * cell + 1] > 64)
                            big_cell_medians<I64>(L, a, s_rgb, true, cell, base, r_lo, r_hi);
                }
            }
            __syncthreads();
            c_begin = c_end;
        }
    }

    // ---- closed-form maps: thread -> (set, cell) ----
    if (threadIdx.x < 3 * TCELLS) {
        const int s = threadIdx.x / TCELLS, cell = threadIdx.x % TCELLS;
        const int kp = 2 * cell, kf = 2 * cell + 1;
        uint32_t n, n_r, n_d;
        long long ihi, ilo;
        double zmin = 0.0;
        if (s < 2) {
            const int k = s ? kf : kp;
            n = L.cnt[k]; n_r = L.road[k]; n_d = L.dyn[k]; ihi = (long long)L.ihi[k]; ilo = (long long)L.ilo[k];
            if (n) zmin = f64_from_order_key(L.zk[k]);
        } else {
            n = L.cnt[kp] + L.cnt[kf]; n_r = L.road[kp] + L.road[kf]; n_d = L.dyn[kp] + L.dyn[kf];
            ihi = (long long)L.ihi[kp] + (long long)L.ihi[kf];
            ilo = (long long)L.ilo[kp] + (long long)L.ilo[kf];
            const unsigned long long zk = L.zk[kp] < L.zk[kf] ? L.zk[kp] : L.zk[kf];
            if (n) zmin = f64_from_order_key(zk);
        }
        double o[7];
        finalize_cell(q, n, n_r, n_d, ihi, ilo, zmin, L.med2[s][cell], o);
#pragma unroll
        for (int k = 0; k < 7; ++k) s_out[7 * s + k][cell] = o[k];
    }
    __syncthreads();
    const int row0 = (tile / a.tx) * TS, col0 = (tile % a.tx) * TS;
    const int64_t ncell = (int64_t)q.px * q.px;
    for (int idx = threadIdx.x; idx < 21 * TCELLS; idx += C_THREADS) {
        const int plane = idx / TCELLS, lc = idx % TCELLS;
        const int row = row0 + lc / TS, col = col0 + lc % TS;
        if (row >= q.px || col >= q.px) continue;
        const double v = s_out[plane][lc];
        const int64_t o = (int64_t)plane * ncell + (int64_t)row * q.px + col;
        if (a.planes) a.planes[o] = v;
        if (a.planes_f16) a.planes_f16[o] = f64_to_f16_bits(v);
    }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static inline int64_t align256(int64_t v) { return (v + 255) & ~255ll; }
static inline int tiles_x(int px) { return (px + TS - 1) / TS; }
static inline int n_groups(int64_t max_points)
{
    static int max_g = 0, per_g = 0;
    if (!max_g) {                                           // PCA_BEV_G / PCA_BEV_CHUNK: tuning overrides
        const char *e = getenv("PCA_BEV_G"), *c = getenv("PCA_BEV_CHUNK");
        max_g = e ? atoi(e) : MAX_G;
        per_g = c ? atoi(c) : 8192;
        if (max_g < 1 || max_g > 1024) max_g = MAX_G;
        if (per_g < 1024) per_g = 8192;
    }
    int64_t g = (max_points + per_g - 1) / per_g;
    return (int)(g < 1 ? 1 : (g > max_g ? max_g : g));
}

extern "C" {

int64_t pca_bev_workspace_bytes(int64_t max_points, int px)
{
    if (max_points < 1) max_points = 1;
    const int64_t T = (int64_t)tiles_x(px) * tiles_x(px), G = n_groups(max_points);
    return align256(max_points * 4) + 2 * align256(G * T * 4) + align256((T + 1) * 4) + align256(max_points * 24) + 512;
}

int pca_bev_generate(pca_ctx *ctx, const pca_store *store, const double *intensity64, const int64_t *frame_off,
                     int slot_begin, int slot_split, int slot_end, int64_t max_points, const pca_bev_params *prm,
                     const double *pending_T, int pending_slot_end, void *workspace, int64_t workspace_bytes,
                     double *planes, uint16_t *planes_f16, void *stream)
{
    if (!ctx) return -1;
    if (!store || !frame_off || !prm || !workspace || (!planes && !planes_f16)) { ctx->err = "bev: bad arguments"; return -1; }
    if (prm->px < 1 || prm->px > 1024) { ctx->err = "bev: px must be in 1..1024"; return -1; }
    if (!(slot_begin <= slot_split && slot_split <= slot_end)) { ctx->err = "bev: need slot_begin <= slot_split <= slot_end"; return -1; }
    if (max_points < 1) max_points = 1;
    if (max_points >= (1ll << 32)) { ctx->err = "bev: window too large for 32-bit positions"; return -1; }
    if (workspace_bytes < pca_bev_workspace_bytes(max_points, prm->px)) { ctx->err = "bev: workspace too small"; return -1; }
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    BevArgs a;
    a.st = *store;
    a.intensity64 = intensity64;
    a.frame_off = frame_off;
    a.slot_begin = slot_begin; a.slot_split = slot_split; a.slot_end = slot_end;
    a.max_points = max_points;
    a.prm = *prm;
    a.pend_slot_end = slot_begin;
    for (int i = 0; i < 12; ++i) a.pend_T.m[i] = 0.0;
    if (pending_T && pending_slot_end > slot_begin) {
        if (pending_slot_end > slot_end) { ctx->err = "bev: pending_slot_end beyond the window"; return -1; }
        a.pend_slot_end = pending_slot_end;
        for (int i = 0; i < 12; ++i) a.pend_T.m[i] = pending_T[i];
    }
    a.tx = tiles_x(prm->px);
    a.T = a.tx * a.tx;
    a.G = n_groups(max_points);
    char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    a.key = reinterpret_cast<uint32_t *>(w); w += align256(max_points * 4);
    a.bh = reinterpret_cast<uint32_t *>(w); w += align256((int64_t)a.G * a.T * 4);
    a.boff = reinterpret_cast<uint32_t *>(w); w += align256((int64_t)a.G * a.T * 4);
    a.tile_off = reinterpret_cast<uint32_t *>(w); w += align256((int64_t)(a.T + 1) * 4);
    a.recs = w;
    a.planes = planes;
    a.planes_f16 = planes_f16;
    const int64_t n_scan = (int64_t)a.T * a.G;
    a.scan_tiles = (int)((n_scan + SCAN_TILE - 1) / SCAN_TILE);
    if (pca_ctx_reserve_tiles(ctx, a.scan_tiles, s)) return -1;
    a.state = ctx->tile_state;
    a.ticket = ctx->ticket;
    a.epoch = pca_ctx_next_epoch(ctx, s);
    const size_t lds = (size_t)a.T * 4;
    PCA_LAUNCH_SHM(ctx, PCA_K_BEV_BIN, bev_tile_hist, dim3(a.G), dim3(AB_THREADS), lds, s, a);
    PCA_LAUNCH(ctx, PCA_K_BEV_SCAN, bev_tile_scan, dim3(a.scan_tiles), dim3(SCAN_THREADS), s, a);
    if (intensity64) {
        PCA_LAUNCH_SHM(ctx, PCA_K_BEV_SCATTER, bev_tile_scatter<true>, dim3(a.G), dim3(AB_THREADS), lds, s, a);
        PCA_LAUNCH(ctx, PCA_K_BEV_CELLS, bev_tile_cells<true>, dim3(a.T), dim3(C_THREADS), s, a);
    } else {
        PCA_LAUNCH_SHM(ctx, PCA_K_BEV_SCATTER, bev_tile_scatter<false>, dim3(a.G), dim3(AB_THREADS), lds, s, a);
        PCA_LAUNCH(ctx, PCA_K_BEV_CELLS, bev_tile_cells<false>, dim3(a.T), dim3(C_THREADS), s, a);
    }
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
