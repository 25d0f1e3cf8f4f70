// pca_icp.hip -- point-to-plane ICP between two lidar sweeps on the device.
//
// Replaces the reference's only remaining native dependency on the KITTI-360 flow:
//     sem_pc_accum.py:310-315        pc2pcd: Open3D PointCloud + estimate_normals()   (default: 30 nearest neighbours)
//     kitti360_sem_pc_accum.py:115-127  registration_icp(source = previous sweep, target = new sweep, threshold, init,
//                                       TransformationEstimationPointToPlane())        (defaults: <= 30 iterations,
//                                       relative fitness / rmse 1e-6)
// Open3D is third-party and unpinned (README.md:14): PARITY IS UNPINNED.  The tests check known motions and a CPU model
// of this same algorithm (exact k-d tree neighbours), not Open3D.
//
//   icp_grid_count / icp_cell_scan / icp_grid_fill   two uniform grids around the sensor (0.5 m and 0.25 m cells):
//                     counting sort of the target points by cell -- a cell's points are contiguous, a row of cells is
//                     one range, and a neighbour search streams them (four loads in flight).  One kernel each for both
//                     grids; only the slabs (cells along z) the cloud occupies are scanned and searched
//   icp_normals       per target point (8 lanes each): the K = 30 nearest neighbours (shell-by-shell grid search, exact
//                     within the search cap), covariance, eigenvector of the smallest eigenvalue (cyclic Jacobi, f64)
//   icp_match         per source point (8 lanes each): q = T p; from the third pass on, first the question whether the partner
//                     of the last pass provably is still the unique nearest target (then no search); else the nearest target
//                     (exact within the cap; the previous partner bounds the search from the start); the correspondence's
//                     point-to-plane row r = (q - t).n, J = [q x n, n]; per workgroup one column of partial sums of J^T J,
//                     J^T r, |q - t|^2, pair count (fixed order)
//   icp_solve         30 workgroups, one per accumulator, add the columns up in a fixed order (deterministic); the last to
//                     arrive takes the step: 6x6 Cholesky solve, T <- exp(x) T, fitness / rmse, convergence flag -- two
//                     launches per iteration, and the host looks at the flag only every twelfth pass.
// Measured on two 120 k-point sweeps: round 2 10.9 ms -> 2.8 ms per registration; round 4 2.71 -> 2.1 ms (grids 278 -> 131 us,
// rows + solve 35 -> 9 us per pass, late passes 130 -> 80 us); what changed is in the comments of icp_cell_scan, icp_visit_shell,
// icp_match and icp_solve_step, the measurements in profiles/r04_experiments/icp_second_session.txt.
#include "pca_common.h"
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <chrono>

// Two uniform grids of 512 x 512 x 64 cells around the sensor: level 0 = 0.5 m cells (256 m x 256 m x 32 m), level 1 =
// 0.25 m cells (128 m x 128 m x 16 m).  Next to the sensor a 0.5 m cell holds ~300 returns: a query settles in the fine
// grid (<= 1 m search radius) and only the sparse far field expands through the coarse one.
#define ICP_NX 512
#define ICP_NY 512
#define ICP_NZ 64
#define ICP_CELLS ((int64_t)ICP_NX * ICP_NY * ICP_NZ)
template <int LV> struct IcpLevel {
    static constexpr double cell = LV ? 0.25 : 0.5;
    static constexpr double ox = LV ? -64.0 : -128.0, oy = LV ? -64.0 : -128.0, oz = LV ? -8.0 : -16.0;
};
#define ICP_K 30                 // neighbours of a normal (Open3D's default KDTreeSearchParamKNN)
#define ICP_FINE_RINGS 4         // fine-grid search radius: 1 m
#define ICP_NORMAL_RINGS 6       // search cap of the normals: 3 m (coarse rings)
#define ICP_MATCH_RINGS 8        // search cap of a correspondence: 4 m (the reference passes 1e3 m = everything)
#define ICP_THREADS 256
// Waves per SIMD the compiler is asked for on the search kernels (their 104-108 VGPRs allow four; five costs 6-16 spilled
// registers and measured 1.47 against 1.50 ms per registration, six 1.59: profiles/r05_experiments/icp_round5.txt).
// -DICP_WAVES_PER_EU=<n>: A/B builds (tools/experiments/icp_ab.sh)
#ifndef ICP_WAVES_PER_EU
#define ICP_WAVES_PER_EU 5
#endif
#define ICP_OCC __attribute__((amdgpu_waves_per_eu(ICP_WAVES_PER_EU)))
#define ICP_NACC 30              // 21 (J^T J upper) + 6 (J^T r) + sum d^2 + inliers + sum r^2
#define ICP_CHECK_EVERY 12        // the host looks at the convergence flag after every 12th pass (a look = a copy + a wait: 45 us of idle
                                  // GPU; a pass after convergence = two early-exit launches: 9 us.  Registrations take 5-15 updates)

struct IcpGrid {
    uint32_t *cnt;               // [cells] points per cell (counting pass), all zero again after the fill pass
    uint32_t *start;             // [cells + 1] first sorted position of the cell; start[cells] = points inside the grid
    float4 *spts;                // [n_tgt] target points sorted by cell: x, y, z, original index (bits)
};

struct IcpArgs {
    const float *src;            // [n_src,4]
    const float *tgt;            // [n_tgt,4]
    int n_src, n_tgt;
    IcpGrid g[2];                // [0] coarse, [1] fine
    float *normal;               // [n_tgt,4]  nx, ny, nz, valid -- by ORIGINAL index
    int32_t *nn_prev;            // [n_src] original index of the previous iteration's correspondence, -1 = none
    int32_t *nn_cell;            // [n_src] coarse cell the query lay in when it was searched last, -1 = outside the grid
    float *nn_slack;             // [n_src] how far the query may still move before its partner has to be searched again (icp_match)
    int no_skip;                 // PCA_ICP_NO_SKIP=1: every pass searches every query (A/B)
    int dbg;                     // PCA_ICP_DBG bit 0: searched queries per pass counted in state[48 + pass] and printed by the host;
                                 // bits 1, 2: ablations of icp_normals (no eigen decomposition / no covariance pass), timing only
    uint64_t *lb_state;          // decoupled look-back of the cell scan
    uint32_t epoch;
    double *partial;             // [ICP_NACC][grid] partial sums, one column per workgroup of icp_match
    double *state;               // [0..15] T (row-major), [16] fitness, [17] rmse, [18] prev fitness, [19] prev rmse,
                                 // [20] converged flag, [21] iterations done, [24..25] zr, [32..43] T of the pass before
    uint32_t *zr;                // [4] occupied slab range of the two grids: zmin, zmax of level 0, then of level 1 (cells along z;
                                 // zmin > zmax: no point inside).  Counting fills it; only these slabs are scanned and searched
    uint32_t *status;            // context status word
    uint32_t *zr_part;           // [workgroups of icp_grid_count][4] their zmin, zmax + 1 per grid
    int n_count_blocks;
    uint32_t *arrived;           // workgroups of the running icp_solve that have written their sum
    double max_dist2;
    double rel_fitness, rel_rmse;
    int grid;
    double *host;                // mapped host memory [32]: what icp_solve_step stores into state[0..21], and [31] = the tag
    uint64_t tag;                // (call << 16) of this registration; icp_solve publishes tag | ended << 8 | passes done
    int pass;                    // number of this pass (0 = first)
};

template <int LV>
__device__ __forceinline__ bool icp_cell_of(double x, double y, double z, int &cx, int &cy, int &cz)
{
    using G = IcpLevel<LV>;
    const double fx = floor((x - G::ox) / G::cell), fy = floor((y - G::oy) / G::cell), fz = floor((z - G::oz) / G::cell);
    if (!(fx >= 0 && fx < ICP_NX && fy >= 0 && fy < ICP_NY && fz >= 0 && fz < ICP_NZ)) return false;
    cx = (int)fx; cy = (int)fy; cz = (int)fz;
    return true;
}
__device__ __forceinline__ int icp_cell_index(int cx, int cy, int cz) { return (cz * ICP_NY + cy) * ICP_NX + cx; }
// the slabs of a grid that hold points (icp_grid_count): lo > hi = none
struct IcpSlab { int lo, hi; };

// Runs of equal cells among CONSECUTIVE lanes.  A sweep delivers its points ring by ring: next to the sensor thirty neighbours in
// azimuth share a 0.5 m cell, and a cell there receives ~300 returns -- 300 same-address atomics, which the memory side executes
// one after the other (icp_grid_count 42 us, icp_grid_fill 54 us for 120 k points were mostly that).  The first lane of a run does
// ONE atomic for the whole run.  key < 0: the lane takes no part.  All 64 lanes must call it (shuffles).
struct IcpRun { bool head; int first, rank, len; };
__device__ __forceinline__ IcpRun icp_run(int key)
{
    const int lane = threadIdx.x & 63;
    const int prev = __shfl_up(key, 1, 64);
    IcpRun r;
    r.head = key >= 0 && (lane == 0 || prev != key);
    const uint64_t heads = __ballot(r.head), stops = __ballot(r.head || key < 0);
    const uint64_t upto = ~0ull >> (63 - lane);             // bits 0..lane
    const uint64_t below = heads & upto;
    r.first = below ? 63 - __clzll((long long)below) : lane;
    const uint64_t above = r.first < 63 ? stops & (~0ull << (r.first + 1)) : 0ull;
    const int end = above ? (int)__ffsll((long long)above) - 1 : 64;
    r.rank = lane - r.first;
    r.len = end - r.first;
    return r;
}

#define ICP_SCAN_THREADS 1024
#define ICP_SCAN_PER 16                                    // cells per thread of the scan
#define ICP_SCAN_TILE (ICP_SCAN_PER * ICP_SCAN_THREADS)

// one thread per target point: its cell in both grids counted, and the slabs (cells along z) the cloud occupies -- the
// grids are 64 slabs of 512 x 512 cells and a lidar sweep fills a quarter to a half of them: only those are scanned
// (icp_cell_scan) and searched (icp_visit_shell treats every other slab as outside the grid; its start[] is never read)
__global__ __launch_bounds__(ICP_THREADS) void icp_grid_count(const IcpArgs a)
{
    const int p = blockIdx.x * ICP_THREADS + threadIdx.x;
    uint32_t zlo0 = 0xffffffffu, zhi0 = 0u, zlo1 = 0xffffffffu, zhi1 = 0u;   // (zhi + 1, so that 0 = none)
    int c0 = -1, c1 = -1;                                   // the point's cell in both grids, -1 = outside
    if (p < a.n_tgt) {
        const float4 v = reinterpret_cast<const float4 *>(a.tgt)[p];
        int cx, cy, cz;
        if (icp_cell_of<0>(v.x, v.y, v.z, cx, cy, cz)) { c0 = icp_cell_index(cx, cy, cz); zlo0 = (uint32_t)cz; zhi0 = (uint32_t)cz + 1u; }
        if (icp_cell_of<1>(v.x, v.y, v.z, cx, cy, cz)) { c1 = icp_cell_index(cx, cy, cz); zlo1 = (uint32_t)cz; zhi1 = (uint32_t)cz + 1u; }
    }
    {   // one atomic per run of consecutive lanes in the same cell (icp_run)
        const IcpRun r0 = icp_run(c0), r1 = icp_run(c1);
        if (r0.head) atomicAdd(&a.g[0].cnt[c0], (uint32_t)r0.len);
        if (r1.head) atomicAdd(&a.g[1].cnt[c1], (uint32_t)r1.len);
    }
    // the workgroup's range into its row of zr_part: icp_cell_scan reduces the rows (no atomics: every wave of the launch runs at
    // once and sees the initial range, so even "only if it widens the range" meant 7500 atomics on four words -- 50-90 us)
    __shared__ uint32_t s_z[ICP_THREADS / 64][4];
    zlo0 = wave_reduce_min(zlo0); zhi0 = wave_reduce_max(zhi0); zlo1 = wave_reduce_min(zlo1); zhi1 = wave_reduce_max(zhi1);
    if ((threadIdx.x & 63) == 0) { uint32_t *r = s_z[threadIdx.x >> 6]; r[0] = zlo0; r[1] = zhi0; r[2] = zlo1; r[3] = zhi1; }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t v = s_z[0][threadIdx.x];
        for (int w = 1; w < ICP_THREADS / 64; ++w) v = (threadIdx.x & 1) ? (s_z[w][threadIdx.x] > v ? s_z[w][threadIdx.x] : v) : (s_z[w][threadIdx.x] < v ? s_z[w][threadIdx.x] : v);
        a.zr_part[4 * blockIdx.x + threadIdx.x] = v;
    }
}

// the range of occupied slabs from the rows icp_grid_count left (every workgroup of the scan for itself; 1024 threads)
__device__ __forceinline__ void icp_reduce_range(const IcpArgs &a, uint32_t (&zr)[4])
{
    __shared__ uint32_t s_r[ICP_SCAN_THREADS / 64][4];
    __shared__ uint32_t s_zr[4];
    uint32_t v[4] = {0xffffffffu, 0u, 0xffffffffu, 0u};
    for (int b = threadIdx.x; b < a.n_count_blocks; b += ICP_SCAN_THREADS) {
        const uint4 r = reinterpret_cast<const uint4 *>(a.zr_part)[b];
        v[0] = r.x < v[0] ? r.x : v[0]; v[1] = r.y > v[1] ? r.y : v[1]; v[2] = r.z < v[2] ? r.z : v[2]; v[3] = r.w > v[3] ? r.w : v[3];
    }
    v[0] = wave_reduce_min(v[0]); v[1] = wave_reduce_max(v[1]); v[2] = wave_reduce_min(v[2]); v[3] = wave_reduce_max(v[3]);
    if ((threadIdx.x & 63) == 0) { uint32_t *r = s_r[threadIdx.x >> 6]; r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3]; }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t m = s_r[0][threadIdx.x];
        for (int w = 1; w < ICP_SCAN_THREADS / 64; ++w) m = (threadIdx.x & 1) ? (s_r[w][threadIdx.x] > m ? s_r[w][threadIdx.x] : m) : (s_r[w][threadIdx.x] < m ? s_r[w][threadIdx.x] : m);
        s_zr[threadIdx.x] = m;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) zr[i] = s_zr[i];
}

// exclusive scan of the per-cell counts of the occupied slabs of both grids (single pass, decoupled look-back; blockIdx.x <
// tiles per grid: level 0, else level 1).  A slab is ICP_NX * ICP_NY / ICP_SCAN_TILE whole tiles, so the range starts and
// ends on tile borders; start[] of the cell behind the range gets the total (the end of the last row).  zr holds zmin and
// zmax + 1 here (icp_grid_count).  Tile = position in the launch: workgroups are dispatched in that order, so a tile's
// predecessors have started when it looks back (the spin is bounded all the same: PCA_STATUS_LOOKBACK_TIMEOUT).  Round 2
// drew tiles by ticket from one word: 4096 returning atomics per grid at ~90 per microsecond were 46 of the scan's 72 us.
#define ICP_SLAB_TILES (ICP_NX * ICP_NY / ICP_SCAN_TILE)
__global__ __launch_bounds__(ICP_SCAN_THREADS) void icp_cell_scan(const IcpArgs a)
{
    __shared__ uint32_t s_w[ICP_SCAN_THREADS / 64];
    __shared__ uint64_t s_excl;
    const int tiles_per_grid = (int)(ICP_CELLS / ICP_SCAN_TILE);
    const int lv = (int)blockIdx.x >= tiles_per_grid ? 1 : 0;
    const IcpGrid &g = a.g[lv];
    const int tile = (int)blockIdx.x - lv * tiles_per_grid;
    // (a workgroup far beyond any possible range leaves before the reduction: 64 slabs at most)
    uint32_t zr[4];
    icp_reduce_range(a, zr);
    if (blockIdx.x == 0 && threadIdx.x < 4) a.zr[threadIdx.x] = zr[threadIdx.x];       // for the kernels that follow (icp_slab)
    const uint32_t zlo = zr[2 * lv], zhi1 = zr[2 * lv + 1];
    if (zhi1 == 0u) return;                                 // no point inside this grid: nothing is ever looked up
    const int n_tiles = (int)(zhi1 - zlo) * ICP_SLAB_TILES;
    if (tile >= n_tiles) return;                            // (uniform) a workgroup beyond the occupied slabs
    uint64_t *lb = a.lb_state + (size_t)lv * tiles_per_grid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (thread t owns ICP_SCAN_PER consecutive cells)
    const int64_t base = ((int64_t)zlo * ICP_SLAB_TILES + tile) * ICP_SCAN_TILE + (int64_t)threadIdx.x * ICP_SCAN_PER;
    uint4 c[ICP_SCAN_PER / 4];
#pragma unroll
    for (int u = 0; u < ICP_SCAN_PER / 4; ++u) c[u] = *reinterpret_cast<const uint4 *>(g.cnt + base + 4 * u);
    uint32_t tsum = 0;
#pragma unroll
    for (int u = 0; u < ICP_SCAN_PER / 4; ++u) tsum += c[u].x + c[u].y + c[u].z + c[u].w;
    const uint32_t inc = wave_incl_scan_add(tsum);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const uint32_t v = lane < ICP_SCAN_THREADS / 64 ? s_w[lane] : 0u;
        const uint32_t winc = wave_incl_scan_add(v);
        if (lane < ICP_SCAN_THREADS / 64) s_w[lane] = winc - v;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)winc, 63);
        const uint64_t e = lb_exclusive_prefix(lb, tile, (uint64_t)total, a.epoch, a.status);
        if (lane == 0) s_excl = (e << 32) | total;
    }
    __syncthreads();
    const uint32_t excl = (uint32_t)(s_excl >> 32), total = (uint32_t)s_excl;
    uint32_t r0 = excl + s_w[wave] + (inc - tsum);
#pragma unroll
    for (int u = 0; u < ICP_SCAN_PER / 4; ++u) {
        *reinterpret_cast<uint4 *>(g.start + base + 4 * u) = make_uint4(r0, r0 + c[u].x, r0 + c[u].x + c[u].y, r0 + c[u].x + c[u].y + c[u].z);
        r0 += c[u].x + c[u].y + c[u].z + c[u].w;
    }
    if (tile == n_tiles - 1 && threadIdx.x == ICP_SCAN_THREADS - 1) g.start[base + ICP_SCAN_PER] = excl + total;
}

__global__ __launch_bounds__(ICP_THREADS) void icp_grid_fill(const IcpArgs a)
{
    const int p = blockIdx.x * ICP_THREADS + threadIdx.x;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    int c0 = -1, c1 = -1, cx, cy, cz;
    if (p < a.n_tgt) {
        v = reinterpret_cast<const float4 *>(a.tgt)[p];
        if (icp_cell_of<0>(v.x, v.y, v.z, cx, cy, cz)) c0 = icp_cell_index(cx, cy, cz);
        if (icp_cell_of<1>(v.x, v.y, v.z, cx, cy, cz)) c1 = icp_cell_index(cx, cy, cz);
    }
    const float4 rec = make_float4(v.x, v.y, v.z, __int_as_float(p));
    // a run of consecutive lanes in one cell takes its places with ONE returning atomic (its first lane), from the back of the
    // cell's range; both grids' lookups and atomics before either record store: two chains side by side
    const IcpRun r0 = icp_run(c0), r1 = icp_run(c1);
    uint32_t s0 = 0u, s1 = 0u, b0 = 0u, b1 = 0u;
    if (c0 >= 0) s0 = a.g[0].start[c0];
    if (c1 >= 0) s1 = a.g[1].start[c1];
    if (r0.head) b0 = atomicSub(&a.g[0].cnt[c0], (uint32_t)r0.len);
    if (r1.head) b1 = atomicSub(&a.g[1].cnt[c1], (uint32_t)r1.len);
    b0 = (uint32_t)__shfl((int)b0, r0.first, 64);
    b1 = (uint32_t)__shfl((int)b1, r1.first, 64);
    if (c0 >= 0) a.g[0].spts[s0 + b0 - 1u - (uint32_t)r0.rank] = rec;
    if (c1 >= 0) a.g[1].spts[s1 + b1 - 1u - (uint32_t)r1.rank] = rec;
}

__device__ __forceinline__ IcpSlab icp_slab(const IcpArgs &a, int lv)
{
    IcpSlab sl;
    sl.lo = (int)a.zr[2 * lv];                              // (0xffffffff -> -1 if the grid is empty: lo > hi either way)
    sl.hi = (int)a.zr[2 * lv + 1] - 1;
    if (a.zr[2 * lv + 1] == 0u) { sl.lo = 1; sl.hi = 0; }
    return sl;
}

// the records [s0, e) of a range of cells, four loads in flight (a one-record-per-trip loop is a chain of memory latencies:
// measured 1 ms per 120 k queries, independent of how much the search is culled)
// SGN > 1: SGN neighbouring lanes share one query; lane `sub` takes every SGN-th record, four loads in flight
template <int SGN, typename F>
__device__ __forceinline__ void icp_scan_range_sg(const float4 *spts, uint32_t s0, uint32_t e, int sub, F &&f)
{
    for (uint32_t q = s0 + (uint32_t)sub; q < e; q += 4 * SGN) {        // (four loads in flight: a dense cell next to the sensor
        const uint32_t q1 = q + SGN, q2 = q + 2 * SGN, q3 = q + 3 * SGN;   //  holds hundreds of records, and its scan is the longest chain of a pass)
        const float4 w0 = spts[q], w1 = spts[q1 < e ? q1 : q], w2 = spts[q2 < e ? q2 : q], w3 = spts[q3 < e ? q3 : q];
        f(w0);
        if (q1 < e) f(w1);
        if (q2 < e) f(w2);
        if (q3 < e) f(w3);
    }
}

template <typename F>
__device__ __forceinline__ void icp_scan_range(const float4 *spts, uint32_t s0, uint32_t e, F &&f)
{
    for (uint32_t q = s0; q < e; q += 4) {
        const uint32_t last = e - 1;
        const float4 w0 = spts[q], w1 = spts[q + 1 < e ? q + 1 : last], w2 = spts[q + 2 < e ? q + 2 : last],
                     w3 = spts[q + 3 < e ? q + 3 : last];
        f(w0);
        if (q + 1 < e) f(w1);
        if (q + 2 < e) f(w2);
        if (q + 3 < e) f(w3);
    }
}

// Visits every target point (its sorted record) of the shell of Chebyshev radius r around cell (cx,cy,cz) of grid LV,
// skipping the cells whose box lies farther from the query (qx,qy,qz) than bound() -- the caller's current search radius
// squared (after the own cell has produced a candidate a few centimetres away, almost every neighbour is culled).
// A shell is walked as
//   (1) its four faces made of whole rows of cells (|dz| = r, or |dy| = r): a row is ONE range of sorted points;
//   (2) its two faces at |dx| = r: (2r-1)^2 single cells each, four lookups in flight.
// SGN = 1: one lane does it all.  SGN = 8: eight neighbouring lanes share the query (sub = the lane's number in its
// group).  The LOOKUPS are dealt out piece by piece -- a far-field query that finds nothing walks ~2000 cells over its
// shells -- and every range a lane finds non-empty is then SCANNED BY ALL EIGHT, record by record -- next to the sensor a
// row of three cells holds hundreds of returns, and one lane scanning it alone was the slowest query of the launch
// (measured per query with s_memrealtime: 0-5 m 36 us mean / 218 us max, beyond 20 m 13 us).  All loops have group-uniform
// trip counts; the eight lanes must be converged at every call.
template <int LV, int SGN, typename B, typename F>
__device__ __forceinline__ void icp_visit_shell(const IcpGrid &g, const IcpSlab sl, int cx, int cy, int cz, int r, double qx, double qy,
                                                double qz, int sub, B &&bound, F &&f, const int rx = 0)
{
    // rx > 0 ("rows mode", group-uniform): the rows of the shell are taken across x in [cx - rx, cx + rx] instead of
    // [cx - r, cx + r] and the two faces at |dx| = r are left out -- with rx >= r the cube of radius r is covered all the
    // same, by 8 r lookups instead of 8 r + 2 (2r - 1)^2.  For a query that has NO candidate yet nothing can be culled: over nine
    // coarse shells that is 289 lookups instead of 1650, and such queries (max-range returns, 245 us) are what a pass waits for.
    using G = IcpLevel<LV>;
    auto gap = [](double q, double lo) { return q < lo ? lo - q : (q > lo + G::cell ? q - (lo + G::cell) : 0.0); };
    // the ranges the lanes of the group hold in (s0, e0), each scanned by the whole group
    auto scan_found = [&](uint32_t s0, uint32_t e0) {
        if (SGN == 1) {
            if (e0 > s0) icp_scan_range(g.spts, s0, e0, f);
            return;
        }
        const int base = (int)(threadIdx.x & 63) & ~(SGN - 1);
        uint32_t bits = (uint32_t)(__ballot(e0 > s0) >> base) & ((1u << SGN) - 1u);
        while (bits) {
            const int l = __ffs(bits) - 1;
            bits &= bits - 1;
            const uint32_t s = (uint32_t)__shfl((int)s0, base + l, 64), e = (uint32_t)__shfl((int)e0, base + l, 64);
            icp_scan_range_sg<SGN>(g.spts, s, e, sub, f);
        }
    };
    const int half = rx > 0 ? rx : r;
    const int x_lo = cx - half < 0 ? 0 : cx - half, x_hi = cx + half >= ICP_NX ? ICP_NX - 1 : cx + half;
    if (r == 0) {
        if (cz < sl.lo || cz > sl.hi) return;               // (group-uniform) an empty slab: its start[] was never written
        const int c0 = icp_cell_index(x_lo, cy, cz);        // (the own cell, or the own row)
        if (SGN == 1) icp_scan_range(g.spts, g.start[c0], g.start[c0 + (x_hi - x_lo) + 1], f);
        else icp_scan_range_sg<SGN>(g.spts, g.start[c0], g.start[c0 + (x_hi - x_lo) + 1], sub, f);
        return;
    }
    const int n = 2 * r + 1, m = 2 * r - 1;
    // (four row lookups in flight per lane, as for the single cells below: one lookup per trip made a far-field query's outer
    // shells -- 8 r rows each -- a chain of r + 1 round trips per shell, and such queries are the last to finish in a pass)
    for (int i0 = 0; i0 < 2 * n + 2 * m; i0 += 4 * SGN) {
        uint32_t s0[4], e0[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s0[u] = e0[u] = 0u;
            const int i = i0 + u * SGN + sub;
            if (i >= 2 * n + 2 * m) continue;
            int dz, dy;
            if (i < 2 * n) { dz = i < n ? -r : r; dy = (i < n ? i : i - n) - r; }
            else { const int j = i - 2 * n; dy = j < m ? -r : r; dz = (j < m ? j : j - m) - (r - 1); }
            const int z = cz + dz, y = cy + dy;
            if (z < sl.lo || z > sl.hi || y < 0 || y >= ICP_NY) continue;
            const double ez = gap(qz, G::oz + z * G::cell), ey = gap(qy, G::oy + y * G::cell);
            if (ey * ey + ez * ez >= bound()) continue;
            const int c0 = icp_cell_index(x_lo, y, z);
            s0[u] = g.start[c0];
            e0[u] = g.start[c0 + (x_hi - x_lo) + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) scan_found(s0[u], e0[u]);
    }
    const int ncell = rx > 0 ? 0 : 2 * m * m;
    for (int k0 = 0; k0 < ncell; k0 += 4 * SGN) {
        uint32_t s0[4], e0[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s0[u] = e0[u] = 0u;
            const int k = k0 + u * SGN + sub;
            if (k >= ncell) continue;
            const int mm = k >> 1, iz = mm / m;
            const int z = cz + iz - (r - 1), y = cy + (mm - iz * m) - (r - 1), x = cx + ((k & 1) ? r : -r);
            if (z < sl.lo || z > sl.hi || y < 0 || y >= ICP_NY || x < 0 || x >= ICP_NX) continue;
            const double ez = gap(qz, G::oz + z * G::cell), ey = gap(qy, G::oy + y * G::cell), ex = gap(qx, G::ox + x * G::cell);
            if (ex * ex + ey * ey + ez * ez >= bound()) continue;
            const int c0 = icp_cell_index(x, y, z);
            s0[u] = g.start[c0];
            e0[u] = g.start[c0 + 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) scan_found(s0[u], e0[u]);
    }
}

// the fine grid can serve a query alone iff its whole search box (ICP_FINE_RINGS cells each way) lies inside the grid
__device__ __forceinline__ bool icp_fine_box(double x, double y, double z, int &cx, int &cy, int &cz)
{
    if (!icp_cell_of<1>(x, y, z, cx, cy, cz)) return false;
    return cx >= ICP_FINE_RINGS && cx < ICP_NX - ICP_FINE_RINGS && cy >= ICP_FINE_RINGS && cy < ICP_NY - ICP_FINE_RINGS &&
           cz >= ICP_FINE_RINGS && cz < ICP_NZ - ICP_FINE_RINGS;
}
// was record w visited by the fine pass of a query whose fine cell is (cx,cy,cz)?
__device__ __forceinline__ bool icp_in_fine_box(const float4 w, int cx, int cy, int cz)
{
    int wx, wy, wz;
    if (!icp_cell_of<1>(w.x, w.y, w.z, wx, wy, wz)) return false;
    return abs(wx - cx) <= ICP_FINE_RINGS && abs(wy - cy) <= ICP_FINE_RINGS && abs(wz - cz) <= ICP_FINE_RINGS;
}

// symmetric 3x3 eigen decomposition by cyclic Jacobi rotations; returns the eigenvector of the smallest eigenvalue
__device__ __forceinline__ void icp_smallest_eigvec(double A[3][3], double n[3])
{
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    // (a rotation changes nothing once the off-diagonal part is below 2^-60 of the diagonal: cyclic Jacobi converges
    // quadratically, so that is one sweep after 1e-9 -- round 4 ran until 1e-300, three to four sweeps of pure rounding noise
    // on ONE lane of eight while the group waits: 100 of the kernel's 423 us)
    const double tiny = (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2])) * 8.7e-19;
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off <= tiny) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            if (fabs(A[p][q]) < 1e-300) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq;
                A[k][q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk;
                A[q][k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq;
                V[k][q] = s * vkp + c * vkq;
            }
        }
    }
    int m = 0;
    if (A[1][1] < A[m][m]) m = 1;
    if (A[2][2] < A[m][m]) m = 2;
    n[0] = V[0][m]; n[1] = V[1][m]; n[2] = V[2][m];
}

// Eight neighbouring lanes share one query (icp_normals, icp_match); sub = threadIdx.x & 7.
#define ICP_SG 8
__device__ __forceinline__ float icp_group_min(float v)
{
    v = fminf(v, __uint_as_float(lane_xor_fetch<1>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(lane_xor_fetch<2>(__float_as_uint(v))));
    return fminf(v, __uint_as_float(lane_xor_fetch<4>(__float_as_uint(v))));
}
__device__ __forceinline__ uint32_t icp_group_sum(uint32_t v)
{
    v += lane_xor_fetch<1>(v);
    v += lane_xor_fetch<2>(v);
    return v + lane_xor_fetch<4>(v);
}
__device__ __forceinline__ double icp_group_sum(double v)
{
#define ICP_SUM_STAGE(S)                                                                                               \
    v += __hiloint2double((int)lane_xor_fetch<S>((uint32_t)__double2hiint(v)), (int)lane_xor_fetch<S>((uint32_t)__double2loint(v)));
    ICP_SUM_STAGE(1) ICP_SUM_STAGE(2) ICP_SUM_STAGE(4)
#undef ICP_SUM_STAGE
    return v;
}

// icp_normals: per target point the K = 30 nearest neighbours, the covariance of everything within the K-th distance,
// the eigenvector of its smallest eigenvalue.  Eight lanes per point: every lane keeps the K smallest distances of the
// records IT scanned (unsorted, in LDS); at the end of a shell the group finds the K-th smallest distance of the union of
// the eight lists exactly (bisection on the bit patterns), which is the cull bound of the next shell and decides whether
// the search has settled.  The covariance pass deals the records out the same way and adds the
// eight partial sums up in a fixed butterfly.
__global__ __launch_bounds__(ICP_THREADS) ICP_OCC void icp_normals(const IcpArgs a)
{
    __shared__ float s_d[ICP_K][ICP_THREADS];               // per lane: the K smallest squared distances it saw (unsorted)
    const int p = (blockIdx.x * ICP_THREADS + threadIdx.x) / ICP_SG;   // original index
    if (p >= a.n_tgt) return;
    const int sub = threadIdx.x & (ICP_SG - 1);
    const IcpSlab sl0 = icp_slab(a, 0), sl1 = icp_slab(a, 1);
    const float4 v = reinterpret_cast<const float4 *>(a.tgt)[p];
    float4 out = make_float4(0.f, 0.f, 1.f, 0.f);
    const int t = threadIdx.x;
    // A lane's list is UNSORTED: the K smallest squared distances it has seen, the largest of them and its place cached in
    // registers.  An arrival below the largest replaces it and the list is read once to find the new largest -- K independent LDS
    // reads.  (Round 2 kept the lists sorted by insertion: every arrival shifted half the list, a chain of ~30 DEPENDENT LDS
    // round trips -- ~100 arrivals per lane made that chain 180 us per wave, the kernel's 670 us.)
    int found = 0;                                          // entries of this lane's list
    float curmax = 0.f;                                     // the largest of them ...
    int maxpos = 0;                                         // ... and where it is
    // kth: an UPPER BOUND of the group's K-th smallest distance while the shells are walked (inf: fewer than K seen) -- all it is
    // used for there is culling and the settle test, and both are safe with anything not below the true value.  The exact K-th
    // distance costs a bisection over the eight lists (31 steps x 30 entries: per-query stamps showed two or three of those per
    // point were most of the kernel) and is found ONCE, after the walk.  The bound: every lane keeps its four smallest
    // distances sorted in registers; if every lane has four that are <= x, the group has 32 >= K that are <= x.
    float kth = __builtin_huge_valf();
    float t0 = __builtin_huge_valf(), t1 = t0, t2 = t0, t3 = t0;
    static_assert(4 * ICP_SG >= ICP_K, "four per lane make K");
    auto bound_k = [&]() { return kth < __builtin_huge_valf() ? (double)kth * (1.0 + 1e-6) + 1e-12 : 1e300; };
    auto offer = [&](float d2) {
        if (d2 > kth) return;
        {
            float x = d2, y;
            y = fminf(t0, x); x = fmaxf(t0, x); t0 = y;
            y = fminf(t1, x); x = fmaxf(t1, x); t1 = y;
            y = fminf(t2, x); x = fmaxf(t2, x); t2 = y;
            t3 = fminf(t3, x);
        }
        if (found < ICP_K) {
            s_d[found][t] = d2;
            if (found == 0 || d2 > curmax) { curmax = d2; maxpos = found; }
            ++found;
            return;
        }
        if (d2 >= curmax) return;
        s_d[maxpos][t] = d2;
        float m = -1.f;
        int mp = 0;
#pragma unroll
        for (int i = 0; i < ICP_K; ++i) {
            const float w = s_d[i][t];
            if (w > m) { m = w; mp = i; }
        }
        curmax = m; maxpos = mp;
    };
    // exact K-th smallest of the union of the eight lists: the smallest bit pattern v (distances are >= 0: patterns order like
    // values) with K or more entries <= v, by bisection between the group's smallest and largest entry
    auto merge_kth = [&]() {
        if (icp_group_sum((uint32_t)found) < ICP_K) return;
        // (the list into registers first -- the walk is over, its registers are free -- so that a step is compares and adds only)
        uint32_t L[ICP_K];
        float mn = __builtin_huge_valf();
#pragma unroll
        for (int i = 0; i < ICP_K; ++i) {
            const float w = i < found ? s_d[i][t] : __builtin_huge_valf();
            L[i] = __float_as_uint(w);                      // (+inf: 0x7f800000, above every pivot)
            mn = fminf(mn, w);
        }
        uint32_t lo = __float_as_uint(icp_group_min(mn));
        uint32_t hi = __float_as_uint(-icp_group_min(found > 0 ? -curmax : __builtin_huge_valf()));
        if (kth < __builtin_huge_valf() && __float_as_uint(kth) < hi) hi = __float_as_uint(kth);   // (the bound: K or more are <= it)
        while (lo < hi) {                                   // (group-uniform)
            const uint32_t mid = lo + ((hi - lo) >> 1);
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < ICP_K; ++i) c += L[i] <= mid ? 1u : 0u;
            if (icp_group_sum(c) >= ICP_K) hi = mid; else lo = mid + 1u;
        }
        kth = __uint_as_float(lo);
    };
    auto bound_up = [&]() {                                 // (after a shell) the cheap bound; never above the previous one
        float ub = -icp_group_min(-t3);                     // max over the lanes of their fourth smallest (inf if a lane has fewer)
        if (!(ub < __builtin_huge_valf()) && icp_group_sum((uint32_t)found) >= ICP_K)
            ub = -icp_group_min(found > 0 ? -curmax : __builtin_huge_valf());    // K or more are stored: none is above the largest stored
        if (ub < kth) kth = ub;
    };
    auto dist2 = [&](const float4 w) { const float dx = w.x - v.x, dy = w.y - v.y, dz = w.z - v.z; return dx * dx + dy * dy + dz * dz; };
    // pass A: the K-th smallest distance.  Fine grid first; every unvisited point is farther than r cells, so the K-th
    // distance is final once it lies inside that radius.
    int fx, fy, fz, cx, cy, cz;
    const bool fine = icp_fine_box(v.x, v.y, v.z, fx, fy, fz);
    bool settled = false;
    if (fine)
        for (int r = 0; r <= ICP_FINE_RINGS && !settled; ++r) {
            icp_visit_shell<1, ICP_SG>(a.g[1], sl1, fx, fy, fz, r, v.x, v.y, v.z, sub, bound_k, [&](const float4 w) { offer(dist2(w)); });
            bound_up();
            const float lim = (float)(r * IcpLevel<1>::cell);
            settled = kth <= lim * lim;
        }
    const bool coarse = icp_cell_of<0>(v.x, v.y, v.z, cx, cy, cz);
    // (a point whose K-th neighbour is 2 m away or farther -- an isolated return at range, a max-range return -- walks its coarse
    // rings as rows, see icp_visit_shell: next to nothing is culled there.  One mode for the whole pass: a cell visited twice
    // would put a point into the lists twice)
    const bool rows_a = !(kth < 4.0f);
    if (!settled && coarse)
        for (int r = 0; r <= ICP_NORMAL_RINGS && !settled; ++r) {
            int rx = 0;
            if (rows_a) { rx = kth < __builtin_huge_valf() ? (int)(sqrtf(kth) * (float)(1.0 / IcpLevel<0>::cell)) + 1 : ICP_NORMAL_RINGS; rx = rx > ICP_NORMAL_RINGS ? ICP_NORMAL_RINGS : rx; }
            icp_visit_shell<0, ICP_SG>(a.g[0], sl0, cx, cy, cz, r, v.x, v.y, v.z, sub, bound_k, [&](const float4 w) {
                if (fine && icp_in_fine_box(w, fx, fy, fz)) return;             // already offered by the fine pass
                offer(dist2(w));
            }, rx);
            bound_up();
            const float lim = (float)(r * IcpLevel<0>::cell);
            settled = kth <= lim * lim;
        }
    const uint32_t total = icp_group_sum((uint32_t)found);
    if (kth < __builtin_huge_valf()) merge_kth();           // the exact K-th distance (at most the bound)
    if (total >= 3 && !(a.dbg & 4)) {                       // (PCA_ICP_DBG=4: ablation, no covariance pass)
        // the search radius of pass B: the K-th distance, or -- fewer than K points inside the search cap -- the largest
        float lim = kth;
        if (!(lim < __builtin_huge_valf())) {
            const float mine = found > 0 ? -curmax : __builtin_huge_valf();
            lim = -icp_group_min(mine);
        }
        // pass B: covariance of every point within that distance (relative to the query: well conditioned)
        double sx = 0, sy = 0, sz = 0, sxx = 0, sxy = 0, sxz = 0, syy = 0, syz = 0, szz = 0;
        uint32_t cnt = 0;
        auto bound_l = [&]() { return (double)lim * (1.0 + 1e-6) + 1e-12; };
        auto add = [&](const float4 w) {
            const float dx = w.x - v.x, dy = w.y - v.y, dz = w.z - v.z;
            if (dx * dx + dy * dy + dz * dz > lim) return;
            const double x = dx, y = dy, z = dz;
            sx += x; sy += y; sz += z;
            sxx += x * x; sxy += x * y; sxz += x * z; syy += y * y; syz += y * z; szz += z * z;
            ++cnt;
        };
        const float fine_reach = (float)(ICP_FINE_RINGS * IcpLevel<1>::cell);
        if (fine && lim <= fine_reach * fine_reach) {                           // the whole ball lies in the fine box
            const int rmax = (int)ceil(sqrt((double)lim) / IcpLevel<1>::cell);
            for (int r = 0; r <= rmax && r <= ICP_FINE_RINGS; ++r) icp_visit_shell<1, ICP_SG>(a.g[1], sl1, fx, fy, fz, r, v.x, v.y, v.z, sub, bound_l, add);
        } else if (coarse) {
            const int rmax = (int)ceil(sqrt((double)lim) / IcpLevel<0>::cell);
            int rx = 0;                                     // (rows for a wide ball, as in pass A; every row once)
            if (lim > 4.0f) { rx = (int)(sqrtf(lim) * (float)(1.0 / IcpLevel<0>::cell)) + 1; rx = rx > ICP_NORMAL_RINGS ? ICP_NORMAL_RINGS : rx; }
            for (int r = 0; r <= rmax && r <= ICP_NORMAL_RINGS; ++r) icp_visit_shell<0, ICP_SG>(a.g[0], sl0, cx, cy, cz, r, v.x, v.y, v.z, sub, bound_l, add, rx);
        }
        cnt = icp_group_sum(cnt);
        sx = icp_group_sum(sx); sy = icp_group_sum(sy); sz = icp_group_sum(sz);
        sxx = icp_group_sum(sxx); sxy = icp_group_sum(sxy); sxz = icp_group_sum(sxz);
        syy = icp_group_sum(syy); syz = icp_group_sum(syz); szz = icp_group_sum(szz);
        if (cnt >= 3 && sub == 0) {
            const double inv = 1.0 / cnt;
            const double mx = sx * inv, my = sy * inv, mz = sz * inv;
            double A[3][3];
            A[0][0] = sxx * inv - mx * mx; A[0][1] = sxy * inv - mx * my; A[0][2] = sxz * inv - mx * mz;
            A[1][1] = syy * inv - my * my; A[1][2] = syz * inv - my * mz; A[2][2] = szz * inv - mz * mz;
            A[1][0] = A[0][1]; A[2][0] = A[0][2]; A[2][1] = A[1][2];
            double n[3] = {0.0, 0.0, 1.0};
            if (!(a.dbg & 2)) icp_smallest_eigvec(A, n);    // (PCA_ICP_DBG=2: ablation, no eigen decomposition)
            const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (len > 0) out = make_float4((float)(n[0] / len), (float)(n[1] / len), (float)(n[2] / len), 1.f);
        }
    }
    if (sub == 0) reinterpret_cast<float4 *>(a.normal)[p] = out;
}

// Gauss-Newton step of one evaluation: fitness / rmse of the current transform, convergence test, 6x6 Cholesky solve,
// T <- exp(x) T.  Called by the first WAVE of icp_solve (all 64 lanes, converged): lane 0 solves, lanes 0-2 take the sine and
// cosine of one angle each.  `prev`: T and the previous evaluation's fitness / rmse / iteration count, loaded by lane 0 at the
// start of the kernel (their round trips run under the summation).  The serial part is a chain of dependent f64 operations on
// one lane: one reciprocal per pivot instead of a division per element (6 instead of 27) and the three sincos side by side
// took the kernel from 50 to ~20 us.
struct IcpPrev { double T[12], fit, rmse, iters; };
__device__ __forceinline__ double icp_lane0(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double icp_from_lane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// (returns true when this evaluation ended the registration: converged, too few pairs, singular system)
__device__ bool icp_solve_body(const IcpArgs &a, const double *sum, const IcpPrev &pv)
{
    // every store goes to the device state AND to its host-visible mirror (mapped memory; lane 0 only)
    struct Both { double *s, *h; __device__ __forceinline__ void put(int k, double v) const { s[k] = v; h[k] = v; } };
    const Both S{a.state, a.host};
    const int lane = threadIdx.x & 63;
    const double inl = sum[28];
    const double fitness = a.n_src > 0 ? inl / a.n_src : 0.0;
    const double rmse = inl > 0 ? sqrt(sum[27] / inl) : 0.0;
    // Open3D evaluates fitness / rmse of the CURRENT transform, then updates; convergence compares successive evaluations
    const bool first = pv.iters == 0.0;
    if (lane == 0) { S.put(16, fitness); S.put(17, rmse); }
    const bool conv = !first && fabs(pv.fit - fitness) < a.rel_fitness && fabs(pv.rmse - rmse) < a.rel_rmse;
    if (icp_lane0(conv ? 1.0 : 0.0) != 0.0) { if (lane == 0) S.put(20, 1.0); return true; }
    if (lane == 0) { S.put(18, fitness); S.put(19, rmse); }
    if (inl < 6) { if (lane == 0) S.put(20, 1.0); return true; }    // (uniform: sum[] is shared)
    // solve (J^T J) x = -J^T r  (Cholesky, upper triangle stored row-wise in sum[0..20]).  Every loop has a constant trip
    // count and no early exit, so the 6x6 system lives in registers (indexed dynamically it sat in scratch memory: ~100
    // dependent scratch round trips, most of the kernel's 55 us)
    double A[6][6], b[6], x[6], inv[6];
    {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) { A[i][j] = A[j][i] = sum[k++]; }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = -sum[21 + i];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            double s = A[i][j];
#pragma unroll
            for (int m = 0; m < j; ++m) s -= A[i][m] * A[j][m];
            if (i == j) { ok = ok && (s > 1e-12); A[i][i] = sqrt(ok ? s : 1.0); inv[i] = 1.0 / A[i][i]; }
            else A[i][j] = s * inv[j];
        }
    }
    if (!ok) { if (lane == 0) S.put(20, 1.0); return true; }        // (uniform)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int m = 0; m < i; ++m) s -= A[i][m] * x[m];
        x[i] = s * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int m = i + 1; m < 6; ++m) s -= A[m][i] * x[m];
        x[i] = s * inv[i];
    }
    // x = (alpha, beta, gamma, tx, ty, tz): R = Rz(gamma) Ry(beta) Rx(alpha); lane l takes sin / cos of angle l
    const double ang = lane == 0 ? x[0] : lane == 1 ? x[1] : x[2];
    const double sn = sin(ang), cs = cos(ang);
    const double sa = icp_from_lane(sn, 0), ca = icp_from_lane(cs, 0), sb = icp_from_lane(sn, 1), cb = icp_from_lane(cs, 1),
                 sg = icp_from_lane(sn, 2), cg = icp_from_lane(cs, 2);
    const double U[12] = {cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa, x[3],
                          sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa, x[4],
                          -sb, cb * sa, cb * ca, x[5]};
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) a.state[32 + i] = pv.T[i];   // (icp_match: how far has a point moved since the last pass?)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                S.put(4 * i + j, U[4 * i] * pv.T[j] + U[4 * i + 1] * pv.T[4 + j] + U[4 * i + 2] * pv.T[8 + j] + (j == 3 ? U[4 * i + 3] : 0.0));
        S.put(21, pv.iters + 1.0);
    }
    return false;
}
// the step, then the news for the host: tag | ended << 8 | passes done, released at system scope behind the stores above --
// pca_icp_register polls that word in mapped memory instead of copying the state back and waiting for the stream
__device__ void icp_solve_step(const IcpArgs &a, const double *sum, const IcpPrev &pv)
{
    const bool ended = icp_solve_body(a, sum, pv);
    if ((threadIdx.x & 63) == 0) {
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<uint64_t *>(a.host + 31), a.tag | (ended ? 0x100ull : 0ull) | (uint64_t)(a.pass + 1),
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}



// icp_match: ICP_SG neighbouring lanes share one source point.  The own cell's records and the rows of every further
// shell are dealt out over the lanes, each lane keeps its own best partner, and at the end of every shell the lanes agree
// on the group's best (DPP butterfly inside the 8 lanes), which tightens everybody's cull bound and decides whether the
// search has settled.  One lane per query was a chain of dependent loads and row walks as long as the worst query of its
// wave (~0.8 ms per pass at < 2 waves per SIMD).  Ties: lowest ORIGINAL index (the order inside a cell varies from run
// to run).  The partner's coordinates travel with its distance (they are in the record that was scanned: re-reading the
// target by index afterwards is one more memory round trip at the end of the query's chain).
struct IcpBest { double d2; int idx; float x, y, z; };
template <int SG>
__device__ __forceinline__ void icp_group_best(IcpBest &b)
{
#define ICP_BEST_STAGE(S)                                                                                              \
    {                                                                                                                  \
        const uint32_t lo = lane_xor_fetch<S>((uint32_t)__double2loint(b.d2)), hi = lane_xor_fetch<S>((uint32_t)__double2hiint(b.d2)); \
        const int oi = (int)lane_xor_fetch<S>((uint32_t)b.idx);                                                        \
        const float ox = __uint_as_float(lane_xor_fetch<S>(__float_as_uint(b.x))), oy = __uint_as_float(lane_xor_fetch<S>(__float_as_uint(b.y))), \
                    oz = __uint_as_float(lane_xor_fetch<S>(__float_as_uint(b.z)));                                     \
        const double od = __hiloint2double((int)hi, (int)lo);                                                          \
        if (oi >= 0 && (b.idx < 0 || od < b.d2 || (od == b.d2 && oi < b.idx))) { b.d2 = od; b.idx = oi; b.x = ox; b.y = oy; b.z = oz; } \
    }
    ICP_BEST_STAGE(1) ICP_BEST_STAGE(2) ICP_BEST_STAGE(4)
    if (SG > 8) ICP_BEST_STAGE(8)
#undef ICP_BEST_STAGE
}

// (the group's nearest G, and the smallest distance any lane has seen of a target that is not G)
template <int SG>
__device__ __forceinline__ void icp_group_merge(IcpBest &b, double &other)
{
    static_assert(SG == 8 || SG == 16, "the butterflies cover 8 or 16 lanes");
    const IcpBest mine = b;
    icp_group_best<SG>(b);
    // this lane's nearest that is not G: its best -- unless that IS G (it found G, or a second copy of it)
    double c = other;
    if (mine.idx >= 0 && mine.idx != b.idx && mine.d2 < c) c = mine.d2;
#define ICP_MIN_STAGE(S)                                                                                               \
    {                                                                                                                  \
        const double od = __hiloint2double((int)lane_xor_fetch<S>((uint32_t)__double2hiint(c)), (int)lane_xor_fetch<S>((uint32_t)__double2loint(c))); \
        c = od < c ? od : c;                                                                                           \
    }
    ICP_MIN_STAGE(1) ICP_MIN_STAGE(2) ICP_MIN_STAGE(4)
    if (SG > 8) ICP_MIN_STAGE(8)
#undef ICP_MIN_STAGE
    other = c;
}

// One pass = icp_match + icp_solve.  icp_match also turns every correspondence into its point-to-plane row
// (r = (q - t).n, J = [q x n, n]) and leaves the workgroup's sums of J^T J, J^T r, |q - t|^2, pair count, r^2 as one column
// of partial sums (fixed order: queries of a workgroup in ascending order, four at a time); icp_solve adds the columns up
// (fixed order again: the result does not depend on timing) and takes the Gauss-Newton step.  (Round 2 had a third kernel
// between the two -- one thread per source point re-reading partner and normal by index, 512 rows, the last workgroup to
// arrive adding them up and solving: 35 us per pass, most of it that serial tail.)
//
// From the second pass on a query first asks whether the result of the last pass PROVABLY stands.  The last search left a
// slack: with d1 the distance to the nearest target, every OTHER target was at least d1 + 2 slack away.  The query has moved
// by delta = |T p - T_old p| since: the old partner is at most d1 + delta away now, every other target at least
// d1 + 2 slack - delta -- while the movements since the search add up to less than the slack the partner is the unique
// nearest: no search, the slack shrinks by delta.  A query WITHOUT a partner (nothing within the distance cap inside its
// search box) stays without one while it stays in its coarse cell -- the box is a function of that cell -- and has moved less
// than the margin by which the box's points lay beyond the cap.  To have a slack, the search culls cells against
// (d1 + ICP_MARGIN)^2 instead of d1^2 -- everything within ICP_MARGIN of the nearest is seen, at the price of the few cells
// that 2 cm more reach -- and keeps the smallest distance `other` of a seen target that is not the nearest: every other target
// is at least min(other, d1 + ICP_MARGIN, r cell) away, r the last ring visited.
// Measured on consecutive ring-model sweeps (tools/experiments/icp_skip_potential.py, PCA_ICP_DBG=1): the updates move the
// points by 0.67 m, 0.14 m, 23 mm, 4 mm, 0.7 mm, 0.17 mm, ... against a median slack of 4.7 mm; searched queries per pass:
// 120 000 x 4, 53 301, 18 831, 6 753, 2 584, 594.  The skipped search would have returned the same partner (it is the unique
// nearest), and distance and row are computed by the same expressions: the pose is bit-identical with and without the shortcut
// (PCA_ICP_NO_SKIP=1 switches it off: A/B).
// SG lanes share a query: 8 from the second pass on; 16 in the FIRST pass, whose searches start from a partner decimetres off
// and read several times the records (measured, 4 / 8 / 16 lanes: first pass 460 / 280 / 246 us, later full passes 178 / 112 / 125)
#define ICP_QPW_MAX (ICP_THREADS / 8)     // queries per workgroup of icp_match, at most
#define ICP_MARGIN 0.02                  // [m]
template <int SG>
__global__ __launch_bounds__(ICP_THREADS) ICP_OCC void icp_match(const IcpArgs a)
{
    constexpr int ICP_QPW = ICP_THREADS / SG;
    __shared__ double s_row[ICP_QPW][8];                    // per query: J[0..5], r, |q - t|^2
    __shared__ uint32_t s_flag[ICP_QPW];                    // bit 0: has a partner, bit 1: the partner has a normal
    __shared__ double s_part[8][32];
    if (a.state[20] != 0.0) return;                         // converged: the remaining passes are no-ops
    const double *T = a.state;
    const int sub = threadIdx.x & (SG - 1);
    const int ql = threadIdx.x / SG;                    // query of the workgroup
    const int p = blockIdx.x * ICP_QPW + ql;
    IcpBest b;
    b.d2 = a.max_dist2; b.idx = -1; b.x = b.y = b.z = 0.f;
    double qx = 0, qy = 0, qz = 0;
    bool need = false;
    if (p < a.n_src) {                                      // (uniform inside a group of eight lanes, as is everything up to the search)
        const float4 v = reinterpret_cast<const float4 *>(a.src)[p];
        int prev = a.nn_prev[p];
        const int cell_ref = a.nn_cell[p];
        const float slack0 = a.nn_slack[p];
        qx = T[0] * v.x + T[1] * v.y + T[2] * v.z + T[3];
        qy = T[4] * v.x + T[5] * v.y + T[6] * v.z + T[7];
        qz = T[8] * v.x + T[9] * v.y + T[10] * v.z + T[11];
        const bool first = prev == (int)0xfefefefe;
        if (first) prev = p < a.n_tgt ? p : -1;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (prev >= 0) t = reinterpret_cast<const float4 *>(a.tgt)[prev];
        int cx, cy, cz;
        const bool in_coarse = icp_cell_of<0>(qx, qy, qz, cx, cy, cz);
        need = true;
        if (!first && !a.no_skip && slack0 > 0.f) {
            const double *O = a.state + 32;                 // T of the last pass
            const double mx = qx - (O[0] * v.x + O[1] * v.y + O[2] * v.z + O[3]);
            const double my = qy - (O[4] * v.x + O[5] * v.y + O[6] * v.z + O[7]);
            const double mz = qz - (O[8] * v.x + O[9] * v.y + O[10] * v.z + O[11]);
            const double left = (double)slack0 - sqrt(mx * mx + my * my + mz * mz) * (1.0 + 1e-9) - 1e-12;
            if (left > 0.0) {
                if (prev >= 0) {
                    const double dx = t.x - qx, dy = t.y - qy, dz = t.z - qz;
                    const double d2 = dx * dx + dy * dy + dz * dz;
                    // (the query's cell is compared as in the branch below: the search box is a function of that cell, and a step
                    // across a cell border -- however short -- moves the box by half a metre: a partner 3.5-4 m away could leave it
                    // or a nearer target enter it.  The comparison is free: the cell is known)
                    if (d2 < a.max_dist2 && (in_coarse ? icp_cell_index(cx, cy, cz) : -1) == cell_ref) { need = false; b.d2 = d2; b.idx = prev; b.x = t.x; b.y = t.y; b.z = t.z; }   // (else the search decides)
                } else if ((in_coarse ? icp_cell_index(cx, cy, cz) : -1) == cell_ref) {
                    need = false;                           // still nothing within the cap in the same box
                }
                if (!need && sub == 0) a.nn_slack[p] = (float)(left * (1.0 - 1e-6));
            }
        }
        if (need) {
            const IcpSlab sl0 = icp_slab(a, 0), sl1 = icp_slab(a, 1);
            const double max_dist = sqrt(a.max_dist2);
            // (no slack is collected in the first two passes -- the updates that follow them move the points by decimetres, no
            // partner survives that -- so they cull against d1 itself, as a search without the shortcut would)
            const double mg = a.state[21] >= 2.0 ? ICP_MARGIN : 0.0;
            double other = 1e300;                           // smallest squared distance of a seen target that is not the nearest
            double bnd = (max_dist + mg) * (max_dist + mg);     // cull bound: (d1 + margin)^2, d1 = the cap while there is no partner
            auto bound = [&]() { return bnd; };
            auto offer = [&](const float4 w) {
                const int wi = __float_as_int(w.w);
                if (wi == b.idx) return;                    // the nearest so far, met again (both grids hold every point)
                const double dx = w.x - qx, dy = w.y - qy, dz = w.z - qz;
                const double d2 = dx * dx + dy * dy + dz * dz;
                if (d2 < b.d2 || (d2 == b.d2 && b.idx >= 0 && wi < b.idx)) {
                    if (b.idx >= 0) other = b.d2;           // (the old nearest: not farther than anything seen before)
                    b.d2 = d2; b.idx = wi; b.x = w.x; b.y = w.y; b.z = w.z;
                    // (d1 + margin)^2 from above: d2 + 2 margin s + margin^2 with s >= sqrt(d2) (a single-precision root, rounded up)
                    const double s1 = (double)(sqrtf((float)d2) * 1.000001f) + 1e-30;
                    bnd = (d2 + 2.0 * mg * s1 + mg * mg) * (1.0 + 1e-12) + 1e-300;
                } else if (d2 < other) {
                    other = d2;
                }
            };
            // warm start: the partner of the last pass bounds the search from the first cell on (the transform moved by a
            // fraction of a cell), so nearly every cell is culled; the result is still the exact nearest neighbour
            // (first pass: the target point of the same index -- consecutive sweeps share their scan order -- provided the search
            // below would reach it: inside the coarse grid and within the match cap; any such point is a valid upper bound)
            if (prev >= 0) {
                const double dx = t.x - qx, dy = t.y - qy, dz = t.z - qz;
                constexpr double cap = ICP_MATCH_RINGS * IcpLevel<0>::cell;
                int tx, ty, tz;
                if (!first || (dx * dx + dy * dy + dz * dz <= cap * cap && icp_cell_of<0>(t.x, t.y, t.z, tx, ty, tz)))
                    offer(make_float4(t.x, t.y, t.z, __int_as_float(prev)));
            }
            int fx, fy, fz;
            bool settled = false;
            double reach = 0.0;                             // every target within this distance has been offered (or culled)
            auto merged = [&]() {                           // the group's nearest, its `other`, the bound that follows
                icp_group_merge<SG>(b, other);
                if (b.idx >= 0) {
                    const double s1 = (double)(sqrtf((float)b.d2) * 1.000001f) + 1e-30;
                    bnd = (b.d2 + 2.0 * mg * s1 + mg * mg) * (1.0 + 1e-12) + 1e-300;
                }
            };
            const bool fine_done = icp_fine_box(qx, qy, qz, fx, fy, fz);
            if (fine_done)
                for (int r = 0; r <= ICP_FINE_RINGS && !settled; ++r) {
                    icp_visit_shell<1, SG>(a.g[1], sl1, fx, fy, fz, r, qx, qy, qz, sub, bound, offer);
                    merged();
                    reach = r * IcpLevel<1>::cell;
                    settled = b.idx >= 0 && b.d2 <= reach * reach;
                }
            // (after a complete fine pass the coarse rings 0 and 1 hold nothing new: that cube reaches 0.75 m from the centre of the
            // query's coarse cell, i.e. at most 0.875 m from the centre of its fine cell, and the fine box reached 1.125 m)
            // -- for cubic shells.  In rows mode rings 0 and 1 are the rows that reach beyond that cube in x: they stay.
            const int r_first = (fine_done && b.idx >= 0 && bnd <= 4.0) ? 2 : 0;
            if (!settled && in_coarse)
                for (int r = r_first; r <= ICP_MATCH_RINGS && !settled; ++r) {
                    // (no candidate yet: rows mode -- nothing can be culled, and the cube is covered by rows alone; with a candidate:
                    // cubic shells, culled against it.  Rows first, shells later keeps every inner cube covered.  Re-offering a point is harmless)
                    // (the same while the candidate is more than 2 m away: the bound culls next to nothing then -- a max-range return
                    // whose partner is 3.9 m away walked all 1650 cells, 74 us, and a late pass lasts as long as its slowest query --
                    // with the rows as wide as the bound reaches)
                    int rx = 0;
                    if (b.idx < 0) rx = ICP_MATCH_RINGS;
                    else if (bnd > 4.0) { rx = (int)(sqrt(bnd) * (1.0 / IcpLevel<0>::cell)) + 1; rx = rx > ICP_MATCH_RINGS ? ICP_MATCH_RINGS : rx; }
                    icp_visit_shell<0, SG>(a.g[0], sl0, cx, cy, cz, r, qx, qy, qz, sub, bound, offer, rx);
                    merged();
                    reach = r * IcpLevel<0>::cell;
                    settled = b.idx >= 0 && b.d2 <= reach * reach;
                }
            if (sub == 0) {
                a.nn_prev[p] = b.idx;
                a.nn_cell[p] = in_coarse ? icp_cell_index(cx, cy, cz) : -1;
                double slack;
                const double oth = sqrt(other);             // (1e150 if nothing else was seen)
                if (b.idx >= 0) {
                    // every other target is at least min(other, d1 + margin, reach) away: half of what that leaves above d1
                    const double d1 = sqrt(b.d2);
                    double far = d1 + mg;
                    far = oth < far ? oth : far;
                    far = reach < far ? reach : far;
                    slack = 0.5 * (far - d1);
                } else {
                    // nothing within the cap: every target of the box is at least min(other, cap + margin) away
                    double far = max_dist + mg;
                    far = oth < far ? oth : far;
                    slack = far - max_dist;
                }
                slack = slack * (1.0 - 1e-6) - 1e-9;
                a.nn_slack[p] = slack > 0.0 ? (float)(slack * (1.0 - 1e-6)) : 0.f;
            }
        }
    }
    if (a.dbg & 1) { const uint64_t m = __ballot(need && sub == 0); if ((threadIdx.x & 63) == 0 && m) { const int it = (int)a.state[21]; atomicAdd(&a.state[48 + (it < 15 ? it : 15)], (double)__popcll(m)); } }
    // the query's row (one lane of its group)
    if (sub == 0) {
        uint32_t flag = 0u;
        double J[6] = {0, 0, 0, 0, 0, 0}, r = 0.0;
        if (b.idx >= 0) {
            flag = 1u;                                      // Open3D: fitness / rmse over all correspondences
            const float4 nn = reinterpret_cast<const float4 *>(a.normal)[b.idx];
            if (nn.w != 0.f) {                              // (no normal: the pair carries no point-to-plane row)
                flag = 3u;
                const double nx = nn.x, ny = nn.y, nz = nn.z;
                r = (qx - b.x) * nx + (qy - b.y) * ny + (qz - b.z) * nz;
                J[0] = qy * nz - qz * ny; J[1] = qz * nx - qx * nz; J[2] = qx * ny - qy * nx; J[3] = nx; J[4] = ny; J[5] = nz;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) s_row[ql][i] = J[i];
        s_row[ql][6] = r;
        s_row[ql][7] = b.d2;
        s_flag[ql] = flag;
    }
    __syncthreads();
    // accumulator k (the 21 products of J^T J row-wise, the 6 of J^T r, |q - t|^2, pairs, r^2) of four queries per thread
    {
        const int k = threadIdx.x & 31, part = threadIdx.x >> 5;        // 8 parts of ICP_QPW / 8 queries
        int i = 0, j = 0;                                   // k < 21: the pair (i, j), i <= j, row-wise
        {
            int kk = k;
            for (i = 0; i < 6 && kk >= 6 - i; ++i) kk -= 6 - i;
            j = i + kk;
        }
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < ICP_QPW / 8; ++u) {
            const int q = part * (ICP_QPW / 8) + u;
            const uint32_t f = s_flag[q];
            double term = 0.0;
            if (k < 21) term = (f & 2u) ? s_row[q][i] * s_row[q][j] : 0.0;
            else if (k < 27) term = (f & 2u) ? s_row[q][k - 21] * s_row[q][6] : 0.0;
            else if (k == 27) term = (f & 1u) ? s_row[q][7] : 0.0;
            else if (k == 28) term = (f & 1u) ? 1.0 : 0.0;
            else if (k == 29) term = (f & 2u) ? s_row[q][6] * s_row[q][6] : 0.0;
            acc += term;
        }
        s_part[part][k] = acc;
    }
    __syncthreads();
    if (threadIdx.x < ICP_NACC) {
        double v = 0.0;
#pragma unroll
        for (int part = 0; part < 8; ++part) v += s_part[part][threadIdx.x];
        a.partial[(size_t)threadIdx.x * a.grid + blockIdx.x] = v;   // [accumulator][workgroup]: icp_solve reads rows of it coalesced
    }
}

// icp_solve: workgroup k adds up accumulator k over the workgroups of icp_match in a fixed order (thread t: columns t, t + 256,
// ... ascending; lanes: butterfly; waves: ascending); the last workgroup to finish takes the Gauss-Newton step.  (One
// workgroup reading all thirty rows -- 450 KB through one CU's 64 B per clock -- and then solving took 30 us.)
#define ICP_SOLVE_THREADS 256
__global__ __launch_bounds__(ICP_SOLVE_THREADS) void icp_solve(const IcpArgs a)
{
    __shared__ double s_red[ICP_SOLVE_THREADS / 64];
    __shared__ double s_sum[32];
    __shared__ int s_last;
    if (a.state[20] != 0.0) return;
    IcpPrev pv;
    if (threadIdx.x < 64) {                                 // (every lane of wave 0 holds them: the values are uniform)
#pragma unroll
        for (int i = 0; i < 12; ++i) pv.T[i] = a.state[i];
        pv.fit = a.state[18]; pv.rmse = a.state[19]; pv.iters = a.state[21];
    }
    const int k = blockIdx.x;
    const double *row = a.partial + (size_t)k * a.grid;
    double acc = 0.0;
    for (int b0 = 0; b0 < a.grid; b0 += 4 * ICP_SOLVE_THREADS) {        // four loads in flight, added in ascending order
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int b = b0 + u * ICP_SOLVE_THREADS + (int)threadIdx.x; v[u] = b < a.grid ? row[b] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) s_red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = 0.0;
        for (int w = 0; w < ICP_SOLVE_THREADS / 64; ++w) v += s_red[w];
        double *sums = a.state + 64;
        __hip_atomic_store(&sums[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();                                    // release at agent scope: the sum, then the arrival
        const uint32_t t = atomicAdd(a.arrived, 1u);
        s_last = (t == gridDim.x - 1);
        if (s_last) __hip_atomic_store(a.arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                        // acquire: the sums the other XCDs released
    if (threadIdx.x < 32) s_sum[threadIdx.x] = threadIdx.x < ICP_NACC ? __hip_atomic_load(a.state + 64 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    __syncthreads();
    if (threadIdx.x < 64) icp_solve_step(a, s_sum, pv);
}

// state block and "first pass" marks in one launch (a 512-byte copy from pageable memory and a memset before)
struct IcpInit { double st[32]; };
__global__ __launch_bounds__(ICP_THREADS) void icp_init(const IcpArgs a, const IcpInit in)
{
    const int p = blockIdx.x * ICP_THREADS + threadIdx.x;
    if (p < a.n_src) a.nn_prev[p] = (int)0xfefefefe;        // "first pass" (see icp_match)
    if (blockIdx.x == 0 && threadIdx.x < 64) a.state[threadIdx.x] = threadIdx.x < 32 ? in.st[threadIdx.x] : 0.0;
}

extern "C" {

static inline int64_t icp_align(int64_t v) { return (v + 255) & ~255ll; }
static inline int icp_grid(int n, int sg = 16) { const int qpw = ICP_THREADS / sg; return (n + qpw - 1) / qpw; }   // workgroups of icp_match = columns of partial sums

int64_t pca_icp_workspace_bytes(int32_t max_points)
{
    if (max_points < 1) max_points = 1;
    return 2 * (icp_align(ICP_CELLS * 4) + icp_align((ICP_CELLS + 1) * 4) + icp_align((int64_t)max_points * 16)) +
           icp_align((int64_t)max_points * 16) + 3 * icp_align((int64_t)max_points * 4) +
           icp_align((int64_t)icp_grid(max_points) * ICP_NACC * 8) + icp_align((int64_t)((max_points + ICP_THREADS - 1) / ICP_THREADS) * 16) +
           icp_align(128 * 8) + 512;
}

int pca_icp_register(pca_ctx *ctx, const float *src_pts, int32_t n_src, const float *tgt_pts, int32_t n_tgt,
                     double max_corr_dist, const double init[16], int max_iter, double rel_fitness, double rel_rmse,
                     void *workspace, int64_t workspace_bytes, double T_out[16], double *fitness, double *rmse,
                     int *iterations, void *stream)
{
    if (!ctx) return -1;
    if (!src_pts || !tgt_pts || n_src < 1 || n_tgt < 1 || !workspace || !T_out) { ctx->err = "icp: bad arguments"; return -1; }
    if (workspace_bytes < pca_icp_workspace_bytes(n_tgt > n_src ? n_tgt : n_src)) { ctx->err = "icp: workspace too small"; return -1; }
    if (max_iter < 1) max_iter = 30;
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    IcpArgs a;
    a.src = src_pts; a.tgt = tgt_pts; a.n_src = n_src; a.n_tgt = n_tgt;
    char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    const int64_t cells = (int64_t)ICP_NX * ICP_NY * ICP_NZ;
    // the count tables belong to the context: the fill pass leaves them all zero, so only a first use (or a call that
    // failed half way) clears them -- 2 x 67 MB of memset per registration otherwise
    if (!ctx->icp_cnt) {
        PCA_CHECK(ctx, hipMalloc(&ctx->icp_cnt, (size_t)(2 * cells * 4)));
        ctx->icp_cnt_dirty = true;
    }
    if (!ctx->icp_host) {
        PCA_CHECK(ctx, hipHostMalloc(&ctx->icp_host, 32 * sizeof(double), hipHostMallocMapped));
        memset(ctx->icp_host, 0, 32 * sizeof(double));
        PCA_CHECK(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->icp_host_dev), ctx->icp_host, 0));
    }
    if (ctx->icp_cnt_dirty) PCA_CHECK(ctx, hipMemsetAsync(ctx->icp_cnt, 0, (size_t)(2 * cells * 4), s));
    ctx->icp_cnt_dirty = true;                              // until this call has come through
    for (int lv = 0; lv < 2; ++lv) {
        a.g[lv].cnt = ctx->icp_cnt + (size_t)lv * cells; w += icp_align(cells * 4);      // (the workspace's own slot stays unused)
        a.g[lv].start = reinterpret_cast<uint32_t *>(w); w += icp_align((cells + 1) * 4);
        a.g[lv].spts = reinterpret_cast<float4 *>(w); w += icp_align((int64_t)n_tgt * 16);
    }
    a.normal = reinterpret_cast<float *>(w); w += icp_align((int64_t)n_tgt * 16);
    a.nn_prev = reinterpret_cast<int32_t *>(w); w += icp_align((int64_t)n_src * 4);
    a.nn_cell = reinterpret_cast<int32_t *>(w); w += icp_align((int64_t)n_src * 4);
    a.nn_slack = reinterpret_cast<float *>(w); w += icp_align((int64_t)n_src * 4);
    { static int ns = -1; if (ns < 0) { const char *e = getenv("PCA_ICP_NO_SKIP"); ns = e ? atoi(e) : 0; } a.no_skip = ns; }
    { static int dg = -1; if (dg < 0) { const char *e = getenv("PCA_ICP_DBG"); dg = e ? atoi(e) : 0; } a.dbg = dg; }
    a.partial = reinterpret_cast<double *>(w); w += icp_align((int64_t)icp_grid(n_src) * ICP_NACC * 8);
    a.n_count_blocks = (n_tgt + ICP_THREADS - 1) / ICP_THREADS;
    a.zr_part = reinterpret_cast<uint32_t *>(w); w += icp_align((int64_t)a.n_count_blocks * 16);
    a.state = reinterpret_cast<double *>(w);
    a.zr = reinterpret_cast<uint32_t *>(a.state + 24);                 // initialised with the state block
    a.status = ctx->ticket + 1;
    a.arrived = reinterpret_cast<uint32_t *>(a.state + 27);            // zero with the state block
    a.max_dist2 = max_corr_dist * max_corr_dist;
    a.rel_fitness = rel_fitness; a.rel_rmse = rel_rmse;
    a.grid = icp_grid(n_src);
    a.host = ctx->icp_host_dev;
    a.pass = 0;
    // the word the host polls: tag of THIS call, nothing done yet.  (A kernel of an earlier registration that could still
    // store into the mirror does not exist: the passes behind the one that ended a registration leave at their first line.)
    ctx->icp_call = (ctx->icp_call + 1) & 0xffffffu;
    a.tag = (uint64_t)ctx->icp_call << 16;
    uint64_t *const tag_word = reinterpret_cast<uint64_t *>(ctx->icp_host + 31);
    __atomic_store_n(tag_word, a.tag, __ATOMIC_RELEASE);
    IcpInit in;
    memset(&in, 0, sizeof in);
    static const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < 16; ++i) in.st[i] = init ? init[i] : eye[i];
    {
        const uint32_t zr0[4] = {0xffffffffu, 0u, 0xffffffffu, 0u};       // zmin, zmax + 1 of both grids: nothing seen yet
        memcpy(&in.st[24], zr0, sizeof zr0);
    }
    for (int i = 0; i < 22; ++i) ctx->icp_host[i] = in.st[i];            // (a registration that ends before any update returns init)
    const int scan_tiles = (int)(cells / ICP_SCAN_TILE);
    if (pca_ctx_reserve_tiles(ctx, 2 * scan_tiles, s)) return -1;
    a.lb_state = ctx->tile_state;
    const dim3 per_point((n_tgt + ICP_THREADS - 1) / ICP_THREADS);
    PCA_LAUNCH(ctx, PCA_K_ICP, icp_init, dim3((n_src + ICP_THREADS - 1) / ICP_THREADS), dim3(ICP_THREADS), s, a, in);
    PCA_LAUNCH(ctx, PCA_K_ICP, icp_grid_count, per_point, dim3(ICP_THREADS), s, a);
    a.epoch = pca_ctx_next_epoch(ctx, s);
    PCA_LAUNCH(ctx, PCA_K_ICP, icp_cell_scan, dim3(2 * scan_tiles), dim3(ICP_SCAN_THREADS), s, a);
    PCA_LAUNCH(ctx, PCA_K_ICP, icp_grid_fill, per_point, dim3(ICP_THREADS), s, a);
    PCA_LAUNCH(ctx, PCA_K_ICP, icp_normals, dim3(((int64_t)n_tgt * ICP_SG + ICP_THREADS - 1) / ICP_THREADS), dim3(ICP_THREADS), s, a);
    // One more evaluation than updates: Open3D reports fitness / rmse of the final transform.  The host stays ICP_AHEAD passes
    // ahead of the device -- pass k is launched once pass k - ICP_AHEAD has reported in through the mapped word -- so the queue
    // never runs dry (a pass lasts 50-250 us, its two launches 10) and at most ICP_AHEAD no-op passes are in flight when the
    // flag comes up.  Round 4 copied the state back and waited every twelfth pass: three no-op passes per registration (30 us),
    // 45 us of idle GPU per look, and a copy + wait at the end.
    constexpr int ICP_AHEAD = 2;
    constexpr double ICP_POLL_TIMEOUT_S = 2.0;
    auto news = [&]() -> uint64_t {                         // low 16 bits of the word if it carries this call's tag, else 0
        const uint64_t t = __atomic_load_n(tag_word, __ATOMIC_ACQUIRE);
        return (t >> 16) == (a.tag >> 16) ? (t & 0xffffu) : 0u;
    };
    bool ended = false, timed_out = false;
    auto wait_for = [&](int passes_done) {                  // until that many passes have reported in, or the registration has ended
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0;; ++spins) {
            const uint64_t v = news();
            if (v & 0x100u) { ended = true; return; }
            if ((int)(v & 0xffu) >= passes_done) return;
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if ((spins & 1023u) == 1023u &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > ICP_POLL_TIMEOUT_S) { timed_out = true; return; }
        }
    };
    int launched = 0;
    for (int it = 0; it <= max_iter && !ended && !timed_out; ++it) {
        if (it >= ICP_AHEAD) wait_for(it - ICP_AHEAD + 1);
        if (ended || timed_out) break;
        if (it == max_iter) a.rel_fitness = a.rel_rmse = 1e300;              // last pass only evaluates
        a.pass = it;
        if (it == 0) { a.grid = icp_grid(n_src, 16); PCA_LAUNCH(ctx, PCA_K_ICP, icp_match<16>, dim3(a.grid), dim3(ICP_THREADS), s, a); }
        else { a.grid = icp_grid(n_src, 8); PCA_LAUNCH(ctx, PCA_K_ICP, icp_match<8>, dim3(a.grid), dim3(ICP_THREADS), s, a); }
        PCA_LAUNCH(ctx, PCA_K_ICP, icp_solve, dim3(ICP_NACC), dim3(ICP_SOLVE_THREADS), s, a);
        ++launched;
    }
    while (!ended && !timed_out) wait_for(launched + 1);    // (the last pass launched ends the registration: it only evaluates)
    double st[64] = {0};
    if (timed_out || (a.dbg & 1)) {
        // nothing heard for seconds (a launch that failed, a device in trouble), or the diagnostics' counters are wanted: the
        // plain way -- copy the state back behind everything on the stream and wait
        PCA_CHECK(ctx, hipMemcpyAsync(st, a.state, sizeof st, hipMemcpyDeviceToHost, s));
        PCA_CHECK(ctx, hipStreamSynchronize(s));
    } else {
        for (int i = 0; i < 22; ++i) st[i] = ctx->icp_host[i];           // (behind the acquire of the word that said "ended")
    }
    for (int i = 0; i < 12; ++i) T_out[i] = st[i];
    if (a.dbg & 1) { fprintf(stderr, "icp: searched queries per pass:"); for (int i = 0; i < 16; ++i) fprintf(stderr, " %.0f", st[48 + i]); fprintf(stderr, "\n"); }
    T_out[12] = T_out[13] = T_out[14] = 0.0; T_out[15] = 1.0;
    if (fitness) *fitness = st[16];
    if (rmse) *rmse = st[17];
    if (iterations) *iterations = (int)st[21];
    ctx->icp_cnt_dirty = timed_out;                         // the fill pass has run: every count is zero again
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
