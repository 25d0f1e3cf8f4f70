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
//                     one range, and a neighbour search streams them (four loads in flight)
//   icp_normals       per target point (8 lanes each): the K = 30 nearest neighbours (shell-by-shell grid search, exact
//                     within the search cap), covariance, eigenvector of the smallest eigenvalue (cyclic Jacobi, f64)
//   icp_match         per source point (8 lanes each): q = T p, nearest target point (exact within the cap; the previous
//                     iteration's partner bounds the search from the start)
//   icp_accumulate    per source point: r = (q - t).n, J = [q x n, n]; per-workgroup partial sums of J^T J, J^T r,
//                     |q - t|^2, inlier count.  The last workgroup to finish adds the partials up in workgroup order
//                     (deterministic) and takes the step: 6x6 Cholesky solve, T <- exp(x) T, fitness / rmse, convergence
//                     flag -- two launches per iteration, and the host looks at the flag only every sixth pass.
// Measured on two 120 k-point sweeps (round 2): 10.9 ms -> 2.8 ms per registration (normals 1.63 -> 0.67 ms, a pass
// 0.91 -> 0.17 ms); what changed is in the comments of icp_visit_shell, icp_match and icp_solve_step.
#include "pca_common.h"

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
#define ICP_NACC 30              // 21 (J^T J upper) + 6 (J^T r) + sum d^2 + inliers + sum r^2
#define ICP_CHECK_EVERY 6         // the host looks at the convergence flag after every 6th pass
#define ICP_MAX_GRID 512          // workgroups of icp_accumulate = rows of partial sums its last workgroup adds up

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
    double *nn_d2;               // [n_src] its squared distance (icp_match -> icp_accumulate)
    uint64_t *lb_state;          // decoupled look-back of the cell scan
    uint32_t *ticket;
    uint32_t epoch;
    int scan_level;              // grid the scan kernel works on
    double *partial;             // [grid][ICP_NACC]
    double *state;               // [0..15] T (row-major), [16] fitness, [17] rmse, [18] prev fitness, [19] prev rmse,
                                 // [20] converged flag, [21] iterations done
    uint32_t *arrived;           // workgroups of the running icp_accumulate that have written their partial sums
    double max_dist2;
    double rel_fitness, rel_rmse;
    int grid;
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

#define ICP_SCAN_THREADS 1024
#define ICP_SCAN_TILE (4 * ICP_SCAN_THREADS)

template <int LV>
__global__ __launch_bounds__(ICP_THREADS) void icp_grid_count(const IcpArgs a)
{
    const int p = blockIdx.x * ICP_THREADS + threadIdx.x;
    if (p >= a.n_tgt) return;
    const float4 v = reinterpret_cast<const float4 *>(a.tgt)[p];
    int cx, cy, cz;
    if (icp_cell_of<LV>(v.x, v.y, v.z, cx, cy, cz)) atomicAdd(&a.g[LV].cnt[icp_cell_index(cx, cy, cz)], 1u);   // else: not in this grid
}

// exclusive scan of the per-cell counts (single pass, decoupled look-back; tiles handed out by ticket)
__global__ __launch_bounds__(ICP_SCAN_THREADS) void icp_cell_scan(const IcpArgs a)
{
    __shared__ int s_tile;
    __shared__ uint32_t s_w[ICP_SCAN_THREADS / 64];
    __shared__ uint64_t s_excl;
    const IcpGrid &g = a.g[a.scan_level];
    const int n_tiles = (int)(ICP_CELLS / ICP_SCAN_TILE);
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(a.ticket, 1u);
        if ((int)t == n_tiles - 1) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tile = (int)t;
    }
    __syncthreads();
    const int tile = s_tile;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = (int64_t)tile * ICP_SCAN_TILE + threadIdx.x * 4;
    const uint4 c = *reinterpret_cast<const uint4 *>(g.cnt + base);
    const uint32_t tsum = c.x + c.y + c.z + c.w;
    const uint32_t inc = wave_incl_scan_add(tsum);
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const uint32_t v = lane < ICP_SCAN_THREADS / 64 ? s_w[lane] : 0u;
        const uint32_t winc = wave_incl_scan_add(v);
        if (lane < ICP_SCAN_THREADS / 64) s_w[lane] = winc - v;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)winc, 63);
        const uint64_t e = lb_exclusive_prefix(a.lb_state, tile, (uint64_t)total, a.epoch, a.ticket + 1);
        if (lane == 0) s_excl = (e << 32) | total;
    }
    __syncthreads();
    const uint32_t excl = (uint32_t)(s_excl >> 32), total = (uint32_t)s_excl;
    const uint32_t r0 = excl + s_w[wave] + (inc - tsum);
    *reinterpret_cast<uint4 *>(g.start + base) = make_uint4(r0, r0 + c.x, r0 + c.x + c.y, r0 + c.x + c.y + c.z);
    if (tile == n_tiles - 1 && threadIdx.x == ICP_SCAN_THREADS - 1) g.start[ICP_CELLS] = excl + total;
}

template <int LV>
__global__ __launch_bounds__(ICP_THREADS) void icp_grid_fill(const IcpArgs a)
{
    const int p = blockIdx.x * ICP_THREADS + threadIdx.x;
    if (p >= a.n_tgt) return;
    const float4 v = reinterpret_cast<const float4 *>(a.tgt)[p];
    int cx, cy, cz;
    if (!icp_cell_of<LV>(v.x, v.y, v.z, cx, cy, cz)) return;
    const IcpGrid &g = a.g[LV];
    const int cell = icp_cell_index(cx, cy, cz);
    const uint32_t pos = g.start[cell] + atomicSub(&g.cnt[cell], 1u) - 1u;     // fills the cell's range from the back
    g.spts[pos] = make_float4(v.x, v.y, v.z, __int_as_float(p));
}

// the records [s0, e) of a range of cells, four loads in flight (a one-record-per-trip loop is a chain of memory latencies:
// measured 1 ms per 120 k queries, independent of how much the search is culled)
// SGN > 1: SGN neighbouring lanes share one query; lane `sub` takes every SGN-th record, two loads in flight
template <int SGN, typename F>
__device__ __forceinline__ void icp_scan_range_sg(const float4 *spts, uint32_t s0, uint32_t e, int sub, F &&f)
{
    for (uint32_t q = s0 + (uint32_t)sub; q < e; q += 2 * SGN) {
        const uint32_t q1 = q + SGN;
        const float4 w0 = spts[q], w1 = spts[q1 < e ? q1 : q];
        f(w0);
        if (q1 < e) f(w1);
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
__device__ __forceinline__ void icp_visit_shell(const IcpGrid &g, int cx, int cy, int cz, int r, double qx, double qy,
                                                double qz, int sub, B &&bound, F &&f)
{
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
    if (r == 0) {
        const int c0 = icp_cell_index(cx, cy, cz);
        if (SGN == 1) icp_scan_range(g.spts, g.start[c0], g.start[c0 + 1], f);
        else icp_scan_range_sg<SGN>(g.spts, g.start[c0], g.start[c0 + 1], sub, f);
        return;
    }
    const int n = 2 * r + 1, m = 2 * r - 1;
    const int x_lo = cx - r < 0 ? 0 : cx - r, x_hi = cx + r >= ICP_NX ? ICP_NX - 1 : cx + r;
    for (int i0 = 0; i0 < 2 * n + 2 * m; i0 += SGN) {
        const int i = i0 + sub;
        uint32_t s0 = 0u, e0 = 0u;
        if (i < 2 * n + 2 * m) {
            int dz, dy;
            if (i < 2 * n) { dz = i < n ? -r : r; dy = (i < n ? i : i - n) - r; }
            else { const int j = i - 2 * n; dy = j < m ? -r : r; dz = (j < m ? j : j - m) - (r - 1); }
            const int z = cz + dz, y = cy + dy;
            if (z >= 0 && z < ICP_NZ && y >= 0 && y < ICP_NY) {
                const double ez = gap(qz, G::oz + z * G::cell), ey = gap(qy, G::oy + y * G::cell);
                if (ey * ey + ez * ez < bound()) {
                    const int c0 = icp_cell_index(x_lo, y, z);
                    s0 = g.start[c0];
                    e0 = g.start[c0 + (x_hi - x_lo) + 1];
                }
            }
        }
        scan_found(s0, e0);
    }
    const int ncell = 2 * m * m;
    for (int k0 = 0; k0 < ncell; k0 += 4 * SGN) {
        uint32_t s0[4], e0[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s0[u] = e0[u] = 0u;
            const int k = k0 + u * SGN + sub;
            if (k >= ncell) continue;
            const int mm = k >> 1, iz = mm / m;
            const int z = cz + iz - (r - 1), y = cy + (mm - iz * m) - (r - 1), x = cx + ((k & 1) ? r : -r);
            if (z < 0 || z >= ICP_NZ || y < 0 || y >= ICP_NY || x < 0 || x >= ICP_NX) continue;
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
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        if (off < 1e-300) break;
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
// records IT scanned (sorted, in LDS); at the end of a shell the group merges the eight lists far enough to know the
// K-th smallest distance of their union exactly (K steps of "smallest head"), which is the cull bound of the next shell
// and decides whether the search has settled.  The covariance pass deals the records out the same way and adds the
// eight partial sums up in a fixed butterfly.
__global__ __launch_bounds__(ICP_THREADS) void icp_normals(const IcpArgs a)
{
    __shared__ float s_d[ICP_K][ICP_THREADS];               // per lane: the K smallest squared distances it saw, ascending
    const int p = (blockIdx.x * ICP_THREADS + threadIdx.x) / ICP_SG;   // original index
    if (p >= a.n_tgt) return;
    const int sub = threadIdx.x & (ICP_SG - 1);
    const int base = (int)(threadIdx.x & 63) & ~(ICP_SG - 1);
    const float4 v = reinterpret_cast<const float4 *>(a.tgt)[p];
    float4 out = make_float4(0.f, 0.f, 1.f, 0.f);
    const int t = threadIdx.x;
    int found = 0;                                          // entries of this lane's list
    float kth = __builtin_huge_valf();                      // K-th smallest distance of the group so far (inf: fewer than K)
    auto bound_k = [&]() { return kth < __builtin_huge_valf() ? (double)kth * (1.0 + 1e-6) + 1e-12 : 1e300; };
    auto offer = [&](float d2) {                            // insertion into the sorted list of the lane's K smallest
        if (d2 > kth || (found == ICP_K && d2 >= s_d[ICP_K - 1][t])) return;
        int i = found < ICP_K ? found : ICP_K - 1;
        while (i > 0 && s_d[i - 1][t] > d2) { s_d[i][t] = s_d[i - 1][t]; --i; }
        s_d[i][t] = d2;
        if (found < ICP_K) ++found;
    };
    auto merge_kth = [&]() {                                // exact K-th smallest of the union of the eight lists
        if (icp_group_sum((uint32_t)found) < ICP_K) return;
        int head = 0;
        float g = 0.f;
        for (int step = 0; step < ICP_K; ++step) {
            const float mine = head < found ? s_d[head][t] : __builtin_huge_valf();
            g = icp_group_min(mine);
            const uint32_t eq = (uint32_t)(__ballot(mine == g) >> base) & 0xffu;      // one of the equal heads advances
            if (sub == __ffs(eq) - 1) ++head;
        }
        kth = g;
    };
    auto dist2 = [&](const float4 w) { const float dx = w.x - v.x, dy = w.y - v.y, dz = w.z - v.z; return dx * dx + dy * dy + dz * dz; };
    // pass A: the K-th smallest distance.  Fine grid first; every unvisited point is farther than r cells, so the K-th
    // distance is final once it lies inside that radius.
    int fx, fy, fz, cx, cy, cz;
    const bool fine = icp_fine_box(v.x, v.y, v.z, fx, fy, fz);
    bool settled = false;
    if (fine)
        for (int r = 0; r <= ICP_FINE_RINGS && !settled; ++r) {
            icp_visit_shell<1, ICP_SG>(a.g[1], fx, fy, fz, r, v.x, v.y, v.z, sub, bound_k, [&](const float4 w) { offer(dist2(w)); });
            merge_kth();
            const float lim = (float)(r * IcpLevel<1>::cell);
            settled = kth <= lim * lim;
        }
    const bool coarse = icp_cell_of<0>(v.x, v.y, v.z, cx, cy, cz);
    if (!settled && coarse)
        for (int r = 0; r <= ICP_NORMAL_RINGS && !settled; ++r) {
            icp_visit_shell<0, ICP_SG>(a.g[0], cx, cy, cz, r, v.x, v.y, v.z, sub, bound_k, [&](const float4 w) {
                if (fine && icp_in_fine_box(w, fx, fy, fz)) return;             // already offered by the fine pass
                offer(dist2(w));
            });
            merge_kth();
            const float lim = (float)(r * IcpLevel<0>::cell);
            settled = kth <= lim * lim;
        }
    const uint32_t total = icp_group_sum((uint32_t)found);
    if (total >= 3) {
        // the search radius of pass B: the K-th distance, or -- fewer than K points inside the search cap -- the largest
        float lim = kth;
        if (!(lim < __builtin_huge_valf())) {
            const float mine = found > 0 ? -s_d[found - 1][t] : __builtin_huge_valf();
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
            for (int r = 0; r <= rmax && r <= ICP_FINE_RINGS; ++r) icp_visit_shell<1, ICP_SG>(a.g[1], fx, fy, fz, r, v.x, v.y, v.z, sub, bound_l, add);
        } else if (coarse) {
            const int rmax = (int)ceil(sqrt((double)lim) / IcpLevel<0>::cell);
            for (int r = 0; r <= rmax && r <= ICP_NORMAL_RINGS; ++r) icp_visit_shell<0, ICP_SG>(a.g[0], cx, cy, cz, r, v.x, v.y, v.z, sub, bound_l, add);
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
            double n[3];
            icp_smallest_eigvec(A, n);
            const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            if (len > 0) out = make_float4((float)(n[0] / len), (float)(n[1] / len), (float)(n[2] / len), 1.f);
        }
    }
    if (sub == 0) reinterpret_cast<float4 *>(a.normal)[p] = out;
}

// Gauss-Newton step of one evaluation: fitness / rmse of the current transform, convergence test, 6x6 Cholesky solve,
// T <- exp(x) T.  One thread.
__device__ void icp_solve_step(const IcpArgs &a, const double *sum)
{
    double *S = a.state;
    const double inl = sum[28];
    const double fitness = a.n_src > 0 ? inl / a.n_src : 0.0;
    const double rmse = inl > 0 ? sqrt(sum[27] / inl) : 0.0;
    // Open3D evaluates fitness / rmse of the CURRENT transform, then updates; convergence compares successive evaluations
    const bool first = S[21] == 0.0;
    S[16] = fitness; S[17] = rmse;
    if (!first && fabs(S[18] - fitness) < a.rel_fitness && fabs(S[19] - rmse) < a.rel_rmse) { S[20] = 1.0; return; }
    S[18] = fitness; S[19] = rmse;
    if (inl < 6) { S[20] = 1.0; return; }
    // solve (J^T J) x = -J^T r  (Cholesky, upper triangle stored row-wise in sum[0..20]).  Every loop has a constant trip
    // count and no early exit, so the 6x6 system lives in registers (indexed dynamically it sat in scratch memory: ~100
    // dependent scratch round trips, most of the kernel's 55 us)
    double A[6][6], b[6], x[6];
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
            if (i == j) { ok = ok && (s > 1e-12); A[i][i] = sqrt(ok ? s : 1.0); }
            else A[i][j] = s / A[j][j];
        }
    }
    if (!ok) { S[20] = 1.0; return; }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int m = 0; m < i; ++m) s -= A[i][m] * x[m];
        x[i] = s / A[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = x[i];
#pragma unroll
        for (int m = i + 1; m < 6; ++m) s -= A[m][i] * x[m];
        x[i] = s / A[i][i];
    }
    // x = (alpha, beta, gamma, tx, ty, tz): R = Rz(gamma) Ry(beta) Rx(alpha)
    const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
    const double U[12] = {cg * cb, cg * sb * sa - sg * ca, cg * sb * ca + sg * sa, x[3],
                          sg * cb, sg * sb * sa + cg * ca, sg * sb * ca - cg * sa, x[4],
                          -sb, cb * sa, cb * ca, x[5]};
    double N[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            N[4 * i + j] = U[4 * i] * S[j] + U[4 * i + 1] * S[4 + j] + U[4 * i + 2] * S[8 + j] + (j == 3 ? U[4 * i + 3] : 0.0);
#pragma unroll
    for (int i = 0; i < 12; ++i) S[i] = N[i];
    S[21] += 1.0;
}

// icp_match: ICP_SG neighbouring lanes share one source point.  The own cell's records and the rows of every further
// shell are dealt out over the lanes, each lane keeps its own best partner, and at the end of every shell the lanes agree
// on the group's best (DPP butterfly inside the 8 lanes), which tightens everybody's cull bound and decides whether the
// search has settled.  One lane per query was a chain of dependent loads and row walks as long as the worst query of its
// wave (~0.8 ms per pass at < 2 waves per SIMD).  Ties: lowest ORIGINAL index (the order inside a cell varies from run
// to run).
__device__ __forceinline__ void icp_group_best(double &best, int &bidx)
{
#define ICP_BEST_STAGE(S)                                                                                              \
    {                                                                                                                  \
        const uint32_t lo = lane_xor_fetch<S>((uint32_t)__double2loint(best)), hi = lane_xor_fetch<S>((uint32_t)__double2hiint(best)); \
        const int oi = (int)lane_xor_fetch<S>((uint32_t)bidx);                                                         \
        const double od = __hiloint2double((int)hi, (int)lo);                                                          \
        if (oi >= 0 && (bidx < 0 || od < best || (od == best && oi < bidx))) { best = od; bidx = oi; }                 \
    }
    ICP_BEST_STAGE(1) ICP_BEST_STAGE(2) ICP_BEST_STAGE(4)
#undef ICP_BEST_STAGE
}

__global__ __launch_bounds__(ICP_THREADS) void icp_match(const IcpArgs a)
{
    if (a.state[20] != 0.0) return;                         // converged: the remaining passes are no-ops
    const double *T = a.state;
    const int sub = threadIdx.x & (ICP_SG - 1);
    const int p = (blockIdx.x * ICP_THREADS + threadIdx.x) / ICP_SG;
    if (p >= a.n_src) return;
    const float4 v = reinterpret_cast<const float4 *>(a.src)[p];
    const double qx = T[0] * v.x + T[1] * v.y + T[2] * v.z + T[3];
    const double qy = T[4] * v.x + T[5] * v.y + T[6] * v.z + T[7];
    const double qz = T[8] * v.x + T[9] * v.y + T[10] * v.z + T[11];
    double best = a.max_dist2;
    int bidx = -1;
    auto bound = [&]() { return best * (1.0 + 1e-12) + 1e-300; };
    auto offer = [&](const float4 w) {
        const double dx = w.x - qx, dy = w.y - qy, dz = w.z - qz;
        const double d2 = dx * dx + dy * dy + dz * dz;
        const int wi = __float_as_int(w.w);
        if (d2 < best || (d2 == best && bidx >= 0 && wi < bidx)) { best = d2; bidx = wi; }
    };
    // warm start: the previous iteration's partner bounds the search from the first cell on (the transform moved by a
    // fraction of a cell), so nearly every cell is culled; the result is still the exact nearest neighbour
    // (first pass: the target point of the same index -- consecutive sweeps share their scan order -- provided the search
    // below would reach it: inside the coarse grid and within the match cap; any such point is a valid upper bound)
    int prev = a.nn_prev[p];
    const bool first = prev == (int)0xfefefefe;
    if (first) prev = p < a.n_tgt ? p : -1;
    if (prev >= 0) {
        const float4 t = reinterpret_cast<const float4 *>(a.tgt)[prev];
        const double dx = t.x - qx, dy = t.y - qy, dz = t.z - qz;
        constexpr double cap = ICP_MATCH_RINGS * IcpLevel<0>::cell;
        int tx, ty, tz;
        if (!first || (dx * dx + dy * dy + dz * dz <= cap * cap && icp_cell_of<0>(t.x, t.y, t.z, tx, ty, tz)))
            offer(make_float4(t.x, t.y, t.z, __int_as_float(prev)));
    }
    int fx, fy, fz, cx, cy, cz;
    bool settled = false;
    if (icp_fine_box(qx, qy, qz, fx, fy, fz))
        for (int r = 0; r <= ICP_FINE_RINGS && !settled; ++r) {
            icp_visit_shell<1, ICP_SG>(a.g[1], fx, fy, fz, r, qx, qy, qz, sub, bound, offer);
            icp_group_best(best, bidx);
            settled = bidx >= 0 && best <= (r * IcpLevel<1>::cell) * (r * IcpLevel<1>::cell);
        }
    if (!settled && icp_cell_of<0>(qx, qy, qz, cx, cy, cz))
        for (int r = 0; r <= ICP_MATCH_RINGS && !settled; ++r) {
            icp_visit_shell<0, ICP_SG>(a.g[0], cx, cy, cz, r, qx, qy, qz, sub, bound, offer);   // re-offering a point is harmless
            icp_group_best(best, bidx);
            settled = bidx >= 0 && best <= (r * IcpLevel<0>::cell) * (r * IcpLevel<0>::cell);
        }
    if (sub == 0) { a.nn_prev[p] = bidx; a.nn_d2[p] = best; }
}

// icp_accumulate: one thread per source point turns its correspondence into a point-to-plane row; fixed-order reduction:
// lanes (butterfly), waves (serial), workgroups (serial, by the last one to arrive) -> deterministic; that last workgroup
// also takes the Gauss-Newton step, so an iteration is two launches and the loop never leaves the device.
__global__ __launch_bounds__(ICP_THREADS) void icp_accumulate(const IcpArgs a)
{
    __shared__ double s_red[ICP_THREADS / 64][ICP_NACC];
    __shared__ double s_sum[ICP_NACC];
    __shared__ int s_last;
    if (a.state[20] != 0.0) return;
    double acc[ICP_NACC];
#pragma unroll
    for (int k = 0; k < ICP_NACC; ++k) acc[k] = 0.0;
    const double *T = a.state;
    for (int p = blockIdx.x * ICP_THREADS + threadIdx.x; p < a.n_src; p += a.grid * ICP_THREADS) {
        const int bidx = a.nn_prev[p];
        if (bidx < 0) continue;
        const float4 v = reinterpret_cast<const float4 *>(a.src)[p];
        const float4 w = reinterpret_cast<const float4 *>(a.tgt)[bidx];
        const float4 nn = reinterpret_cast<const float4 *>(a.normal)[bidx];
        const double qx = T[0] * v.x + T[1] * v.y + T[2] * v.z + T[3];
        const double qy = T[4] * v.x + T[5] * v.y + T[6] * v.z + T[7];
        const double qz = T[8] * v.x + T[9] * v.y + T[10] * v.z + T[11];
        acc[27] += a.nn_d2[p];                              // Open3D: fitness / rmse over all correspondences
        acc[28] += 1.0;
        if (nn.w == 0.f) continue;                          // no normal: the pair carries no point-to-plane row
        const double nx = nn.x, ny = nn.y, nz = nn.z;
        const double r = (qx - w.x) * nx + (qy - w.y) * ny + (qz - w.z) * nz;
        const double J[6] = {qy * nz - qz * ny, qz * nx - qx * nz, qx * ny - qy * nx, nx, ny, nz};
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) acc[k++] += J[i] * J[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[21 + i] += J[i] * r;
        acc[29] += r * r;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto block_sum = [&](double *out, bool to_global) {     // acc[] of the workgroup -> out[0..NACC)
#pragma unroll
        for (int k = 0; k < ICP_NACC; ++k) {
            double v = acc[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) s_red[wave][k] = v;
        }
        __syncthreads();
        if (threadIdx.x < ICP_NACC) {
            double v = 0.0;
            for (int w = 0; w < ICP_THREADS / 64; ++w) v += s_red[w][threadIdx.x];
            out[threadIdx.x] = v;
            if (to_global) __threadfence();                 // release at agent scope by the (one) wave that wrote the row
        }
    };
    // the row of partial sums is released by the wave that wrote it; the arrival counter follows after the barrier
    block_sum(a.partial + (size_t)blockIdx.x * ICP_NACC, true);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(a.arrived, 1u);
        s_last = (t == gridDim.x - 1);
        if (s_last) __hip_atomic_store(a.arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                        // acquire: the rows the other XCDs released
    // thread t adds up the rows t, t + 256, ... (ascending), then the same lanes -> waves reduction: a fixed order
#pragma unroll
    for (int k = 0; k < ICP_NACC; ++k) acc[k] = 0.0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += ICP_THREADS) {
        const double2 *row = reinterpret_cast<const double2 *>(a.partial + (size_t)b * ICP_NACC);
        double2 rv[ICP_NACC / 2];
#pragma unroll
        for (int k = 0; k < ICP_NACC / 2; ++k) rv[k] = row[k];
#pragma unroll
        for (int k = 0; k < ICP_NACC / 2; ++k) { acc[2 * k] += rv[k].x; acc[2 * k + 1] += rv[k].y; }
    }
    __syncthreads();
    block_sum(s_sum, false);
    __syncthreads();
    if (threadIdx.x == 0) icp_solve_step(a, s_sum);
}

extern "C" {

static inline int64_t icp_align(int64_t v) { return (v + 255) & ~255ll; }
static inline int icp_grid(int n) { const int g = (n + ICP_THREADS - 1) / ICP_THREADS; return g < 1 ? 1 : (g > ICP_MAX_GRID ? ICP_MAX_GRID : g); }

int64_t pca_icp_workspace_bytes(int32_t max_points)
{
    if (max_points < 1) max_points = 1;
    return 2 * (icp_align(ICP_CELLS * 4) + icp_align((ICP_CELLS + 1) * 4) + icp_align((int64_t)max_points * 16)) +
           icp_align((int64_t)max_points * 16) + icp_align((int64_t)max_points * 4) + icp_align((int64_t)max_points * 8) +
           icp_align((int64_t)ICP_MAX_GRID * ICP_NACC * 8) +
           icp_align(32 * 8) + 512;
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
    for (int lv = 0; lv < 2; ++lv) {
        a.g[lv].cnt = reinterpret_cast<uint32_t *>(w); w += icp_align(cells * 4);
        a.g[lv].start = reinterpret_cast<uint32_t *>(w); w += icp_align((cells + 1) * 4);
        a.g[lv].spts = reinterpret_cast<float4 *>(w); w += icp_align((int64_t)n_tgt * 16);
    }
    a.normal = reinterpret_cast<float *>(w); w += icp_align((int64_t)n_tgt * 16);
    a.nn_prev = reinterpret_cast<int32_t *>(w); w += icp_align((int64_t)n_src * 4);
    a.nn_d2 = reinterpret_cast<double *>(w); w += icp_align((int64_t)n_src * 8);
    a.partial = reinterpret_cast<double *>(w); w += icp_align((int64_t)ICP_MAX_GRID * ICP_NACC * 8);
    a.state = reinterpret_cast<double *>(w);
    a.arrived = reinterpret_cast<uint32_t *>(a.state + 24);            // zero with the state block
    a.max_dist2 = max_corr_dist * max_corr_dist;
    a.rel_fitness = rel_fitness; a.rel_rmse = rel_rmse;
    a.grid = icp_grid(n_src);
    double st[32] = {0};
    static const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < 16; ++i) st[i] = init ? init[i] : eye[i];
    PCA_CHECK(ctx, hipMemsetAsync(a.g[0].cnt, 0, (size_t)cells * 4, s));
    PCA_CHECK(ctx, hipMemsetAsync(a.g[1].cnt, 0, (size_t)cells * 4, s));
    PCA_CHECK(ctx, hipMemsetAsync(a.nn_prev, 0xfe, (size_t)n_src * 4, s));        // 0xfefefefe: "first pass" (see icp_match)
    PCA_CHECK(ctx, hipMemcpyAsync(a.state, st, sizeof st, hipMemcpyHostToDevice, s));
    const int scan_tiles = (int)(cells / ICP_SCAN_TILE);
    if (pca_ctx_reserve_tiles(ctx, scan_tiles, s)) return -1;
    a.lb_state = ctx->tile_state;
    a.ticket = ctx->ticket;
    const dim3 per_point((n_tgt + ICP_THREADS - 1) / ICP_THREADS);
    PCA_LAUNCH(ctx, PCA_K_ICP, icp_grid_count<0>, per_point, dim3(ICP_THREADS), s, a);
    PCA_LAUNCH(ctx, PCA_K_ICP, icp_grid_count<1>, per_point, dim3(ICP_THREADS), s, a);
    for (int lv = 0; lv < 2; ++lv) {
        a.scan_level = lv;
        a.epoch = pca_ctx_next_epoch(ctx, s);
        PCA_LAUNCH(ctx, PCA_K_ICP, icp_cell_scan, dim3(scan_tiles), dim3(ICP_SCAN_THREADS), s, a);
    }
    PCA_LAUNCH(ctx, PCA_K_ICP, icp_grid_fill<0>, per_point, dim3(ICP_THREADS), s, a);
    PCA_LAUNCH(ctx, PCA_K_ICP, icp_grid_fill<1>, per_point, dim3(ICP_THREADS), s, a);
    PCA_LAUNCH(ctx, PCA_K_ICP, icp_normals, dim3(((int64_t)n_tgt * ICP_SG + ICP_THREADS - 1) / ICP_THREADS), dim3(ICP_THREADS), s, a);
    // one more evaluation than updates: Open3D reports fitness / rmse of the final transform.  The loop never leaves the
    // device inside a group of ICP_CHECK_EVERY passes; between groups the host reads the convergence flag (a pass after
    // convergence is a no-op, the check only saves launching the rest of the 31)
    for (int it = 0; it <= max_iter; ++it) {
        if (it == max_iter) a.rel_fitness = a.rel_rmse = 1e300;              // last pass only evaluates
        PCA_LAUNCH(ctx, PCA_K_ICP, icp_match, dim3(((int64_t)n_src * ICP_SG + ICP_THREADS - 1) / ICP_THREADS), dim3(ICP_THREADS), s, a);
        PCA_LAUNCH(ctx, PCA_K_ICP, icp_accumulate, dim3(a.grid), dim3(ICP_THREADS), s, a);
        if (it % ICP_CHECK_EVERY == ICP_CHECK_EVERY - 1 && it < max_iter) {
            PCA_CHECK(ctx, hipMemcpyAsync(st, a.state, sizeof st, hipMemcpyDeviceToHost, s));
            PCA_CHECK(ctx, hipStreamSynchronize(s));
            if (st[20] != 0.0) break;
        }
    }
    PCA_CHECK(ctx, hipMemcpyAsync(st, a.state, sizeof st, hipMemcpyDeviceToHost, s));
    PCA_CHECK(ctx, hipStreamSynchronize(s));
    for (int i = 0; i < 12; ++i) T_out[i] = st[i];
    T_out[12] = T_out[13] = T_out[14] = 0.0; T_out[15] = 1.0;
    if (fitness) *fitness = st[16];
    if (rmse) *rmse = st[17];
    if (iterations) *iterations = (int)st[21];
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
