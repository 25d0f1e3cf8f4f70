// pca_host.hip -- host-only part of the C ABI (no device work): the per-frame bookkeeping of the accumulator's pose track.
//
// integrate() of the reference does, per frame, on Python lists (sem_pc_accum.py:156-228): re-express every stored pose in
// the new ego frame (update_poses), append the new pose, append the newest path segment, sum the path, and drop the oldest
// frames beyond the horizon (remove_observations / comp_incr_path_dist).  pca_amd/host_logic.py:PoseTrack keeps those as
// numpy expressions (~30 us of interpreter per frame); this is the same arithmetic in one call.
//
// Bit equality with the numpy form is by construction, not by imitation:
//   * the two matrix products -- (4,4) @ (4,1) per pose and tri(n) @ d -- are issued through the SAME cblas_dgemv numpy
//     itself calls for them (the caller hands over the entry point of the OpenBLAS that numpy has loaded);
//   * np.sum is numpy's pairwise summation (8 interleaved partial sums per block of <= 128), restated below;
//   * everything else is single IEEE operations.
// pca_amd/host_logic.py checks both against numpy on random data when it binds the library and keeps the numpy form if
// anything differs (or if the BLAS entry point cannot be found).
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <chrono>
#include <unistd.h>
#include <hip/hip_runtime.h>

#include "../../include/pca.h"
#include "pca_common.h"
#include "pca_stage_pool.h"

typedef void (*cblas_dgemv64_fn)(int order, int trans, int64_t m, int64_t n, double alpha, const double *A, int64_t lda,
                                 const double *x, int64_t incx, double beta, double *y, int64_t incy);
#define CBLAS_ROW_MAJOR 101
#define CBLAS_NO_TRANS 111

// y = T @ x for one 4x4 row-major T.  numpy issues one dgemv per pose; through the BLAS interface that is ~150 ns of call
// overhead per pose (30 us for a 200-frame window).  What the dgemv kernel computes for a 4-long dot product is a fixed
// short expression, but WHICH one depends on the kernel OpenBLAS picked for this CPU.  The candidates below are tried
// against numpy when the library is bound (pca_amd/host_logic.py); mode 0 -- the real dgemv -- stays if none matches.
static int g_gemv4_mode = 0;
static inline void gemv4(cblas_dgemv64_fn gemv, int mode, const double *T, const double *x, double *y)
{
    if (mode == 0) {
        y[0] = y[1] = y[2] = y[3] = 0.0;
        gemv(CBLAS_ROW_MAJOR, CBLAS_NO_TRANS, 4, 4, 1.0, T, 4, x, 1, 0.0, y, 1);
        return;
    }
    for (int r = 0; r < 4; ++r) {
        const double *t = T + 4 * r;
        const double p0 = t[0] * x[0], p1 = t[1] * x[1], p2 = t[2] * x[2], p3 = t[3] * x[3];
        switch (mode) {
        case 1: y[r] = (p0 + p2) + (p1 + p3); break;                           // 4 lanes multiplied, folded high onto low
        case 2: y[r] = fma(t[2], x[2], p0) + fma(t[3], x[3], p1); break;       // two 2-lane fma accumulators
        case 3: y[r] = fma(t[3], x[3], fma(t[2], x[2], fma(t[1], x[1], p0))); break;   // scalar fma chain
        case 4: y[r] = ((p0 + p1) + p2) + p3; break;                           // scalar, no fma
        default: y[r] = (p0 + p1) + (p2 + p3); break;                          // pairwise
        }
    }
}

struct pca_host_track {
    cblas_dgemv64_fn gemv = nullptr;
    std::vector<double> H;         // [cap][4]  homogeneous poses x, y, z, 1 of slots [head, head + n)
    std::vector<double> D;         // [cap]     segment distances of slots [head, head + n - 1)
    std::vector<double> tri;       // np.tri(tri_n), row-major
    std::vector<double> incr;      // scratch [n]
    std::vector<double> rows;      // scratch [8][n]: a group of rows of np.tri(n)
    int64_t tri_n = -1;
    int64_t head = 0, n = 0, nd = 0;
};

// numpy's pairwise summation of a contiguous f64 vector (numpy/_core/src/umath/loops_utils.h.src: pairwise_sum)
static double np_pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double r = 0.0;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

static void track_room(pca_host_track *t)
{
    // slots are consumed at the tail and released at the head: slide the live range to the front when the tail is reached
    const int64_t cap = (int64_t)t->D.size();
    if (t->head + t->n + 1 <= cap) return;
    if (t->n + 1 > cap / 2) {
        const int64_t ncap = cap * 2 > 1024 ? cap * 2 : 1024;
        std::vector<double> H2((size_t)ncap * 4), D2((size_t)ncap);
        memcpy(H2.data(), t->H.data() + 4 * t->head, sizeof(double) * 4 * (size_t)t->n);
        memcpy(D2.data(), t->D.data() + t->head, sizeof(double) * (size_t)t->nd);
        t->H.swap(H2); t->D.swap(D2);
    } else {
        memmove(t->H.data(), t->H.data() + 4 * t->head, sizeof(double) * 4 * (size_t)t->n);
        memmove(t->D.data(), t->D.data() + t->head, sizeof(double) * (size_t)t->nd);
    }
    t->head = 0;
}

static const double *track_tri(pca_host_track *t, int64_t n)
{
    if (t->tri_n != n) {
        t->tri.assign((size_t)(n * n), 0.0);
        for (int64_t i = 0; i < n; ++i)
            for (int64_t j = 0; j <= i; ++j) t->tri[(size_t)(i * n + j)] = 1.0;
        t->tri_n = n;
    }
    return t->tri.data();
}

// Rows [r0, r1) of np.tri(nd) @ D, r0 a multiple of 8, r1 - r0 <= 8 -- without the n x n matrix: the dgemv kernels walk the
// output rows in fixed groups from row 0, so a group of 8 comes out as in the full product (probed against numpy when the
// library is bound; mode 0 = always the full product).  A full 200 x 200 dgemv is ~9 us (OpenBLAS goes multi-threaded
// above ~9000 elements); eviction needs the first rows only, the trigger the last row and a crossing found by bisection.
static int g_incr_blocks = 0;
static void incr_block(pca_host_track *t, int64_t r0, int64_t r1, double *out)
{
    const int64_t n = t->nd, m = r1 - r0;
    t->rows.assign((size_t)(8 * n), 0.0);
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j <= r0 + i; ++j) t->rows[(size_t)(i * n + j)] = 1.0;
    t->gemv(CBLAS_ROW_MAJOR, CBLAS_NO_TRANS, m, n, 1.0, t->rows.data(), n, t->D.data() + t->head, 1, 0.0, out, 1);
}
// first index i with incr[i] - thr > 0 (the prefix sums of non-negative distances never decrease, rounded or not:
// bisection over the groups of 8 finds what a scan finds), or -1
static int64_t incr_first_above(pca_host_track *t, double thr)
{
    const int64_t n = t->nd, nb = (n + 7) / 8;
    double v[8];
    int64_t lo = 0, hi = nb;                                 // first group whose LAST row is above: in [lo, hi]
    while (lo < hi) {
        const int64_t mid = (lo + hi) / 2, r0 = 8 * mid, r1 = r0 + 8 < n ? r0 + 8 : n;
        incr_block(t, r0, r1, v);
        if (v[r1 - r0 - 1] - thr > 0.0) hi = mid; else lo = mid + 1;
    }
    if (lo >= nb) return -1;
    const int64_t r0 = 8 * lo, r1 = r0 + 8 < n ? r0 + 8 : n;
    incr_block(t, r0, r1, v);
    for (int64_t i = 0; i < r1 - r0; ++i)
        if (v[i] - thr > 0.0) return r0 + i;
    return -1;
}
static double incr_at(pca_host_track *t, int64_t i)
{
    double v[8];
    const int64_t r0 = 8 * (i / 8), r1 = r0 + 8 < t->nd ? r0 + 8 : t->nd;
    incr_block(t, r0, r1, v);
    return v[i - r0];
}

extern "C" {

// rows [r0, r1) of tri(n) @ d as the group form computes them (r0 % 8 == 0, r1 - r0 <= 8): what the binding compares with
// numpy's full product
int pca_host_incr_probe(void *cblas_dgemv64, const double *d, int64_t n, int64_t r0, int64_t r1, double *out)
{
    if (!cblas_dgemv64 || r0 % 8 || r1 - r0 > 8 || r1 > n || r0 < 0 || r1 <= r0) return -1;
    pca_host_track t;
    t.gemv = reinterpret_cast<cblas_dgemv64_fn>(cblas_dgemv64);
    t.D.assign(d, d + n); t.head = 0; t.nd = n;
    incr_block(&t, r0, r1, out);
    return 0;
}
// 1: eviction and trigger may use the group form; 0: always the full product.  Returns the previous setting.
int pca_host_incr_blocks(int on)
{
    const int old = g_incr_blocks;
    if (on == 0 || on == 1) g_incr_blocks = on;
    return old;
}

// (4,4) @ (4,) products for n pairs with candidate `mode` (0 = the real dgemv): what the binding compares with numpy
int pca_host_gemv4_probe(void *cblas_dgemv64, int mode, const double *T, const double *x, int64_t n, double *y)
{
    if (!cblas_dgemv64 || mode < 0 || mode > 5) return -1;
    for (int64_t i = 0; i < n; ++i) gemv4(reinterpret_cast<cblas_dgemv64_fn>(cblas_dgemv64), mode, T + 16 * i, x + 4 * i, y + 4 * i);
    return 0;
}
// selects the form pca_host_track_transform uses from now on (process-wide); returns the previous one
int pca_host_gemv4_mode(int mode)
{
    const int old = g_gemv4_mode;
    if (mode >= 0 && mode <= 5) g_gemv4_mode = mode;
    return old;
}

int pca_host_track_create(pca_host_track **out, void *cblas_dgemv64)
{
    if (!out || !cblas_dgemv64) return -1;
    pca_host_track *t = new pca_host_track();
    t->gemv = reinterpret_cast<cblas_dgemv64_fn>(cblas_dgemv64);
    t->H.resize(4 * 1024); t->D.resize(1024);
    *out = t;
    return 0;
}

void pca_host_track_destroy(pca_host_track *t) { delete t; }

int64_t pca_host_track_len(const pca_host_track *t) { return t ? t->n : 0; }
int64_t pca_host_track_n_segments(const pca_host_track *t) { return t ? t->nd : 0; }
const double *pca_host_track_poses(const pca_host_track *t) { return t ? t->H.data() + 4 * t->head : nullptr; }
const double *pca_host_track_segments(const pca_host_track *t) { return t ? t->D.data() + t->head : nullptr; }

// Replaces the whole content: poses [n][3], segment distances [nd]
int pca_host_track_set(pca_host_track *t, const double *poses, int64_t n, const double *segs, int64_t nd)
{
    if (!t || n < 0 || nd < 0) return -1;
    const int64_t need = (n > nd ? n : nd) + 1;
    if ((int64_t)t->D.size() < 2 * need) { t->H.assign((size_t)(8 * need), 0.0); t->D.assign((size_t)(2 * need), 0.0); }
    t->head = 0; t->n = n; t->nd = nd;
    for (int64_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) t->H[(size_t)(4 * i + k)] = poses[3 * i + k];
        t->H[(size_t)(4 * i + 3)] = 1.0;
    }
    if (nd) memcpy(t->D.data(), segs, sizeof(double) * (size_t)nd);
    return 0;
}

// update_poses (sem_pc_accum.py:156-165): every stored pose p <- (T @ [p, 1])[:3], one gemv per pose as numpy issues them
int pca_host_track_transform(pca_host_track *t, const double T[16])
{
    if (!t || !T) return -1;
    double *H = t->H.data() + 4 * t->head;
    const int mode = g_gemv4_mode;
    for (int64_t f = 0; f < t->n; ++f) {
        double y[4];
        gemv4(t->gemv, mode, T, H + 4 * f, y);
        H[4 * f + 0] = y[0]; H[4 * f + 1] = y[1]; H[4 * f + 2] = y[2]; H[4 * f + 3] = 1.0;
    }
    return 0;
}

int pca_host_track_append(pca_host_track *t, const double pose[3])
{
    if (!t || !pose) return -1;
    track_room(t);
    double *h = t->H.data() + 4 * (t->head + t->n);
    h[0] = pose[0]; h[1] = pose[1]; h[2] = pose[2]; h[3] = 1.0;
    t->n++;
    return 0;
}

// remove_observations, first half (sem_pc_accum.py:185-196): appends dist(newest, one before) = sqrt(sum((p1 - p0)^2));
// returns np.sum of all segment distances through *path_length
int pca_host_track_push_segment(pca_host_track *t, double *path_length)
{
    if (!t || t->n < 2) return -1;
    const double *p1 = t->H.data() + 4 * (t->head + t->n - 1), *p0 = p1 - 4;
    double s = 0.0;
    for (int k = 0; k < 3; ++k) { const double d = p1[k] - p0[k]; s += d * d; }
    t->D[(size_t)(t->head + t->nd)] = sqrt(s);
    t->nd++;
    if (path_length) *path_length = np_pairwise_sum(t->D.data() + t->head, t->nd);
    return 0;
}

// comp_incr_path_dist (sem_pc_accum.py:211-228): np.tri(n) @ d through numpy's own gemv.  out: [n_segments]
int pca_host_track_incr(pca_host_track *t, double *out)
{
    if (!t || !out) return -1;
    if (t->nd == 0) return 0;
    t->gemv(CBLAS_ROW_MAJOR, CBLAS_NO_TRANS, t->nd, t->nd, 1.0, track_tri(t, t->nd), t->nd, t->D.data() + t->head, 1, 0.0, out, 1);
    return 0;
}

// remove_observations, second half (sem_pc_accum.py:197-209): frames to drop so that the remaining path fits the horizon
int64_t pca_host_track_evict_beyond(pca_host_track *t, double horizon, double path_length)
{
    if (!t) return -1;
    if (!(path_length > horizon)) return 0;
    const double excess = path_length - horizon;
    int64_t k = 0;
    if (g_incr_blocks) {
        const int64_t first = incr_first_above(t, excess);
        k = first < 0 ? 0 : first;                                          // argmax of a boolean array: first True, else 0
    } else {
        t->incr.resize((size_t)(t->nd > 0 ? t->nd : 1));
        pca_host_track_incr(t, t->incr.data());
        for (int64_t i = 0; i < t->nd; ++i)
            if (t->incr[(size_t)i] - excess > 0.0) { k = i; break; }
    }
    t->head += k; t->n -= k; t->nd -= k;
    return k;
}

// The whole bookkeeping of one integrate() (kitti360_sem_pc_accum.py:41-88 minus the per-point work):
//   if frames are stored: transform them by T_new_prev; append the new pose [0,0,0]; if now more than one: push the segment
//   and evict beyond the horizon.  Returns the number of evicted frames; *path_length = the path before the eviction (what
//   the reference prints), NaN when no segment exists yet.
int64_t pca_host_track_step(pca_host_track *t, const double T_new_prev[16], double horizon, double *path_length)
{
    if (!t || !T_new_prev) return -1;
    static const double origin[3] = {0.0, 0.0, 0.0};
    if (t->n > 0) pca_host_track_transform(t, T_new_prev);
    pca_host_track_append(t, origin);
    double pl = NAN;
    int64_t k = 0;
    if (t->n > 1) {
        pca_host_track_push_segment(t, &pl);
        k = pca_host_track_evict_beyond(t, horizon, pl);
    }
    if (path_length) *path_length = pl;
    return k;
}

// The sample trigger of the KITTI-360 driver (run_kitti360_bev_gen.py:218-240), on the track as it stands after an
// integrate(): returns the present index, or -1 when one of the three conditions says "no sample now".
//   (1) the path is at least bev_horizon long;  (2) present = first frame whose distance from the start exceeds ...
//   incr = tri @ d;  if incr[-1] < H: no;  idx = argmax((incr - H) > 0);  if incr[-1] - incr[idx] < H: no;
//   (3) dist(pose[previous_idx], pose[idx]) >= min_step, else no (previous_idx as the driver keeps it: it may have gone
//       negative through evictions and then counts from the end; out of range: -2).
int64_t pca_host_track_trigger(pca_host_track *t, double bev_horizon, int64_t previous_idx, double min_step)
{
    if (!t || t->n < 2 || t->nd < 1) return -1;
    double last, at_idx;
    int64_t idx = 0;
    if (g_incr_blocks) {
        last = incr_at(t, t->nd - 1);
        if (last < bev_horizon) return -1;
        const int64_t first = incr_first_above(t, bev_horizon);
        idx = first < 0 ? 0 : first;
        at_idx = incr_at(t, idx);
    } else {
        t->incr.resize((size_t)t->nd);
        pca_host_track_incr(t, t->incr.data());
        last = t->incr[(size_t)(t->nd - 1)];
        if (last < bev_horizon) return -1;
        for (int64_t i = 0; i < t->nd; ++i)
            if (t->incr[(size_t)i] - bev_horizon > 0.0) { idx = i; break; }
        at_idx = t->incr[(size_t)idx];
    }
    if (last - at_idx < bev_horizon) return -1;
    if (previous_idx < 0) previous_idx += t->n;              // the driver indexes a Python list: negative = from the end
    if (previous_idx < 0 || previous_idx >= t->n) return -2; // IndexError in the driver
    {
        const double *a = t->H.data() + 4 * (t->head + previous_idx), *b = t->H.data() + 4 * (t->head + idx);
        double s = 0.0;
        for (int k = 0; k < 3; ++k) { const double d = b[k] - a[k]; s += d * d; }
        if (sqrt(s) < min_step) return -1;
    }
    return idx;
}

}  // extern "C"


// ---------------------------------------------------------------------------------------------
// Host arrays -> device, the way an unchanged driver hands observations over (numpy arrays in pageable memory): copy into
// pinned staging blocks on a few threads at once, then one asynchronous H2D copy per array.  What this replaces on the
// Python side (three np.copyto + three tensor.copy_ per KITTI observation) cost ~0.15 ms of a 0.34 ms step.
// The pool (pca_stage_pool.h) claims slices one at a time, every claim tied to its job; its threads spin for
// PCA_STAGING_SPIN_US (default 2000) after a job before they go to sleep: a driver stepping every 0.2-0.3 ms finds them
// awake, an idle process does not burn cores.
// ---------------------------------------------------------------------------------------------
namespace {
using StageSlice = pca_stage::Slice;
using StagePool = pca_stage::Pool;
static StagePool *stage_pool()
{
    static std::mutex once;
    static StagePool *pool = nullptr;                       // (heap: joined by the handler below, not by static destruction order)
    static pid_t owner = 0;
    std::lock_guard<std::mutex> lk(once);
    if (pool && owner != getpid()) pool = nullptr;          // a forked child: the threads stayed with the parent (leaks the object)
    if (!pool) {
        owner = getpid();
        // default: a third of this rank's share of the host's cores (sched_getaffinity-blind: cgroup limits are not seen,
        // which is why slices are claimed and not dealt), at most 7 helpers, at least one
        int threads = 3;
        long ranks = 1;
        if (const char *e = getenv("LOCAL_WORLD_SIZE")) ranks = atol(e) > 0 ? atol(e) : 1;
        const long cores = sysconf(_SC_NPROCESSORS_ONLN);
        if (cores > 0) { const long t = cores / ranks / 3; threads = (int)(t < 1 ? 1 : (t > 7 ? 7 : t)); }
        if (const char *e = getenv("PCA_STAGING_THREADS")) threads = atoi(e);
        if (threads < 0) threads = 0;
        if (threads > 15) threads = 15;
        pool = new StagePool();
        pool->start(threads);
        atexit([] { if (pool && owner == getpid()) { delete pool; pool = nullptr; } });
    }
    return pool;
}
}  // namespace

// host copies of `n` ranges side by side on the staging pool (also used by pca_kitti_integrate)
int pca_stage_copy(const void *const *src, void *const *dst, const int64_t *bytes, int n)
{
    constexpr size_t SLICE = 128 * 1024;
    std::vector<StageSlice> slices;
    for (int k = 0; k < n; ++k)
        for (size_t o = 0; o < (size_t)bytes[k]; o += SLICE)
            slices.push_back({(char *)dst[k] + o, (const char *)src[k] + o, (size_t)bytes[k] - o < SLICE ? (size_t)bytes[k] - o : SLICE});
    static std::mutex serial;                               // one job at a time (callers on several threads take turns)
    std::lock_guard<std::mutex> lk(serial);
    StagePool *pool = stage_pool();
    if (slices.size() < 2) { for (auto &sl : slices) pca_stage::copy_slice(sl.dst, sl.src, sl.n); pca_stage::stream_fence(); }
    else pool->run(slices.data(), (int)slices.size());
    return 0;
}

extern "C" int pca_host_stage_h2d(int n, const void *const *src, void *const *pinned, void *const *dev, const int64_t *bytes,
                                  void *stream)
{
    if (n < 0 || (n > 0 && (!src || !pinned || !dev || !bytes))) return -1;
    for (int k = 0; k < n; ++k)
        if (bytes[k] < 0 || (bytes[k] > 0 && (!src[k] || !pinned[k] || !dev[k]))) return -1;
    pca_stage_copy(src, pinned, bytes, n);
    // arrays that lie back to back in BOTH the pinned block and the device buffer (the six camera images of a NuScenes
    // observation) leave as ONE copy: a copy command has a fixed cost of tens of microseconds on this stack
    for (int k = 0; k < n;) {
        int64_t len = bytes[k];
        int m = k + 1;
        while (m < n && (const char *)pinned[m] == (const char *)pinned[k] + len && (const char *)dev[m] == (const char *)dev[k] + len) len += bytes[m++];
        if (len > 0 && hipMemcpyAsync(dev[k], pinned[k], (size_t)len, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return -2;
        k = m;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// A device result on its way to the host without the caller's stream waiting for it: the copy runs on a side stream of
// the context, behind everything enqueued on `stream` so far; the ticket names the event that marks its end.
// (In Python this was a stream context, wait_stream, copy_, an Event and record_stream per sample: 0.04 ms of the
// unchanged driver's 0.24 ms step.)
// ---------------------------------------------------------------------------------------------
extern "C" int pca_host_d2h_async(pca_ctx *ctx, const void *dev, void *pinned, int64_t bytes, void *stream)
{
    if (!ctx) return -1;
    if (!dev || !pinned || bytes < 0) { ctx->err = "d2h_async: bad arguments"; return -1; }
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    if (!ctx->d2h_stream) {
        PCA_CHECK(ctx, hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
        PCA_CHECK(ctx, hipEventCreateWithFlags(&ctx->d2h_go, hipEventDisableTiming));
    }
    const uint32_t k = ctx->d2h_next++ & 63u;
    if (!ctx->d2h_done[k]) PCA_CHECK(ctx, hipEventCreateWithFlags(&ctx->d2h_done[k], hipEventDisableTiming));
    PCA_CHECK(ctx, hipEventRecord(ctx->d2h_go, (hipStream_t)stream));
    PCA_CHECK(ctx, hipStreamWaitEvent(ctx->d2h_stream, ctx->d2h_go, 0));
    if (bytes > 0) PCA_CHECK(ctx, hipMemcpyAsync(pinned, dev, (size_t)bytes, hipMemcpyDeviceToHost, ctx->d2h_stream));
    PCA_CHECK(ctx, hipEventRecord(ctx->d2h_done[k], ctx->d2h_stream));
    return (int)k;
}

// Waits for the copy of `ticket` (a ticket older than 64 copies waits for a later copy of the same stream: still after its own).
extern "C" int pca_host_d2h_wait(pca_ctx *ctx, int ticket)
{
    if (!ctx) return -1;
    if (ticket < 0 || ticket > 63 || !ctx->d2h_done[ticket]) { ctx->err = "d2h_wait: bad ticket"; return -1; }
    PCA_CHECK(ctx, hipEventSynchronize(ctx->d2h_done[ticket]));
    return 0;
}
