// pca_accum.hip -- per-frame "integrate" kernels for gfx950 (MI355X)  (K1 lives in pca_k1.hip):
//   K1n nusc_sample_filter_transform    K0n nusc_project_cams
//   K2  retransform                      K3  mark_dynamic                  voxel de-duplication
// All are HBM-bound streaming kernels (no MFMA: the only contractions are 3x4 / 4x4 per point).
#include "pca_common.h"
#include <cstdlib>

// Compaction tiles: 4096 points per workgroup of 1024 threads.  Large tiles keep the number of tiles in flight
// (= the distance the decoupled look-back has to walk when all workgroups run in lock step) small and
// amortise ticket / look-back / barrier costs over more points.
#define PPT 4              // points per thread; point (k, t) of a tile is tile*TILE + k*BLK + t
#define CBLK 512           // threads per compaction workgroup (tile = 4 * CBLK points)
#define SBLK 256           // workgroup size of the plain streaming kernels (K0n, K2, K3)

// loads through the global address space (pointers that arrive inside structs are generic to the compiler)
template <typename T>
__device__ __forceinline__ T ldg(const T *p)
{
    return *reinterpret_cast<const __attribute__((address_space(1))) T *>(reinterpret_cast<uintptr_t>(p));
}
// a value at a workgroup-uniform address in memory that nobody writes during the launch: one scalar load (s_load)
template <typename T>
__device__ __forceinline__ T sload(const T *p)
{
    const uintptr_t u = reinterpret_cast<uintptr_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(((uintptr_t)hi << 32) | lo);
}
struct __attribute__((packed)) U32u { uint32_t v; };
__device__ __forceinline__ uint32_t ldg_u32_unaligned(const uint8_t *p)     // one global_load_dword at any byte address
{
    return reinterpret_cast<const __attribute__((address_space(1))) U32u *>(reinterpret_cast<uintptr_t>(p))->v;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg4(const float *p)      // one 16-byte global load
{
    const f32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) f32x4 *>(reinterpret_cast<uintptr_t>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// ---------------------------------------------------------------------------------------------
// Stable block-level compaction: given keep[k] for the PPT points of each thread (point order =
// k-major, then thread), returns the position of each kept point among the tile's kept points and
// the global exclusive prefix of the tile obtained by decoupled look-back.
// ---------------------------------------------------------------------------------------------
template <int NP>
struct TileScanT {
    uint32_t local[NP];    // rank of the point inside the tile (valid where keep)
    uint32_t total;        // kept points of the tile
    uint64_t excl;         // kept points of all earlier tiles of the launch
};
typedef TileScanT<PPT> TileScan;

template <int BLK, int NP = PPT>
__device__ __forceinline__ TileScanT<NP> tile_compact(const bool keep[NP], uint64_t *state, int tile, uint32_t epoch,
                                                      uint32_t *status)
{
    constexpr int NW = BLK / PCA_WAVE;
    static_assert(NP * NW <= 64, "the per-(row, wave) totals are scanned by one wave");
    __shared__ uint32_t s_wtot[NP * NW];       // kept points per (row k, wave), k-major = point order
    __shared__ uint32_t s_woff[NP * NW + 1];   // exclusive prefix of s_wtot, [NP*NW] = tile total
    __shared__ uint64_t s_excl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TileScanT<NP> r;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const uint64_t b = __ballot(keep[k]);
        r.local[k] = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_wtot[k * NW + wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    if (wave == 0) {
        const uint32_t v = lane < NP * NW ? s_wtot[lane] : 0u;
        const uint32_t inc = wave_incl_scan_add(v);
        if (lane < NP * NW) s_woff[lane] = inc - v;
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (lane == 0) s_woff[NP * NW] = total;
        const uint64_t e = lb_exclusive_prefix(state, tile, (uint64_t)total, epoch, status);
        if (lane == 0) s_excl = e;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NP; ++k) r.local[k] += s_woff[k * NW + wave];
    r.total = s_woff[NP * NW];
    r.excl = s_excl;
    return r;
}

__device__ __forceinline__ int draw_tile(uint32_t *ticket, int total_tiles)
{
    __shared__ int s_tile;
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(ticket, 1u);
        if ((int)t == total_tiles - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tile = (int)t;
    }
    __syncthreads();
    return s_tile;
}

// =============================================================================================
// K1n NuScenes oracle: nearest sample (6 cameras) + invalid/filter + stable append + ego->world
// =============================================================================================
struct K1nArgs {
    const double *pc;          // [n,7]
    const int64_t *cam_idx;    // [n]
    int n, total_tiles;
    const uint8_t *imgs;       // [ncam,H,W,3]
    const uint8_t *sems;       // [ncam,H,W]
    int ncam, H, W;
    int sample_mode;           // 0 nearest (the reference); 1 bilinear rgb (opt-in), class stays nearest
    int static_tiles;          // grid <= CUs: every workgroup is resident, tile = blockIdx (no ticket)
    Mat44 T;
    ClassMask filt;
    pca_store st;
    int64_t *frame_off;
    int slot;
    uint64_t *state;
    uint32_t *ticket;
    uint32_t epoch;
};

#define K1N_BLK 256
#define K1N_PPT 2            // 512-point tiles: a 35 k-point sweep spreads over 68 CUs; the kernel is a latency chain

__global__ __launch_bounds__(K1N_BLK) void k1n_nusc(const K1nArgs a)
{
    constexpr int TILE_PTS = K1N_PPT * K1N_BLK;
    const int tile = a.static_tiles ? (int)blockIdx.x : draw_tile(a.ticket, a.total_tiles);
    const int64_t base_pt = (int64_t)tile * TILE_PTS;
    // every load that does not depend on another goes out first: camera index and the whole 7 x f64 row
    int64_t cam[K1N_PPT];
    double row[K1N_PPT][7];
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) {
        const int64_t p = base_pt + k * K1N_BLK + threadIdx.x;
        cam[k] = -1;
#pragma unroll
        for (int i = 0; i < 7; ++i) row[k][i] = 0.0;
        if (p < a.n) {
            cam[k] = ldg(a.cam_idx + p);
#pragma unroll
            for (int i = 0; i < 7; ++i) row[k][i] = ldg(a.pc + p * 7 + i);
        }
    }
    bool valid[K1N_PPT], keep[K1N_PPT];
    uint32_t packed[K1N_PPT];
    unsigned cls[K1N_PPT];
    bool bad_uv = false;
    const int64_t last = (int64_t)a.ncam * a.H * a.W * 3 - 4;           // last legal 4-byte window of the images
    auto rgb_at = [&](int64_t pix) -> uint32_t {
        int64_t off = pix * 3;
        const int sh = off > last ? (int)(off - last) * 8 : 0;
        off = off > last ? last : off;
        return (ldg_u32_unaligned(a.imgs + off) >> sh) & 0xffffffu;
    };
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) {
        const int64_t c = cam[k];
        const double u = row[k][4], v = row[k][5];
        valid[k] = c >= 0 && c < a.ncam;                   // else: features stay -1 -> invalid
        if (valid[k] && !(u > 1.0 && u < (double)a.W - 1.0 && v > 1.0 && v < (double)a.H - 1.0)) { bad_uv = true; valid[k] = false; }
        const int ui = valid[k] ? (int)rint(u) : 0, vi = valid[k] ? (int)rint(v) : 0;
        const int64_t img0 = (valid[k] ? c : 0) * a.H;
        const int64_t pix = (img0 + vi) * a.W + ui;        // pixel 0 of camera 0 for invalid points: a legal address
        cls[k] = ldg(a.sems + pix);
        if (!a.sample_mode) {
            packed[k] = rgb_at(pix);
        } else {
            const Bilin b = bilin_weights<false>(valid[k] ? u : 0.0, valid[k] ? v : 0.0);
            const int u0 = (int)b.u0, u1 = (int)b.u1, v0 = (int)b.v0, v1 = (int)b.v1;   // inside the image: 1 < u < W-1
            auto at = [&](int vv, int uu) { return rgb_at(valid[k] ? (img0 + vv) * a.W + uu : 0); };   // never out of the stack
            packed[k] = bilin_rgb(b, at(v0, u0), at(v1, u1), at(v1, u0), at(v0, u1));
        }
    }
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) {
        keep[k] = valid[k] && !in_mask(a.filt, cls[k]);
        packed[k] |= cls[k] << 24;
    }
    if (bad_uv) pca_raise(a.ticket + 1, PCA_STATUS_UV_OUT_OF_IMAGE);

    const TileScanT<K1N_PPT> sc = tile_compact<K1N_BLK, K1N_PPT>(keep, a.state, tile, a.epoch, a.ticket + 1);
    const int64_t tile_base = a.frame_off[a.slot] + (int64_t)sc.excl;
    bool overflow = false;
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) {
        if (!keep[k]) continue;
        const int64_t o = tile_base + sc.local[k];
        if (o >= a.st.capacity) { overflow = true; continue; }
        const double x = row[k][0], y = row[k][1], z = row[k][2];
        a.st.x[o] = row4(a.T.m + 0, x, y, z);
        a.st.y[o] = row4(a.T.m + 4, x, y, z);
        a.st.z[o] = row4(a.T.m + 8, x, y, z);
        a.st.intensity[o] = (float)row[k][3];
        a.st.rgbs[o] = packed[k];
        a.st.inst[o] = (int32_t)row[k][6];
        a.st.dyn[o] = 0;
    }
    if (overflow) pca_raise(a.ticket + 1, PCA_STATUS_STORE_OVERFLOW);
    if (threadIdx.x == 0 && tile == a.total_tiles - 1) a.frame_off[a.slot + 1] = tile_base + sc.total;
}

// ---------------------------------------------------------------------------------------------
// K1n for a BATCH of frames (a whole scene: run_nuscenes_bev_gen.py:236-237 integrates all ~40 samples before the first
// BEV).  One frame is 68 tiles -- a launch that leaves three quarters of the chip idle and is all latency (9 us); the
// batch is one front launch over the tiles of all frames and one append launch, in the SPLIT form of K1 (csrc/pca_k1.hip):
// the front writes each tile's kept records (final values: transformed x, y, z, intensity, rgb | class, instance) at
// tile-local places of a staging area plus the tile's count, nobody waits for anybody; the append adds up the counts before
// its tile and streams the records into the store, closing the frames' segments.
// ---------------------------------------------------------------------------------------------
struct K1nFrame {              // device-side descriptor of one frame of the batch
    const double *pc;          // [n,7]
    const int64_t *cam_idx;    // [n]
    const uint8_t *imgs;       // [ncam,H,W,3]
    const uint8_t *sems;       // [ncam,H,W]
    int32_t n, tile0;          // points; first tile of the frame
    double T[12];              // top three rows of T_ego_world
};
struct K1nBatchArgs {
    const K1nFrame *frames;    // dev [n_frames]
    const int32_t *tile_frame; // dev [total_tiles]
    int n_frames, total_tiles;
    int ncam, H, W, sample_mode;
    ClassMask filt;
    double *sx, *sy, *sz;      // staging [total_tiles * TILE]
    float *si;
    uint32_t *sc;
    int32_t *sn;
    uint32_t *counts;          // [total_tiles]
    int32_t *lastf;            // [total_tiles] frame index if the tile is the last of its frame, else -1
    pca_store st;
    int64_t *frame_off;
    int first_slot;
    uint32_t *status;
};

__global__ __launch_bounds__(K1N_BLK) void k1n_front_batch(const K1nBatchArgs a)
{
    constexpr int TILE_PTS = K1N_PPT * K1N_BLK, NW = K1N_BLK / PCA_WAVE;
    __shared__ uint32_t s_wtot[K1N_PPT * NW], s_woff[K1N_PPT * NW + 1];
    __shared__ __align__(16) double s_rows[TILE_PTS * 7];   // the tile's rows [x y z i u v inst], as they lie in memory (28 KB)
    const int tile = blockIdx.x;
    // tile -> frame -> descriptor through the scalar cache (both tables are written before the launch, never during it):
    // two short hops instead of two vector-memory round trips in front of the first point load
    const int f = sload(a.tile_frame + tile);
    const K1nFrame *fp = a.frames + f;
    const double *pc = sload(&fp->pc);
    const int64_t *cam_idx = sload(&fp->cam_idx);
    const uint8_t *imgs = sload(&fp->imgs), *sems = sload(&fp->sems);
    const int n = sload(&fp->n), tile0 = sload(&fp->tile0);
    const int64_t base_pt = (int64_t)(tile - tile0) * TILE_PTS;
    const int n_here = n - base_pt < TILE_PTS ? (int)(n - base_pt) : TILE_PTS;      // points of this tile (0: an empty frame)
    // The rows are an array of structures (7 x f64 = 56 B per point): a lane-per-point load touches 28 cache lines per
    // instruction, seven times over.  The tile's rows are ONE contiguous 28 KB range instead: 16-byte loads, consecutive
    // lanes on consecutive addresses, into LDS; every lane then picks its own rows out of LDS.
    {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const double *src = pc + base_pt * 7;
        const int total = n_here * 7;
        for (int i = 2 * (int)threadIdx.x; i < total; i += 2 * K1N_BLK) {
            if (i + 1 < total) {
                const d2 v = *reinterpret_cast<const __attribute__((address_space(1))) d2 *>(reinterpret_cast<uintptr_t>(src + i));
                s_rows[i] = v.x; s_rows[i + 1] = v.y;
            } else {
                s_rows[i] = ldg(src + i);
            }
        }
    }
    int64_t cam[K1N_PPT];
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) {
        const int p = k * K1N_BLK + (int)threadIdx.x;
        cam[k] = p < n_here ? ldg(cam_idx + base_pt + p) : -1;
    }
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = sload(&fp->T[i]);
    __syncthreads();
    double row[K1N_PPT][7];
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) {
        const int p = k * K1N_BLK + (int)threadIdx.x;
#pragma unroll
        for (int i = 0; i < 7; ++i) row[k][i] = p < n_here ? s_rows[p * 7 + i] : 0.0;
    }
    bool valid[K1N_PPT], keep[K1N_PPT];
    uint32_t packed[K1N_PPT];
    unsigned cls[K1N_PPT];
    int64_t pixk[K1N_PPT];
    bool bad_uv = false;
    const int64_t last = (int64_t)a.ncam * a.H * a.W * 3 - 4;           // last legal 4-byte window of the images
    auto rgb_at = [&](int64_t pix) -> uint32_t {
        int64_t off = pix * 3;
        const int sh = off > last ? (int)(off - last) * 8 : 0;
        off = off > last ? last : off;
        return (ldg_u32_unaligned(imgs + off) >> sh) & 0xffffffu;
    };
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) {
        const int64_t c = cam[k];
        const double u = row[k][4], v = row[k][5];
        valid[k] = c >= 0 && c < a.ncam;                   // else: features stay -1 -> invalid
        if (valid[k] && !(u > 1.0 && u < (double)a.W - 1.0 && v > 1.0 && v < (double)a.H - 1.0)) { bad_uv = true; valid[k] = false; }
        const int ui = valid[k] ? (int)rint(u) : 0, vi = valid[k] ? (int)rint(v) : 0;
        const int64_t img0 = (valid[k] ? c : 0) * a.H;
        pixk[k] = (img0 + vi) * a.W + ui;                  // pixel 0 of camera 0 for invalid points: a legal address
        cls[k] = ldg(sems + pixk[k]);
    }
    // the colour is gathered for the KEPT points only (a second, dependent gather: a wave-wide gather pulls 64 lines through
    // the L1 for 64 x 4 useful bytes -- what bounds this kernel on scattered points, as it bounds K1's batch form)
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) {
        keep[k] = valid[k] && !in_mask(a.filt, cls[k]);
        const int64_t pix = keep[k] ? pixk[k] : 0;
        if (!a.sample_mode) {
            packed[k] = rgb_at(pix);
        } else {
            const double u = row[k][4], v = row[k][5];
            const Bilin b = bilin_weights<false>(keep[k] ? u : 0.0, keep[k] ? v : 0.0);
            const int u0 = (int)b.u0, u1 = (int)b.u1, v0 = (int)b.v0, v1 = (int)b.v1;
            const int64_t img0 = (keep[k] ? cam[k] : 0) * a.H;
            auto at = [&](int vv, int uu) { return rgb_at(keep[k] ? (img0 + vv) * a.W + uu : 0); };
            packed[k] = bilin_rgb(b, at(v0, u0), at(v1, u1), at(v1, u0), at(v0, u1));
        }
    }
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) packed[k] |= cls[k] << 24;
    if (bad_uv) pca_raise(a.status, PCA_STATUS_UV_OUT_OF_IMAGE);
    // stable ranks inside the tile (point order = k-major, then thread)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t local[K1N_PPT];
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) {
        const uint64_t b = __ballot(keep[k]);
        local[k] = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_wtot[k * NW + wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    if (wave == 0) {
        const uint32_t v = lane < K1N_PPT * NW ? s_wtot[lane] : 0u;
        const uint32_t inc = wave_incl_scan_add(v);
        if (lane < K1N_PPT * NW) s_woff[lane] = inc - v;
        if (lane == 63) s_woff[K1N_PPT * NW] = inc;
    }
    __syncthreads();
    const int64_t sbase = (int64_t)tile * TILE_PTS;
#pragma unroll
    for (int k = 0; k < K1N_PPT; ++k) {
        if (!keep[k]) continue;
        const int64_t o = sbase + s_woff[k * NW + wave] + local[k];
        const double x = row[k][0], y = row[k][1], z = row[k][2];
        a.sx[o] = row4(T + 0, x, y, z);
        a.sy[o] = row4(T + 4, x, y, z);
        a.sz[o] = row4(T + 8, x, y, z);
        a.si[o] = (float)row[k][3];
        a.sc[o] = packed[k];
        a.sn[o] = (int32_t)row[k][6];
    }
    if (threadIdx.x == 0) {
        a.counts[tile] = s_woff[K1N_PPT * NW];
        const int ftiles = n > 0 ? (n + TILE_PTS - 1) / TILE_PTS : 1;
        a.lastf[tile] = tile - tile0 == ftiles - 1 ? f : -1;
    }
}

__global__ __launch_bounds__(256) void k1n_append_batch(const K1nBatchArgs a)
{
    constexpr int BLK = 256, TILE_PTS = K1N_PPT * K1N_BLK;
    const int tile = blockIdx.x;
    __shared__ uint32_t s_w[BLK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t sum = 0;
    for (int t = threadIdx.x; t < tile; t += BLK) sum += ldg(a.counts + t);
    sum = wave_reduce_add(sum);
    if (lane == 0) s_w[wave] = sum;
    const uint32_t c = a.counts[tile];
    const int32_t lf = a.lastf[tile];
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < BLK / 64; ++w) before += s_w[w];
    const int64_t base = a.frame_off[a.first_slot] + before;
    if (threadIdx.x == 0 && lf >= 0) a.frame_off[a.first_slot + lf + 1] = base + c;
    const int64_t sbase = (int64_t)tile * TILE_PTS;
    bool overflow = false;
    for (uint32_t j = threadIdx.x; j < c; j += BLK) {
        const int64_t o = base + j;
        if (o >= a.st.capacity) { overflow = true; continue; }
        a.st.x[o] = ldg(a.sx + sbase + j);
        a.st.y[o] = ldg(a.sy + sbase + j);
        a.st.z[o] = ldg(a.sz + sbase + j);
        a.st.intensity[o] = ldg(a.si + sbase + j);
        a.st.rgbs[o] = ldg(a.sc + sbase + j);
        a.st.inst[o] = ldg(a.sn + sbase + j);
        a.st.dyn[o] = 0;
    }
    if (overflow) pca_raise(a.status, PCA_STATUS_STORE_OVERFLOW);
}

// pts_feat_from_img(pts_uv, img, 'bilinear') of the reference (datasets/nuscenes_utils.py:181-210) for a 2-D map:
// the reference's arithmetic to the letter (bilin_weights<true>); raises the UV status bit where the reference asserts.
struct BilinArgs { const double *map; int H, W; const double *uv; int n; double *out; uint32_t *status; };

__global__ __launch_bounds__(SBLK) void sample_bilinear(const BilinArgs a)
{
    const int p = blockIdx.x * SBLK + threadIdx.x;
    if (p >= a.n) return;
    const double u = a.uv[2 * p], v = a.uv[2 * p + 1];
    if (!(u > 1.0 && u < (double)a.W - 1.0 && v > 1.0 && v < (double)a.H - 1.0)) { pca_raise(a.status, PCA_STATUS_UV_OUT_OF_IMAGE); a.out[p] = 0.0; return; }
    const Bilin b = bilin_weights<true>(u, v);
    const int u0 = (int)b.u0, u1 = (int)b.u1, v0 = (int)b.v0, v1 = (int)b.v1;
    a.out[p] = bilin_value(b, a.map[(int64_t)v0 * a.W + u0], a.map[(int64_t)v1 * a.W + u1], a.map[(int64_t)v1 * a.W + u0],
                           a.map[(int64_t)v0 * a.W + u1]);
}

// =============================================================================================
// K0n NuScenes: lidar -> ego -> global -> cameras, pinhole projection, last camera wins
// =============================================================================================
#define MAX_CAMS 8
struct K0nArgs {
    const double *pc;   // [n,3]
    int n, ncam;
    Mat44 T_ego_from_lidar, T_glob_from_ego;
    Mat44 T_cam_from_glob[MAX_CAMS];
    double K[MAX_CAMS][9];
    double wh[MAX_CAMS][2];
    double *pc_in_ego;  // [n,3]
    double *uv;         // [n,2]
    int64_t *cam_idx;   // [n]
};

__global__ __launch_bounds__(SBLK) void k0n_project(const K0nArgs a)
{
    for (int64_t p = (int64_t)blockIdx.x * SBLK + threadIdx.x; p < a.n; p += (int64_t)gridDim.x * SBLK) {
        const double x = a.pc[3 * p], y = a.pc[3 * p + 1], z = a.pc[3 * p + 2];
        const double ex = row4(a.T_ego_from_lidar.m + 0, x, y, z);
        const double ey = row4(a.T_ego_from_lidar.m + 4, x, y, z);
        const double ez = row4(a.T_ego_from_lidar.m + 8, x, y, z);
        a.pc_in_ego[3 * p] = ex; a.pc_in_ego[3 * p + 1] = ey; a.pc_in_ego[3 * p + 2] = ez;
        const double gx = row4(a.T_glob_from_ego.m + 0, ex, ey, ez);
        const double gy = row4(a.T_glob_from_ego.m + 4, ex, ey, ez);
        const double gz = row4(a.T_glob_from_ego.m + 8, ex, ey, ez);
        double ou = 0.0, ov = 0.0;
        int64_t oc = -1;
        for (int j = 0; j < a.ncam; ++j) {
            const double *Tc = a.T_cam_from_glob[j].m;
            const double cx = row4(Tc + 0, gx, gy, gz), cy = row4(Tc + 4, gx, gy, gz), cz = row4(Tc + 8, gx, gy, gz);
            if (!(cz > 1e-3)) continue;
            const double *Kj = a.K[j];
            // viewpad row . [x y z 1] with a zero fourth coefficient
            double px = Kj[0] * cx; px = fma(Kj[1], cy, px); px = fma(Kj[2], cz, px); px = fma(0.0, 1.0, px);
            double py = Kj[3] * cx; py = fma(Kj[4], cy, py); py = fma(Kj[5], cz, py); py = fma(0.0, 1.0, py);
            double pz = Kj[6] * cx; pz = fma(Kj[7], cy, pz); pz = fma(Kj[8], cz, pz); pz = fma(0.0, 1.0, pz);
            const double u = px / pz, v = py / pz;
            if (u > 1.0 && u < a.wh[j][0] - 1.0 && v > 1.0 && v < a.wh[j][1] - 1.0) { ou = u; ov = v; oc = j; }
        }
        a.uv[2 * p] = ou; a.uv[2 * p + 1] = ov;
        a.cam_idx[p] = oc;
    }
}

// =============================================================================================
// K2  in-place re-transform of stored frames (chain of n_T rigid transforms, applied in order)
// =============================================================================================
#define MAX_CHAIN 16
struct K2Args {
    double *x, *y, *z;
    const int64_t *frame_off;
    int slot_begin, slot_end;
    int n_T;
    Mat34 T[MAX_CHAIN];     // top three rows of each 4x4
};

__device__ __forceinline__ void apply_chain(const K2Args &a, double &x, double &y, double &z)
{
    for (int t = 0; t < a.n_T; ++t) {
        const double *m = a.T[t].m;
        const double nx = row4(m + 0, x, y, z), ny = row4(m + 4, x, y, z), nz = row4(m + 8, x, y, z);
        x = nx; y = ny; z = nz;
    }
}

__global__ __launch_bounds__(SBLK) void k2_retransform(const K2Args a)
{
    const int64_t lo = a.frame_off[a.slot_begin], hi = a.frame_off[a.slot_end];
    // 16-byte vector body over even-aligned pairs, scalar head/tail
    const int64_t lo2 = (lo + 1) & ~1ll, hi2 = hi & ~1ll;
    const int64_t gtid = (int64_t)blockIdx.x * SBLK + threadIdx.x, gsz = (int64_t)gridDim.x * SBLK;
    if (gtid == 0) {
        if (lo < lo2 && lo < hi) { double x = a.x[lo], y = a.y[lo], z = a.z[lo]; apply_chain(a, x, y, z); a.x[lo] = x; a.y[lo] = y; a.z[lo] = z; }
        if (hi2 < hi && hi2 >= lo2) { double x = a.x[hi2], y = a.y[hi2], z = a.z[hi2]; apply_chain(a, x, y, z); a.x[hi2] = x; a.y[hi2] = y; a.z[hi2] = z; }
    }
    double2 *X = reinterpret_cast<double2 *>(a.x), *Y = reinterpret_cast<double2 *>(a.y),
            *Z = reinterpret_cast<double2 *>(a.z);
    for (int64_t i = lo2 / 2 + gtid; i < hi2 / 2; i += gsz) {
        double2 vx = X[i], vy = Y[i], vz = Z[i];
        apply_chain(a, vx.x, vy.x, vz.x);
        apply_chain(a, vx.y, vy.y, vz.y);
        X[i] = vx; Y[i] = vy; Z[i] = vz;
    }
}

// Batched integrate: frame i of a batch of k frames appended in one K1 call still owes the transforms of the frames
// that came after it, T[i+1] .. T[k-1] (sem_pc_accum.py:167-183 applied step by step).  One launch per pass of
// MAX_CHAIN transforms: blockIdx.y = frame of the batch, which applies the transforms t >= i + 1 of the pass.
struct K2TailArgs {
    double *x, *y, *z;
    const int64_t *frame_off;
    int first_slot;          // slot of frame 0 of the batch
    int t_base, n_T;         // the pass holds transforms t_base .. t_base + n_T - 1
    Mat34 T[MAX_CHAIN];
};

__global__ __launch_bounds__(SBLK) void k2_retransform_tail(const K2TailArgs a)
{
    const int i = blockIdx.y;
    int t0 = i + 1 - a.t_base;                   // first transform of this pass that frame i owes
    if (t0 < 0) t0 = 0;
    if (t0 >= a.n_T) return;
    const int64_t lo = a.frame_off[a.first_slot + i], hi = a.frame_off[a.first_slot + i + 1];
    for (int64_t p = lo + (int64_t)blockIdx.x * SBLK + threadIdx.x; p < hi; p += (int64_t)gridDim.x * SBLK) {
        double x = a.x[p], y = a.y[p], z = a.z[p];
        for (int t = t0; t < a.n_T; ++t) {
            const double *m = a.T[t].m;
            const double nx = row4(m + 0, x, y, z), ny = row4(m + 4, x, y, z), nz = row4(m + 8, x, y, z);
            x = nx; y = ny; z = nz;
        }
        a.x[p] = x; a.y[p] = y; a.z[p] = z;
    }
}

// =============================================================================================
// K3  flag points of an instance as dynamic
// =============================================================================================
#define MAX_PAIRS 32
struct K3Args {
    const int32_t *inst;
    uint8_t *dyn;
    const int64_t *frame_off;
    int n_pairs;
    int32_t slot[MAX_PAIRS];
    int32_t inst_idx[MAX_PAIRS];
};

__global__ __launch_bounds__(SBLK) void k3_mark_dynamic(const K3Args a)
{
    const int pr = blockIdx.y;
    const int64_t lo = a.frame_off[a.slot[pr]], hi = a.frame_off[a.slot[pr] + 1];
    const int32_t want = a.inst_idx[pr];
    for (int64_t p = lo + (int64_t)blockIdx.x * SBLK + threadIdx.x; p < hi; p += (int64_t)gridDim.x * SBLK)
        if (a.inst[p] == want) a.dyn[p] = 1;
}

// =============================================================================================
// Voxel de-duplication of the running accumulation buffer (opt-in; the reference only evicts whole
// frames, sem_pc_accum.py:185-209).  Of all stored points of the window that fall into one voxel
// floor(xyz / size) the first in store order (the oldest observation) stays; the segments are
// compacted in place, stably, and frame_off follows.
//   dedup_insert   open-addressing table keyed by the packed voxel index: value = min point index
//   dedup_compact  keep = (table value == own index); single-pass stable compaction (decoupled
//                  look-back).  In place is safe: a tile's destination range ends before its own
//                  source range ends, and it is written only after every earlier tile has published
//                  its aggregate -- which each workgroup does after its own loads have completed.
//   dedup_offsets  frame_off <- new segment boundaries
// =============================================================================================
struct DedupArgs {
    pca_store st;
    int64_t *frame_off;
    int64_t *new_off;         // [slot_end - slot_begin + 2]; last entry = kept total
    int slot_begin, slot_end;
    int64_t max_points;
    double size;
    unsigned long long *keys; // [cap] 0 = empty
    uint32_t *vals;           // [cap] min window-relative index
    uint64_t cap_mask;
    uint64_t *state;
    uint32_t *ticket;
    uint32_t epoch;
    int total_tiles;
};

__device__ __forceinline__ unsigned long long voxel_key(double x, double y, double z, double size)
{
    auto q = [&](double v) -> unsigned long long {
        double f = floor(v / size) + 1048576.0;
        f = f < 0.0 ? 0.0 : (f > 2097151.0 ? 2097151.0 : f);
        return (unsigned long long)f;
    };
    return (1ull << 63) | (q(x) << 42) | (q(y) << 21) | q(z);
}
__device__ __forceinline__ uint64_t voxel_hash(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return k;
}
__device__ __forceinline__ void dedup_window(const DedupArgs &a, int64_t &lo, int64_t &hi)
{
    lo = a.frame_off[a.slot_begin];
    hi = a.frame_off[a.slot_end];
    if (hi - lo > a.max_points) hi = lo + a.max_points;
}

__global__ __launch_bounds__(SBLK) void dedup_insert(const DedupArgs a)
{
    int64_t lo, hi;
    dedup_window(a, lo, hi);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.frame_off[a.slot_end] - lo > a.max_points)
        pca_raise(a.ticket + 1, PCA_STATUS_STORE_OVERFLOW);
    for (int64_t p = lo + (int64_t)blockIdx.x * SBLK + threadIdx.x; p < hi; p += (int64_t)gridDim.x * SBLK) {
        const unsigned long long key = voxel_key(a.st.x[p], a.st.y[p], a.st.z[p], a.size);
        uint64_t h = voxel_hash(key) & a.cap_mask;
        for (uint64_t probe = 0; probe <= a.cap_mask; ++probe) {             // load factor <= 1/2: ends early
            const unsigned long long prev = atomicCAS(&a.keys[h], 0ull, key);
            if (prev == 0ull || prev == key) { atomicMin(&a.vals[h], (uint32_t)(p - lo)); break; }
            h = (h + 1) & a.cap_mask;
        }
    }
}

template <int BLK>
__global__ __launch_bounds__(BLK) void dedup_compact(const DedupArgs a)
{
    constexpr int TILE = PPT * BLK;
    __shared__ uint32_t s_rank[TILE];
    const int tile = draw_tile(a.ticket, a.total_tiles);
    int64_t lo, hi;
    dedup_window(a, lo, hi);
    const int64_t t_lo = lo + (int64_t)tile * TILE;
    bool keep[PPT];
    double X[PPT], Y[PPT], Z[PPT];
    float I[PPT];
    uint32_t C[PPT];
    int32_t N[PPT];
    uint8_t D[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int64_t p = t_lo + k * BLK + threadIdx.x;
        keep[k] = false;
        X[k] = Y[k] = Z[k] = 0.0; I[k] = 0.f; C[k] = 0; N[k] = 0; D[k] = 0;
        if (p < hi) {
            X[k] = a.st.x[p]; Y[k] = a.st.y[p]; Z[k] = a.st.z[p];
            I[k] = a.st.intensity[p]; C[k] = a.st.rgbs[p]; N[k] = a.st.inst[p]; D[k] = a.st.dyn[p];
        }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int64_t p = t_lo + k * BLK + threadIdx.x;
        if (p >= hi) continue;
        const unsigned long long key = voxel_key(X[k], Y[k], Z[k], a.size);
        uint64_t h = voxel_hash(key) & a.cap_mask;
        for (uint64_t probe = 0; probe <= a.cap_mask; ++probe) {
            const unsigned long long kk = a.keys[h];
            if (kk == key) { keep[k] = a.vals[h] == (uint32_t)(p - lo); break; }
            if (kk == 0ull) break;                                             // cannot happen after dedup_insert
            h = (h + 1) & a.cap_mask;
        }
    }
    // every load of this workgroup has returned before its aggregate becomes visible (see the header)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const TileScan sc = tile_compact<BLK>(keep, a.state, tile, a.epoch, a.ticket + 1);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        s_rank[k * BLK + threadIdx.x] = sc.local[k];
        if (!keep[k]) continue;
        const int64_t d = lo + (int64_t)sc.excl + sc.local[k];
        a.st.x[d] = X[k]; a.st.y[d] = Y[k]; a.st.z[d] = Z[k];
        a.st.intensity[d] = I[k]; a.st.rgbs[d] = C[k]; a.st.inst[d] = N[k]; a.st.dyn[d] = D[k];
    }
    __syncthreads();
    // segment boundaries that fall into this tile: first frame f > slot_begin with frame_off[f] >= t_lo
    const int64_t t_hi = t_lo + TILE < hi ? t_lo + TILE : hi;
    int f0 = a.slot_begin + 1, f1 = a.slot_end + 1;
    while (f0 < f1) {
        const int m = (f0 + f1) >> 1;
        if (a.frame_off[m] < t_lo) f0 = m + 1; else f1 = m;
    }
    for (int f = f0 + threadIdx.x; f <= a.slot_end; f += BLK) {
        const int64_t b = a.frame_off[f];
        if (b >= t_hi) break;
        a.new_off[f - a.slot_begin] = lo + (int64_t)sc.excl + s_rank[b - t_lo];
    }
    if (tile == a.total_tiles - 1 && threadIdx.x == 0) a.new_off[a.slot_end - a.slot_begin + 1] = lo + (int64_t)sc.excl + sc.total;
}

__global__ __launch_bounds__(SBLK) void dedup_offsets(const DedupArgs a)
{
    int64_t lo, hi;
    dedup_window(a, lo, hi);
    const int64_t end = a.new_off[a.slot_end - a.slot_begin + 1];
    __shared__ int64_t s_old_hi;
    if (threadIdx.x == 0) s_old_hi = hi;
    __syncthreads();
    // boundaries at or beyond the old window end (the end itself, empty trailing frames) move to the new end
    for (int f = a.slot_begin + 1 + threadIdx.x; f <= a.slot_end; f += SBLK) {
        const int64_t b = a.frame_off[f];
        a.new_off[f - a.slot_begin] = b >= s_old_hi ? end : a.new_off[f - a.slot_begin];
    }
    __syncthreads();
    for (int f = a.slot_begin + 1 + threadIdx.x; f <= a.slot_end; f += SBLK) a.frame_off[f] = a.new_off[f - a.slot_begin];
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int pca_nusc_sample_filter_transform(pca_ctx *ctx, const double *pc, const int64_t *cam_idx, int32_t n,
                                     const uint8_t *imgs, const uint8_t *sems, int ncam, int H, int W,
                                     const double T[16], const uint64_t filter_mask[4], const pca_store *store,
                                     int64_t *frame_off, int slot, void *stream)
{
    return pca_nusc_sample_filter_transform_ex(ctx, pc, cam_idx, n, imgs, sems, ncam, H, W, T, filter_mask, store, frame_off,
                                               slot, PCA_SAMPLE_NEAREST, stream);
}

int pca_sample_bilinear(pca_ctx *ctx, const double *map, int H, int W, const double *uv, int32_t n, double *out, void *stream)
{
    if (!ctx) return -1;
    if (n < 0 || (n > 0 && (!map || !uv || !out)) || H < 1 || W < 1) { ctx->err = "bilinear: bad arguments"; return -1; }
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    BilinArgs a;
    a.map = map; a.H = H; a.W = W; a.uv = uv; a.n = n; a.out = out; a.status = ctx->ticket + 1;
    hipLaunchKernelGGL(sample_bilinear, dim3((n + SBLK - 1) / SBLK), dim3(SBLK), 0, s, a);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

int pca_nusc_sample_filter_transform_ex(pca_ctx *ctx, const double *pc, const int64_t *cam_idx, int32_t n,
                                        const uint8_t *imgs, const uint8_t *sems, int ncam, int H, int W,
                                        const double T[16], const uint64_t filter_mask[4], const pca_store *store,
                                        int64_t *frame_off, int slot, int sample_mode, void *stream)
{
    if (!ctx) return -1;
    if (pca_k1_flush_pending(ctx)) return -1;               // a deferred K1 of this context comes first
    if (sample_mode != PCA_SAMPLE_NEAREST && sample_mode != PCA_SAMPLE_BILINEAR) { ctx->err = "k1n: unknown sample_mode"; return -1; }
    if (ncam < 1 || H < 1 || W < 1 || (int64_t)ncam * H * W * 3 < 4) { ctx->err = "k1n: bad image stack"; return -1; }
    if (n < 0 || (n > 0 && (!pc || !cam_idx || !imgs || !sems)) || !store || !frame_off) { ctx->err = "k1n: bad arguments"; return -1; }
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    const int total = n > 0 ? (n + K1N_PPT * K1N_BLK - 1) / (K1N_PPT * K1N_BLK) : 1;
    if (pca_ctx_reserve_tiles(ctx, total, s)) return -1;
    K1nArgs a;
    a.sample_mode = sample_mode;
    a.static_tiles = total <= ctx->n_cu;
    a.pc = pc; a.cam_idx = cam_idx; a.n = n; a.total_tiles = total;
    a.imgs = imgs; a.sems = sems; a.ncam = ncam; a.H = H; a.W = W;
    for (int i = 0; i < 16; ++i) a.T.m[i] = T[i];
    for (int i = 0; i < 4; ++i) a.filt.w[i] = filter_mask ? filter_mask[i] : 0;
    a.st = *store; a.frame_off = frame_off; a.slot = slot;
    a.state = ctx->tile_state; a.ticket = ctx->ticket;
    a.epoch = pca_ctx_next_epoch(ctx, s);
    PCA_LAUNCH(ctx, PCA_K_NUSC, k1n_nusc, dim3(total), dim3(K1N_BLK), s, a);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

static int accum_grow(pca_ctx *ctx, void **p, int64_t *cap, int64_t need, hipStream_t s)
{
    if (need <= *cap) return 0;
    PCA_CHECK(ctx, hipStreamSynchronize(s));
    if (*p) PCA_CHECK(ctx, hipFree(*p));
    *p = nullptr; *cap = 0;
    const int64_t want = need + need / 4;
    PCA_CHECK(ctx, hipMalloc(p, (size_t)want));
    *cap = want;
    return 0;
}

#define K1N_MAX_BATCH_TILES 16384       // k1n_append_batch adds up the counts before its tile
int pca_nusc_sample_filter_transform_batch(pca_ctx *ctx, const pca_nusc_frame *frames, int n_frames, int ncam, int H, int W,
                                           const uint64_t filter_mask[4], const pca_store *store, int64_t *frame_off,
                                           int first_slot, int sample_mode, void *stream)
{
    if (!ctx) return -1;
    if (pca_k1_flush_pending(ctx)) return -1;               // a deferred K1 of this context comes first
    if (sample_mode != PCA_SAMPLE_NEAREST && sample_mode != PCA_SAMPLE_BILINEAR) { ctx->err = "k1n: unknown sample_mode"; return -1; }
    if (ncam < 1 || H < 1 || W < 1 || (int64_t)ncam * H * W * 3 < 4) { ctx->err = "k1n: bad image stack"; return -1; }
    if (!frames || n_frames < 1 || !store || !frame_off) { ctx->err = "k1n: bad arguments"; return -1; }
    constexpr int TILE_PTS = K1N_PPT * K1N_BLK;
    int64_t total = 0;
    for (int k = 0; k < n_frames; ++k) {
        const pca_nusc_frame &f = frames[k];
        // (the image stacks are required for EMPTY frames too: an empty frame still runs one tile, whose gathers read
        // pixel 0 -- "always a legal address" -- of that frame's stacks)
        if (f.n < 0 || (f.n > 0 && (!f.pc || !f.cam_idx)) || !f.imgs || !f.sems || !f.T) { ctx->err = "k1n: bad frame (points, camera indices, image and class stacks, T)"; return -1; }
        total += f.n > 0 ? (f.n + TILE_PTS - 1) / TILE_PTS : 1;
    }
    if (total > K1N_MAX_BATCH_TILES) { ctx->err = "k1n: batch too large (split it: at most 16384 tiles of 512 points)"; return -1; }
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    // descriptors + tile -> frame table: built in pinned memory, one asynchronous upload
    const int64_t desc_bytes = (((int64_t)sizeof(K1nFrame) * n_frames + 255) & ~255ll), table_bytes = ((total * 4 + 255) & ~255ll);
    const int64_t up_bytes = desc_bytes + table_bytes;
    if (ctx->k1n_busy) { PCA_CHECK(ctx, hipEventSynchronize(ctx->k1n_ev)); ctx->k1n_busy = false; }
    if (up_bytes > ctx->k1n_pin_cap) {
        if (ctx->k1n_pin) PCA_CHECK(ctx, hipHostFree(ctx->k1n_pin));
        ctx->k1n_pin = nullptr; ctx->k1n_pin_cap = 0;
        PCA_CHECK(ctx, hipHostMalloc(&ctx->k1n_pin, (size_t)(2 * up_bytes), hipHostMallocMapped));
        ctx->k1n_pin_cap = 2 * up_bytes;
    }
    if (!ctx->k1n_ev) PCA_CHECK(ctx, hipEventCreateWithFlags(&ctx->k1n_ev, hipEventDisableTiming));
    if (accum_grow(ctx, &ctx->k1n_desc_dev, &ctx->k1n_desc_cap, up_bytes, s)) return -1;
    K1nFrame *hf = reinterpret_cast<K1nFrame *>(ctx->k1n_pin);
    int32_t *ht = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(ctx->k1n_pin) + desc_bytes);
    int32_t tile0 = 0;
    for (int k = 0; k < n_frames; ++k) {
        const pca_nusc_frame &f = frames[k];
        hf[k].pc = f.pc; hf[k].cam_idx = f.cam_idx; hf[k].imgs = f.imgs; hf[k].sems = f.sems;
        hf[k].n = f.n; hf[k].tile0 = tile0;
        for (int i = 0; i < 12; ++i) hf[k].T[i] = f.T[i];
        const int nt = f.n > 0 ? (f.n + TILE_PTS - 1) / TILE_PTS : 1;
        for (int t = 0; t < nt; ++t) ht[tile0 + t] = k;
        tile0 += nt;
    }
    // staging: sx sy sz f64 | si f32 | sc u32 | sn i32 | counts u32 | lastf i32
    const int64_t slots = total * TILE_PTS;
    const int64_t need = 3 * slots * 8 + 3 * slots * 4 + 2 * ((total * 4 + 255) & ~255ll) + 1024;
    if (accum_grow(ctx, &ctx->k1n_ws, &ctx->k1n_ws_cap, need, s)) return -1;
    if (ctx->profiling == 1) pca_prof_begin(ctx, PCA_K_NUSC, s);
    // (descriptors + tile table, ~20 KB: fetched by a kernel from the mapped host block -- a copy command of this size was
    // 13-17 us of a 60-95 us call; PCA_SMALL_COPY=1 restores it for A/B)
    static int small_copy = -1;
    if (small_copy < 0) { const char *e = getenv("PCA_SMALL_COPY"); small_copy = e ? atoi(e) : 0; }
    if (small_copy) PCA_CHECK(ctx, hipMemcpyAsync(ctx->k1n_desc_dev, ctx->k1n_pin, (size_t)up_bytes, hipMemcpyHostToDevice, s));
    else if (pca_fetch_block(ctx, ctx->k1n_pin, 0, ctx->k1n_desc_dev, up_bytes, s)) return -1;
    PCA_CHECK(ctx, hipEventRecord(ctx->k1n_ev, s));
    ctx->k1n_busy = true;
    K1nBatchArgs a;
    a.frames = reinterpret_cast<const K1nFrame *>(ctx->k1n_desc_dev);
    a.tile_frame = reinterpret_cast<const int32_t *>(reinterpret_cast<char *>(ctx->k1n_desc_dev) + desc_bytes);
    a.n_frames = n_frames; a.total_tiles = (int)total;
    a.ncam = ncam; a.H = H; a.W = W; a.sample_mode = sample_mode;
    for (int i = 0; i < 4; ++i) a.filt.w[i] = filter_mask ? filter_mask[i] : 0;
    char *w = reinterpret_cast<char *>(ctx->k1n_ws);
    a.sx = reinterpret_cast<double *>(w); w += slots * 8;
    a.sy = reinterpret_cast<double *>(w); w += slots * 8;
    a.sz = reinterpret_cast<double *>(w); w += slots * 8;
    a.si = reinterpret_cast<float *>(w); w += slots * 4;
    a.sc = reinterpret_cast<uint32_t *>(w); w += slots * 4;
    a.sn = reinterpret_cast<int32_t *>(w); w += slots * 4;
    a.counts = reinterpret_cast<uint32_t *>(w); w += (total * 4 + 255) & ~255ll;
    a.lastf = reinterpret_cast<int32_t *>(w);
    a.st = *store; a.frame_off = frame_off; a.first_slot = first_slot;
    a.status = ctx->ticket + 1;
    hipLaunchKernelGGL(k1n_front_batch, dim3((unsigned)total), dim3(K1N_BLK), 0, s, a);
    hipLaunchKernelGGL(k1n_append_batch, dim3((unsigned)total), dim3(256), 0, s, a);
    if (ctx->profiling == 1) pca_prof_end(ctx, s);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

int pca_nusc_project_cams(pca_ctx *ctx, const double *pc_lidar, int32_t n, const double T_ego_from_lidar[16],
                          const double T_glob_from_ego[16], const double *T_cam_from_glob, const double *K,
                          const double *wh, int ncam, double *pc_in_ego, double *uv, int64_t *cam_idx, void *stream)
{
    if (!ctx) return -1;
    if (ncam < 0 || ncam > MAX_CAMS) { ctx->err = "k0n: ncam out of range"; return -1; }
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    K0nArgs a;
    a.pc = pc_lidar; a.n = n; a.ncam = ncam;
    for (int i = 0; i < 16; ++i) { a.T_ego_from_lidar.m[i] = T_ego_from_lidar[i]; a.T_glob_from_ego.m[i] = T_glob_from_ego[i]; }
    for (int j = 0; j < ncam; ++j) {
        for (int i = 0; i < 16; ++i) a.T_cam_from_glob[j].m[i] = T_cam_from_glob[16 * j + i];
        for (int i = 0; i < 9; ++i) a.K[j][i] = K[9 * j + i];
        a.wh[j][0] = wh[2 * j]; a.wh[j][1] = wh[2 * j + 1];
    }
    a.pc_in_ego = pc_in_ego; a.uv = uv; a.cam_idx = cam_idx;
    const int grid = (n + SBLK - 1) / SBLK < 2048 ? (n + SBLK - 1) / SBLK : 2048;
    PCA_LAUNCH(ctx, PCA_K_PROJECT_CAMS, k0n_project, dim3(grid), dim3(SBLK), s, a);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

int pca_retransform(pca_ctx *ctx, const pca_store *store, const int64_t *frame_off, int slot_begin, int slot_end,
                    const double *Ts, int n_T, void *stream)
{
    if (!ctx) return -1;
    if (pca_k1_flush_pending(ctx)) return -1;               // a deferred K1 of this context comes first
    if (!store || !frame_off || !Ts || n_T < 0 || slot_end < slot_begin) { ctx->err = "k2: bad arguments"; return -1; }
    if (n_T == 0 || slot_end == slot_begin) return 0;
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    for (int t0 = 0; t0 < n_T; t0 += MAX_CHAIN) {
        K2Args a;
        a.x = store->x; a.y = store->y; a.z = store->z;
        a.frame_off = frame_off; a.slot_begin = slot_begin; a.slot_end = slot_end;
        a.n_T = (n_T - t0) < MAX_CHAIN ? (n_T - t0) : MAX_CHAIN;
        for (int t = 0; t < a.n_T; ++t)
            for (int i = 0; i < 12; ++i) a.T[t].m[i] = Ts[(int64_t)(t0 + t) * 16 + i];
        PCA_LAUNCH(ctx, PCA_K_RETRANSFORM, k2_retransform, dim3(2048), dim3(SBLK), s, a);
        PCA_CHECK(ctx, hipGetLastError());
    }
    return 0;
}

int pca_retransform_batch_tail(pca_ctx *ctx, const pca_store *store, const int64_t *frame_off, int first_slot, int n_frames,
                               const double *Ts, void *stream)
{
    if (!ctx) return -1;
    if (pca_k1_flush_pending(ctx)) return -1;               // a deferred K1 of this context comes first
    if (!store || !frame_off || !Ts || n_frames < 0 || n_frames > 65535) { ctx->err = "k2 tail: bad arguments"; return -1; }
    if (n_frames < 2) return 0;                  // a single frame owes nothing
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    for (int t0 = 1; t0 < n_frames; t0 += MAX_CHAIN) {          // transform 0 is owed by nobody in the batch
        K2TailArgs a;
        a.x = store->x; a.y = store->y; a.z = store->z;
        a.frame_off = frame_off; a.first_slot = first_slot;
        a.t_base = t0;
        a.n_T = (n_frames - t0) < MAX_CHAIN ? (n_frames - t0) : MAX_CHAIN;
        for (int t = 0; t < a.n_T; ++t)
            for (int i = 0; i < 12; ++i) a.T[t].m[i] = Ts[(int64_t)(t0 + t) * 16 + i];
        // frames t0 + n_T - 1 and later owe nothing of this pass
        const int rows = t0 + a.n_T - 1 < n_frames ? t0 + a.n_T - 1 : n_frames;
        PCA_LAUNCH(ctx, PCA_K_RETRANSFORM, k2_retransform_tail, dim3(32, rows), dim3(SBLK), s, a);
        PCA_CHECK(ctx, hipGetLastError());
    }
    return 0;
}

// (H,W,3) u8 -> (3,H,W) f32, (x / 255 - mean) / std: the input normalisation of the reference's semseg wrapper
// (utils/onnx_utils.py:26-29, torchvision ToTensor + Normalize) with IEEE f32 divisions, so the CNN sees the bits the
// reference feeds it while the image never leaves the device.
struct NormArgs { const uint8_t *rgb; float *out; int H, W; float mean[3], std[3]; };

__global__ __launch_bounds__(SBLK) void image_to_nchw_f32(const NormArgs a)
{
    const int64_t n = (int64_t)a.H * a.W;
    for (int64_t p = (int64_t)blockIdx.x * SBLK + threadIdx.x; p < n; p += (int64_t)gridDim.x * SBLK) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = (float)a.rgb[3 * p + c] / 255.0f;
            a.out[c * n + p] = (x - a.mean[c]) / a.std[c];
        }
    }
}

int pca_image_to_nchw_f32(pca_ctx *ctx, const uint8_t *rgb, int H, int W, const float mean[3], const float std[3], float *out,
                          void *stream)
{
    if (!ctx) return -1;
    if (!rgb || !out || !mean || !std || H < 1 || W < 1) { ctx->err = "normalise: bad arguments"; return -1; }
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    NormArgs a;
    a.rgb = rgb; a.out = out; a.H = H; a.W = W;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.std[c] = std[c]; }
    const int64_t n = (int64_t)H * W;
    const int grid = (int)((n + SBLK - 1) / SBLK < 4096 ? (n + SBLK - 1) / SBLK : 4096);
    hipLaunchKernelGGL(image_to_nchw_f32, dim3(grid), dim3(SBLK), 0, s, a);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

int pca_mark_dynamic(pca_ctx *ctx, const pca_store *store, const int64_t *frame_off, const int32_t *slots,
                     const int32_t *inst_idx, int n_pairs, void *stream)
{
    if (!ctx) return -1;
    if (pca_k1_flush_pending(ctx)) return -1;               // a deferred K1 of this context comes first
    if (!store || !frame_off || n_pairs < 0 || (n_pairs > 0 && (!slots || !inst_idx))) { ctx->err = "k3: bad arguments"; return -1; }
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    for (int p0 = 0; p0 < n_pairs; p0 += MAX_PAIRS) {
        K3Args a;
        a.inst = store->inst; a.dyn = store->dyn; a.frame_off = frame_off;
        a.n_pairs = (n_pairs - p0) < MAX_PAIRS ? (n_pairs - p0) : MAX_PAIRS;
        for (int i = 0; i < a.n_pairs; ++i) { a.slot[i] = slots[p0 + i]; a.inst_idx[i] = inst_idx[p0 + i]; }
        PCA_LAUNCH(ctx, PCA_K_MARK_DYNAMIC, k3_mark_dynamic, dim3(64, a.n_pairs), dim3(SBLK), s, a);
        PCA_CHECK(ctx, hipGetLastError());
    }
    return 0;
}


static inline int64_t dd_align(int64_t v) { return (v + 255) & ~255ll; }
static inline int64_t dedup_capacity(int64_t max_points)
{
    int64_t cap = 1024;
    while (cap < 2 * max_points) cap <<= 1;
    return cap;
}

int64_t pca_voxel_dedup_workspace_bytes(int64_t max_points, int n_slots)
{
    if (max_points < 1) max_points = 1;
    const int64_t cap = dedup_capacity(max_points);
    return dd_align(cap * 8) + dd_align(cap * 4) + dd_align((int64_t)(n_slots + 2) * 8) + 512;
}

int pca_voxel_dedup(pca_ctx *ctx, const pca_store *store, int64_t *frame_off, int slot_begin, int slot_end,
                    double voxel_size, int64_t max_points, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!ctx) return -1;
    if (pca_k1_flush_pending(ctx)) return -1;               // a deferred K1 of this context comes first
    if (!store || !frame_off || !workspace || slot_end < slot_begin) { ctx->err = "dedup: bad arguments"; return -1; }
    if (!(voxel_size > 0.0)) { ctx->err = "dedup: voxel_size must be positive"; return -1; }
    if (slot_end == slot_begin) return 0;
    if (max_points < 1) max_points = 1;
    if (max_points >= (1ll << 31)) { ctx->err = "dedup: window too large"; return -1; }
    if (workspace_bytes < pca_voxel_dedup_workspace_bytes(max_points, slot_end - slot_begin)) { ctx->err = "dedup: workspace too small"; return -1; }
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    DedupArgs a;
    a.st = *store;
    a.frame_off = frame_off;
    a.slot_begin = slot_begin; a.slot_end = slot_end;
    a.max_points = max_points;
    a.size = voxel_size;
    const int64_t cap = dedup_capacity(max_points);
    char *w = reinterpret_cast<char *>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    a.keys = reinterpret_cast<unsigned long long *>(w); w += dd_align(cap * 8);
    a.vals = reinterpret_cast<uint32_t *>(w); w += dd_align(cap * 4);
    a.new_off = reinterpret_cast<int64_t *>(w);
    a.cap_mask = (uint64_t)cap - 1;
    constexpr int BLK = 256;
    a.total_tiles = (int)((max_points + PPT * BLK - 1) / (PPT * BLK));
    if (pca_ctx_reserve_tiles(ctx, a.total_tiles, s)) return -1;
    a.state = ctx->tile_state;
    a.ticket = ctx->ticket;
    a.epoch = pca_ctx_next_epoch(ctx, s);
    PCA_CHECK(ctx, hipMemsetAsync(a.keys, 0, (size_t)cap * 8, s));
    PCA_CHECK(ctx, hipMemsetAsync(a.vals, 0xff, (size_t)cap * 4, s));
    const int64_t g = (max_points + SBLK - 1) / SBLK;
    PCA_LAUNCH(ctx, PCA_K_DEDUP, dedup_insert, dim3((unsigned)(g < 4096 ? g : 4096)), dim3(SBLK), s, a);
    PCA_LAUNCH(ctx, PCA_K_DEDUP, dedup_compact<BLK>, dim3(a.total_tiles), dim3(BLK), s, a);
    PCA_LAUNCH(ctx, PCA_K_DEDUP, dedup_offsets, dim3(1), dim3(SBLK), s, a);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

}  // extern "C"
