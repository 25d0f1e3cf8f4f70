// pca_k1_body.h -- the device code of K1 (kitti_project_sample_filter): helpers, argument block and the tile body k1_body.
// Included by pca_k1.hip (the K1 kernels) and by pca_bev.hip (round 5: level 1 of the raster can carry the K1 of the frame that
// was integrated just before it in the same launch, see bev_tile_bin).  The description of the algorithm is in pca_k1.hip.
#pragma once
#include "pca_common.h"

template <typename T>
__device__ __forceinline__ T k1_ldg(const T *p)
{
    return *reinterpret_cast<const __attribute__((address_space(1))) T *>(reinterpret_cast<uintptr_t>(p));
}
struct __attribute__((packed)) K1U32u { uint32_t v; };
__device__ __forceinline__ uint32_t k1_ldg_u32_unaligned(const uint8_t *p)     // one global_load_dword at any byte address
{
    return reinterpret_cast<const __attribute__((address_space(1))) K1U32u *>(reinterpret_cast<uintptr_t>(p))->v;
}
typedef float k1_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 k1_ldg4(const float *p)      // one 16-byte global load
{
    const k1_f32x4 v = *reinterpret_cast<const __attribute__((address_space(1))) k1_f32x4 *>(reinterpret_cast<uintptr_t>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t k1_xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }
__device__ __forceinline__ int64_t k1_uniform_i64(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ uint32_t k1_mbcnt(uint64_t m)      // set bits of m below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---- cache-policy knobs of the batched front kernel (compile time; tools/experiments/k1_variants.sh builds and times them) ----
// K1_STREAM_NT: bit 0 = the point stream is loaded non-temporally, bit 1 = the staging records are stored non-temporally
//               (both are touched once by this kernel: they should not push the frame's image lines out of its L2)
// K1_GATHER_MODE: how the class byte and the colour dword are gathered: 0 plain, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1
#ifndef K1_STREAM_NT
#define K1_STREAM_NT 0
#endif
#ifndef K1_GATHER_MODE
#define K1_GATHER_MODE 0
#endif
template <typename T>
__device__ __forceinline__ T k1_gather(const T *p)
{
    typedef const __attribute__((address_space(1))) T *G;           // (global_load_*, not flat_load_*)
    G g = reinterpret_cast<G>(reinterpret_cast<uintptr_t>(p));
#if K1_GATHER_MODE == 1
    return __builtin_nontemporal_load(g);
#elif K1_GATHER_MODE == 2
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#elif K1_GATHER_MODE == 3
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#elif K1_GATHER_MODE == 4
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    return *g;
#endif
}
__device__ __forceinline__ uint32_t k1_gather_u32_unaligned(const uint8_t *p)
{
#if K1_GATHER_MODE == 0
    return k1_ldg_u32_unaligned(p);
#else
    return k1_gather(reinterpret_cast<const uint32_t *>(p));      // (experiment builds only: the hardware takes the unaligned address)
#endif
}
__device__ __forceinline__ float4 k1_load_point(const float *p, bool split)
{
#if K1_STREAM_NT & 1
    if (split) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) f4 *>(reinterpret_cast<uintptr_t>(p)));
        return make_float4(v.x, v.y, v.z, v.w);
    }
#endif
    return k1_ldg4(p);
}

#define K1_MAXQ 8
#define K1_APPEND_BLK 256
#define K1_SCAN_BLK 1024

struct K1Args {
    const K1Frame *frames;              // dev [n_frames], or nullptr -> `one`.  FUSED: frame order; SPLIT: sorted by queue
    K1Frame one;
    int n_frames;
    int n_queues;
    int qframe0[K1_MAXQ + 1];           // SPLIT: frames of queue q = [qframe0[q], qframe0[q+1]) of `frames` ...
    int qbase, qrem;                    // ... = q * qbase + min(q, qrem) (+ qbase + (q < qrem)): n_frames / Q, n_frames % Q
    int qtiles[K1_MAXQ];                // SPLIT: tiles in queue q
    Mat34 P;
    int H, W;
    float cull[16];                     // f32 rows x[4] y[4] d[4] of P, then s, c (error bound = s*max|xyz| + c), W-.5, H-.5
    ClassMask filt;
    pca_store st;                       // FUSED
    int64_t *frame_off;
    int first_slot;
    uint64_t *state;
    uint32_t *status;
    uint32_t epoch;
    int sample_mode;                    // 0 nearest (the reference); 1 bilinear rgb (opt-in), class stays nearest
    int tpf;                            // tiles per frame if every frame of the launch has the same, else 0
    float4 *rec_p;                      // SPLIT: [tiles][TILE] kept records: x, y, z, intensity (f32, as loaded)
    uint32_t *rec_c;                    // SPLIT: [tiles][TILE]               rgb | class << 24
    uint32_t *counts;                   // SPLIT: [tiles] kept points
    int32_t *lastf;                     // SPLIT: [tiles] frame index if the tile is the last of its frame, else -1
    unsigned long long *dbg;            // diagnostic stamps (PCA_K1_STAMPS=1): 8 words per workgroup, else nullptr
};

#define K1_FLAT_BLOCK ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x)
#define K1_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[K1_FLAT_BLOCK * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// velo2frame + velo2img of one point: pixel index v*W+u, or -1 if outside the frustum (sem_pc_accum.py:347-394).
// P rows as fma chains in k order (= the dgemm numpy runs), IEEE f64 divide, np.round = rint.
__device__ __forceinline__ int k1_project_pixel(const Mat34 &P, float xf, float yf, float zf, int W, int H,
                                                double *uq = nullptr, double *vq = nullptr)
{
    const double x = (double)xf, y = (double)yf, z = (double)zf;
    const double fx = row4(P.m + 0, x, y, z);
    const double fy = row4(P.m + 4, x, y, z);
    double d = row4(P.m + 8, x, y, z);
    if (d == 0.0) d = -1e-6;
    const double ad = fabs(d);
    const double qu = fx / ad, qv = fy / ad;
    const double uf = rint(qu);
    const double vf = rint(qv);
    if (uq) { *uq = qu; *vq = qv; }
    const bool ok = (uf >= 0.0) && (uf < (double)W) && (vf >= 0.0) && (vf < (double)H) && (d > 0.0) && (d < __builtin_huge_val());
    return ok ? (int)vf * W + (int)uf : -1;
}

// The frame whose key (tile0 or qpos0, ascending over frames[f_lo, f_hi)) is the last one <= pos.  Every lane loads one
// whole 48-byte descriptor, so the lookup is a single memory round trip; the holder hands it out through readlane.
template <bool BY_TILE0>
__device__ __forceinline__ K1Frame k1_find_frame(const K1Frame *frames, int f_lo, int f_hi, int pos, int lane)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t best[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int f0 = f_lo; f0 < f_hi; f0 += 64) {
        const int f = f0 + lane;
        u32x4 w[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        if (f < f_hi) {
            const __attribute__((address_space(1))) u32x4 *p =
                reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(reinterpret_cast<uintptr_t>(frames + f));
            w[0] = p[0]; w[1] = p[1]; w[2] = p[2];
        }
        const int key = (int)(BY_TILE0 ? w[2].y : w[2].z);
        const int c = (int)__popcll(__ballot(f < f_hi && key <= pos));
        if (c > 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                best[4 * i + 0] = (uint32_t)__builtin_amdgcn_readlane((int)w[i].x, c - 1);
                best[4 * i + 1] = (uint32_t)__builtin_amdgcn_readlane((int)w[i].y, c - 1);
                best[4 * i + 2] = (uint32_t)__builtin_amdgcn_readlane((int)w[i].z, c - 1);
                best[4 * i + 3] = (uint32_t)__builtin_amdgcn_readlane((int)w[i].w, c - 1);
            }
        }
        if (c < 64) break;
    }
    K1Frame fr;
    auto ptr = [&](int i) { return (uintptr_t)(((uint64_t)best[i + 1] << 32) | best[i]); };
    fr.pts = reinterpret_cast<const float *>(ptr(0));
    fr.rgb = reinterpret_cast<const uint8_t *>(ptr(2));
    fr.sem = reinterpret_cast<const uint8_t *>(ptr(4));
    fr.sem_gt = reinterpret_cast<const uint8_t *>(ptr(6));
    fr.n = (int32_t)best[8]; fr.tile0 = (int32_t)best[9]; fr.qpos0 = (int32_t)best[10]; fr.f = (int32_t)best[11];
    return fr;
}
static_assert(sizeof(K1Frame) == 48, "k1_find_frame reads a descriptor as three 16-byte words");

// One descriptor at a workgroup-uniform address, read through the scalar cache (s_load_dwordx4 x 3: one short hop instead
// of vector loads from the kernel-argument segment -- 2.6 us of a tile's 12.5 were spent waiting for those).  The
// descriptors are written before the launch (kernel arguments, or an upload on the same stream), never during it.
__device__ __forceinline__ K1Frame k1_frame_uniform(const K1Frame *f)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)reinterpret_cast<uintptr_t>(f));
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(reinterpret_cast<uintptr_t>(f) >> 32));
    const __attribute__((address_space(4))) u32x4 *p =
        reinterpret_cast<const __attribute__((address_space(4))) u32x4 *>(((uintptr_t)hi << 32) | lo);
    const u32x4 w0 = p[0], w1 = p[1], w2 = p[2];
    auto ptr = [](uint32_t a, uint32_t b) { return (uintptr_t)(((uint64_t)b << 32) | a); };
    K1Frame fr;
    fr.pts = reinterpret_cast<const float *>(ptr(w0.x, w0.y));
    fr.rgb = reinterpret_cast<const uint8_t *>(ptr(w0.z, w0.w));
    fr.sem = reinterpret_cast<const uint8_t *>(ptr(w1.x, w1.y));
    fr.sem_gt = reinterpret_cast<const uint8_t *>(ptr(w1.z, w1.w));
    fr.n = (int32_t)w2.x; fr.tile0 = (int32_t)w2.y; fr.qpos0 = (int32_t)w2.z; fr.f = (int32_t)w2.w;
    return fr;
}
// qframe0[q] of the kernel arguments without a dynamically indexed load (a select chain on scalar registers)
template <int N>
__device__ __forceinline__ int k1_pick(const int (&v)[N], int q)
{
    int r = v[0];
#pragma unroll
    for (int i = 1; i < N; ++i) r = q == i ? v[i] : r;
    return r;
}

#define K1_INLINE_FRAMES 64
struct K1InlineFrames { K1Frame f[K1_INLINE_FRAMES]; };   // descriptors of a small batch travel in the kernel arguments
// k1_kitti_inl(K1Args a, K1InlineFrames inl): explicit arguments lie in the kernel-argument segment in order, from offset 0
static_assert(sizeof(K1Args) % alignof(K1InlineFrames) == 0, "inl follows a without padding");
__device__ __forceinline__ const K1Frame *k1_inline_frames()
{
    auto ka = __builtin_amdgcn_kernarg_segment_ptr();
    return reinterpret_cast<const K1Frame *>(reinterpret_cast<uintptr_t>(ka) + sizeof(K1Args));
}

// Tail: called by every thread at the end of the FUSED path with the tile's kept points as the lanes hold them -- p[r] (x, y, z,
// intensity as loaded), packed[r] (rgb | class << 24), km[r] (ballot of the lanes that keep round r's candidate) -- for a caller
// that has more to do with them than the append (level 1 of the raster bins them); K1NoTail: nothing.
struct K1NoTail { template <int PPT> __device__ __forceinline__ void operator()(const float4 (&)[PPT], const uint32_t (&)[PPT], const uint64_t (&)[PPT]) const {} };
template <int BLK, int PPT, bool SPLIT, bool BILIN, bool INL, typename Tail = K1NoTail>
__device__ __forceinline__ void k1_body(const K1Args &a, const Tail &tail = Tail())
{
    constexpr int TILE = BLK * PPT, NW = BLK / PCA_WAVE, NC = PPT * NW;
    static_assert(NC <= 64, "the per-(row, wave) counts are scanned by one wave");
    static_assert(TILE <= 65536, "tile-local indices are 16 bits");
    static_assert(PPT % 2 == 0, "phase 1 tests two rows per packed instruction");
    __shared__ float4 s_candp[TILE];       // the candidates (x, y, z, intensity), in point order
    __shared__ uint16_t s_cand[TILE];      // their tile-local indices (use_gt_sem only)
    __shared__ uint32_t s_cnt[NC];         // per (row, wave) counts: candidates, later kept points
    __shared__ uint32_t s_filt[8];         // 256-bit class filter
    __shared__ long long s_excl;           // FUSED: exclusive prefix of the tile
    __shared__ uint32_t s_box[6];          // FUSED: the box of the tile's kept points (ordered encodings; pca_store.frame_box)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    K1_STAMP(0);

    // ---------------- which tile (no divisions: the grid is shaped by the host) ----------------
    //   FUSED                  grid (tiles)                          block = tile in output order
    //   SPLIT, equal frames    grid (Q, tiles per frame, frames / Q) x = queue, y = tile of the frame, z = frame of the queue
    //   SPLIT, ragged          grid (Q, longest queue)               x = queue, y = position in the queue
    K1Frame fr;
    int tin;
    if (!SPLIT) {
        fr = a.frames ? k1_find_frame<true>(a.frames, 0, a.n_frames, (int)blockIdx.x, lane) : a.one;
        tin = (int)blockIdx.x - fr.tile0;
    } else {
        // frames k = q (mod Q) form queue q, queues laid out one after the other: its range in `frames` in closed form (no load)
        const int q = blockIdx.x, f_lo = q * a.qbase + (q < a.qrem ? q : a.qrem), f_hi = f_lo + a.qbase + (q < a.qrem ? 1 : 0);
        if (a.tpf) {
            if (f_lo + (int)blockIdx.z >= f_hi) return;                   // queues of unequal length
            // (inline descriptors: addressed through the kernel-argument segment pointer -- taking the address of the
            // by-value parameter itself would make the compiler copy all 3 KB of it to scratch)
            fr = INL ? k1_frame_uniform(k1_inline_frames() + f_lo + blockIdx.z)
                     : a.frames ? k1_frame_uniform(a.frames + f_lo + blockIdx.z) : a.one;
            tin = blockIdx.y;
        } else {
            if ((int)blockIdx.y >= k1_pick(a.qtiles, q)) return;
            fr = k1_find_frame<false>(a.frames, f_lo, f_hi, (int)blockIdx.y, lane);
            tin = (int)blockIdx.y - fr.qpos0;
        }
    }
    if (threadIdx.x < 8) {                                 // (a select chain over scalar registers: a dynamically indexed
        uint32_t w = (uint32_t)a.filt.w[0];                //  read of the kernel arguments is a vector load that wave 0 would
#pragma unroll                                             //  wait for before it issues its point loads)
        for (int i = 1; i < 8; ++i) w = (int)threadIdx.x == i ? (uint32_t)(a.filt.w[i >> 1] >> (32 * (i & 1))) : w;
        s_filt[threadIdx.x] = w;
    }
    const int tile = fr.tile0 + tin;                       // index in output (frame-major) order
    const int ftiles = fr.n > 0 ? (fr.n + TILE - 1) / TILE : 1;
    const int base_pt = tin * TILE;
    const int n_here = fr.n - base_pt < TILE ? fr.n - base_pt : TILE;     // points of this tile (0 for an empty frame)
    const bool gt = fr.sem_gt != nullptr;                  // use_gt_sem: no projection, rgb = 0
    const float *pts = fr.pts + 4 * (int64_t)base_pt;
    K1_STAMP(1);

    // ---------------- phase 1: conservative frustum test of every point ----------------
    // f32 estimates of the three projection rows, two points per packed instruction; a point is dropped only if one of
    //   depth, u + 0.5 depth, (W - 0.5) depth - u, v + 0.5 depth, (H - 0.5) depth - v
    // is below minus the error bound (NaN / inf never drop a point: every compare is false)
    typedef float f2 __attribute__((ext_vector_type(2)));
    uint64_t cm[PPT];
    float4 v[PPT];
    if (n_here > 0) {                                      // (uniform) all PPT loads of a lane back to back, no branches:
#pragma unroll                                             // a lane past the end re-reads the tile's last point
        for (int k = 0; k < PPT; ++k) {
            const int idx = k * BLK + (int)threadIdx.x;
            v[k] = k1_load_point(pts + 4 * (idx < n_here ? idx : n_here - 1), SPLIT);   // 16 B / lane, fully coalesced
        }
    } else {
#pragma unroll
        for (int k = 0; k < PPT; ++k) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    {
        const float *cu = a.cull;
        auto sp = [](float c) { f2 r; r.x = c; r.y = c; return r; };
#pragma unroll
        for (int k = 0; k < PPT; k += 2) {
            bool cand0 = k * BLK + (int)threadIdx.x < n_here, cand1 = (k + 1) * BLK + (int)threadIdx.x < n_here;
            if (!gt) {
                f2 x, y, z, m;
                x.x = v[k].x; x.y = v[k + 1].x; y.x = v[k].y; y.y = v[k + 1].y; z.x = v[k].z; z.y = v[k + 1].z;
                m.x = fmaxf(fmaxf(fabsf(x.x), fabsf(y.x)), fabsf(z.x));
                m.y = fmaxf(fmaxf(fabsf(x.y), fabsf(y.y)), fabsf(z.y));
                const f2 M = __builtin_elementwise_fma(sp(cu[12]), m, sp(cu[13]));       // error bound of every form
                const f2 fx = __builtin_elementwise_fma(sp(cu[2]), z, __builtin_elementwise_fma(sp(cu[1]), y, __builtin_elementwise_fma(sp(cu[0]), x, sp(cu[3]))));
                const f2 fy = __builtin_elementwise_fma(sp(cu[6]), z, __builtin_elementwise_fma(sp(cu[5]), y, __builtin_elementwise_fma(sp(cu[4]), x, sp(cu[7]))));
                const f2 d = __builtin_elementwise_fma(sp(cu[10]), z, __builtin_elementwise_fma(sp(cu[9]), y, __builtin_elementwise_fma(sp(cu[8]), x, sp(cu[11]))));
                const f2 t1 = __builtin_elementwise_fma(sp(0.5f), d, fx), t2 = __builtin_elementwise_fma(sp(cu[14]), d, -fx);
                const f2 t3 = __builtin_elementwise_fma(sp(0.5f), d, fy), t4 = __builtin_elementwise_fma(sp(cu[15]), d, -fy);
                // bitwise on purpose: compares, no branches
                const int rej0 = (int)(d.x < -M.x) | (int)(t1.x < -M.x) | (int)(t2.x < -M.x) | (int)(t3.x < -M.x) | (int)(t4.x < -M.x);
                const int rej1 = (int)(d.y < -M.y) | (int)(t1.y < -M.y) | (int)(t2.y < -M.y) | (int)(t3.y < -M.y) | (int)(t4.y < -M.y);
                cand0 = cand0 && !rej0;
                cand1 = cand1 && !rej1;
            }
            cm[k] = __ballot(cand0);
            cm[k + 1] = __ballot(cand1);
            if (lane == 0) { s_cnt[k * NW + wave] = (uint32_t)__popcll(cm[k]); s_cnt[(k + 1) * NW + wave] = (uint32_t)__popcll(cm[k + 1]); }
        }
    }
    __syncthreads();
    K1_STAMP(2);
    uint32_t ncand;
    {
        const uint32_t c = lane < NC ? s_cnt[lane] : 0u;
        const uint32_t inc = wave_incl_scan_add(c);
        const uint32_t exc = inc - c;
        ncand = (uint32_t)__builtin_amdgcn_readlane((int)inc, NC - 1);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)exc, k * NW + wave);
            if ((cm[k] >> lane) & 1ull) {
                const uint32_t o = off + k1_mbcnt(cm[k]);
                s_candp[o] = v[k];
                if (gt) s_cand[o] = (uint16_t)(k * BLK + threadIdx.x);
            }
        }
    }
    __syncthreads();
    K1_STAMP(3);

    // ---------------- phase 2: exact projection, gathers and class filter of the candidates ----------------
    // round r handles candidates r*BLK .. ; three sweeps over the rounds keep the loads of all rounds in flight together
    uint64_t km[PPT];
    uint32_t packed[PPT];
    float4 p[PPT];
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        p[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        const uint32_t j = (uint32_t)(r * BLK) + threadIdx.x;
        if (j < ncand) p[r] = s_candp[j];
    }
    if (gt) {
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            const uint32_t j = (uint32_t)(r * BLK) + threadIdx.x;
            unsigned c = 0;
            const bool act = j < ncand;
            if (act) c = k1_ldg(fr.sem_gt + base_pt + s_cand[j]);
            packed[r] = c << 24;
            km[r] = __ballot(act && !((s_filt[c >> 5] >> (c & 31u)) & 1u));
        }
    } else {
        unsigned cls[PPT], rgb[PPT], rsh[PPT];
        int pixr[PPT];
        bool ok[PPT];
        // Batches: the colour is gathered for the KEPT points only, after the class filter (a second, dependent round trip,
        // 24 % fewer colour gathers).  What bounds the batch form on scattered points is the number of distinct lines its
        // gathers pull through the L1s -- 64 per wave instruction, a whole line for 1 or 4 useful bytes: every gather at
        // pixel 0 instead takes the front kernel from 57 to 31 us.  One frame (FUSED) is latency-bound: both at once.
        constexpr bool dep = SPLIT;
        const int last = a.H * a.W * 3 < 4 ? 0 : a.H * a.W * 3 - 4;      // last legal 4-byte window of the image
        // r | g<<8 | b<<16 of one pixel as ONE unaligned dword + the shift that brings the pixel to bit 0.  The shift is
        // applied in the third sweep: nothing in this one consumes a gathered value, so the gathers of ALL rounds of a
        // wave are in flight together (consumed inside the round, each round waited for its own round trip).
        // (an image of fewer than four bytes reaches the kernel as a padded copy: pca_kitti_project_sample_filter_ex)
        auto rgb_raw = [&](int pix, unsigned &sh) -> uint32_t {
            const int off = pix * 3;
            sh = off > last ? (unsigned)(off - last) * 8u : 0u;
            return SPLIT ? k1_gather_u32_unaligned(fr.rgb + (off > last ? last : off)) : k1_ldg_u32_unaligned(fr.rgb + (off > last ? last : off));
        };
        auto rgb_at = [&](int pix) -> uint32_t { unsigned sh; const uint32_t w = rgb_raw(pix, sh); return (w >> sh) & 0xffffffu; };
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            ok[r] = false; cls[r] = 0; rgb[r] = 0; rsh[r] = 0;
            if ((uint32_t)(r * BLK + wave * PCA_WAVE) < ncand) {       // wave-uniform: a wave without candidates in this round skips it
                const uint32_t j = (uint32_t)(r * BLK) + threadIdx.x;
                double qu = 0.0, qv = 0.0;
                const int px = BILIN ? k1_project_pixel(a.P, p[r].x, p[r].y, p[r].z, a.W, a.H, &qu, &qv)
                                     : k1_project_pixel(a.P, p[r].x, p[r].y, p[r].z, a.W, a.H);
                ok[r] = j < ncand && px >= 0;
                const int pix = ok[r] ? px : 0;                           // pixel 0 is always a valid address
                // two gathers per point: the class byte and ONE unaligned dword holding r,g,b
                cls[r] = SPLIT ? k1_gather(fr.sem + pix) : k1_ldg(fr.sem + pix);
                pixr[r] = pix;
                if (!BILIN) {
                    if (!dep) rgb[r] = rgb_raw(pix, rsh[r]);
                } else {                                                   // opt-in: bilinear colour, neighbours clamped to the image
                    const Bilin b = bilin_weights<false>(ok[r] ? qu : 0.0, ok[r] ? qv : 0.0);
                    auto cl = [](double v, int n) { return (int)(v < 0.0 ? 0.0 : (v > (double)(n - 1) ? (double)(n - 1) : v)); };
                    const int u0 = cl(b.u0, a.W), u1 = cl(b.u1, a.W), v0 = cl(b.v0, a.H), v1 = cl(b.v1, a.H);
                    rgb[r] = bilin_rgb(b, rgb_at(v0 * a.W + u0), rgb_at(v1 * a.W + u1), rgb_at(v1 * a.W + u0), rgb_at(v0 * a.W + u1));
                }
            }
        }
        if (!BILIN && dep) {
#pragma unroll
            for (int r = 0; r < PPT; ++r) {
                const unsigned c = cls[r];
                ok[r] = ok[r] && !((s_filt[c >> 5] >> (c & 31u)) & 1u);
                if (ok[r]) rgb[r] = rgb_raw(pixr[r], rsh[r]);
            }
#pragma unroll
            for (int r = 0; r < PPT; ++r) {
                packed[r] = ((rgb[r] >> rsh[r]) & 0xffffffu) | (cls[r] << 24);
                km[r] = __ballot(ok[r]);
            }
        } else {
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            const unsigned c = cls[r];
            packed[r] = ((rgb[r] >> rsh[r]) & 0xffffffu) | (c << 24);
            km[r] = __ballot(ok[r] && !((s_filt[c >> 5] >> (c & 31u)) & 1u));
        }
        }
    }
#pragma unroll
    for (int r = 0; r < PPT; ++r)
        if (lane == 0) s_cnt[r * NW + wave] = (uint32_t)__popcll(km[r]);
    if (!SPLIT && threadIdx.x < 6) s_box[threadIdx.x] = 0u;
    __syncthreads();
    K1_STAMP(4);
    uint32_t koff[PPT], total;
    {
        const uint32_t c = lane < NC ? s_cnt[lane] : 0u;
        const uint32_t inc = wave_incl_scan_add(c);
        const uint32_t exc = inc - c;
        total = (uint32_t)__builtin_amdgcn_readlane((int)inc, NC - 1);
#pragma unroll
        for (int r = 0; r < PPT; ++r) koff[r] = (uint32_t)__builtin_amdgcn_readlane((int)exc, r * NW + wave);
    }

    if (SPLIT) {
        // ---------------- the tile's kept records, in point order ----------------
        float4 *rp = a.rec_p + (size_t)tile * TILE;
        uint32_t *rc = a.rec_c + (size_t)tile * TILE;
#pragma unroll
        for (int r = 0; r < PPT; ++r)
            if ((km[r] >> lane) & 1ull) {
                const uint32_t o = koff[r] + k1_mbcnt(km[r]);
#if K1_STREAM_NT & 2
                typedef float f4 __attribute__((ext_vector_type(4)));
                f4 w; w.x = p[r].x; w.y = p[r].y; w.z = p[r].z; w.w = p[r].w;
                __builtin_nontemporal_store(w, reinterpret_cast<__attribute__((address_space(1))) f4 *>(reinterpret_cast<uintptr_t>(rp + o)));
                __builtin_nontemporal_store(packed[r], reinterpret_cast<__attribute__((address_space(1))) uint32_t *>(reinterpret_cast<uintptr_t>(rc + o)));
#else
                rp[o] = p[r];
                rc[o] = packed[r];
#endif
            }
        if (threadIdx.x == 0) { a.counts[tile] = total; a.lastf[tile] = tin == ftiles - 1 ? fr.f : -1; }
        K1_STAMP(5);
        K1_STAMP(6);
        return;
    }

    // ---------------- FUSED: look-back, then append the kept records ----------------
    // The box of the frame's kept points, in the frame's own coordinates (pca.h: pca_store.frame_box): six atomic max on
    // order-preserving encodings of the f32 values (upper bounds) and their complements (lower bounds); 0 = nothing yet.  Per wave
    // in registers, per tile through LDS, six global atomics per TILE -- per wave they were 2 880 same-address atomics per frame,
    // and the L2 channel they hit served nobody else for ~20 us (every first-round workgroup of level 1 beside it took 43 us
    // instead of 25).
    if (a.st.frame_box) {                                   // (uniform)
        uint32_t lo[3] = {0u, 0u, 0u}, hi[3] = {0u, 0u, 0u};
#pragma unroll
        for (int r = 0; r < PPT; ++r)
            if ((km[r] >> lane) & 1ull) {
                const float c[3] = {p[r].x, p[r].y, p[r].z};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const uint32_t o = pca_f32_ordered(c[k]);
                    hi[k] = hi[k] > o ? hi[k] : o;
                    lo[k] = lo[k] > ~o ? lo[k] : ~o;
                }
            }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t l = wave_reduce_max(lo[k]), h = wave_reduce_max(hi[k]);
            if (lane == 0 && h) { atomicMax(&s_box[2 * k], l); atomicMax(&s_box[2 * k + 1], h); }
        }
    }
    if (wave == 0) {
        const uint64_t e = lb_exclusive_prefix(a.state, tile, (uint64_t)total, a.epoch, a.status);
        if (lane == 0) s_excl = (long long)e;
    }
    __syncthreads();
    K1_STAMP(5);
    if (a.st.frame_box && threadIdx.x < 6 && s_box[threadIdx.x]) atomicMax(a.st.frame_box + (size_t)(a.first_slot + fr.f) * 6 + threadIdx.x, s_box[threadIdx.x]);
    const int64_t tile_base = a.frame_off[a.first_slot] + k1_uniform_i64(s_excl);
    const int64_t room64 = a.st.capacity - tile_base;
    const uint32_t room = room64 <= 0 ? 0u : (room64 > 0x7fffffffll ? 0x7fffffffu : (uint32_t)room64);
    double *xb = a.st.x + tile_base, *yb = a.st.y + tile_base, *zb = a.st.z + tile_base;
    float *ib = a.st.intensity + tile_base;
    uint32_t *cb = a.st.rgbs + tile_base;
    int32_t *nb = a.st.inst + tile_base;
    uint8_t *db = a.st.dyn + tile_base;
    bool overflow = false;
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
        if (km[r] == 0) continue;                          // uniform
        if ((km[r] >> lane) & 1ull) {
            const uint32_t o = koff[r] + k1_mbcnt(km[r]);
            if (o >= room) { overflow = true; continue; }
            xb[o] = (double)p[r].x;
            yb[o] = (double)p[r].y;
            zb[o] = (double)p[r].z;
            ib[o] = p[r].w;
            cb[o] = packed[r];
            nb[o] = 0;
            db[o] = 0;
        }
    }
    if (overflow) pca_raise(a.status, PCA_STATUS_STORE_OVERFLOW);
    if (threadIdx.x == 0 && tin == ftiles - 1)             // the last tile of a frame closes its segment
        a.frame_off[a.first_slot + fr.f + 1] = tile_base + total;
    K1_STAMP(6);
    tail.template operator()<PPT>(p, packed, km);
}

