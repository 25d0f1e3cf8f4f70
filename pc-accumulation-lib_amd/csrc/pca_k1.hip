// pca_k1.hip -- K1  kitti_project_sample_filter for gfx950 (MI355X)
//   fused  velo2frame -> velo2img (frustum mask) -> nearest sample of semseg + RGB -> class filter -> stable append
//   (sem_pc_accum.py:317-402, kitti360_sem_pc_accum.py:132-156 of the reference), one launch for a batch of frames.
//
// What bounds it (measured): per frame the kernel moves 1.9 MB of points, the image lines its ~60 k gathers
// touch and 1 MB of kept records; it does ~30 f64 operations for the 29 % of the points that fall into the frustum.
// On scattered points the batch form is bound by its GATHERS, not by HBM: a wave-wide gather pulls 64 distinct lines
// through the CU's L1 for 64 x (1 or 4) useful bytes (with every gather at pixel 0 the front kernel runs in 31 us instead
// of 57; without phase 2 at all in 23 us = 5.3 TB/s).  So the design is about (1) not fetching an image into eight L2s,
// (2) not spending vector issue slots on the 71 % of the points that are outside the frustum, (3) as few gathered lines
// as the reference's semantics allow, (4) enough independent workgroups in flight to cover the chain
// descriptor -> points -> class gather -> colour gather -> stores, with every link a single round trip.
//
//   * one workgroup = one tile of BLK*PPT consecutive points of one frame.
//   * phase 1 (every point): 16-byte load, f32 estimate of the three projection rows with a certified error bound,
//     conservative frustum test -> candidates (a superset of the in-frustum points), compacted into an LDS list.
//   * phase 2 (candidates only, dense lanes): the exact f64 path -- fma chain, IEEE divide, rint, 6-way mask -- two
//     gathers (class byte, one unaligned dword for r,g,b), 256-bit class filter from LDS.  The rounds of a tile are
//     software-pipelined: all point re-loads, then all projections and gathers, then all filters; no sweep consumes what
//     it gathers, so the gathers of all rounds of a wave are in flight together.  Batches gather the colour after the
//     class filter, for the kept points only (24 % fewer colour gathers; one frame gathers both at once: latency).
//   * stable compaction inside the tile: ballot ranks + one 64-lane DPP scan per workgroup.
//   * across tiles, two forms:
//       FUSED (a launch of at most one tile per CU: every workgroup is resident, tile = blockIdx): decoupled
//         look-back (8-byte {flag, epoch, value} granules, relaxed agent-scope atomics, bounded spin), then the SoA
//         stores of the kept records.  One launch: this is what integrate() of one frame runs.
//       SPLIT (batches): no workgroup ever waits for another.  The front kernel writes each tile's kept records
//         (x, y, z, intensity as loaded + rgb | class: 20 B per kept point) and its count; k1_append adds up the counts
//         before its tile and streams the records into the SoA store, fully coalesced.  Grid x = queue, queue q holding
//         the frames f = q (mod Q): with the round-robin placement of workgroups on the 8 XCDs a frame's image lines are
//         pulled into ONE L2 instead of eight (placement is a speed matter only, nothing depends on it).  The frame
//         descriptor comes through the scalar cache from a closed-form index (no dependent vector loads).
//     Measured on 64 x 120 k points: chaining the tiles inside one launch -- tickets + look-back over all
//     tiles (round 1), or per-frame sums of published counts + frame totals (round 3, 'LINKED') -- makes every tile wait
//     for the slowest of the ~1000 tiles in flight before it may store (10 us of a 24 us tile lifetime): 107 us against
//     72 for the split form, which pays 20 B written + 20 B read per kept point instead and never waits.
#include "pca_common.h"
#include <cstdlib>
#include <cstring>
#include <vector>

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

template <int BLK, int PPT, bool SPLIT, bool BILIN, bool INL>
__device__ __forceinline__ void k1_body(const K1Args &a)
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
    if (wave == 0) {
        const uint64_t e = lb_exclusive_prefix(a.state, tile, (uint64_t)total, a.epoch, a.status);
        if (lane == 0) s_excl = (long long)e;
    }
    __syncthreads();
    K1_STAMP(5);
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
}

template <int BLK, int PPT, bool SPLIT, bool BILIN>
__global__ __launch_bounds__(BLK) void k1_kitti(const K1Args a) { k1_body<BLK, PPT, SPLIT, BILIN, false>(a); }

// same, with the frame descriptors in the kernel arguments (no upload before the launch): equal-sized frames, <= 64 of them
template <int BLK, int PPT, bool BILIN>
__global__ __launch_bounds__(BLK) void k1_kitti_inl(const K1Args a, const K1InlineFrames inl) { k1_body<BLK, PPT, true, BILIN, true>(a); }


// SPLIT, second kernel: streams a tile's kept records into the SoA store (consecutive lanes write consecutive
// records: every store instruction is fully coalesced).  The tile's store offset is the sum of the counts of the
// tiles before it, which every workgroup adds up for itself (<= K1_MAX_SPLIT_TILES values from L2: no scan kernel,
// no atomics, nobody waits); the last tile of a frame also closes the frame's segment.
#define K1_MAX_SPLIT_TILES 16384
struct K1AppendArgs {
    const float4 *rec_p;
    const uint32_t *rec_c;
    const uint32_t *counts;
    const int32_t *lastf;
    int tile_points;
    int tiles;                 // tiles of the (sub-)batch
    int group;                 // consecutive tiles per workgroup
    pca_store st;
    int64_t *frame_off;
    int first_slot;
    uint32_t *status;
};

// A workgroup takes `group` consecutive tiles (round 5; one before): the sum over the counts in front of it is taken once per
// group -- tile 15 000 of a 64-frame batch reads 60 KB of counts before its first store, and 15 000 workgroups did that --
// and the staging records, written once by the front kernel and read once here, are read with the non-temporal hint, the SoA
// stores (nobody re-reads them before the raster's next pass over the whole store) likewise when `nt` says so.
template <bool NT>
__device__ __forceinline__ void k1_append_body(const K1AppendArgs &a)
{
    constexpr int BLK = K1_APPEND_BLK;
    const int tile0 = (int)blockIdx.x * a.group;
    const int tile1 = tile0 + a.group < a.tiles ? tile0 + a.group : a.tiles;
    __shared__ uint32_t s_w[BLK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t sum = 0;
    for (int t = threadIdx.x; t < tile0; t += BLK) sum += k1_ldg(a.counts + t);
    sum = wave_reduce_add(sum);
    if (lane == 0) s_w[wave] = sum;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < BLK / 64; ++w) before += s_w[w];
    bool overflow = false;
    for (int tile = tile0; tile < tile1; ++tile) {          // (uniform)
        const uint32_t c = a.counts[tile];
        const int32_t lf = a.lastf[tile];
        const int64_t base = a.frame_off[a.first_slot] + before;
        if (threadIdx.x == 0 && lf >= 0) a.frame_off[a.first_slot + lf + 1] = base + c;
        const float4 *rp = a.rec_p + (size_t)tile * a.tile_points;
        const uint32_t *rc = a.rec_c + (size_t)tile * a.tile_points;
        for (uint32_t j = threadIdx.x; j < c; j += BLK) {
            float4 p;
            uint32_t col;
            if (NT) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(rp + j));
                p = make_float4(v.x, v.y, v.z, v.w);
                col = __builtin_nontemporal_load(rc + j);
            } else {
                p = k1_ldg4(reinterpret_cast<const float *>(rp + j));
                col = k1_ldg(rc + j);
            }
            const int64_t o = base + j;
            if (o >= a.st.capacity) { overflow = true; continue; }
            if (NT) {
                __builtin_nontemporal_store((double)p.x, a.st.x + o);
                __builtin_nontemporal_store((double)p.y, a.st.y + o);
                __builtin_nontemporal_store((double)p.z, a.st.z + o);
                __builtin_nontemporal_store(p.w, a.st.intensity + o);
                __builtin_nontemporal_store(col, a.st.rgbs + o);
                __builtin_nontemporal_store((int32_t)0, a.st.inst + o);
                __builtin_nontemporal_store((uint8_t)0, a.st.dyn + o);
            } else {
                a.st.x[o] = (double)p.x;
                a.st.y[o] = (double)p.y;
                a.st.z[o] = (double)p.z;
                a.st.intensity[o] = p.w;
                a.st.rgbs[o] = col;
                a.st.inst[o] = 0;
                a.st.dyn[o] = 0;
            }
        }
        before += c;
    }
    if (overflow) pca_raise(a.status, PCA_STATUS_STORE_OVERFLOW);
}
__global__ __launch_bounds__(K1_APPEND_BLK) void k1_append(const K1AppendArgs a) { k1_append_body<false>(a); }
__global__ __launch_bounds__(K1_APPEND_BLK) void k1_append_nt(const K1AppendArgs a) { k1_append_body<true>(a); }

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

// diagnostic: copies the stamps of the last K1 launch (8 words per workgroup) to `out`; returns the workgroup count
int pca_debug_k1_stamps(pca_ctx *ctx, unsigned long long *out, int max_blocks)
{
    if (!ctx || !ctx->dbg) return 0;
    const int n = ctx->dbg_blocks < max_blocks ? ctx->dbg_blocks : max_blocks;
    if (hipMemcpy(out, ctx->dbg, sizeof(unsigned long long) * 8 * n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}

#define K1_FUSED_BLK 256        // FUSED: many small tiles (one frame = 118 workgroups), latency matters
#define K1_FUSED_PPT 4

static void k1_split_config(int *blk, int *ppt)
{
    static int eb = -1, ep = -1;
    if (eb < 0) {
        eb = 512; ep = 4;
        const char *e = getenv("PCA_K1_CFG");                 // "BLKxPPT", tuning only
        if (e) { if (sscanf(e, "%dx%d", &eb, &ep) != 2) { eb = 512; ep = 4; } }
    }
    *blk = eb; *ppt = ep;
}

static int k1_grow(pca_ctx *ctx, void **p, int64_t *cap, int64_t need, hipStream_t s)
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

static int64_t k1_ws_need(int64_t total_tiles, int tile_pts)     // bytes of SPLIT workspace for a sub-batch (see k1_prepare)
{
    const int64_t slots = total_tiles * tile_pts;
    const int64_t o_lastf = (total_tiles * 4 + 255) & ~255ll, o_c = (o_lastf + total_tiles * 4 + 255) & ~255ll,
                  o_p = (o_c + slots * 4 + 255) & ~255ll;
    return o_p + slots * 16;
}

struct K1Plan {
    K1Args fa;
    K1AppendArgs pa;
    dim3 grid_front;     // FUSED: (tiles); SPLIT: see k1_kitti
    int tiles;
    bool append_nt;
    const K1Frame *host_frames;   // the plan's descriptors on the host
    bool inline_frames;           // they fit the kernel arguments: no device copy needed
};

// Fills the launch arguments of one (sub-)batch and its frame descriptors hf[0..n_frames) (the caller uploads them to
// dev_frames before the launch).  ws_slot: which half of the SPLIT workspace.
static int k1_prepare(pca_ctx *ctx, const pca_kitti_frame *frames, int n_frames, const double P[12], int H, int W,
                      const uint64_t filter_mask[4], const pca_store *store, int64_t *frame_off, int first_slot,
                      bool fused, int blk, int ppt, int ws_slot, K1Frame *hf, const K1Frame *dev_frames, K1Plan *plan,
                      int sample_mode, hipStream_t s)
{
    const int tile_pts = blk * ppt;
    auto tiles_of = [&](int n) { return n > 0 ? (n + tile_pts - 1) / tile_pts : 1; };
    K1Args &a = plan->fa;
    // SPLIT: frame f -> queue f % Q, block b -> position b / Q of queue b % Q (a frame's image lines stay in one L2)
    int Q = 1;
    if (!fused && n_frames > 1) Q = n_frames < K1_MAXQ ? n_frames : K1_MAXQ;
    if (const char *e = getenv("PCA_K1_QUEUES")) {
        const int v = atoi(e);
        if (!fused && v >= 1 && v <= K1_MAXQ) Q = v < n_frames ? v : n_frames;
    }
    std::vector<int> tile0(n_frames);
    int total = 0;
    bool equal = true;
    for (int k = 0; k < n_frames; ++k) {
        tile0[k] = total;
        total += tiles_of(frames[k].n);
        equal = equal && tiles_of(frames[k].n) == tiles_of(frames[0].n);
    }
    int w = 0, maxq = 0;
    for (int q = 0; q < K1_MAXQ; ++q) {
        a.qframe0[q] = w;
        int qpos = 0;
        if (q < Q)
            for (int k = q; k < n_frames; k += Q) {
                K1Frame &d = hf[w++];
                d.pts = frames[k].pts; d.rgb = frames[k].rgb; d.sem = frames[k].sem; d.sem_gt = frames[k].sem_gt;
                d.n = frames[k].n; d.tile0 = tile0[k]; d.qpos0 = qpos; d.f = k;
                qpos += tiles_of(frames[k].n);
            }
        a.qtiles[q] = qpos;
        if (qpos > maxq) maxq = qpos;
    }
    a.qframe0[K1_MAXQ] = w;
    a.n_queues = Q;
    a.qbase = n_frames / Q; a.qrem = n_frames % Q;
    a.n_frames = n_frames;
    a.tpf = equal ? tiles_of(frames[0].n) : 0;
    a.sample_mode = sample_mode;
    a.frames = n_frames > 1 ? dev_frames : nullptr;
    a.one = hf[0];
    plan->host_frames = hf;
    plan->inline_frames = !fused && equal && n_frames > 1 && n_frames <= K1_INLINE_FRAMES && !getenv("PCA_K1_NO_INLINE");
    for (int i = 0; i < 12; ++i) a.P.m[i] = P[i];
    a.H = H; a.W = W;
    {   // f32 rows and the error bound of the conservative test: 2^-19 relative is 6x the worst case of three
        // rounded coefficients and four fma roundings per form (< 2^-21.6), so a point is only ever culled when its
        // exact f64 projection is outside the frustum by a wide margin
        const double g = 1.0 / 524288.0;
        const double wh = (double)(W > H ? W : H) + 1.0;
        double sx = 0, sy = 0, sd = 0;
        for (int j = 0; j < 3; ++j) { sx += fabs(P[j]); sy += fabs(P[4 + j]); sd += fabs(P[8 + j]); }
        for (int i = 0; i < 12; ++i) a.cull[i] = (float)P[i];
        a.cull[12] = (float)(g * (sx + sy + wh * sd) * 1.0000002);
        a.cull[13] = (float)(g * (fabs(P[3]) + fabs(P[7]) + wh * fabs(P[11])) * 1.0000002 + 1e-30);
        a.cull[14] = (float)((double)W - 0.5);
        a.cull[15] = (float)((double)H - 0.5);
    }
    for (int i = 0; i < 4; ++i) a.filt.w[i] = filter_mask ? filter_mask[i] : 0;
    a.st = *store;
    a.frame_off = frame_off;
    a.first_slot = first_slot;
    a.state = nullptr; a.epoch = 0;
    a.status = ctx->ticket + 1;
    a.rec_p = nullptr; a.rec_c = nullptr; a.counts = nullptr; a.lastf = nullptr;
    a.dbg = nullptr;
    plan->tiles = total;
    int maxf = 0;
    for (int q = 0; q < Q; ++q) maxf = a.qframe0[q + 1] - a.qframe0[q] > maxf ? a.qframe0[q + 1] - a.qframe0[q] : maxf;
    plan->grid_front = fused ? dim3(total) : a.tpf ? dim3(Q, a.tpf, maxf) : dim3(Q, maxq);
    if (!fused && (maxq > 65535 || maxf > 65535)) { ctx->err = "k1: batch too large"; return -1; }
    const int64_t nblocks = (int64_t)plan->grid_front.x * plan->grid_front.y * plan->grid_front.z;
    if (getenv("PCA_K1_STAMPS")) {
        if (!ctx->dbg) PCA_CHECK(ctx, hipMalloc(&ctx->dbg, sizeof(unsigned long long) * 8 * 65536));
        if (nblocks <= 65536) { a.dbg = ctx->dbg; PCA_CHECK(ctx, hipMemsetAsync(ctx->dbg, 0, sizeof(unsigned long long) * 8 * nblocks, s)); }
        ctx->dbg_blocks = (int)(nblocks < 65536 ? nblocks : 65536);
    }
    if (fused) {
        if (pca_ctx_reserve_tiles(ctx, total, s)) return -1;
        a.state = ctx->tile_state;
        a.epoch = pca_ctx_next_epoch(ctx, s);
        return 0;
    }
    // workspace: counts u32[total] | lastf i32[total] | rec_c u32[total * tile_pts] | rec_p float4[total * tile_pts]
    // (grown by the caller for the largest sub-batch BEFORE any plan is prepared: the plans hold pointers into it)
    const int64_t slots = (int64_t)total * tile_pts;
    const int64_t o_counts = 0, o_lastf = ((int64_t)total * 4 + 255) & ~255ll, o_c = (o_lastf + (int64_t)total * 4 + 255) & ~255ll,
                  o_p = (o_c + slots * 4 + 255) & ~255ll, need = o_p + slots * 16;
    if (need > ctx->k1_ws_cap[ws_slot]) { ctx->err = "k1: workspace not grown for this sub-batch"; return -1; }
    char *ws = reinterpret_cast<char *>(ctx->k1_ws[ws_slot]);
    a.counts = reinterpret_cast<uint32_t *>(ws + o_counts);
    a.lastf = reinterpret_cast<int32_t *>(ws + o_lastf);
    a.rec_c = reinterpret_cast<uint32_t *>(ws + o_c);
    a.rec_p = reinterpret_cast<float4 *>(ws + o_p);
    K1AppendArgs &pa = plan->pa;
    pa.rec_p = a.rec_p; pa.rec_c = a.rec_c; pa.counts = a.counts; pa.lastf = a.lastf; pa.tile_points = tile_pts;
    pa.tiles = total;
    {   // tiles per workgroup of k1_append and the non-temporal hint: PCA_K1_APPEND="<group>[,nt]" (A/B; results are identical).
        // Default: one tile per workgroup (4 and 8 measured the same), plain loads and stores.  Non-temporal: the kernel itself
        // takes 23 instead of 17 us; it wins only where the NEXT call's inputs would otherwise be pushed out of the Infinity Cache
        // (64 frames cycled: 69.9 -> 60-64 us), loses on cached inputs (8 frames: 49.7 -> 54.1) and changes nothing when the
        // inputs come from HBM (two alternating batches: 75.3 / 75.4): profiles/r05_experiments/k1_round5.txt
        static int group = -1, nt = 0;
        if (group < 0) {
            group = 1;
            if (const char *e = getenv("PCA_K1_APPEND")) { group = atoi(e); nt = strstr(e, "nt") != nullptr; }
            if (group < 1) group = 1;
            if (group > 64) group = 64;
        }
        pa.group = total / group >= 4 * ctx->n_cu ? group : 1;     // (small batches: as many workgroups as there are tiles)
        plan->append_nt = nt != 0;
    }
    pa.st = *store; pa.frame_off = frame_off; pa.first_slot = first_slot; pa.status = a.status;
    return 0;
}

static int k1_launch_split(pca_ctx *ctx, int blk, int ppt, const K1Plan *plan, hipStream_t s)
{
    bool launched = false;
    K1InlineFrames inl;
    if (plan->inline_frames) memcpy(inl.f, plan->host_frames, sizeof(K1Frame) * plan->fa.n_frames);
#define K1_CASE(B, Pp) if (blk == B && ppt == Pp) { \
        if (plan->inline_frames) { \
            if (plan->fa.sample_mode) hipLaunchKernelGGL((k1_kitti_inl<B, Pp, true>), plan->grid_front, dim3(B), 0, s, plan->fa, inl); \
            else hipLaunchKernelGGL((k1_kitti_inl<B, Pp, false>), plan->grid_front, dim3(B), 0, s, plan->fa, inl); \
        } else if (plan->fa.sample_mode) hipLaunchKernelGGL((k1_kitti<B, Pp, true, true>), plan->grid_front, dim3(B), 0, s, plan->fa); \
        else hipLaunchKernelGGL((k1_kitti<B, Pp, true, false>), plan->grid_front, dim3(B), 0, s, plan->fa); \
        launched = true; }
    K1_CASE(256, 4) K1_CASE(512, 4) K1_CASE(1024, 4)
#undef K1_CASE
    if (!launched) { ctx->err = "k1: unsupported PCA_K1_CFG"; return -1; }
    const dim3 agrid((plan->tiles + plan->pa.group - 1) / plan->pa.group);
    if (plan->append_nt) hipLaunchKernelGGL(k1_append_nt, agrid, dim3(K1_APPEND_BLK), 0, s, plan->pa);
    else hipLaunchKernelGGL(k1_append, agrid, dim3(K1_APPEND_BLK), 0, s, plan->pa);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

int pca_kitti_project_sample_filter(pca_ctx *ctx, const pca_kitti_frame *frames, int n_frames, const double P[12],
                                    int H, int W, const uint64_t filter_mask[4], const pca_store *store,
                                    int64_t *frame_off, int first_slot, void *stream)
{
    return pca_kitti_project_sample_filter_ex(ctx, frames, n_frames, P, H, W, filter_mask, store, frame_off, first_slot,
                                              PCA_SAMPLE_NEAREST, stream);
}

int pca_kitti_project_sample_filter_ex(pca_ctx *ctx, const pca_kitti_frame *frames, int n_frames, const double P[12],
                                       int H, int W, const uint64_t filter_mask[4], const pca_store *store,
                                       int64_t *frame_off, int first_slot, int sample_mode, void *stream)
{
    if (!ctx) return -1;
    if (sample_mode != PCA_SAMPLE_NEAREST && sample_mode != PCA_SAMPLE_BILINEAR) { ctx->err = "k1: unknown sample_mode"; return -1; }
    if (!frames || n_frames <= 0 || !store || !frame_off || !P) { ctx->err = "k1: bad arguments"; return -1; }
    if (H < 0 || W < 0 || (int64_t)H * W * 3 >= (1ll << 31)) { ctx->err = "k1: image too large"; return -1; }
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < n_frames; ++k) {
        if (frames[k].n < 0 || (frames[k].n > 0 && !frames[k].pts)) { ctx->err = "k1: bad frame"; return -1; }
        if (!frames[k].sem_gt && frames[k].n > 0 && (!frames[k].rgb || !frames[k].sem || H * W == 0)) { ctx->err = "k1: frame needs rgb+sem or sem_gt"; return -1; }
    }
    // The colour gather is one 4-byte load per point: a one-pixel image (3 bytes) is handed to the kernel as a 4-byte copy
    std::vector<pca_kitti_frame> padded;
    if (H * W * 3 < 4 && H * W > 0) {
        bool any = false;
        for (int k = 0; k < n_frames; ++k) any = any || (frames[k].rgb && !frames[k].sem_gt);
        if (any) {
            if (k1_grow(ctx, &ctx->k1_tiny, &ctx->k1_tiny_cap, (int64_t)n_frames * 4, s)) return -1;
            PCA_CHECK(ctx, hipMemsetAsync(ctx->k1_tiny, 0, (size_t)n_frames * 4, s));
            padded.assign(frames, frames + n_frames);
            for (int k = 0; k < n_frames; ++k)
                if (frames[k].rgb && !frames[k].sem_gt) {
                    uint8_t *dst = reinterpret_cast<uint8_t *>(ctx->k1_tiny) + 4 * k;
                    PCA_CHECK(ctx, hipMemcpyAsync(dst, frames[k].rgb, (size_t)H * W * 3, hipMemcpyDeviceToDevice, s));
                    padded[k].rgb = dst;
                }
            frames = padded.data();
        }
    }
    // FUSED when every workgroup of the launch is resident at once (one tile per CU at most), else SPLIT
    auto count_tiles = [&](int k0, int k1, int tile_pts) {
        int64_t t = 0;
        for (int k = k0; k < k1; ++k) t += frames[k].n > 0 ? (frames[k].n + tile_pts - 1) / tile_pts : 1;
        return t;
    };
    bool fused = count_tiles(0, n_frames, K1_FUSED_BLK * K1_FUSED_PPT) <= ctx->n_cu;
    if (const char *e = getenv("PCA_K1_MODE")) { if (!strcmp(e, "split")) fused = false; }
    // frame descriptors of the whole call: built in pinned memory, one asynchronous upload
    const int slot = ctx->k1_pin_next;
    ctx->k1_pin_next ^= 1;
    if (ctx->k1_pin_busy[slot]) { PCA_CHECK(ctx, hipEventSynchronize(ctx->k1_pin_ev[slot])); ctx->k1_pin_busy[slot] = false; }
    if (n_frames > ctx->k1_pin_cap[slot]) {
        if (ctx->k1_pin[slot]) PCA_CHECK(ctx, hipHostFree(ctx->k1_pin[slot]));
        ctx->k1_pin[slot] = nullptr; ctx->k1_pin_cap[slot] = 0;
        PCA_CHECK(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->k1_pin[slot]), sizeof(K1Frame) * (size_t)n_frames * 2, hipHostMallocMapped));
        ctx->k1_pin_cap[slot] = n_frames * 2;
    }
    if (!ctx->k1_pin_ev[slot]) PCA_CHECK(ctx, hipEventCreateWithFlags(&ctx->k1_pin_ev[slot], hipEventDisableTiming));
    K1Frame *hf = ctx->k1_pin[slot];
    if (k1_grow(ctx, &ctx->k1_frames_dev, &ctx->k1_frames_cap, (int64_t)sizeof(K1Frame) * n_frames, s)) return -1;
    const K1Frame *df = reinterpret_cast<const K1Frame *>(ctx->k1_frames_dev);
    bool prof_open = false;
    bool need_upload = true;
    auto upload = [&]() -> int {
        if (ctx->profiling == 1) { pca_prof_begin(ctx, PCA_K_KITTI, s); prof_open = true; }   // one event pair around the unit's GPU work
        if (n_frames > 1 && need_upload) {
            // (fetched by a kernel from the mapped host block: a copy command of a few KB costs 13-17 us, see pca_fetch_block)
            if (pca_fetch_block(ctx, hf, 0, ctx->k1_frames_dev, (int64_t)sizeof(K1Frame) * n_frames, s)) return -1;
            PCA_CHECK(ctx, hipEventRecord(ctx->k1_pin_ev[slot], s));
            ctx->k1_pin_busy[slot] = true;
        }
        return 0;
    };
    int rc = 0;
    if (fused) {
        K1Plan plan;
        rc = k1_prepare(ctx, frames, n_frames, P, H, W, filter_mask, store, frame_off, first_slot, true, K1_FUSED_BLK, K1_FUSED_PPT, 0, hf, df, &plan, sample_mode, s);
        if (rc == 0) rc = upload();
        if (rc == 0) {
            if (sample_mode) hipLaunchKernelGGL((k1_kitti<K1_FUSED_BLK, K1_FUSED_PPT, false, true>), plan.grid_front, dim3(K1_FUSED_BLK), 0, s, plan.fa);
            else hipLaunchKernelGGL((k1_kitti<K1_FUSED_BLK, K1_FUSED_PPT, false, false>), plan.grid_front, dim3(K1_FUSED_BLK), 0, s, plan.fa);
            if (hipGetLastError() != hipSuccess) { ctx->err = "k1: launch failed"; rc = -1; }
        }
    } else {
        int blk, ppt;
        k1_split_config(&blk, &ppt);
        // sub-batches of at most K1_MAX_SPLIT_TILES tiles (k1_append adds up the counts before its tile)
        std::vector<K1Plan> plans;
        std::vector<int> cuts(1, 0);
        int64_t max_tiles = 0;
        for (int k0 = 0; k0 < n_frames && rc == 0;) {
            int k1 = k0 + 1;
            int64_t t = count_tiles(k0, k1, blk * ppt);
            if (t > (1 << 28)) { ctx->err = "k1: frame too large"; rc = -1; break; }
            while (k1 < n_frames) {
                const int64_t tn = count_tiles(k1, k1 + 1, blk * ppt);
                if (t + tn > K1_MAX_SPLIT_TILES) break;
                t += tn; ++k1;
            }
            max_tiles = t > max_tiles ? t : max_tiles;
            cuts.push_back(k1);
            k0 = k1;
        }
        // the sub-batches run one after the other on the stream and share ONE workspace: it is grown once, for the
        // largest of them, before any plan takes pointers into it
        if (rc == 0 && k1_grow(ctx, &ctx->k1_ws[0], &ctx->k1_ws_cap[0], k1_ws_need(max_tiles, blk * ppt), s)) rc = -1;
        for (size_t i = 0; i + 1 < cuts.size() && rc == 0; ++i) {
            const int k0 = cuts[i], k1 = cuts[i + 1];
            plans.emplace_back();
            rc = k1_prepare(ctx, frames + k0, k1 - k0, P, H, W, filter_mask, store, frame_off, first_slot + k0, false, blk, ppt,
                            0, hf + k0, df + k0, &plans.back(), sample_mode, s);
        }
        need_upload = false;
        for (const K1Plan &pl : plans) need_upload = need_upload || !pl.inline_frames;
        if (rc == 0) rc = upload();
        for (size_t i = 0; i < plans.size() && rc == 0; ++i) rc = k1_launch_split(ctx, blk, ppt, &plans[i], s);
    }
    if (prof_open) pca_prof_end(ctx, s);
    return rc;
}

}  // extern "C"
