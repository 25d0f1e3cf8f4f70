// pca_common.h -- shared host/device helpers of the gfx950 library (not part of the C ABI).
//
// Numerics contract: numpy's small matmuls are f64 FMA chains in k order -> explicit
// fma(); everything else is un-fused -> this library is compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>

#include <hip/hip_fp16.h>
#include "../../include/pca.h"
#include "pca_wave.h"

#define PCA_WAVE 64

struct K1Frame {           // device-side frame descriptor of K1 (pca_k1.hip); the array is sorted by (queue, frame)
    const float *pts;
    const uint8_t *rgb, *sem, *sem_gt;
    int32_t n;
    int32_t tile0;         // first tile of the frame in OUTPUT (frame-major) order = index into the look-back state
    int32_t qpos0;         // first position of the frame in its queue
    int32_t f;             // frame index in the launch: slot = first_slot + f
};

struct pca_ctx {
    int device = 0;
    int n_cu = 256;                   // compute units of the device (MI355X: 256)
    std::string err;
    // decoupled look-back workspace (stable compaction / scans)
    uint64_t *tile_state = nullptr;   // dev [tile_cap]
    int64_t tile_cap = 0;
    uint32_t *ticket = nullptr;       // dev: one PcaStatusBlock (below) -- [0] ticket counter, [1] status bits, [2..3] device address of the mirror
    uint32_t *status_mirror = nullptr;     // pinned, device-visible [PCA_STATUS_BITS]: word b != 0 <=> bit b was raised
    uint32_t *status_mirror_dev = nullptr;
    uint32_t epoch = 0;               // 22-bit launch tag of tile_state entries
    void *k1_frames_dev = nullptr;    // dev: K1's frame descriptors of a batched launch
    int64_t k1_frames_cap = 0;        // bytes
    void *k1n_ws = nullptr;           // dev: staging of the batched K1n (kept records + counts)
    int64_t k1n_ws_cap = 0;
    void *k1n_desc_dev = nullptr;     // dev: its frame descriptors + tile -> frame table
    int64_t k1n_desc_cap = 0;
    void *k1n_pin = nullptr;          // pinned staging of those
    int64_t k1n_pin_cap = 0;
    hipEvent_t k1n_ev = nullptr;
    bool k1n_busy = false;
    void *bevm_pin = nullptr;         // argument blocks of pca_bev_generate_many (pinned staging)
    int64_t bevm_cap = 0;
    hipEvent_t bevm_ev = nullptr;
    bool bevm_busy = false;
    // pca_host_d2h_async: a side stream for results on their way to the host, the event that lets it start and a ring of
    // completion events (a ticket = a place in the ring)
    hipStream_t d2h_stream = nullptr;
    hipEvent_t d2h_go = nullptr;
    hipEvent_t d2h_done[64] = {};
    uint32_t d2h_next = 0;
    // pca_kitti_integrate: the copy stream an observation's upload leaves on, and the event K1 waits for
    hipStream_t h2d_stream[1] = {nullptr};
    hipEvent_t h2d_done[1] = {nullptr};
    // pca_kitti_integrate: host observations on their way to the device -- a ring of pinned + device blocks, each free
    // again when the K1 launch that read it has finished
    struct Stage { void *pin = nullptr; void *dev = nullptr; int64_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };
#define PCA_STAGE_DEPTH 4
    Stage stage[PCA_STAGE_DEPTH];
    uint32_t stage_next = 0;
    void *k1_tiny = nullptr;          // dev: 4-byte copies of images smaller than the 4-byte colour gather
    int64_t k1_tiny_cap = 0;
    void *k1_ws[2] = {nullptr, nullptr};   // dev: counts / kept records of K1's split form, one per sub-batch in flight
    int64_t k1_ws_cap[2] = {0, 0};         // bytes
    K1Frame *k1_pin[2] = {nullptr, nullptr};   // pinned staging of the descriptors, alternating between calls
    int k1_pin_cap[2] = {0, 0};
    hipEvent_t k1_pin_ev[2] = {nullptr, nullptr};
    bool k1_pin_busy[2] = {false, false};
    int k1_pin_next = 0;
    unsigned long long *dbg = nullptr; // diagnostic stamps of the last K1 launch (PCA_K1_STAMPS)
    int dbg_blocks = 0;
    int heavy_cooldown = 0;           // calls for which bev_tile_cells_heavy is still launched after the last heavy tile
    // device ICP (pca_icp.hip): the per-cell count tables of its grids, owned by the context because they are all zero again
    // after every registration (no 134 MB of memsets per call), and the registration's state as the host sees it --
    // mapped host memory the solve kernel stores into: the host polls it instead of copying and waiting
    uint32_t *icp_cnt = nullptr;      // dev [levels][cells]
    bool icp_cnt_dirty = true;        // not known to be all zero (first use, or a call that failed half way)
    double *icp_host = nullptr;       // pinned + mapped [32]: T, fitness, rmse, iterations, flag; [31] = (call << 8 | pass) tag
    double *icp_host_dev = nullptr;
    uint32_t icp_call = 0;
    // K1 deferred into the next raster (round 5): pca_kitti_integrate leaves the frame's K1 here when the mode is on; the next
    // pca_bev_generate_chain on this context whose window ends with that frame runs it INSIDE level 1's launch (K1 alone is a
    // 9 us latency-bound launch of 118 small workgroups in front of a 55 us kernel; as 30 of level 1's 512 workgroups it costs
    // ~1.5 us); every other entry point that touches a store runs it first, on its own (pca_k1_flush_pending).
    struct K1Pending {
        bool valid = false;
        pca_kitti_frame fr;
        double P[12];
        int H = 0, W = 0;
        uint64_t filt[4] = {0, 0, 0, 0};
        pca_store store;
        int64_t *frame_off = nullptr;
        int slot = 0, sample_mode = 0;
        hipStream_t stream = nullptr;
        int stage_idx = -1;           // the staging block K1 reads (its `done` event is recorded behind the K1 that does), or -1
    };
    bool k1_defer = false;
    K1Pending k1_pend;
    uint32_t *status_host = nullptr;  // pinned
    uint32_t *heavy_hint = nullptr;   // pinned, device-visible: heavy-tile count of the latest rasteriser call
    uint32_t *heavy_hint_dev = nullptr;
    int bin_first = 0, bin_end = 0;   // pca_bev_bin_range: the slots the next single raster bins (bin_valid); forgotten by that call
    bool bin_valid = false;
    // optional per-kernel event timing
    struct Ev { hipEvent_t a, b; int kid; };
    int profiling = 0;                // 0 off, 1 every kernel launch, 2 whole units only (pca_profile_enable)
    std::vector<Ev> evs;              // recorded pairs not yet folded into the totals
    std::vector<Ev> free_evs;         // recycled events
    double prof_ms[PCA_K_COUNT] = {0};
    int64_t prof_n[PCA_K_COUNT] = {0};
};

// Launch wrapper: plain launch, or bracketed by events while profiling is on.
void pca_prof_begin(pca_ctx *ctx, int kid, hipStream_t s);
void pca_prof_end(pca_ctx *ctx, hipStream_t s);
#define PCA_LAUNCH_SHM(ctx, kid, kernel, grid, block, shm, stream, ...)        \
    do {                                                                        \
        if ((ctx)->profiling == 1) pca_prof_begin((ctx), (kid), (stream));      \
        hipLaunchKernelGGL(kernel, grid, block, shm, stream, __VA_ARGS__);      \
        if ((ctx)->profiling == 1) pca_prof_end((ctx), (stream));               \
    } while (0)
#define PCA_LAUNCH(ctx, kid, kernel, grid, block, stream, ...) \
    PCA_LAUNCH_SHM(ctx, kid, kernel, grid, block, 0, stream, __VA_ARGS__)

#define PCA_CHECK(ctx, expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                    \
            return -1;                                                                         \
        }                                                                                      \
    } while (0)

// internal (pca_k1.hip): runs a deferred K1 now, on its own (no-op without one).  Every entry point that reads or writes a
// store, frame_off or the status word calls it first.
int pca_k1_flush_pending(pca_ctx *ctx);
// internal (pca_api.hip)
// Small argument blocks (frame descriptors, raster parameters: a few KB) from MAPPED host memory into device memory by a
// KERNEL that reads the host block over PCIe: a copy command of that size costs 13-17 us on this stack (measured as the gap
// between the HIP-event time of a batched K1n call and the sum of its kernels), a launch 2-3.  `mapped_host`: from
// hipHostMalloc(..., hipHostMallocMapped) -- the START of the allocation, the block begins `offset` bytes into it (the device
// address is asked for the allocation, not for an interior pointer); offset and bytes: multiples of 16.
int pca_fetch_block(pca_ctx *ctx, const void *mapped_host, int64_t offset, void *dev, int64_t bytes, hipStream_t s);
int pca_ctx_reserve_tiles(pca_ctx *ctx, int64_t tiles, hipStream_t s);
uint32_t pca_ctx_next_epoch(pca_ctx *ctx, hipStream_t s);

struct Mat34 { double m[12]; };
struct Mat44 { double m[16]; };
struct ClassMask { uint64_t w[4]; };

// (a select chain over the four words: `m.w[c >> 6]` with a per-lane class is a dynamically indexed read of the KERNEL
// ARGUMENTS, which compiles to a vector load from the argument segment -- a memory round trip in the middle of the chain
// class gather -> filter -> colour gather)
__device__ __forceinline__ bool in_mask(const ClassMask &m, unsigned c)
{
    const unsigned w = c >> 6;
    uint64_t v = m.w[0];
    v = w == 1 ? m.w[1] : v;
    v = w == 2 ? m.w[2] : v;
    v = w == 3 ? m.w[3] : v;
    return (v >> (c & 63)) & 1ull;
}

// row . [x y z 1], k order, fma chain  (== OpenBLAS dgemm for K = 4)
__device__ __forceinline__ double row4(const double *r, double x, double y, double z)
{
    double a = r[0] * x;
    a = fma(r[1], y, a);
    a = fma(r[2], z, a);
    a = fma(r[3], 1.0, a);
    return a;
}

// Bilinear sampling weights of pts_feat_from_img(..., 'bilinear') (datasets/nuscenes_utils.py:197-210 of the
// reference): neighbours floor / ceil, weights as un-fused f64 expressions, the fourth weight as 1 - (sum of the others),
// value = ((w_ff a(v0,u0) + w_cc a(v1,u1)) + w_cf a(v1,u0)) + w_fc a(v0,u1).
// STRICT = the reference's arithmetic to the letter (an integer coordinate makes floor == ceil, area 0, NaN);
// otherwise the opt-in sample mode of K1 / K1n: the upper neighbour is floor + 1 (its weight is then exactly 0).
struct Bilin { double u0, u1, v0, v1, w_ff, w_cc, w_cf, w_fc; };
template <bool STRICT>
__device__ __forceinline__ Bilin bilin_weights(double u, double v)
{
    Bilin b;
    b.u0 = floor(u); b.v0 = floor(v);
    b.u1 = STRICT ? ceil(u) : b.u0 + 1.0;
    b.v1 = STRICT ? ceil(v) : b.v0 + 1.0;
    const double area = (b.u1 - b.u0) * (b.v1 - b.v0);
    b.w_ff = (b.u1 - u) * (b.v1 - v) / area;
    b.w_cc = (u - b.u0) * (v - b.v0) / area;
    b.w_fc = (u - b.u0) * (b.v1 - v) / area;
    b.w_cf = 1.0 - (b.w_ff + b.w_cc + b.w_fc);
    return b;
}
__device__ __forceinline__ double bilin_value(const Bilin &b, double a_ff, double a_cc, double a_cf, double a_fc)
{
    return b.w_ff * a_ff + b.w_cc * a_cc + b.w_cf * a_cf + b.w_fc * a_fc;
}
// r,g,b of four corner pixels (packed r | g<<8 | b<<16) -> packed bilinear colour, every channel rounded half-to-even
__device__ __forceinline__ uint32_t bilin_rgb(const Bilin &b, uint32_t c_ff, uint32_t c_cc, uint32_t c_cf, uint32_t c_fc)
{
    uint32_t out = 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const int sh = 8 * ch;
        double v = rint(bilin_value(b, (double)((c_ff >> sh) & 255u), (double)((c_cc >> sh) & 255u),
                                    (double)((c_cf >> sh) & 255u), (double)((c_fc >> sh) & 255u)));
        v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
        out |= (uint32_t)v << sh;
    }
    return out;
}

// f64 -> f16 bits, round-to-nearest-even straight from the double (numpy astype(np.float16)).
// Shipped form: f64 -> f32 with round-to-ODD (truncate, then set the last bit if anything was lost: the sticky bit
// survives), then the hardware's f32 -> f16 round-to-nearest-even.  Rounding to odd into a format with more than
// 2*11+2 significant bits makes the second rounding see exactly what a direct rounding would; checked bit for bit
// against the integer routine below on 16.7 M values incl. halfway cases and subnormals (tools/experiments/f16_cvt.hip).
__device__ __forceinline__ uint16_t f64_to_f16_bits_reference(double d);
__device__ __forceinline__ uint16_t f64_to_f16_bits(double d)
{
    float t = __double2float_rz(d);
    if ((double)t != d) t = __uint_as_float(__float_as_uint(t) | 1u);
    return __half_as_ushort(__float2half_rn(t));
}
__device__ __forceinline__ uint16_t f64_to_f16_bits_reference(double d)
{
    uint64_t b = (uint64_t)__double_as_longlong(d);
    uint16_t sign = (uint16_t)((b >> 48) & 0x8000u);
    uint64_t absb = b & 0x7fffffffffffffffull;
    if (absb >= 0x7ff0000000000000ull)
        return (uint16_t)(sign | 0x7c00u | ((absb > 0x7ff0000000000000ull) ? 0x200u : 0u));
    if (absb == 0) return sign;
    int e = (int)(absb >> 52) - 1023;
    if (e >= 16) return (uint16_t)(sign | 0x7c00u);
    uint64_t man = (absb & 0xfffffffffffffull) | 0x10000000000000ull;
    int shift, he;
    if (e >= -14) { shift = 42; he = e + 15; }
    else { shift = 42 + (-14 - e); he = 0; }
    if (shift > 63) return sign;
    uint64_t keep = man >> shift;
    uint64_t rem = man & ((1ull << shift) - 1);
    uint64_t half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (keep & 1))) keep++;
    uint32_t h = (he > 0) ? (uint32_t)((he - 1) << 10) + (uint32_t)keep : (uint32_t)keep;
    if (h >= 0x7c00u) h = 0x7c00u;
    return (uint16_t)(sign | h);
}

// f32 -> u32, order-preserving (a < b  <=>  ordered(a) < ordered(b)), never 0 for a number: the frame boxes of
// pca_store.frame_box are atomic maxima of ordered(v) (upper bound) and ~ordered(v) (lower bound), 0 = nothing yet
__host__ __device__ __forceinline__ uint32_t pca_f32_ordered(float v)
{
    union { float f; uint32_t u; } c; c.f = v;
    return c.u ^ ((c.u >> 31) ? 0xffffffffu : 0x80000000u);
}
__host__ __device__ __forceinline__ float pca_f32_from_ordered(uint32_t o)
{
    union { float f; uint32_t u; } c; c.u = o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu);
    return c.f;
}

// Raises a PCA_STATUS_* bit: the device word (read and cleared by pca_status, which synchronises) and its host-visible
// mirror -- a plain store into mapped host memory, whose address sits next to the status word -- that pca_status_peek
// reads without touching the stream.  Only ever executed on the error paths.
#define PCA_STATUS_BITS 8
// The 16-byte device block behind pca_ctx::ticket.  Kernels are handed `&block->status` (ctx->ticket + 1) as their status
// word; pca_raise finds the mirror's device address in the SAME block.  Whoever hands a kernel a status word must hand it this
// one; pca_status clears `status` alone (4 bytes) and nothing may write `mirror` after pca_ctx_create.
struct PcaStatusBlock {
    uint32_t ticket;       // ticket counter of the look-back users
    uint32_t status;       // PCA_STATUS_* bits (read and cleared by pca_status)
    uint32_t *mirror;      // device address of the host-visible mirror words [PCA_STATUS_BITS] (mapped host memory), or null
};
static_assert(sizeof(PcaStatusBlock) == 16 && offsetof(PcaStatusBlock, status) == 4 && offsetof(PcaStatusBlock, mirror) == 8,
              "pca_ctx::ticket is four u32: [0] ticket, [1] status, [2..3] the mirror's address");
__device__ __forceinline__ void pca_raise(uint32_t *status, uint32_t bit)
{
    atomicOr(status, bit);
    uint32_t *mirror = reinterpret_cast<const PcaStatusBlock *>(status - 1)->mirror;
    if (mirror) __hip_atomic_store(mirror + (__ffs((int)bit) - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------
// Decoupled look-back (single-pass chained scan).  Tiles are handed out by an atomic ticket so a
// tile only ever waits on tiles whose workgroups are already running (no dispatch-order assumption).
// State word (one 8-byte granule, relaxed agent-scope atomic store / load, so flag and value can
// never be seen torn and no separate payload needs a release/acquire):
//     [63:62] flag (0 invalid, 1 aggregate, 2 inclusive prefix)  [61:40] epoch  [39:0] value
// `epoch` tags the launch, so the array never has to be cleared between launches.
// ---------------------------------------------------------------------------------------------
#define LB_FLAG_AGG 1ull
#define LB_FLAG_PFX 2ull
#define LB_VAL_MASK ((1ull << 40) - 1)

__device__ __forceinline__ uint64_t lb_pack(uint64_t flag, uint32_t epoch, uint64_t v)
{
    return (flag << 62) | ((uint64_t)(epoch & 0x3fffffu) << 40) | (v & LB_VAL_MASK);
}

__device__ __forceinline__ void lb_store(uint64_t *p, uint64_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t lb_load(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Publishes the tile's aggregate (tile 0: its inclusive prefix straight away).  One lane.
__device__ __forceinline__ void lb_publish_aggregate(uint64_t *state, int tile, uint64_t aggregate, uint32_t epoch)
{
    lb_store(&state[tile], lb_pack(tile == 0 ? LB_FLAG_PFX : LB_FLAG_AGG, epoch, aggregate));
}

// Called by ALL lanes of ONE wave after the aggregate was published: walks back over the predecessors,
// publishes the inclusive prefix, returns the exclusive prefix.  The spin is bounded: a predecessor that does not
// publish within LB_MAX_POLLS fetches (seconds; a protocol or dispatch failure, never seen) raises
// PCA_STATUS_LOOKBACK_TIMEOUT in *status and the walk returns what it has, so the grid always drains.
#define LB_MAX_POLLS (1u << 21)
__device__ __forceinline__ uint64_t lb_walk(uint64_t *state, int tile, uint64_t aggregate, uint32_t epoch,
                                            uint32_t *status = nullptr)
{
    const int lane = threadIdx.x & 63;
    if (tile == 0) return 0;
    uint64_t excl = 0;
    int hi = tile - 1;                      // newest predecessor of the current window
    // LB_W windows of 64 predecessors are fetched per round trip (independent loads), then consumed in order:
    // when all workgroups run in lock step the walk is as long as the number of tiles in flight.
    constexpr int LB_W = 4;
    bool done = false;
    uint32_t polls = 0;
    while (!done) {
        if (++polls > LB_MAX_POLLS) {
            if (status && lane == 0) pca_raise(status, PCA_STATUS_LOOKBACK_TIMEOUT);
            break;
        }
        uint64_t w[LB_W];
#pragma unroll
        for (int j = 0; j < LB_W; ++j) {
            const int idx = hi - 64 * j - lane;
            w[j] = idx >= 0 ? lb_load(&state[idx]) : 0;
        }
#pragma unroll
        for (int j = 0; j < LB_W; ++j) {
            if (done) continue;
            const int idx = hi - lane;      // lane 0 looks at the closest predecessor of this window
            bool valid = true, pfx = false;
            if (idx >= 0) {
                const uint64_t flag = w[j] >> 62;
                const bool mine = (uint32_t)((w[j] >> 40) & 0x3fffffu) == (epoch & 0x3fffffu);
                valid = mine && flag != 0;
                pfx = valid && flag == LB_FLAG_PFX;
            }
            const uint64_t pfx_mask = __ballot(pfx);
            const uint64_t inv_mask = __ballot(!valid);
            // lanes up to the first prefix (or all 64) must be valid before the window can be consumed
            const int first_pfx = pfx_mask ? (int)__ffsll((unsigned long long)pfx_mask) - 1 : 64;
            const uint64_t need = (first_pfx >= 63) ? ~0ull : ((2ull << first_pfx) - 1ull);
            if (inv_mask & need) {          // somebody has not published yet: fetch again from this window on
                __builtin_amdgcn_s_sleep(1);
                break;
            }
            uint64_t v = (idx >= 0 && lane <= first_pfx) ? (w[j] & LB_VAL_MASK) : 0;
            // values are below 2^40: two 20-bit halves, each summed over the wave on the VALU
            excl += ((uint64_t)wave_reduce_add((uint32_t)(v >> 20)) << 20) + wave_reduce_add((uint32_t)v & 0xfffffu);
            if (first_pfx < 64 || hi - 64 < 0) done = true;
            else hi -= 64;
        }
    }
    if (lane == 0) lb_store(&state[tile], lb_pack(LB_FLAG_PFX, epoch, excl + aggregate));
    return excl;
}

// publish + walk in one go (non-pipelined users)
__device__ __forceinline__ uint64_t lb_exclusive_prefix(uint64_t *state, int tile, uint64_t aggregate, uint32_t epoch,
                                                        uint32_t *status = nullptr)
{
    if ((threadIdx.x & 63) == 0) lb_publish_aggregate(state, tile, aggregate, epoch);
    return lb_walk(state, tile, aggregate, epoch, status);
}
