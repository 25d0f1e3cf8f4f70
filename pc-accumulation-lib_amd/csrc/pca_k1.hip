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

#include "pca_k1_body.h"

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

}  // extern "C"

// ---- deferred K1 (pca_common.h: K1Pending) ----
int pca_k1_flush_pending(pca_ctx *ctx)
{
    if (!ctx || !ctx->k1_pend.valid) return 0;
    pca_ctx::K1Pending pd = ctx->k1_pend;
    ctx->k1_pend.valid = false;                             // (first: the call below comes back through here)
    const int rc = pca_kitti_project_sample_filter_ex(ctx, &pd.fr, 1, pd.P, pd.H, pd.W, pd.filt, &pd.store, pd.frame_off, pd.slot,
                                                      pd.sample_mode, pd.stream);
    if (pd.stage_idx >= 0) {                                // the staging block is free again once K1 has read it
        pca_ctx::Stage &st = ctx->stage[pd.stage_idx];
        if (hipEventRecord(st.done, pd.stream) == hipSuccess) st.busy = true;
        else (void)hipStreamSynchronize(pd.stream);
    }
    return rc;
}
// The deferred frame as level 1 of the raster takes it: the argument block of a FUSED launch with 1024 x 4 tiles (look-back
// state reserved, epoch drawn).  *n_tiles = its workgroups.  The caller launches, then calls pca_k1_pending_launched.
#define K1_TAIL_BLK 1024
#define K1_TAIL_PPT 4
int pca_k1_prepare_pending(pca_ctx *ctx, K1Args *out, int *n_tiles, hipStream_t s)
{
    pca_ctx::K1Pending &pd = ctx->k1_pend;
    if (!pd.valid) return -1;
    K1Plan plan;
    K1Frame hf;
    if (k1_prepare(ctx, &pd.fr, 1, pd.P, pd.H, pd.W, pd.filt, &pd.store, pd.frame_off, pd.slot, true, K1_TAIL_BLK, K1_TAIL_PPT, 0, &hf,
                   nullptr, &plan, pd.sample_mode, s))
        return -1;
    *out = plan.fa;
    *n_tiles = plan.tiles;
    return 0;
}
void pca_k1_pending_launched(pca_ctx *ctx, hipStream_t s)
{
    pca_ctx::K1Pending &pd = ctx->k1_pend;
    if (pd.stage_idx >= 0) {
        pca_ctx::Stage &st = ctx->stage[pd.stage_idx];
        if (hipEventRecord(st.done, s) == hipSuccess) st.busy = true;
        else (void)hipStreamSynchronize(s);
    }
    pd.valid = false;
}

extern "C" {

// K1 of pca_kitti_integrate deferred into the next raster of this context (see pca.h); on = 0 also runs what is pending.
int pca_k1_defer(pca_ctx *ctx, int on)
{
    if (!ctx) return -1;
    ctx->k1_defer = on != 0;
    return on ? 0 : pca_k1_flush_pending(ctx);
}
int pca_k1_flush(pca_ctx *ctx) { return ctx ? pca_k1_flush_pending(ctx) : -1; }

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
    if (pca_k1_flush_pending(ctx)) return -1;               // (an earlier frame whose K1 was deferred: it comes first)
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
