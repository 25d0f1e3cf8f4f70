// pca_session.hip -- one library call per driver call (host-side fusion; no kernels of its own).
//
// The unchanged KITTI-360 driver makes two calls per frame: integrate(observations) and, when its trigger fires,
// generate_bev(present_idx, ...) (run_kitti360_bev_gen.py:186-273 of the reference).  Behind each sat a dozen Python
// functions and three to five ctypes calls (stage + upload, K1, pose track; trajectories, raster, device -> host copy):
// ~0.1 ms of interpreter per step next to 0.1 ms of GPU work.  The two entry points below do each call's library work in
// one go; Python keeps the argument marshalling, the slot bookkeeping of the store and the dict shell of the result.
#include <vector>

#include "pca_common.h"

int pca_stage_copy(const void *const *src, void *const *dst, const int64_t *bytes, int n);      // pca_host.hip

static inline int64_t up256(int64_t v) { return (v + 255) & ~255ll; }

extern "C" {

// integrate() of one KITTI-360 observation (kitti360_sem_pc_accum.py:41-88 of the reference, per-point work = K1):
// host arrays are staged through a ring of pinned + device blocks owned by the context (ONE asynchronous H2D copy for the
// whole observation), K1 appends the frame into `slot`, and the pose track takes its step.
int pca_kitti_integrate(pca_ctx *ctx, const pca_kitti_obs *obs, const double P[12], int H, int W,
                        const uint64_t filter_mask[4], const pca_store *store, int64_t *frame_off, int slot, int sample_mode,
                        pca_host_track *track, const double *T_new_prev, double horizon, int64_t *evicted,
                        double *path_length, void *stream)
{
    if (!ctx) return -1;
    if (!obs || obs->n < 0 || !P || !filter_mask || !store || !frame_off) { ctx->err = "kitti_integrate: bad arguments"; return -1; }
    if (track && (!T_new_prev || !evicted || !path_length)) { ctx->err = "kitti_integrate: the pose track needs T_new_prev and its outputs"; return -1; }
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    if (pca_k1_flush_pending(ctx)) return -1;               // the frame before this one, if its K1 is still owed
    const void *in[4] = {obs->pts, obs->rgb, obs->sem, obs->sem_gt};
    const int64_t size[4] = {16ll * obs->n, 3ll * H * W, 1ll * H * W, 1ll * obs->n};
    const void *dev[4] = {in[0], in[1], in[2], in[3]};
    // ---- host arrays: one pinned block, one device block, one copy ----
    int64_t off[4] = {0, 0, 0, 0}, total = 0;
    for (int k = 0; k < 4; ++k) {
        if (!(in[k] && ((obs->host_mask >> k) & 1u))) continue;
        if (size[k] > 0) { off[k] = total; total += up256(size[k]); }
        else dev[k] = nullptr;                             // an empty host array: nothing to stage, nothing the device may touch
    }
    pca_ctx::Stage *slot_st = nullptr;
    if (total > 0) {
        pca_ctx::Stage &st = ctx->stage[ctx->stage_next++ % PCA_STAGE_DEPTH];
        if (st.busy) { PCA_CHECK(ctx, hipEventSynchronize(st.done)); st.busy = false; }     // PCA_STAGE_DEPTH calls ago: long done
        if (st.cap < total) {
            if (st.pin) PCA_CHECK(ctx, hipHostFree(st.pin));
            if (st.dev) PCA_CHECK(ctx, hipFree(st.dev));
            st.pin = st.dev = nullptr; st.cap = 0;
            const int64_t cap = total + total / 4;
            PCA_CHECK(ctx, hipHostMalloc(&st.pin, (size_t)cap));
            PCA_CHECK(ctx, hipMalloc(&st.dev, (size_t)cap));
            st.cap = cap;
        }
        if (!st.done) PCA_CHECK(ctx, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
        const void *src[4]; void *dst[4]; int64_t bytes[4]; int m = 0;
        for (int k = 0; k < 4; ++k)
            if (in[k] && ((obs->host_mask >> k) & 1u) && size[k] > 0) {
                src[m] = in[k]; dst[m] = (char *)st.pin + off[k]; bytes[m] = size[k]; ++m;
                dev[k] = (char *)st.dev + off[k];
            }
        pca_stage_copy(src, dst, bytes, m);
        // The upload does NOT go onto the caller's stream: there it would queue behind the raster of the previous step and
        // K1 behind it -- copy and kernels one after the other, 4 MB at PCIe speed being as long as the kernels of a step.
        // It leaves, as ONE copy, on a copy stream of the context while the compute stream is still busy with the previous
        // step; K1 waits for its event.  (The device block needs no guard against earlier readers: its last reader, the K1
        // of PCA_STAGE_DEPTH calls ago, was waited for above.)  Measured on the unchanged driver's step, one box, us per step:
        // copy on the caller's stream 202-212; one copy stream 182; two copy streams with half the block each 234-240 (the
        // second event hop and two short DMAs cost more than a second SDMA engine returns); 1 MB pieces staged and sent in
        // turn on one copy stream 213-219 (five pool jobs and five copy calls instead of one each).
        static int64_t side_min = -1;
        if (side_min < 0) { const char *e = getenv("PCA_H2D_SIDE_MIN"); side_min = e ? atoll(e) : (256ll << 10); }
        if (total >= side_min) {
            if (!ctx->h2d_stream[0]) PCA_CHECK(ctx, hipStreamCreateWithFlags(&ctx->h2d_stream[0], hipStreamNonBlocking));
            if (!ctx->h2d_done[0]) PCA_CHECK(ctx, hipEventCreateWithFlags(&ctx->h2d_done[0], hipEventDisableTiming));
            PCA_CHECK(ctx, hipMemcpyAsync(st.dev, st.pin, (size_t)total, hipMemcpyHostToDevice, ctx->h2d_stream[0]));
            PCA_CHECK(ctx, hipEventRecord(ctx->h2d_done[0], ctx->h2d_stream[0]));
            PCA_CHECK(ctx, hipStreamWaitEvent(s, ctx->h2d_done[0], 0));
        } else {
            PCA_CHECK(ctx, hipMemcpyAsync(st.dev, st.pin, (size_t)total, hipMemcpyHostToDevice, s));
        }
        slot_st = &st;
    }
    pca_kitti_frame fr;
    fr.pts = (const float *)dev[0]; fr.rgb = (const uint8_t *)dev[1]; fr.sem = (const uint8_t *)dev[2];
    fr.sem_gt = (const uint8_t *)dev[3]; fr.n = obs->n; fr.reserved = 0;
    // Deferred (pca_k1_defer): K1 is left for the next raster of this context, which runs it inside level 1's launch -- or for
    // whoever touches a store first (pca_k1_flush_pending).  Only the plain case: points there, nearest sampling, a real image.
    const bool plain = obs->n > 0 && sample_mode == PCA_SAMPLE_NEAREST && (fr.sem_gt || ((int64_t)H * W * 3 >= 4 && fr.rgb && fr.sem)) &&
                       obs->n <= (1 << 20);
    if (ctx->k1_defer && plain) {
        pca_ctx::K1Pending &pd = ctx->k1_pend;
        pd.fr = fr;
        for (int i = 0; i < 12; ++i) pd.P[i] = P[i];
        pd.H = H; pd.W = W;
        for (int i = 0; i < 4; ++i) pd.filt[i] = filter_mask[i];
        pd.store = *store; pd.frame_off = frame_off; pd.slot = slot; pd.sample_mode = sample_mode; pd.stream = s;
        pd.stage_idx = slot_st ? (int)(slot_st - ctx->stage) : -1;
        pd.valid = true;
    } else {
        const int rc = pca_kitti_project_sample_filter_ex(ctx, &fr, 1, P, H, W, filter_mask, store, frame_off, slot, sample_mode, stream);
        if (slot_st) {                                     // the blocks are free again once K1 has read them
            const hipError_t e = hipEventRecord(slot_st->done, s);
            if (e == hipSuccess) slot_st->busy = true;
            else {
                // no event to wait for: the block must not be staged into again while K1 may still read it -- wait here, once,
                // and say what happened
                (void)hipStreamSynchronize(s);
                if (rc == 0) { ctx->err = std::string("kitti_integrate: hipEventRecord(stage done): ") + hipGetErrorString(e); return -1; }
            }
        }
        if (rc != 0) return rc;
    }
    if (track) *evicted = pca_host_track_step(track, T_new_prev, horizon, path_length);
    return 0;
}

// generate_bev(present_idx, 1, gen_future=True) of the KITTI-360 flow without augmentation (kitti360_sem_pc_accum.py:166-243,
// bev_generator.py:63-125 of the reference): the three ego polylines in grid coordinates (poses - origin -> rotate,
// translate, clip, floor: pca_host_ego_to_grid), the raster with the owed re-transforms riding along
// (pca_bev_generate_chain), and the planes' way to the host on the context's side stream (pca_host_d2h_async).
// prm carries origin, R, dx, dy, view of the sample (the caller evaluates the heading with numpy, as the reference does).
// traj_rows [2 (F - 1)][3], traj_start [F] (F = poses of the track); host_planes may be NULL (planes stay in HBM).
// Returns the copy's ticket (>= 0; 0 without a copy) or -1.
int pca_kitti_generate_bev(pca_ctx *ctx, const pca_store *store, const int64_t *frame_off, int slot_begin, int slot_split,
                           int slot_end, int64_t max_points, const pca_bev_params *prm, const double *pending_Ts,
                           const int *pending_slot_ends, int n_pending, int write_back, void *workspace,
                           int64_t workspace_bytes, uint16_t *planes_f16, void *host_planes, const pca_host_track *track,
                           double *traj_rows, int32_t *traj_start, int32_t *n_rows, void *stream)
{
    if (!ctx) return -1;
    if (!prm || !planes_f16 || !track || !traj_rows || !traj_start || !n_rows) { ctx->err = "kitti_generate_bev: bad arguments"; return -1; }
    const int64_t F = pca_host_track_len(track);
    if (F != slot_end - slot_begin) { ctx->err = "kitti_generate_bev: the pose track and the window disagree about the number of frames"; return -1; }
    // rel = poses - origin, element by element (numpy: poses - origin), then the reference's trajectory transform
    static thread_local std::vector<double> rel;
    rel.resize((size_t)(3 * (F > 0 ? F : 1)));
    const double *Hp = pca_host_track_poses(track);
    for (int64_t f = 0; f < F; ++f)
        for (int k = 0; k < 3; ++k) rel[(size_t)(3 * f + k)] = Hp[4 * f + k] - prm->origin[k];
    *n_rows = pca_host_ego_to_grid(rel.data(), (int)F, prm->R, prm->dx, prm->dy, prm->view, prm->px, traj_rows, traj_start);
    if (pca_bev_generate_chain(ctx, store, nullptr, frame_off, slot_begin, slot_split, slot_end, max_points, prm, pending_Ts,
                               pending_slot_ends, n_pending, write_back, workspace, workspace_bytes, nullptr, planes_f16,
                               nullptr, stream) != 0)
        return -1;
    if (!host_planes) return 0;
    return pca_host_d2h_async(ctx, planes_f16, host_planes, 21ll * prm->px * prm->px * 2, stream);
}

int pca_kitti_integrate_v(pca_ctx *ctx, pca_kitti_integrate_args *a)
{
    if (!ctx) return -1;
    if (!a) { ctx->err = "kitti_integrate_v: no argument block"; return -1; }
    return pca_kitti_integrate(ctx, a->obs, a->P, a->H, a->W, a->filter_mask, a->store, a->frame_off, a->slot, a->sample_mode,
                               a->track, a->T_new_prev, a->horizon, &a->evicted, &a->path_length, a->stream);
}

int pca_kitti_generate_bev_v(pca_ctx *ctx, pca_kitti_generate_bev_args *a)
{
    if (!ctx) return -1;
    if (!a) { ctx->err = "kitti_generate_bev_v: no argument block"; return -1; }
    a->hinted = 0;
    if (a->hint_F > 0 && !(a->n_pending > 0 && a->write_back)) {
        const int rc = pca_bev_view_hint(ctx, a->hint_slot0, a->hint_F, a->hint_then, a->hint_box, a->hint_cone, a->hint_now, a->prm);
        if (rc < 0) { ctx->err = "kitti_generate_bev_v: bad view hint"; return -1; }
        a->hinted = rc;
    }
    return pca_kitti_generate_bev(ctx, a->store, a->frame_off, a->slot_begin, a->slot_split, a->slot_end, a->max_points, a->prm,
                                  a->pending_Ts, a->pending_slot_ends, a->n_pending, a->write_back, a->workspace,
                                  a->workspace_bytes, a->planes_f16, a->host_planes, a->track, a->traj_rows, a->traj_start,
                                  &a->n_rows, a->stream);
}

}  // extern "C"
