/*
 * pca.h -- C ABI of the MI355X (gfx950) semantic point-cloud accumulator + BEV rasteriser.
 *
 * The reference (robin-karlsson0/pc-accumulation-lib) is pure Python and has no FFI layer: its
 * "plugin interface" for this path is the Python classes in sem_pc_accum.py / bev_generator/.  This
 * header is the boundary a maintainer binds underneath those classes (ctypes stub: INTEGRATION.md);
 * every entry point names the reference code it replaces (file:line relative to the reference root).
 *
 * Conventions
 *   - all pointers marked "dev" are device (HBM) pointers, everything else is host memory read
 *     synchronously during the call;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls only enqueue work and
 *     never synchronise unless stated;
 *   - return value 0 = ok, negative = error (text via pca_last_error); no exceptions cross the boundary;
 *   - one pca_ctx per host thread / GPU; calls on one ctx must target one stream at a time.
 *
 * Point store (HBM layout): structure-of-arrays, 37 B per stored point
 *     x,y,z      f64   reference sem_pc columns 0..2
 *     intensity  f32   RAW lidar intensity; column 3 = raw (KITTI) or raw/255. (NuScenes), the division
 *                      is applied by consumers (pca_bev_params.intensity_div255) exactly as the reference's
 *                      f64 division would
 *     rgbs       u32   r | g<<8 | b<<16 | sem<<24 (columns 4..7, small non-negative integers)
 *     inst       i32   column 8
 *     dyn        u8    column 9
 * Frames are contiguous segments; `frame_off` (dev, int64[max_frames+1]) holds the segment boundaries
 * and lives on the device so that integrate() never has to read a count back.
 */
#ifndef PCA_H
#define PCA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCA_VERSION 2

typedef struct pca_ctx pca_ctx;

typedef struct {
    double *x, *y, *z;   /* dev */
    float *intensity;    /* dev */
    uint32_t *rgbs;      /* dev */
    int32_t *inst;       /* dev */
    uint8_t *dyn;        /* dev */
    int64_t capacity;    /* points */
    /* Optional (version 2; NULL = absent): dev, six words per slot, ZEROED by the caller whenever a slot is (re)used.  K1 (one-frame
     * launches) leaves there the box of the points the frame kept, in the frame's own coordinates: words 2k / 2k+1 = lower /
     * upper bound of coordinate k as an order-preserving code of the f32 (pca_f32_box_decode turns a row into six floats;
     * upper word 0 = the frame kept nothing, or K1 has not run yet).  A caller that reads rows back -- whenever it likes: a row
     * it has not seen only means "unknown" -- can tell which frames cannot reach a raster's view (pca_host_view_hull,
     * pca_bev_bin_range). */
    uint32_t *frame_box;
} pca_store;

/* status bits accumulated on the device, read back with pca_status() */
#define PCA_STATUS_STORE_OVERFLOW 1u /* a kernel wanted to write past store.capacity (points dropped)     */
#define PCA_STATUS_UV_OUT_OF_IMAGE 2u /* NuScenes: pixel coords outside (1, wh-1): reference AssertionError */
#define PCA_STATUS_NEGATIVE_INTENSITY 4u /* BEV: a negative f32 intensity (pass intensity64 for such data)   */
#define PCA_STATUS_LOOKBACK_TIMEOUT 8u /* a compaction workgroup gave up waiting for a predecessor (output invalid) */

int pca_version(void);
int pca_ctx_create(int device, pca_ctx **out);
void pca_ctx_destroy(pca_ctx *ctx);
const char *pca_last_error(pca_ctx *ctx);
/* Synchronises `stream`, returns the OR of PCA_STATUS_* bits raised since the last call and clears them. */
int pca_status(pca_ctx *ctx, void *stream, uint32_t *status_out);
/* The same bits as the host can see them now, WITHOUT touching the stream (every raise also stores into a mapped host
 * word): a kernel's raise shows here at the latest once that kernel has finished.  Does not clear; pca_status does.
 * The drop-in classes peek on every integrate() / generate_bev() and where they wait for a result anyway, so that a
 * dropped point or an invalid compaction surfaces as the exception the reference would have raised synchronously
 * (IndexError of bev_generator/sem_bev.py:543-551, AssertionError of datasets/nuscenes_utils.py:191-195). */
int pca_status_peek(pca_ctx *ctx, uint32_t *status_out);
/* the mirror itself: host address of 8 words, word b != 0 <=> bit b raised (for bindings that want a zero-cost read) */
const uint32_t *pca_status_mirror(pca_ctx *ctx);

/* ------------------------------------------------------------------------------------------------
 * K1  KITTI-360 fused  project -> frustum mask -> nearest sample (rgb + semseg) -> class filter ->
 *     stable compaction -> append.   Replaces, per frame,
 *       sem_pc_accum.py:347-402 (velo2frame, velo2img), :323-345 (gen_semantic_pc, called twice from
 *       kitti360_sem_pc_accum.py:132-135), :317-321 (filter_semseg_pc), kitti360_sem_pc_accum.py:136-156.
 *     One launch handles a BATCH of frames (frame k appends into slot first_slot+k); kept points keep
 *     their input order (the reference's boolean-mask compaction is stable).
 *     If frame.sem_gt != NULL the use_gt_sem branch (kitti360_sem_pc_accum.py:139-144) is taken:
 *     no projection, rgb = 0, class = sem_gt[p].
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    const float *pts;      /* dev [n,4] f32 x,y,z,intensity                                    */
    const uint8_t *rgb;    /* dev [H,W,3] u8, or NULL with sem_gt                              */
    const uint8_t *sem;    /* dev [H,W]   u8 class map, or NULL with sem_gt                    */
    const uint8_t *sem_gt; /* dev [n] u8 per-point class, or NULL                              */
    int32_t n;             /* points in this frame                                             */
    int32_t reserved;      /* ignored                                                          */
} pca_kitti_frame;

/* frames: HOST array of n_frames descriptors (copied into the launch).  P: 3x4 row-major f64
 * (P_velo_frame).  filter_mask: 256-bit class set.  frame_off[first_slot] must hold the append position;
 * frame_off[first_slot+k+1] is written for every frame k. */
int pca_kitti_project_sample_filter(pca_ctx *ctx, const pca_kitti_frame *frames, int n_frames, const double P[12],
                                    int H, int W, const uint64_t filter_mask[4], const pca_store *store,
                                    int64_t *frame_off /*dev*/, int first_slot, void *stream);

/* Sample modes of K1 / K1n.  NEAREST is what the reference's callers use (sem_pc_accum.py:338-341 rounds the pixel,
 * nuscenes_oracle_sem_pc_accum.py:466-469 passes 'nearest').  BILINEAR is opt-in (the reference has the branch,
 * datasets/nuscenes_utils.py:197-210, but no caller): r, g, b = round-half-even of the bilinear mix of the four
 * neighbours, weights exactly as that branch computes them (pinned through pca_sample_bilinear), neighbours clamped to
 * the image, an integer coordinate takes floor + 1 as its upper neighbour (weight 0) instead of the reference's 0 / 0;
 * the class is always the nearest pixel's. */
#define PCA_SAMPLE_NEAREST 0
#define PCA_SAMPLE_BILINEAR 1
int pca_kitti_project_sample_filter_ex(pca_ctx *ctx, const pca_kitti_frame *frames, int n_frames, const double P[12],
                                       int H, int W, const uint64_t filter_mask[4], const pca_store *store,
                                       int64_t *frame_off /*dev*/, int first_slot, int sample_mode, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K1n NuScenes (oracle pose)  nearest sample from 6 camera images -> invalid / class filter ->
 *     stable compaction -> ego->world transform -> append.
 *     Replaces nuscenes_oracle_sem_pc_accum.py:457-501 and datasets/nuscenes_utils.py:181-214 ('nearest').
 *     pc: dev [n,7] f64 rows x,y,z,intensity,u,v,inst;  cam_idx: dev [n] int64 (-1: on no image);
 *     imgs: dev [ncam,H,W,3] u8; sems: dev [ncam,H,W] u8;  T: 4x4 row-major f64 (T_ego_world).
 * ------------------------------------------------------------------------------------------------ */
int pca_nusc_sample_filter_transform(pca_ctx *ctx, const double *pc, const int64_t *cam_idx, int32_t n,
                                     const uint8_t *imgs, const uint8_t *sems, int ncam, int H, int W,
                                     const double T[16], const uint64_t filter_mask[4], const pca_store *store,
                                     int64_t *frame_off /*dev*/, int slot, void *stream);
int pca_nusc_sample_filter_transform_ex(pca_ctx *ctx, const double *pc, const int64_t *cam_idx, int32_t n,
                                        const uint8_t *imgs, const uint8_t *sems, int ncam, int H, int W,
                                        const double T[16], const uint64_t filter_mask[4], const pca_store *store,
                                        int64_t *frame_off /*dev*/, int slot, int sample_mode, void *stream);

/* The same for a BATCH of frames (a whole scene: run_nuscenes_bev_gen.py:236-237 integrates every sample of a scene before
 * the first BEV): one front launch over the tiles of all frames + one append launch, the stored rows being exactly those of
 * n_frames single calls in order (slots first_slot .. first_slot + n_frames - 1; the K3 marks of the tracker are applied
 * afterwards, in tracker order).  Every frame has its own point rows, camera indices, image stack and T_ego_world; the
 * image stacks share ncam, H, W.  At most 16384 tiles of 512 points per call. */
typedef struct {
    const double *pc;          /* dev [n,7] f64 rows x,y,z,intensity,u,v,inst */
    const int64_t *cam_idx;    /* dev [n] */
    const uint8_t *imgs;       /* dev [ncam,H,W,3] */
    const uint8_t *sems;       /* dev [ncam,H,W] */
    int32_t n;
    int32_t reserved;
    const double *T;           /* host 4x4 row-major f64 (T_ego_world) */
} pca_nusc_frame;
int pca_nusc_sample_filter_transform_batch(pca_ctx *ctx, const pca_nusc_frame *frames, int n_frames, int ncam, int H, int W,
                                           const uint64_t filter_mask[4], const pca_store *store, int64_t *frame_off /*dev*/,
                                           int first_slot, int sample_mode, void *stream);

/* pts_feat_from_img(pts_uv, img, method='bilinear') of the reference for a 2-D feature map (its bilinear branch only
 * works for those): datasets/nuscenes_utils.py:181-210, the arithmetic to the letter.  map: dev [H,W] f64; uv: dev [n,2]
 * f64 (u along x); out: dev [n] f64.  Raises PCA_STATUS_UV_OUT_OF_IMAGE where the reference's assert fails. */
int pca_sample_bilinear(pca_ctx *ctx, const double *map /*dev*/, int H, int W, const double *uv /*dev*/, int32_t n,
                        double *out /*dev*/, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K0n NuScenes lidar -> ego -> global -> N cameras, pinhole projection, last camera wins.
 *     Replaces obs_dataloaders/nuscenes_obs_dataloader.py:162-202 and
 *     datasets/nuscenes_utils.py:46-60, :112-136 (view_points = nuscenes-devkit, restated).
 *     pc_lidar dev [n,3] f64.  T_cam_from_glob [ncam,16], K [ncam,9], wh [ncam,2] host arrays.
 *     Outputs dev: pc_in_ego [n,3], uv [n,2] f64, cam_idx [n] int64.
 * ------------------------------------------------------------------------------------------------ */
int pca_nusc_project_cams(pca_ctx *ctx, const double *pc_lidar, int32_t n, const double T_ego_from_lidar[16],
                          const double T_glob_from_ego[16], const double *T_cam_from_glob, const double *K,
                          const double *wh, int ncam, double *pc_in_ego, double *uv, int64_t *cam_idx,
                          void *stream);

/* ------------------------------------------------------------------------------------------------
 * K2  in-place rigid re-transform of every stored point of frames [slot_begin, slot_end).
 *     Replaces sem_pc_accum.py:167-183 (update_sem_pcs).  Ts: n_T 4x4 row-major matrices applied one
 *     after the other (n_T = 1 is the reference's per-step call; n_T > 1 applies a backlog of steps in
 *     one pass over HBM with identical roundings).
 * ------------------------------------------------------------------------------------------------ */
int pca_retransform(pca_ctx *ctx, const pca_store *store, const int64_t *frame_off /*dev*/, int slot_begin,
                    int slot_end, const double *Ts, int n_T, void *stream);

/* Batched integrate: k frames appended by ONE K1 call into slots first_slot .. first_slot+k-1, with the per-frame
 * transforms Ts[0..k) (Ts[i] = T_new_prev of frame i).  Frame i still owes Ts[i+1] .. Ts[k-1] (the reference applies
 * them one integrate() at a time, sem_pc_accum.py:167-183); this applies them, frame by frame, in ceil((k-1)/16)
 * launches with the roundings of the step-by-step form.  (Frames stored BEFORE the batch owe all k: pca_retransform.) */
int pca_retransform_batch_tail(pca_ctx *ctx, const pca_store *store, const int64_t *frame_off /*dev*/, int first_slot,
                               int n_frames, const double *Ts, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K3  dyn[p] = 1 where inst[p] == inst_idx, for n_pairs (slot, inst_idx) pairs.
 *     Replaces nuscenes_oracle_sem_pc_accum.py:223-229, :243-250.
 * ------------------------------------------------------------------------------------------------ */
int pca_mark_dynamic(pca_ctx *ctx, const pca_store *store, const int64_t *frame_off /*dev*/, const int32_t *slots,
                     const int32_t *inst_idx, int n_pairs, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Voxel de-duplication of the accumulation buffer (opt-in; no reference counterpart -- the reference only
 *     evicts whole frames by horizon, sem_pc_accum.py:185-209).  Over the stored points of frames
 *     [slot_begin, slot_end): of all points in one voxel floor(xyz / voxel_size) the first in store order (the
 *     oldest observation) stays.  Segments are compacted in place, order-preserving; frame_off[slot_begin+1 ..
 *     slot_end] is rewritten.  slot_end must be the last used slot (later segments are not moved).
 * ------------------------------------------------------------------------------------------------ */
int64_t pca_voxel_dedup_workspace_bytes(int64_t max_points, int n_slots);
int pca_voxel_dedup(pca_ctx *ctx, const pca_store *store, int64_t *frame_off /*dev*/, int slot_begin, int slot_end,
                    double voxel_size, int64_t max_points, void *workspace /*dev*/, int64_t workspace_bytes,
                    void *stream);

/* ------------------------------------------------------------------------------------------------
 * Point-to-plane ICP between two lidar sweeps.  Replaces the Open3D calls
 *       sem_pc_accum.py:310-315 (estimate_normals, default 30 nearest neighbours) and
 *       kitti360_sem_pc_accum.py:115-127 (registration_icp(source = previous sweep, target = new sweep, threshold,
 *       init, PointToPlane), defaults: <= 30 iterations, relative fitness / rmse 1e-6).
 *     Open3D is a third-party, unpinned dependency of the reference: parity is UNPINNED; the tests check known
 *     motions and a k-d-tree CPU model of this algorithm.  Neighbour searches are exact inside a cap (normals 3 m,
 *     correspondences min(max_corr_dist, 4 m)); sweeps must lie within +-128 m (x, y), -16..16 m (z) of the sensor.
 *     src_pts / tgt_pts: dev [n,4] f32 rows x,y,z,(ignored).  init / T_out: host 4x4 row-major; T_out maps source
 *     coordinates into the target frame (T_new_prev).  Synchronises `stream` before returning.
 * ------------------------------------------------------------------------------------------------ */
int64_t pca_icp_workspace_bytes(int32_t max_points);   /* max_points >= max(n_src, n_tgt) */
int pca_icp_register(pca_ctx *ctx, const float *src_pts /*dev*/, int32_t n_src, const float *tgt_pts /*dev*/,
                     int32_t n_tgt, double max_corr_dist, const double init[16], int max_iter, double rel_fitness,
                     double rel_rmse, void *workspace /*dev*/, int64_t workspace_bytes, double T_out[16],
                     double *fitness, double *rmse, int *iterations, void *stream);

/* ------------------------------------------------------------------------------------------------
 * K4-K7  BEV rasteriser: window [slot_begin, slot_end), 'present' = [slot_begin, slot_split),
 *     'future' = [slot_split, slot_end), 'full' = both.  Replaces
 *       kitti360_sem_pc_accum.py:189-213 / nuscenes_oracle_sem_pc_accum.py:535-581 (window assembly, origin),
 *       bev_generator/bev_generator.py:127-160, :207-255, :737-747 (rotate, translate, crop, height, floor),
 *       bev_generator/sem_bev.py:57-118 (static partition, maps), :535-554 (min z), :619-669 (rgb median),
 *       bev_generator/bev_generator.py:373-480 (counts, dirichlet, intensity), sem_bev.py:593-617, :204-257 (fp16).
 *     Output (dev): planes f64 [21,px,px] and/or planes_f16 [21,px,px] (either may be NULL), order
 *       set-major {present,future,full} x {road,intensity,r,g,b,dynamic,elevation}.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    double origin[3];         /* bev_frame_coords                                                   */
    double R[9];              /* rotation_matrix_3d(rot_ang) evaluated on the host (np.cos/np.sin);
                                 must be a rotation about z (checked)                                */
    double dx, dy;            /* augmentation translation                                           */
    double view;              /* zoom_scalar * view_size                                            */
    double height_filter;     /* keep z < height_filter; NaN disables                               */
    double int_scaler, int_sep_scaler, int_mid_threshold;
    double rgb_fill;          /* value of an empty cell before /255                                 */
    int32_t px;
    int32_t road_class;
    uint64_t dynobj_mask[4];  /* classes counted in the 'dynamic' plane                             */
    int32_t intensity_div255;
    int32_t pad;
} pca_bev_params;

/* bytes of scratch the rasteriser needs for a window of at most max_points and a px x px grid */
int64_t pca_bev_workspace_bytes(int64_t max_points, int px);

/* intensity64 (dev, may be NULL): f64 intensities indexed like the store, overriding store.intensity
 * (for callers whose column 3 is not f32-representable).  max_points bounds the window size.
 * pending_T (host, 4x4 row-major, may be NULL): a re-transform that is still owed to the frames
 * [slot_begin, pending_slot_end) (sem_pc_accum.py:167-183).  It is applied -- and written back to the store,
 * exactly as pca_retransform would -- by the first kernel of the rasteriser, which reads those coordinates
 * anyway; this saves one full read of the store per step when every integrate() is followed by a BEV. */
int pca_bev_generate(pca_ctx *ctx, const pca_store *store, const double *intensity64,
                     const int64_t *frame_off /*dev*/, int slot_begin, int slot_split, int slot_end,
                     int64_t max_points, const pca_bev_params *prm, const double *pending_T, int pending_slot_end,
                     void *workspace /*dev*/, int64_t workspace_bytes, double *planes /*dev*/,
                     uint16_t *planes_f16 /*dev*/, void *stream);

/* Same rasteriser with EXTRA REDUCERS (opt-in; the reference's live code has none of them -- mean elevation /
 * mean intensity exist only in its unused legacy utils/bev_generation.py:249-294).  extra_planes: dev f64
 * [3 sets][PCA_BEV_EXTRA_PLANES][px][px], per set {max z, mean z} over the static points of a cell (0 where
 * unobserved, like the elevation plane) and {mean raw intensity} over its road points (0 where none).
 * Sums are exact fixed-point integers: the result does not depend on the order of the points. */
#define PCA_BEV_EXTRA_PLANES 3
int pca_bev_generate_ex(pca_ctx *ctx, const pca_store *store, const double *intensity64,
                        const int64_t *frame_off /*dev*/, int slot_begin, int slot_split, int slot_end,
                        int64_t max_points, const pca_bev_params *prm, const double *pending_T, int pending_slot_end,
                        void *workspace /*dev*/, int64_t workspace_bytes, double *planes /*dev*/,
                        uint16_t *planes_f16 /*dev*/, double *extra_planes /*dev, may be NULL*/, void *stream);

/* The same with a CHAIN of owed re-transforms (sem_pc_accum.py:167-183 issues one per integrate): transform k (4x4 row-major
 * in pending_Ts[16 k ..], oldest first) is owed by slots [slot_begin, pending_slot_ends[k]) (ascending).  The rasteriser
 * applies them one after the other to the coordinates it reads -- the roundings of one pca_retransform pass each -- and,
 * with write_back != 0, stores the result (the transforms are then no longer owed); with write_back == 0 they stay owed
 * and the store is untouched: the caller passes them again, together with the next one, and saves the 24 B per stored
 * point of the write-back on that call.  n_pending <= PCA_BEV_MAX_CHAIN. */
#define PCA_BEV_MAX_CHAIN 4
int pca_bev_generate_chain(pca_ctx *ctx, const pca_store *store, const double *intensity64,
                           const int64_t *frame_off /*dev*/, int slot_begin, int slot_split, int slot_end,
                           int64_t max_points, const pca_bev_params *prm, const double *pending_Ts,
                           const int *pending_slot_ends, int n_pending, int write_back, void *workspace /*dev*/,
                           int64_t workspace_bytes, double *planes /*dev*/, uint16_t *planes_f16 /*dev*/,
                           double *extra_planes /*dev, may be NULL*/, void *stream);

/* Several rasters of ONE store in one launch of each kernel: the sweep over present_idx of a finished scene
 * (run_nuscenes_bev_gen.py:245-271 walks present_idx over a store that no longer changes) or the bev_num augmented samples
 * of one window (kitti360_sem_pc_accum.py:236-241, a multiprocessing.Pool in the reference).  Each job is what one
 * pca_bev_generate call takes (slots, parameters, output planes); no owed transforms; workspace:
 * n_jobs * round_up(pca_bev_workspace_bytes(max_points, px), 256) + 256 bytes.  Results equal n_jobs single calls. */
typedef struct {
    int32_t slot_begin, slot_split, slot_end, reserved;
    pca_bev_params prm;
    double *planes;            /* dev [21,px,px] f64 or NULL */
    uint16_t *planes_f16;      /* dev [21,px,px] f16 or NULL */
} pca_bev_job;
int pca_bev_generate_many(pca_ctx *ctx, const pca_store *store, const double *intensity64 /*dev or NULL*/,
                          const int64_t *frame_off /*dev*/, const pca_bev_job *jobs, int n_jobs, int64_t max_points,
                          void *workspace /*dev*/, int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Input normalisation of the semseg CNN on the device.  Replaces utils/onnx_utils.py:26-29, :35-36
 *     (torchvision ToTensor + Normalize on the host): out[c][y][x] = (rgb[y][x][c] / 255 - mean[c]) / std[c] in IEEE f32.
 *     rgb: dev [H,W,3] u8; out: dev [3,H,W] f32.  The CNN itself is an external ONNX file (utils/onnx_utils.py binds this
 *     buffer and the class-map output to an onnxruntime session, so neither visits the host).
 * ------------------------------------------------------------------------------------------------ */
int pca_image_to_nchw_f32(pca_ctx *ctx, const uint8_t *rgb /*dev*/, int H, int W, const float mean[3], const float std[3],
                          float *out /*dev*/, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Polynomial warp of finished fp16 planes (--bev_do_warp augmentation).  Replaces
 *       bev_generator/bev_generator.py:482-525 (warp_dense_probmaps: a Python loop over px^2 cells).
 *     out[n][jw][iw] = in[n][clamp(rint(b_1 jw + b_2 jw^2))][clamp(rint(a_1 iw + a_2 iw^2))], the index expressions
 *     evaluated exactly as numpy evaluates them; (a_1, a_2) / (b_1, b_2) come from cal_warp_params (:563-569).
 *     planes_f16, out_f16: dev [n_planes, px, px], distinct buffers.
 * ------------------------------------------------------------------------------------------------ */
int pca_bev_warp(pca_ctx *ctx, const uint16_t *planes_f16 /*dev*/, uint16_t *out_f16 /*dev*/, int n_planes, int px,
                 double a_1, double a_2, double b_1, double b_2, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Host helper (no device work): the ego trajectory of one BEV sample -- rotate, translate, clip to the view box
 * with bisected border crossings, grid coordinates.  Replaces bev_generator/bev_generator.py:207-371, :737-747 for
 * the ego polyline (plain IEEE arithmetic, bit-identical to the numpy form in pca_amd/host_logic.py).
 * full: host [F,3] f64; R: 3x3 row-major; rows: host [2(F-1),3] out; start: host [F] out (first row of every
 * edge; start[F-1] = number of rows).  Returns the number of rows.
 * ------------------------------------------------------------------------------------------------ */
int pca_host_ego_to_grid(const double *full, int F, const double R[9], double dx, double dy, double view, int px,
                         double *rows, int32_t *start);

/* ------------------------------------------------------------------------------------------------
 * One library call per driver call (KITTI-360 flow).  The reference's driver calls integrate(observations) per frame and
 * generate_bev(present_idx, bev_num, gen_future) when its trigger fires (run_kitti360_bev_gen.py:186-273); each of the two
 * below does everything the library contributes to one such call.
 *
 * pca_kitti_integrate  replaces kitti360_sem_pc_accum.py:41-88: K1 on the observation (see pca_kitti_project_sample_filter)
 *     + the pose bookkeeping of the frame (pca_host_track_step; track may be NULL: the caller keeps its own).  Inputs named
 *     in host_mask are HOST arrays (what the reference's loader yields: kitti360_obs_dataloader.py:87-106); they are copied
 *     into a pinned block of the context on the staging pool and leave in ONE asynchronous H2D copy; the others are device
 *     pointers.  frame_off[slot] must hold the append position; *evicted / *path_length as pca_host_track_step.
 * pca_kitti_generate_bev  replaces kitti360_sem_pc_accum.py:166-243 + bev_generator.py:63-125 for one un-augmented sample:
 *     ego polylines (pca_host_ego_to_grid on poses - prm->origin; rows / start / *n_rows as there, F = poses of the track
 *     = slot_end - slot_begin), raster (pca_bev_generate_chain, same arguments), and -- host_planes != NULL: page-locked,
 *     21 px px 2 bytes -- the planes' copy to the host (pca_host_d2h_async).  Returns that copy's ticket (0 without), -1 on error.
 * ------------------------------------------------------------------------------------------------ */
typedef struct pca_host_track pca_host_track;
typedef struct {
    const void *pts;       /* [n,4] f32 x,y,z,intensity                                        */
    const void *rgb;       /* [H,W,3] u8, or NULL with sem_gt                                  */
    const void *sem;       /* [H,W] u8 class map, or NULL with sem_gt                          */
    const void *sem_gt;    /* [n] u8 per-point class, or NULL                                  */
    int32_t n;
    uint32_t host_mask;    /* bit 0 pts, 1 rgb, 2 sem, 3 sem_gt: that pointer is a HOST array  */
} pca_kitti_obs;
int pca_kitti_integrate(pca_ctx *ctx, const pca_kitti_obs *obs, const double P[12], int H, int W,
                        const uint64_t filter_mask[4], const pca_store *store, int64_t *frame_off /*dev*/, int slot,
                        int sample_mode, pca_host_track *track, const double *T_new_prev, double horizon, int64_t *evicted,
                        double *path_length, void *stream);
/* Frames that cannot reach the view, skipped by a raster.  The window of a sample spans the accumulation horizon, the view a
 * part of it: the frames far ahead of the sample's pose and the oldest ones put no point into any plane, yet the raster's
 * first kernel loads and transforms them.  The CALLER knows where every frame is:
 * pca_host_view_hull (host only, no GPU): frame f of F (oldest first) was created when the product of all transforms applied to
 *     the store was then[f] (3x4 row-major; the caller multiplies every T onto its running product `now`: the reference's
 *     update_sem_pcs, sem_pc_accum.py:168-183), so its points are, now, at  now * inverse(then[f]) * (where K1 put them);
 *     K1 only keeps points in front of the camera whose pixel lies in the image, i.e. inside the cone `cone` (apex, then the four
 *     rays through the image corners, in the sensor's frame: pca_host_camera_cone; NULL: no camera test, e.g. per-point
 *     labels) and inside box[f] (six floats lo/hi per coordinate, what K1 left in pca_store.frame_box; lo > hi = unknown).  Returns in
 *     *first / *last the first and the last frame for which neither the cone nor the box proves that it misses the view square of prm (1 cm of
 *     margin; -1 / -1: none); every frame outside [first, last] is certain to fail the raster's crop with all its points.
 * pca_bev_bin_range: the NEXT pca_bev_generate_chain / pca_kitti_generate_bev of the context bins the slots [slot_first,
 *     slot_last_plus1) only (clamped to its window; ignored, and forgotten, if that call writes owed transforms back or is a
 *     different kind of raster).  Results do not change when every frame outside has no point in the view. */
void pca_host_camera_cone(const double P[12], int H, int W, double cone[15]);
int pca_host_view_hull(int F, const double *then /*[F][12]*/, const float *box /*[F][6]*/, const double *cone /*[15] or NULL*/,
                       const double now[12], const pca_bev_params *prm, int *first, int *last);
int pca_bev_bin_range(pca_ctx *ctx, int slot_first, int slot_last_plus1);
/* the two in one call (frame 0 of the F frames sits in slot0): 1 = a narrower range was set, 0 = every frame may reach the view */
int pca_bev_view_hint(pca_ctx *ctx, int slot0, int F, const double *then, const float *box, const double *cone, const double now[12],
                      const pca_bev_params *prm);
void pca_f32_box_decode(const uint32_t *rows /*[n][6]*/, int n, float *box /*[n][6]: lo x, hi x, lo y, ...; lo > hi = empty*/);

/* K1 of pca_kitti_integrate left for the raster that follows it.  The reference's driver integrates a frame and, when its
 * trigger fires, rasterises the window that ends with that frame (run_kitti360_bev_gen.py:186-273): with on = 1, a
 * pca_kitti_integrate whose inputs are the plain kind (nearest sampling, rgb + class map) does the pose bookkeeping and the
 * staging as before but only NOTES its K1; the next pca_bev_generate_chain / pca_kitti_generate_bev of the context whose
 * window ends with that slot (same store, frame_off and stream, 5 planes x int32 layout) runs it as the first workgroups
 * of its own first kernel -- the frame's kept points go into the store AND, straight from registers, into the raster's
 * tile lists, so they are not read back.  Every other entry point that reads or writes a store runs a noted K1 first, on
 * its own (the order of effects on `stream` is the order of the calls, as without deferral); so does pca_status.  Until
 * then the DEVICE pointers of the observation must stay valid (host arrays are held in the context's staging block).
 * pca_k1_defer(ctx, 0) turns it off and runs what is noted; pca_k1_flush runs what is noted.  0 / -1. */
int pca_k1_defer(pca_ctx *ctx, int on);
int pca_k1_flush(pca_ctx *ctx);
int pca_kitti_generate_bev(pca_ctx *ctx, const pca_store *store, const int64_t *frame_off /*dev*/, int slot_begin,
                           int slot_split, int slot_end, int64_t max_points, const pca_bev_params *prm,
                           const double *pending_Ts, const int *pending_slot_ends, int n_pending, int write_back,
                           void *workspace /*dev*/, int64_t workspace_bytes, uint16_t *planes_f16 /*dev*/,
                           void *host_planes /*pinned or NULL*/, const pca_host_track *track, double *traj_rows,
                           int32_t *traj_start, int32_t *n_rows, void *stream);

/* The same two calls with their arguments in a block the caller keeps between calls: a binding in an interpreted language
 * (ctypes: ~0.2 us per converted argument, 45 arguments per step) rewrites the few fields that change -- slot numbers, the
 * pose, the output pointers -- and passes two pointers.  Members as the parameters of the same name; `evicted`, `path_length`,
 * `n_rows` are outputs.  pca_kitti_generate_bev_v also takes the view hint (hint_F > 0: the arguments of pca_bev_view_hint;
 * it is skipped, like the hint itself, when the call writes owed transforms back).  Returns what the plain form returns. */
typedef struct {
    const pca_kitti_obs *obs; const double *P; int32_t H, W; const uint64_t *filter_mask; const pca_store *store;
    int64_t *frame_off; int32_t slot, sample_mode; pca_host_track *track; const double *T_new_prev; double horizon;
    int64_t evicted; double path_length; void *stream;
} pca_kitti_integrate_args;
int pca_kitti_integrate_v(pca_ctx *ctx, pca_kitti_integrate_args *a);
typedef struct {
    const pca_store *store; const int64_t *frame_off; int32_t slot_begin, slot_split, slot_end, pad0; int64_t max_points;
    const pca_bev_params *prm; const double *pending_Ts; const int *pending_slot_ends; int32_t n_pending, write_back;
    void *workspace; int64_t workspace_bytes; uint16_t *planes_f16; void *host_planes; const pca_host_track *track;
    double *traj_rows; int32_t *traj_start; int32_t n_rows, pad1; void *stream;
    int32_t hint_slot0, hint_F; const double *hint_then; const float *hint_box; const double *hint_cone; const double *hint_now;
    int32_t hinted, pad2;          /* out: 1 if the hint left frames out */
} pca_kitti_generate_bev_args;
int pca_kitti_generate_bev_v(pca_ctx *ctx, pca_kitti_generate_bev_args *a);

/* ------------------------------------------------------------------------------------------------
 * Host arrays -> device for callers that hold their observations in pageable host memory -- what the reference's drivers
 * pass to integrate() (run_kitti360_bev_gen.py:173-190: numpy arrays straight from the loader).  src[k] (pageable, bytes[k]
 * long) is copied into pinned[k] (page-locked, caller-owned, at least bytes[k]) on a small pool of host threads, then
 * pinned[k] -> dev[k] is enqueued on `stream` (asynchronous).  The pinned blocks may be reused once the stream has passed
 * the copies (record an event after the call).  PCA_STAGING_THREADS (default 3; 0 = the calling thread alone) and
 * PCA_STAGING_SPIN_US (default 2000: how long the pool's threads poll after a job before they sleep) tune the pool.
 * Returns 0, -1 for bad arguments, -2 if a copy could not be enqueued.
 * ------------------------------------------------------------------------------------------------ */
int pca_host_stage_h2d(int n, const void *const *src, void *const *pinned, void *const *dev, const int64_t *bytes,
                       void *stream);

/* ------------------------------------------------------------------------------------------------
 * A device result on its way to the host (the BEV planes of a sample: sem_bev.py hands them back as host float16
 * arrays, bev_generator/sem_bev.py:164-205 of the reference) without the caller's stream waiting for the copy: it runs on
 * a side stream of the context behind everything enqueued on `stream` so far.  `pinned`: page-locked, caller-owned, to be
 * left alone until pca_host_d2h_wait(ticket) has returned.  Returns the ticket (0..63) or -1.
 * ------------------------------------------------------------------------------------------------ */
int pca_host_d2h_async(pca_ctx *ctx, const void *dev, void *pinned, int64_t bytes, void *stream);
int pca_host_d2h_wait(pca_ctx *ctx, int ticket);

/* ------------------------------------------------------------------------------------------------
 * Host helper (no device work): the accumulator's pose track -- the per-frame bookkeeping integrate() does on Python
 * lists in the reference.  Replaces sem_pc_accum.py:156-165 (update_poses), :185-209 (remove_observations), :211-228
 * (comp_incr_path_dist), :404-415 (dist) and, for runners that replay it, the sample trigger of
 * run_kitti360_bev_gen.py:218-240.  Bit-identical to the numpy expressions by construction: `cblas_dgemv64` is the
 * entry point of the cblas_dgemv (64-bit integers) of the OpenBLAS that numpy itself has loaded, so both matrix products
 * ((4,4)@(4,1) per pose, tri(n)@d) run the very kernel numpy runs; np.sum is restated as numpy's pairwise summation.
 *   pca_host_track_step      one integrate(): transform stored poses by T_new_prev, append [0,0,0], push the newest
 *                            segment, evict beyond `horizon`; returns the number of evicted frames, *path_length = the
 *                            path before the eviction (NaN while there is one pose only)
 *   pca_host_track_trigger   present index of a BEV sample, -1 if one of the driver's three conditions says no,
 *                            -2 if previous_idx is out of range (IndexError in the driver)
 *   pca_host_track_incr      tri(n) @ d into out[n_segments]
 * poses(): [len][4] doubles x, y, z, 1 (valid until the next mutating call); segments(): [n_segments].
 * ------------------------------------------------------------------------------------------------ */
typedef struct pca_host_track pca_host_track;
/* The (4,4)@(4,1) product per pose: mode 0 calls cblas_dgemv (exact by construction, ~150 ns of call overhead per pose);
 * modes 1..5 are the closed forms a dgemv kernel can reduce to for a 4-long dot product.  The binding probes them against
 * numpy on random data (pca_host_gemv4_probe) and selects one only if it agrees everywhere (pca_host_gemv4_mode). */
int pca_host_gemv4_probe(void *cblas_dgemv64, int mode, const double *T /*[n][16]*/, const double *x /*[n][4]*/, int64_t n,
                         double *y /*[n][4]*/);
int pca_host_gemv4_mode(int mode);
/* tri(n) @ d: the eviction needs its first rows, the trigger its last row and one crossing.  With blocks on, groups of 8
 * rows (from a multiple of 8) are computed on their own instead of the full n x n product; the binding turns that on only
 * after pca_host_incr_probe reproduced numpy's full product on random data. */
int pca_host_incr_probe(void *cblas_dgemv64, const double *d, int64_t n, int64_t r0, int64_t r1, double *out);
int pca_host_incr_blocks(int on);
int pca_host_track_create(pca_host_track **out, void *cblas_dgemv64);
void pca_host_track_destroy(pca_host_track *t);
int64_t pca_host_track_len(const pca_host_track *t);
int64_t pca_host_track_n_segments(const pca_host_track *t);
const double *pca_host_track_poses(const pca_host_track *t);
const double *pca_host_track_segments(const pca_host_track *t);
int pca_host_track_set(pca_host_track *t, const double *poses /*[n][3]*/, int64_t n, const double *segs, int64_t nd);
int pca_host_track_transform(pca_host_track *t, const double T[16]);
int pca_host_track_append(pca_host_track *t, const double pose[3]);
int pca_host_track_push_segment(pca_host_track *t, double *path_length);
int pca_host_track_incr(pca_host_track *t, double *out);
int64_t pca_host_track_evict_beyond(pca_host_track *t, double horizon, double path_length);
int64_t pca_host_track_step(pca_host_track *t, const double T_new_prev[16], double horizon, double *path_length);
int64_t pca_host_track_trigger(pca_host_track *t, double bev_horizon, int64_t previous_idx, double min_step);

/* ------------------------------------------------------------------------------------------------
 * Optional timing with HIP events recorded on the call's stream.  on = 1: every kernel launch is bracketed
 * (ids PCA_K_KITTI .. PCA_K_DEDUP); on = 2: only whole multi-kernel units are (PCA_K_BEV_UNIT = one
 * pca_bev_generate call, launch gaps included, without the per-kernel events in between); 0: off.  pca_profile_read synchronises, returns the accumulated time / launch count of one
 * kernel id since the last pca_profile_enable(ctx, 1) and keeps recording.  (No reference counterpart.)
 * PCA_K_BEV_SCAN / PCA_K_BEV_SCATTER are kept for the numbering only: level 1 of the rasteriser is one kernel
 * (PCA_K_BEV_BIN) since round 2 and these two never record a launch.
 * ------------------------------------------------------------------------------------------------ */
enum {
    PCA_K_KITTI = 0, PCA_K_NUSC, PCA_K_PROJECT_CAMS, PCA_K_RETRANSFORM, PCA_K_MARK_DYNAMIC,
    PCA_K_BEV_BIN, PCA_K_BEV_SCAN, PCA_K_BEV_SCATTER, PCA_K_BEV_CELLS, PCA_K_BEV_CELLS_HEAVY, PCA_K_DEDUP, PCA_K_BEV_UNIT, PCA_K_ICP, PCA_K_COUNT
};
int pca_profile_enable(pca_ctx *ctx, int on);
int pca_profile_read(pca_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* PCA_H */
