/*
 * pca_oracle.c -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product path (pc-accumulation-lib_amd/) never does.  Parity status: PINNED -- every function
 * below is checked bit-for-bit (integers, masks, coordinates) against golden vectors produced by
 * running the real reference in the build container (tools/make_golden.py -> tests/golden/ npz files,
 * checked by tests/test_oracle_golden.py).
 *
 * Scalar, single-threaded, plain C.  Numerics contract:
 *   - numpy's small matmuls are OpenBLAS dgemm == an f64 FMA chain in k order -> explicit fma();
 *   - every other numpy expression is un-fused -> build with -ffp-contract=off;
 *   - np.round == rint (half to even), astype(int) of a finite value truncates.
 *
 * Each function cites the reference file:line it restates (paths relative to the reference root).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* fma() resolves to the hardware instruction where the CPU has it (ifunc clone), libm otherwise. */
#define ORC_API __attribute__((visibility("default"), target_clones("fma", "default")))

/* ---- shared record layout: structure-of-arrays point store ---------------------------------- */
typedef struct {
    double *x, *y, *z;   /* coordinates, f64 (reference: columns 0..2 of the (M,10) f64 rows)  */
    float *intensity;    /* raw f32 intensity; reference column 3 = raw (KITTI) or raw/255 (NuScenes) */
    uint32_t *rgbs;      /* r | g<<8 | b<<16 | sem<<24   (reference columns 4..7, small integers)  */
    int32_t *inst;       /* reference column 8 */
    uint8_t *dyn;        /* reference column 9 */
} orc_store;

static inline int in_mask(const uint64_t m[4], unsigned c) { return (int)((m[c >> 6] >> (c & 63)) & 1u); }

/* 3x4 / 4x4 row times homogeneous point, k order, fma chain (== dgemm for K=4).
 * sem_pc_accum.py:357-363 (velo2frame), :177-181 (update_sem_pcs), datasets/nuscenes_utils.py:58-60 */
static inline double row4(const double *r, double x, double y, double z)
{
    double a = r[0] * x;
    a = fma(r[1], y, a);
    a = fma(r[2], z, a);
    a = fma(r[3], 1.0, a);
    return a;
}

/* -------------------------------------------------------------------------------------------
 * K1: KITTI fused  velo2frame -> velo2img -> gen_semantic_pc (rgb and semseg) -> filter_semseg_pc
 *   sem_pc_accum.py:347-402 (projection + mask), :323-345 (gather), :317-321 (filter),
 *   kitti360_sem_pc_accum.py:130-156 (column assembly; use_gt_sem branch :139-144).
 * pts [n,4] f32.  If sem_gt != NULL the use_gt_sem branch is taken: no projection, rgb = 0.
 * Per input point (optional, may be NULL): mask_out[n] (in-frustum), u_out[n], v_out[n] (int64 as numpy).
 * Appends kept records to `st` starting at index `base`; returns the number kept.  inst = 0, dyn = 0.
 * ------------------------------------------------------------------------------------------- */
/* Bilinear weights of pts_feat_from_img(..., 'bilinear'), datasets/nuscenes_utils.py:197-210: neighbours floor / ceil,
 * un-fused f64, fourth weight = 1 - (sum of the others), value = w_ff a(v0,u0) + w_cc a(v1,u1) + w_cf a(v1,u0) + w_fc a(v0,u1).
 * strict = the reference to the letter; else (opt-in sample mode of K1 / K1n, no reference caller) the upper neighbour is
 * floor + 1, so an integer coordinate gives weight 0 instead of 0 / 0. */
typedef struct { double u0, u1, v0, v1, w_ff, w_cc, w_cf, w_fc; } orc_bilin;
static orc_bilin bilin_weights(double u, double v, int strict)
{
    orc_bilin b;
    b.u0 = floor(u); b.v0 = floor(v);
    b.u1 = strict ? ceil(u) : b.u0 + 1.0;
    b.v1 = strict ? ceil(v) : b.v0 + 1.0;
    double area = (b.u1 - b.u0) * (b.v1 - b.v0);
    b.w_ff = (b.u1 - u) * (b.v1 - v) / area;
    b.w_cc = (u - b.u0) * (v - b.v0) / area;
    b.w_fc = (u - b.u0) * (b.v1 - v) / area;
    b.w_cf = 1.0 - (b.w_ff + b.w_cc + b.w_fc);
    return b;
}
static double bilin_value(const orc_bilin *b, double a_ff, double a_cc, double a_cf, double a_fc)
{
    return b->w_ff * a_ff + b->w_cc * a_cc + b->w_cf * a_cf + b->w_fc * a_fc;
}
static uint32_t bilin_rgb(const orc_bilin *b, const uint8_t *ff, const uint8_t *cc, const uint8_t *cf, const uint8_t *fc)
{
    uint32_t out = 0;
    for (int ch = 0; ch < 3; ++ch) {
        double v = rint(bilin_value(b, (double)ff[ch], (double)cc[ch], (double)cf[ch], (double)fc[ch]));
        v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
        out |= (uint32_t)v << (8 * ch);
    }
    return out;
}
static int clampi(double v, int n) { return (int)(v < 0.0 ? 0.0 : (v > (double)(n - 1) ? (double)(n - 1) : v)); }

/* pts_feat_from_img(uv, map, 'bilinear') for a 2-D map (the only shape the reference's branch supports); -1 if the
 * reference's assert would fail. */
ORC_API int orc_sample_bilinear(const double *map, int H, int W, const double *uv, int64_t n, double *out)
{
    for (int64_t p = 0; p < n; ++p) {
        double u = uv[2 * p], v = uv[2 * p + 1];
        if (!(u > 1.0 && u < (double)W - 1.0 && v > 1.0 && v < (double)H - 1.0)) return -1;
        orc_bilin b = bilin_weights(u, v, 1);
        int u0 = (int)b.u0, u1 = (int)b.u1, v0 = (int)b.v0, v1 = (int)b.v1;
        out[p] = bilin_value(&b, map[(int64_t)v0 * W + u0], map[(int64_t)v1 * W + u1], map[(int64_t)v1 * W + u0],
                             map[(int64_t)v0 * W + u1]);
    }
    return 0;
}

static int orc_sample_mode = 0;    /* 0 nearest (the reference), 1 bilinear rgb: set by the tests of the opt-in mode */
ORC_API void orc_set_sample_mode(int mode) { orc_sample_mode = mode; }

ORC_API int64_t orc_kitti_project_sample_filter(const float *pts, int64_t n, const double *P, const uint8_t *rgb,
                                                const uint8_t *sem, const uint8_t *sem_gt, int H, int W,
                                                const uint64_t *filter_mask, orc_store *st, int64_t base,
                                                uint8_t *mask_out, int64_t *u_out, int64_t *v_out)
{
    int64_t m = 0;
    for (int64_t p = 0; p < n; ++p) {
        const float *q = pts + 4 * p;
        double x = (double)q[0], y = (double)q[1], z = (double)q[2];
        uint32_t packed;
        if (sem_gt) {
            unsigned c = sem_gt[p];
            if (in_mask(filter_mask, c)) continue;
            packed = (uint32_t)c << 24;
        } else {
            double fx = row4(P + 0, x, y, z);
            double fy = row4(P + 4, x, y, z);
            double d = row4(P + 8, x, y, z);
            if (d == 0.0) d = -1e-6;                       /* :385 */
            double uf = rint(fx / fabs(d));                /* :386 */
            double vf = rint(fy / fabs(d));                /* :387 */
            int ok = (uf >= 0.0) && (uf < (double)W) && (vf >= 0.0) && (vf < (double)H) && (d > 0.0) &&
                     (d < INFINITY);                       /* :390-394 */
            if (mask_out) mask_out[p] = (uint8_t)ok;
            /* astype(int) of non-finite / out-of-range doubles is INT64_MIN on x86-64 */
            if (u_out) u_out[p] = (uf >= -9.2e18 && uf <= 9.2e18) ? (int64_t)uf : INT64_MIN;
            if (v_out) v_out[p] = (vf >= -9.2e18 && vf <= 9.2e18) ? (int64_t)vf : INT64_MIN;
            if (!ok) continue;
            int u = (int)uf, v = (int)vf;
            unsigned c = sem[(int64_t)v * W + u];
            if (in_mask(filter_mask, c)) continue;         /* :317-321 */
            const uint8_t *px = rgb + ((int64_t)v * W + u) * 3;
            packed = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | ((uint32_t)c << 24);
            if (orc_sample_mode) {                         /* opt-in: bilinear colour, neighbours clamped to the image */
                orc_bilin b = bilin_weights(fx / fabs(d), fy / fabs(d), 0);
                int u0 = clampi(b.u0, W), u1 = clampi(b.u1, W), v0 = clampi(b.v0, H), v1 = clampi(b.v1, H);
                packed = bilin_rgb(&b, rgb + ((int64_t)v0 * W + u0) * 3, rgb + ((int64_t)v1 * W + u1) * 3,
                                   rgb + ((int64_t)v1 * W + u0) * 3, rgb + ((int64_t)v0 * W + u1) * 3) | ((uint32_t)c << 24);
            }
        }
        int64_t o = base + m++;
        st->x[o] = x; st->y[o] = y; st->z[o] = z;
        st->intensity[o] = q[3];
        st->rgbs[o] = packed;
        st->inst[o] = 0;
        st->dyn[o] = 0;
    }
    return m;
}

/* K2: in-place rigid re-transform of stored points.  sem_pc_accum.py:167-183 (T is 4x4 row-major). */
ORC_API void orc_retransform(double *x, double *y, double *z, int64_t n, const double *T)
{
    for (int64_t p = 0; p < n; ++p) {
        double a = x[p], b = y[p], c = z[p];
        x[p] = row4(T + 0, a, b, c);
        y[p] = row4(T + 4, a, b, c);
        z[p] = row4(T + 8, a, b, c);
    }
}

/* datasets/nuscenes_utils.py:46-60 homo_transform on an (n,3) row-major array -> (n,3). */
ORC_API void orc_homo_transform(const double *T, const double *pts, int64_t n, double *out)
{
    for (int64_t p = 0; p < n; ++p) {
        double a = pts[3 * p], b = pts[3 * p + 1], c = pts[3 * p + 2];
        out[3 * p + 0] = row4(T + 0, a, b, c);
        out[3 * p + 1] = row4(T + 4, a, b, c);
        out[3 * p + 2] = row4(T + 8, a, b, c);
    }
}

/* -------------------------------------------------------------------------------------------
 * K1n: NuScenes oracle-pose  gather(nearest) -> invalid/filter mask -> compaction -> ego->world.
 *   nuscenes_oracle_sem_pc_accum.py:457-501, datasets/nuscenes_utils.py:181-214 ('nearest').
 * pc [n,7] f64 rows [x,y,z,intensity,u,v,inst]; cam_idx [n] (-1 = on no image);
 * imgs [ncam,H,W,3] u8, sems [ncam,H,W] u8.  Stored intensity is RAW (the /255 of :491 is applied
 * by consumers through the store's intensity mode).  Returns kept count, or -1 if a point assigned
 * to a camera has (u,v) outside the open box (1, wh-1) (reference: AssertionError at utils :195).
 * ------------------------------------------------------------------------------------------- */
ORC_API int64_t orc_nusc_sample_filter_transform(const double *pc, const int64_t *cam_idx, int64_t n,
                                                 const uint8_t *imgs, const uint8_t *sems, int ncam, int H, int W,
                                                 const double *T, const uint64_t *filter_mask, orc_store *st,
                                                 int64_t base)
{
    int64_t m = 0;
    for (int64_t p = 0; p < n; ++p) {
        const double *q = pc + 7 * p;
        int64_t c = cam_idx[p];
        if (c < 0 || c >= ncam) continue;                     /* feats stay -1 -> invalid (:476) */
        double u = q[4], v = q[5];
        if (!(u > 1.0 && u < (double)W - 1.0 && v > 1.0 && v < (double)H - 1.0)) return -1;
        int ui = (int)rint(u), vi = (int)rint(v);             /* utils :212-213 */
        int64_t pix = ((int64_t)c * H + vi) * W + ui;
        unsigned s = sems[pix];
        if (in_mask(filter_mask, s)) continue;                /* :478-480 */
        const uint8_t *px = imgs + pix * 3;
        int64_t o = base + m++;
        st->x[o] = row4(T + 0, q[0], q[1], q[2]);             /* :488 */
        st->y[o] = row4(T + 4, q[0], q[1], q[2]);
        st->z[o] = row4(T + 8, q[0], q[1], q[2]);
        st->intensity[o] = (float)q[3];
        st->rgbs[o] = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16) | ((uint32_t)s << 24);
        if (orc_sample_mode) {                                 /* opt-in: bilinear colour (all four neighbours are inside) */
            orc_bilin b = bilin_weights(u, v, 0);
            const uint8_t *img = imgs + (int64_t)c * H * W * 3;
            int u0 = (int)b.u0, u1 = (int)b.u1, v0 = (int)b.v0, v1 = (int)b.v1;
            st->rgbs[o] = bilin_rgb(&b, img + ((int64_t)v0 * W + u0) * 3, img + ((int64_t)v1 * W + u1) * 3,
                                    img + ((int64_t)v1 * W + u0) * 3, img + ((int64_t)v0 * W + u1) * 3) | ((uint32_t)s << 24);
        }
        st->inst[o] = (int32_t)q[6];
        st->dyn[o] = 0;
    }
    return m;
}

/* K3: nuscenes_oracle_sem_pc_accum.py:223-229, :243-250  sem_pc[sem_pc[:,8]==inst_idx, 9] = 1 */
ORC_API void orc_mark_dynamic(const int32_t *inst, uint8_t *dyn, int64_t begin, int64_t end, int32_t inst_idx)
{
    for (int64_t p = begin; p < end; ++p)
        if (inst[p] == inst_idx) dyn[p] = 1;
}

/* -------------------------------------------------------------------------------------------
 * K0n: lidar -> ego -> global -> 6 cameras, pinhole projection, last camera wins.
 *   obs_dataloaders/nuscenes_obs_dataloader.py:162-202, datasets/nuscenes_utils.py:112-136.
 *   view_points() is third-party (nuscenes-devkit, unpinned, absent): restated from its published
 *   behaviour  uv = (viewpad @ [p;1])[:2] / (viewpad @ [p;1])[2]  -- "parity unpinned" at that call.
 * pc_lidar [n,3] f64; K [ncam,9]; T_cam_from_glob [ncam,16] (already inverted on the host with
 * np.linalg.inv as the reference does); wh [ncam,2].
 * ------------------------------------------------------------------------------------------- */
ORC_API void orc_nusc_project_cams(const double *pc_lidar, int64_t n, const double *T_ego_from_lidar,
                                   const double *T_glob_from_ego, const double *T_cam_from_glob, const double *K,
                                   const double *wh, int ncam, double *pc_in_ego, double *uv, int64_t *cam_idx)
{
    for (int64_t p = 0; p < n; ++p) {
        double a = pc_lidar[3 * p], b = pc_lidar[3 * p + 1], c = pc_lidar[3 * p + 2];
        double ex = row4(T_ego_from_lidar + 0, a, b, c);
        double ey = row4(T_ego_from_lidar + 4, a, b, c);
        double ez = row4(T_ego_from_lidar + 8, a, b, c);
        pc_in_ego[3 * p] = ex; pc_in_ego[3 * p + 1] = ey; pc_in_ego[3 * p + 2] = ez;
        double gx = row4(T_glob_from_ego + 0, ex, ey, ez);
        double gy = row4(T_glob_from_ego + 4, ex, ey, ez);
        double gz = row4(T_glob_from_ego + 8, ex, ey, ez);
        double ou = 0.0, ov = 0.0;
        int64_t oc = -1;
        for (int j = 0; j < ncam; ++j) {
            const double *Tc = T_cam_from_glob + 16 * j;
            const double *Kj = K + 9 * j;
            double cx = row4(Tc + 0, gx, gy, gz), cy = row4(Tc + 4, gx, gy, gz), cz = row4(Tc + 8, gx, gy, gz);
            if (!(cz > 1e-3)) continue;                                  /* utils :125 */
            /* viewpad rows: [K_r0 K_r1 K_r2 0] . [x y z 1] as a k-ordered fma chain */
            double r0[4] = {Kj[0], Kj[1], Kj[2], 0.0}, r1[4] = {Kj[3], Kj[4], Kj[5], 0.0},
                   r2[4] = {Kj[6], Kj[7], Kj[8], 0.0};
            double px = row4(r0, cx, cy, cz), py = row4(r1, cx, cy, cz), pz = row4(r2, cx, cy, cz);
            double u = px / pz, v = py / pz;
            if (u > 1.0 && u < wh[2 * j] - 1.0 && v > 1.0 && v < wh[2 * j + 1] - 1.0) {   /* :134-135 */
                ou = u; ov = v; oc = j;
            }
        }
        uv[2 * p] = ou; uv[2 * p + 1] = ov;
        cam_idx[p] = oc;
    }
}

/* ---- f64 -> f16, round to nearest even directly from the double (numpy astype(np.float16)) ---- */
ORC_API uint16_t orc_f64_to_f16(double d)
{
    uint64_t b; memcpy(&b, &d, 8);
    uint16_t sign = (uint16_t)((b >> 48) & 0x8000u);
    uint64_t absb = b & 0x7fffffffffffffffull;
    if (absb >= 0x7ff0000000000000ull)                     /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((absb > 0x7ff0000000000000ull) ? 0x200u : 0u));
    int e = (int)(absb >> 52) - 1023;                       /* unbiased */
    uint64_t man = (absb & 0xfffffffffffffull) | 0x10000000000000ull;   /* 53 bits, implicit one */
    if (absb == 0) return sign;
    if (e >= 16) return (uint16_t)(sign | 0x7c00u);         /* overflow before rounding */
    int shift;                                              /* bits to drop from the 53-bit mantissa */
    int he;                                                 /* half biased exponent of the result */
    if (e >= -14) { shift = 42; he = e + 15; }              /* normal half: keep 11 bits */
    else { shift = 42 + (-14 - e); he = 0; }                /* subnormal half */
    if (shift > 63) return sign;                            /* far below half of the smallest subnormal */
    uint64_t keep = man >> shift;
    uint64_t rem = man & ((1ull << shift) - 1);
    uint64_t half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (keep & 1))) keep++;
    uint32_t h;
    if (he > 0) h = (uint32_t)((he - 1) << 10) + (uint32_t)keep;   /* keep has the implicit bit at 1<<10 */
    else h = (uint32_t)keep;                                        /* subnormal (may round up into normal) */
    if (h >= 0x7c00u) h = 0x7c00u;
    return (uint16_t)(sign | h);
}

/* -------------------------------------------------------------------------------------------
 * BEV rasteriser (K4-K7), one call = present + future + full.
 *   window assembly        kitti360_sem_pc_accum.py:189-213, nuscenes_oracle_sem_pc_accum.py:535-581
 *   rotate/translate/crop  bev_generator/bev_generator.py:207-255
 *   height filter, floor   :152-157, :737-747
 *   static partition       bev_generator/sem_bev.py:57-58 (dyn == 1 dropped)
 *   counts / dirichlet     bev_generator.py:373-394, :438-480
 *   intensity              bev_generator.py:396-415, sem_bev.py:593-617
 *   elevation (min z)      sem_bev.py:535-554
 *   rgb median             sem_bev.py:619-669
 * Points [0,n_split) are the 'present' set, [n_split,n) 'future', all = 'full' (the reference
 * concatenates frames in order, so set membership is a prefix property).
 * Output planes f64 [21][px][px] in the order (set-major) present, future, full x
 *   {road, intensity, r, g, b, dynamic, elevation};  row = px-1-j, col = i.
 * intraw (optional) [3][px][px]: intensity before road_marking_transform.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    double origin[3];      /* bev_frame_coords (poses[present_idx])                                 */
    double R[9];           /* rotation_matrix_3d(rot_ang), computed on the host with np.cos/np.sin  */
    double dx, dy;         /* translation after rotation                                            */
    double view;           /* aug_view_size = zoom * view_size                                      */
    double height_filter;  /* keep z < height_filter; NaN = disabled                                */
    double int_scaler, int_sep_scaler, int_mid_threshold;
    double rgb_fill;       /* median of an empty cell (sem_bev.py:661-664)                          */
    int32_t px;            /* pixel_size                                                            */
    int32_t road_class;    /* sem_idxs['road']                                                      */
    uint64_t dynobj_mask[4]; /* classes car/truck/bus/motorcycle                                    */
    int32_t intensity_div255; /* 1: reference column 3 = raw/255. (NuScenes), 0: raw (KITTI)        */
    int32_t pad;
} orc_bev_params;

static int cmp_u8(const void *a, const void *b) { return (int)*(const uint8_t *)a - (int)*(const uint8_t *)b; }

static double median_u8(uint8_t *v, int64_t n, double fill)
{
    if (n == 0) return fill;
    qsort(v, (size_t)n, 1, cmp_u8);
    if (n & 1) return (double)v[n / 2];
    return ((double)v[n / 2 - 1] + (double)v[n / 2]) / 2.0;   /* np.median: mean of the two middles */
}

ORC_API int orc_bev(const orc_store *st, const double *intensity64, int64_t n, int64_t n_split,
                    const orc_bev_params *prm, double *planes, uint16_t *planes_f16, double *intraw,
                    int64_t *cell_out /* optional [n]: cell id or -1 */)
{
    const int px = prm->px;
    const int64_t ncell = (int64_t)px * px;
    const double v = prm->view;
    const double lo = -0.5 * v, hi = 0.5 * v;              /* bev_generator.py:248-252 */
    const double half_px = 0.5 * (double)px;
    int32_t *cell = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    double *zz = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!cell || !zz) return -1;
    const double *R = prm->R;
    for (int64_t p = 0; p < n; ++p) {
        double x = st->x[p] - prm->origin[0];              /* kitti360_sem_pc_accum.py:193 */
        double y = st->y[p] - prm->origin[1];
        double z = st->z[p] - prm->origin[2];
        double a = R[0] * x; a = fma(R[1], y, a); a = fma(R[2], z, a);   /* bev_generator.py:227 */
        double b = R[3] * x; b = fma(R[4], y, b); b = fma(R[5], z, b);
        double c = R[6] * x; c = fma(R[7], y, c); c = fma(R[8], z, c);
        a += prm->dx;                                      /* :230-231 */
        b += prm->dy;
        int keep = (a > lo) && (a < hi) && (b > lo) && (b < hi);
        if (keep && !isnan(prm->height_filter)) keep = c < prm->height_filter;   /* :152-154 */
        if (keep && st->dyn[p] == 1) keep = 0;             /* sem_bev.py:57-58 static only */
        if (!keep) { cell[p] = -1; zz[p] = 0; if (cell_out) cell_out[p] = -1; continue; }
        double fi = floor(a / v * (double)px + half_px);   /* :743-745 */
        double fj = floor(b / v * (double)px + half_px);
        int i = (int)fi, j = (int)fj;
        if (i > px - 1) i = px - 1;                        /* measure-zero ulp case, SURVEY.md 4 */
        if (j > px - 1) j = px - 1;
        if (i < 0) i = 0;
        if (j < 0) j = 0;
        cell[p] = (px - 1 - j) * px + i;
        zz[p] = c;
        if (cell_out) cell_out[p] = cell[p];
    }
    /* per set: 0 present [0,n_split), 1 future [n_split,n), 2 full [0,n) */
    int64_t *cnt = (int64_t *)calloc((size_t)ncell + 1, sizeof(int64_t));
    int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncell + 1));
    int64_t *order = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    uint8_t *tmp = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    if (!cnt || !fill || !order || !tmp) return -1;
    for (int s = 0; s < 3; ++s) {
        int64_t b0 = (s == 1) ? n_split : 0, b1 = (s == 0) ? n_split : n;
        double *P = planes + (int64_t)s * 7 * ncell;
        double *road = P, *inten = P + ncell, *r = P + 2 * ncell, *g = P + 3 * ncell, *bl = P + 4 * ncell,
               *dynp = P + 5 * ncell, *elev = P + 6 * ncell;
        memset(cnt, 0, sizeof(int64_t) * (size_t)(ncell + 1));
        for (int64_t c = 0; c < ncell; ++c) { road[c] = 0; inten[c] = 0; dynp[c] = 0; elev[c] = 0; }
        /* counts + sequential (bincount-order) intensity sum + min z */
        int64_t *n_all = cnt;   /* reuse below for bucket offsets; keep separate small arrays for classes */
        double *n_road = road, *n_dyn = dynp;              /* accumulate counts in the output planes first */
        uint8_t *seen = (uint8_t *)calloc((size_t)ncell, 1);
        if (!seen) return -1;
        for (int64_t p = b0; p < b1; ++p) {
            int32_t c = cell[p];
            if (c < 0) continue;
            n_all[c + 1]++;
            unsigned sem = st->rgbs[p] >> 24;
            if ((int)sem == prm->road_class) {
                n_road[c] += 1.0;
                double iv = intensity64 ? intensity64[p]
                                        : (prm->intensity_div255 ? (double)st->intensity[p] / 255.0
                                                                 : (double)st->intensity[p]);
                inten[c] += iv;                            /* bincount: sequential in point order */
            }
            if (in_mask(prm->dynobj_mask, sem)) n_dyn[c] += 1.0;
            if (!seen[c] || zz[p] < elev[c]) { elev[c] = zz[p]; seen[c] = 1; }   /* sem_bev.py:543-552 */
        }
        free(seen);
        /* bucket points per cell for the medians */
        for (int64_t c = 0; c < ncell; ++c) n_all[c + 1] += n_all[c];
        memcpy(fill, n_all, sizeof(int64_t) * (size_t)(ncell + 1));
        for (int64_t p = b0; p < b1; ++p)
            if (cell[p] >= 0) order[fill[cell[p]]++] = p;
        for (int64_t c = 0; c < ncell; ++c) {
            int64_t o = n_all[c], k = n_all[c + 1] - o;
            double na = (double)k, nr = n_road[c], nd = n_dyn[c];
            for (int ch = 0; ch < 3; ++ch) {
                for (int64_t t = 0; t < k; ++t) tmp[t] = (uint8_t)(st->rgbs[order[o + t]] >> (8 * ch));
                double med = median_u8(tmp, k, prm->rgb_fill) / 255.0;    /* sem_bev.py:62-64 */
                (ch == 0 ? r : ch == 1 ? g : bl)[c] = med;
            }
            /* dirichlet expectation (bev_generator.py:468-478): (a+1) / ((a+1) + (b+1)) */
            road[c] = (nr + 1.0) / ((nr + 1.0) + ((na - nr) + 1.0));
            dynp[c] = (nd + 1.0) / ((nd + 1.0) + ((na - nd) + 1.0));
            double iraw = inten[c] / (nr + 1.0);                           /* bev_generator.py:413 */
            if (intraw) intraw[(int64_t)s * ncell + c] = iraw;
            double zarg = prm->int_sep_scaler * (iraw - prm->int_mid_threshold);
            double sg = 1.0 / (1.0 + exp(-zarg));                          /* sem_bev.py:615-617 */
            double iv = prm->int_scaler * sg;
            if (iv > 1.0) iv = 1.0;
            inten[c] = iv;
        }
    }
    if (planes_f16)
        for (int64_t t = 0; t < 21 * ncell; ++t) planes_f16[t] = orc_f64_to_f16(planes[t]);
    free(cell); free(zz); free(cnt); free(fill); free(order); free(tmp);
    return 0;
}
