// pca_api.hip -- context, error reporting and look-back workspace management (C ABI, include/pca.h).
#include <stdio.h>
#include <cmath>
#include "pca_common.h"

__global__ __launch_bounds__(256) void pca_fetch_block_kernel(const uint4 *src, uint4 *dst, int64_t n16)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) dst[i] = src[i];
}

int pca_fetch_block(pca_ctx *ctx, const void *mapped_host, int64_t offset, void *dev, int64_t bytes, hipStream_t s)
{
    if (bytes <= 0) return 0;
    void *src = nullptr;
    PCA_CHECK(ctx, hipHostGetDevicePointer(&src, const_cast<void *>(mapped_host), 0));
    src = reinterpret_cast<char *>(src) + offset;
    const int64_t n16 = (bytes + 15) / 16;
    const int grid = (int)((n16 + 255) / 256 < 64 ? (n16 + 255) / 256 : 64);
    hipLaunchKernelGGL(pca_fetch_block_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const uint4 *>(src),
                       reinterpret_cast<uint4 *>(dev), n16);
    PCA_CHECK(ctx, hipGetLastError());
    return 0;
}

int pca_ctx_reserve_tiles(pca_ctx *ctx, int64_t tiles, hipStream_t s)
{
    if (tiles <= ctx->tile_cap) return 0;
    // growing is rare (first call / bigger batch); it is the only place a launch path synchronises
    PCA_CHECK(ctx, hipStreamSynchronize(s));
    if (ctx->tile_state) PCA_CHECK(ctx, hipFree(ctx->tile_state));
    ctx->tile_cap = tiles * 2 > 4096 ? tiles * 2 : 4096;
    PCA_CHECK(ctx, hipMalloc(&ctx->tile_state, sizeof(uint64_t) * ctx->tile_cap));
    PCA_CHECK(ctx, hipMemsetAsync(ctx->tile_state, 0, sizeof(uint64_t) * ctx->tile_cap, s));
    ctx->epoch = 0;
    return 0;
}

// Launch tag for the look-back state words.  22 bits; when it wraps the array is cleared once so an
// entry left over from 4M launches ago can never be mistaken for a current one.
uint32_t pca_ctx_next_epoch(pca_ctx *ctx, hipStream_t s)
{
    ctx->epoch = (ctx->epoch + 1) & 0x3fffffu;
    if (ctx->epoch == 0) {
        (void)hipMemsetAsync(ctx->tile_state, 0, sizeof(uint64_t) * ctx->tile_cap, s);
        ctx->epoch = 1;
    }
    return ctx->epoch;
}

void pca_prof_begin(pca_ctx *ctx, int kid, hipStream_t s)
{
    pca_ctx::Ev e;
    if (!ctx->free_evs.empty()) { e = ctx->free_evs.back(); ctx->free_evs.pop_back(); }
    else { (void)hipEventCreate(&e.a); (void)hipEventCreate(&e.b); }
    e.kid = kid;
    (void)hipEventRecord(e.a, s);
    ctx->evs.push_back(e);
}

void pca_prof_end(pca_ctx *ctx, hipStream_t s) { (void)hipEventRecord(ctx->evs.back().b, s); }

static void prof_fold(pca_ctx *ctx)
{
    for (auto &e : ctx->evs) {
        (void)hipEventSynchronize(e.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { ctx->prof_ms[e.kid] += ms; ctx->prof_n[e.kid]++; }
        ctx->free_evs.push_back(e);
    }
    ctx->evs.clear();
}

extern "C" {

// ---------------------------------------------------------------------------------------------------------------
// Host helper (no device work): the ego trajectory of one BEV sample in grid coordinates.
//   rotate (numpy's (3,3)@(3,F) product = fma chain in k order) -> translate -> clip to the open view box, walking
//   EDGES a -> b (an inside `a` is emitted; a border crossing additionally emits the crossing point found by midpoint
//   bisection to 1e-4, carrying a's z) -> floor(x / view * px + 0.5 px).  Plain IEEE arithmetic, bit-identical to the
//   numpy / Python-float form in pca_amd/host_logic.py (bev_generator.py:207-371, :737-747 of the reference).
//   rows: [<= 2 (F-1)][3]; start: [F] first output row of every edge, start[F-1] = number of rows.  Returns that number.
// ---------------------------------------------------------------------------------------------------------------
// ---- which frames of a window can reach a raster's view (pca.h) -- host only ----------------------------------------------------
// Camera cone in the sensor's frame from P = [M | p4] (pixels = P (x, 1) / depth, depth = third row): centre C = -inverse(M) p4,
// ray through pixel (u, v) = inverse(M) (u, v, 1) (depth grows along it).  K1 keeps a point if rint(u), rint(v) land in the image
// and depth > 0: the pixel lies in [-0.5, W - 0.5] x [-0.5, H - 0.5]; the corners are taken one pixel further out.
void pca_host_camera_cone(const double P[12], int H, int W, double cone[15])
{
    const double a00 = P[0], a01 = P[1], a02 = P[2], a10 = P[4], a11 = P[5], a12 = P[6], a20 = P[8], a21 = P[9], a22 = P[10];
    const double c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
    const double c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
    const double id = 1.0 / (a00 * c00 + a01 * c10 + a02 * c20);
    const double inv[9] = {c00 * id, c01 * id, c02 * id, c10 * id, c11 * id, c12 * id, c20 * id, c21 * id, c22 * id};
    for (int r = 0; r < 3; ++r) cone[r] = -(inv[3 * r] * P[3] + inv[3 * r + 1] * P[7] + inv[3 * r + 2] * P[11]);
    const double us[2] = {-1.5, (double)W + 0.5}, vs[2] = {-1.5, (double)H + 0.5};
    for (int k = 0; k < 4; ++k) {
        const double u = us[k & 1], v = vs[k >> 1];
        for (int r = 0; r < 3; ++r) cone[3 + 3 * k + r] = inv[3 * r] * u + inv[3 * r + 1] * v + inv[3 * r + 2];
    }
}

void pca_f32_box_decode(const uint32_t *rows, int n, float *box)
{
    for (int f = 0; f < n; ++f)
        for (int k = 0; k < 3; ++k) {
            const uint32_t l = rows[6 * f + 2 * k], h = rows[6 * f + 2 * k + 1];
            const bool have = h != 0u;
            box[6 * f + 2 * k] = have ? pca_f32_from_ordered(~l) : 1.0f;
            box[6 * f + 2 * k + 1] = have ? pca_f32_from_ordered(h) : -1.0f;
        }
}

// Frame f misses the view for sure if its points all lie beyond ONE of the four lines that bound the view square (plus 1 cm):
// either the camera cone does (its apex does, and none of its four edge rays comes back), or the box of the kept points does
// (the linear function at the better end of every coordinate).  Both are affine images under  now inverse(then[f])  and the
// raster's view transform; a number that is not finite proves nothing (the frame counts as visible).
int pca_host_view_hull(int F, const double *then, const float *box, const double *cone, const double now[12],
                       const pca_bev_params *prm, int *first, int *last)
{
    if (F < 0 || !then || !now || !prm || !first || !last) return -1;
    const double h = 0.5 * prm->view + 0.01;
    auto visible = [&](int f) -> bool {
        const double *m = then + 12 * (size_t)f;
        const double a00 = m[0], a01 = m[1], a02 = m[2], a10 = m[4], a11 = m[5], a12 = m[6], a20 = m[8], a21 = m[9], a22 = m[10];
        const double c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
        const double c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
        const double c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
        const double id = 1.0 / (a00 * c00 + a01 * c10 + a02 * c20);
        const double inv[9] = {c00 * id, c01 * id, c02 * id, c10 * id, c11 * id, c12 * id, c20 * id, c21 * id, c22 * id};
        // rows x, y of M = now inverse(then), then the view transform on top: u = Q c + q for a point c of the frame
        double Q[6], q[2];
        {
            double Mr[6], Mt[2];
            for (int r = 0; r < 2; ++r) {
                for (int c = 0; c < 3; ++c) Mr[3 * r + c] = now[4 * r] * inv[c] + now[4 * r + 1] * inv[3 + c] + now[4 * r + 2] * inv[6 + c];
                Mt[r] = now[4 * r + 3] - (Mr[3 * r] * m[3] + Mr[3 * r + 1] * m[7] + Mr[3 * r + 2] * m[11]);
            }
            const double tx = Mt[0] - prm->origin[0], ty = Mt[1] - prm->origin[1];
            q[0] = prm->R[0] * tx + prm->R[1] * ty + prm->dx;
            q[1] = prm->R[3] * tx + prm->R[4] * ty + prm->dy;
            for (int c = 0; c < 3; ++c) {
                Q[c] = prm->R[0] * Mr[c] + prm->R[1] * Mr[3 + c];
                Q[3 + c] = prm->R[3] * Mr[c] + prm->R[4] * Mr[3 + c];
            }
        }
        for (int k = 0; k < 6; ++k) if (!std::isfinite(Q[k])) return true;
        if (!std::isfinite(q[0]) || !std::isfinite(q[1])) return true;
        for (int axis = 0; axis < 2; ++axis)
            for (int side = 0; side < 2; ++side) {
                const double s = side ? -1.0 : 1.0;             // beyond the line  s u[axis] = h ?
                const double *row = Q + 3 * axis;
                if (cone) {
                    const double ua = s * (row[0] * cone[0] + row[1] * cone[1] + row[2] * cone[2] + q[axis]);
                    bool out = ua >= h;
                    for (int k = 0; k < 4 && out; ++k) {
                        const double *d = cone + 3 + 3 * k;
                        out = s * (row[0] * d[0] + row[1] * d[1] + row[2] * d[2]) >= 0.0;
                    }
                    if (out) return false;
                }
                if (box) {
                    const float *b = box + 6 * (size_t)f;
                    if (b[0] <= b[1] && b[2] <= b[3] && b[4] <= b[5]) {
                        double lo = s * q[axis];
                        for (int c = 0; c < 3; ++c) {
                            const double x0 = s * row[c] * (double)b[2 * c], x1 = s * row[c] * (double)b[2 * c + 1];
                            lo += x0 < x1 ? x0 : x1;
                        }
                        if (lo >= h) return false;
                    }
                }
            }
        return true;
    };
    int a = 0, b = F - 1;
    while (a <= b && !visible(a)) ++a;
    while (b >= a && !visible(b)) --b;
    if (a > b) { *first = -1; *last = -1; } else { *first = a; *last = b; }
    return 0;
}

// pca_host_view_hull + pca_bev_bin_range in one call (frame 0 of the F frames sits in slot0).  Returns 1 if a range narrower
// than the F frames was set, 0 if every frame may reach the view (nothing set), -1 on bad arguments.
int pca_bev_view_hint(pca_ctx *ctx, int slot0, int F, const double *then, const float *box, const double *cone, const double now[12],
                      const pca_bev_params *prm)
{
    if (!ctx) return -1;
    int first = 0, last = 0;
    if (pca_host_view_hull(F, then, box, cone, now, prm, &first, &last)) return -1;
    if (first == 0 && last == F - 1) return 0;
    if (first < 0) { first = 0; last = -1; }
    ctx->bin_first = slot0 + first; ctx->bin_end = slot0 + last + 1; ctx->bin_valid = true;
    return 1;
}

int pca_bev_bin_range(pca_ctx *ctx, int slot_first, int slot_last_plus1)
{
    if (!ctx) return -1;
    ctx->bin_first = slot_first; ctx->bin_end = slot_last_plus1; ctx->bin_valid = slot_last_plus1 >= slot_first;
    return 0;
}

static inline bool box_inside(double x, double y, double lo, double hi) { return lo < x && x < hi && lo < y && y < hi; }

int pca_host_ego_to_grid(const double *full, int F, const double R[9], double dx, double dy, double view, int px,
                         double *rows, int32_t *start)
{
    if (F < 2) { if (F == 1 && start) start[0] = 0; return 0; }
    const double h = 0.5 * view, lo = -h, pxd = (double)px, half_px = 0.5 * pxd;
    auto rot = [&](const double *p, double &x, double &y, double &z) {
        x = fma(R[2], p[2], fma(R[1], p[1], R[0] * p[0])) + dx;
        y = fma(R[5], p[2], fma(R[4], p[1], R[3] * p[0])) + dy;
        z = fma(R[8], p[2], fma(R[7], p[1], R[6] * p[0]));
    };
    auto emit = [&](int m, double x, double y, double z) {
        rows[3 * m + 0] = floor(x / view * pxd + half_px);
        rows[3 * m + 1] = floor(y / view * pxd + half_px);
        rows[3 * m + 2] = z;
    };
    int m = 0;
    double ax, ay, az, bx, by, bz;
    rot(full, ax, ay, az);
    bool a_in = box_inside(ax, ay, lo, h);
    for (int k = 0; k + 1 < F; ++k) {
        rot(full + 3 * (k + 1), bx, by, bz);
        const bool b_in = box_inside(bx, by, lo, h);
        start[k] = m;
        if (a_in) emit(m++, ax, ay, az);
        if (a_in != b_in) {
            double x0 = ax, y0 = ay, x1 = bx, y1 = by, xm = 0.0, ym = 0.0, gap = __builtin_huge_val();
            while (gap > 1e-4) {
                xm = 0.5 * (x0 + x1);
                ym = 0.5 * (y0 + y1);
                const bool p0_in = box_inside(x0, y0, lo, h), mid_in = box_inside(xm, ym, lo, h);
                if (mid_in == p0_in) { gap = sqrt((xm - x0) * (xm - x0) + (ym - y0) * (ym - y0)); x0 = xm; y0 = ym; }
                else { gap = sqrt((xm - x1) * (xm - x1) + (ym - y1) * (ym - y1)); x1 = xm; y1 = ym; }
            }
            emit(m++, xm, ym, az);
        }
        ax = bx; ay = by; az = bz; a_in = b_in;
    }
    start[F - 1] = m;
    return m;
}

int pca_profile_enable(pca_ctx *ctx, int on)
{
    if (!ctx) return -1;
    prof_fold(ctx);
    ctx->profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
    if (on) for (int k = 0; k < PCA_K_COUNT; ++k) { ctx->prof_ms[k] = 0; ctx->prof_n[k] = 0; }
    return 0;
}

int pca_profile_read(pca_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches)
{
    if (!ctx || kernel_id < 0 || kernel_id >= PCA_K_COUNT) return -1;
    prof_fold(ctx);
    if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
    if (launches) *launches = ctx->prof_n[kernel_id];
    return 0;
}

int pca_version(void) { return PCA_VERSION; }

int pca_ctx_create(int device, pca_ctx **out)
{
    if (!out) return -1;
    *out = nullptr;
    pca_ctx *ctx = new pca_ctx();
    ctx->device = device;
    // (a failure here has no context to carry its message: it goes to stderr, naming the call)
#define PCA_CREATE_STEP(expr)                                                                                          \
    do {                                                                                                               \
        const hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                        \
            fprintf(stderr, "pca_ctx_create(device %d): %s failed: %s\n", device, #expr, hipGetErrorString(e_));       \
            delete ctx;                                                                                                \
            return -1;                                                                                                 \
        }                                                                                                              \
    } while (0)
    PCA_CREATE_STEP(hipSetDevice(device));
    PCA_CREATE_STEP(hipMalloc(&ctx->ticket, sizeof(PcaStatusBlock)));
    PCA_CREATE_STEP(hipMemset(ctx->ticket, 0, sizeof(PcaStatusBlock)));
    PCA_CREATE_STEP(hipHostMalloc(&ctx->status_host, sizeof(uint32_t)));
    PCA_CREATE_STEP(hipHostMalloc(&ctx->status_mirror, PCA_STATUS_BITS * sizeof(uint32_t), hipHostMallocMapped));
    PCA_CREATE_STEP(hipHostMalloc(&ctx->heavy_hint, 4 * sizeof(uint32_t), hipHostMallocMapped));    // [0] heavy tiles, [1] points binned by the last raster with a bin range
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0) ctx->n_cu = n_cu;
    *ctx->heavy_hint = 1;            // first call: assume heavy tiles exist
    ctx->heavy_hint[1] = ctx->heavy_hint[2] = ctx->heavy_hint[3] = 0;
    for (int b = 0; b < PCA_STATUS_BITS; ++b) ctx->status_mirror[b] = 0;
    PCA_CREATE_STEP(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->heavy_hint_dev), ctx->heavy_hint, 0));
    PCA_CREATE_STEP(hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->status_mirror_dev), ctx->status_mirror, 0));
    PCA_CREATE_STEP(hipMemcpy(reinterpret_cast<char *>(ctx->ticket) + offsetof(PcaStatusBlock, mirror), &ctx->status_mirror_dev, sizeof(void *), hipMemcpyHostToDevice));
#undef PCA_CREATE_STEP
    *out = ctx;
    return 0;
}

void pca_ctx_destroy(pca_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->tile_state) (void)hipFree(ctx->tile_state);
    if (ctx->ticket) (void)hipFree(ctx->ticket);
    for (int i = 0; i < 2; ++i) {
        if (ctx->k1_ws[i]) (void)hipFree(ctx->k1_ws[i]);
        if (ctx->k1_pin[i]) (void)hipHostFree(ctx->k1_pin[i]);
        if (ctx->k1_pin_ev[i]) (void)hipEventDestroy(ctx->k1_pin_ev[i]);
    }
    if (ctx->k1_frames_dev) (void)hipFree(ctx->k1_frames_dev);
    if (ctx->k1_tiny) (void)hipFree(ctx->k1_tiny);
    for (auto &st : ctx->stage) {
        if (st.done) { (void)hipEventSynchronize(st.done); (void)hipEventDestroy(st.done); }
        if (st.pin) (void)hipHostFree(st.pin);
        if (st.dev) (void)hipFree(st.dev);
    }
    if (ctx->bevm_pin) (void)hipHostFree(ctx->bevm_pin);
    if (ctx->bevm_ev) (void)hipEventDestroy(ctx->bevm_ev);
    for (auto &e : ctx->d2h_done) if (e) (void)hipEventDestroy(e);
    if (ctx->d2h_go) (void)hipEventDestroy(ctx->d2h_go);
    if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
    if (ctx->h2d_done[0]) (void)hipEventDestroy(ctx->h2d_done[0]);
    if (ctx->h2d_stream[0]) { (void)hipStreamSynchronize(ctx->h2d_stream[0]); (void)hipStreamDestroy(ctx->h2d_stream[0]); }
    if (ctx->k1n_ws) (void)hipFree(ctx->k1n_ws);
    if (ctx->k1n_desc_dev) (void)hipFree(ctx->k1n_desc_dev);
    if (ctx->k1n_pin) (void)hipHostFree(ctx->k1n_pin);
    if (ctx->k1n_ev) (void)hipEventDestroy(ctx->k1n_ev);
    if (ctx->icp_cnt) (void)hipFree(ctx->icp_cnt);
    if (ctx->icp_host) (void)hipHostFree(ctx->icp_host);
    if (ctx->status_host) (void)hipHostFree(ctx->status_host);
    if (ctx->status_mirror) (void)hipHostFree(ctx->status_mirror);
    if (ctx->heavy_hint) (void)hipHostFree(ctx->heavy_hint);
    prof_fold(ctx);
    for (auto &e : ctx->free_evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    delete ctx;
}

const char *pca_last_error(pca_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int pca_status(pca_ctx *ctx, void *stream, uint32_t *status_out)
{
    if (!ctx || !status_out) return -1;
    if (pca_k1_flush_pending(ctx)) return -1;               // a deferred K1 of this context comes first
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    PCA_CHECK(ctx, hipMemcpyAsync(ctx->status_host, ctx->ticket + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    PCA_CHECK(ctx, hipMemsetAsync(ctx->ticket + 1, 0, sizeof(uint32_t), s));
    PCA_CHECK(ctx, hipStreamSynchronize(s));
    *status_out = *ctx->status_host;
    for (int b = 0; b < PCA_STATUS_BITS; ++b) {             // the mirror may have run ahead of a status word cleared earlier
        if (__atomic_load_n(&ctx->status_mirror[b], __ATOMIC_RELAXED)) *status_out |= 1u << b;
        __atomic_store_n(&ctx->status_mirror[b], 0u, __ATOMIC_RELAXED);
    }
    return 0;
}

// The bits raised so far as the host sees them NOW: no stream operation, no wait.  A kernel's raise is visible here at the
// latest when that kernel has finished (any wait on its stream, an event behind it, a finished copy).  Does not clear:
// pca_status does.
int pca_status_peek(pca_ctx *ctx, uint32_t *status_out)
{
    if (!ctx || !status_out) return -1;
    uint32_t st = 0;
    for (int b = 0; b < PCA_STATUS_BITS; ++b)
        if (__atomic_load_n(&ctx->status_mirror[b], __ATOMIC_RELAXED)) st |= 1u << b;
    *status_out = st;
    return 0;
}

const uint32_t *pca_status_mirror(pca_ctx *ctx) { return ctx ? ctx->status_mirror : nullptr; }

}  // extern "C"
