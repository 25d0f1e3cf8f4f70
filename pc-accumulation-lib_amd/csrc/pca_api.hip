// pca_api.hip -- context, error reporting and look-back workspace management (C ABI, include/pca.h).
#include <stdio.h>
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
    PCA_CREATE_STEP(hipHostMalloc(&ctx->heavy_hint, sizeof(uint32_t), hipHostMallocMapped));
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0) ctx->n_cu = n_cu;
    *ctx->heavy_hint = 1;            // first call: assume heavy tiles exist
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
