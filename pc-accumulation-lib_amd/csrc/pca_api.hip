// pca_api.hip -- context, error reporting and look-back workspace management (C ABI, include/pca.h).
#include "pca_common.h"

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

extern "C" {

int pca_version(void) { return PCA_VERSION; }

int pca_ctx_create(int device, pca_ctx **out)
{
    if (!out) return -1;
    *out = nullptr;
    pca_ctx *ctx = new pca_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&ctx->ticket, 2 * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(ctx->ticket, 0, 2 * sizeof(uint32_t)) != hipSuccess ||
        hipHostMalloc(&ctx->status_host, sizeof(uint32_t)) != hipSuccess) {
        delete ctx;
        return -1;
    }
    *out = ctx;
    return 0;
}

void pca_ctx_destroy(pca_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->tile_state) (void)hipFree(ctx->tile_state);
    if (ctx->ticket) (void)hipFree(ctx->ticket);
    if (ctx->frames_dev) (void)hipFree(ctx->frames_dev);
    if (ctx->status_host) (void)hipHostFree(ctx->status_host);
    delete ctx;
}

const char *pca_last_error(pca_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int pca_status(pca_ctx *ctx, void *stream, uint32_t *status_out)
{
    if (!ctx || !status_out) return -1;
    hipStream_t s = (hipStream_t)stream;
    PCA_CHECK(ctx, hipSetDevice(ctx->device));
    PCA_CHECK(ctx, hipMemcpyAsync(ctx->status_host, ctx->ticket + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    PCA_CHECK(ctx, hipMemsetAsync(ctx->ticket + 1, 0, sizeof(uint32_t), s));
    PCA_CHECK(ctx, hipStreamSynchronize(s));
    *status_out = *ctx->status_host;
    return 0;
}

}  // extern "C"
