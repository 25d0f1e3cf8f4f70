// Issue rate of the vector instructions the rasteriser's level 1 is made of, relative to v_add_u32 (4 cycles per wave64
// on a 16-lane SIMD).  One wave per SIMD and four waves per SIMD; independent destination registers, no memory.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/experiments/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
#define BODY(asm_text)                                                                              \
    for (int it = 0; it < iters; ++it) {                                                            \
        REP8(asm volatile(asm_text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)     \
    }

template <int OP>
__global__ void rate(double *out, int iters, double b, double c)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    if (OP == 0) BODY("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3")
    if (OP == 1) BODY("v_mul_f64 %0, %4, %0\n v_mul_f64 %1, %4, %1\n v_mul_f64 %2, %4, %2\n v_mul_f64 %3, %4, %3")
    if (OP == 2) BODY("v_add_f64 %0, %4, %0\n v_add_f64 %1, %4, %1\n v_add_f64 %2, %4, %2\n v_add_f64 %3, %4, %3")
    if (OP == 3) BODY("v_min_f64 %0, %4, %0\n v_min_f64 %1, %4, %1\n v_min_f64 %2, %4, %2\n v_min_f64 %3, %4, %3")
    if (OP == 4) BODY("v_floor_f64 %0, %0\n v_floor_f64 %1, %1\n v_floor_f64 %2, %2\n v_floor_f64 %3, %3")
    if (OP == 5) BODY("v_cmp_gt_f64 vcc, %0, %4\n v_cmp_gt_f64 vcc, %1, %4\n v_cmp_gt_f64 vcc, %2, %4\n v_cmp_gt_f64 vcc, %3, %4")
    if (OP == 6) BODY("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3")
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}

#define BODY32(asm_text)                                                                            \
    for (int it = 0; it < iters; ++it) {                                                            \
        REP8(asm volatile(asm_text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));)     \
    }
template <int OP>
__global__ void rate32(unsigned *out, int iters, unsigned b, unsigned c)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    if (OP == 0) BODY32("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3")
    if (OP == 1) BODY32("v_mul_lo_u32 %0, %4, %0\n v_mul_lo_u32 %1, %4, %1\n v_mul_lo_u32 %2, %4, %2\n v_mul_lo_u32 %3, %4, %3")
    if (OP == 2) BODY32("v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %1, %4, %1, vcc\n v_cndmask_b32 %2, %4, %2, vcc\n v_cndmask_b32 %3, %4, %3, vcc")
    if (OP == 3) BODY32("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3")
    if (OP == 4) BODY32("v_and_or_b32 %0, %4, %5, %0\n v_and_or_b32 %1, %4, %5, %1\n v_and_or_b32 %2, %4, %5, %2\n v_and_or_b32 %3, %4, %5, %3")
    if (OP == 5) BODY32("v_mad_u32_u24 %0, %4, %5, %0\n v_mad_u32_u24 %1, %4, %5, %1\n v_mad_u32_u24 %2, %4, %5, %2\n v_mad_u32_u24 %3, %4, %5, %3")
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int OP>
__global__ void rate_cvt(unsigned *out, int iters, double b)
{
    double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));) }
        if (OP == 1) { REP8(asm volatile("v_cvt_f64_i32 %4, %0\n v_cvt_f64_i32 %5, %1\n v_cvt_f64_i32 %6, %2\n v_cvt_f64_i32 %7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 2) { REP8(asm volatile("v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (unsigned)(d0 + d1 + d2 + d3);
}

template <typename F>
static double time_ms(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const int iters = 4000, per_iter = 32;
    void *buf; hipMalloc(&buf, 256 * 4 * 1024 * 8);
    for (int wps = 1; wps <= 4; wps *= 4) {                 // waves per SIMD: blocks of 256 threads = 1 wave per SIMD
        const int grid = 256 * wps;
        double base = 0;
#define RUN32(OP, name) { double ms = time_ms([&] { hipLaunchKernelGGL(rate32<OP>, dim3(grid), dim3(256), 0, 0, (unsigned *)buf, iters, 3u, 5u); }); \
            if (OP == 0) base = ms; printf("wps %d %-16s %8.3f ms  %5.2f x v_add_u32  (%.2f cycles at 4/add)\n", wps, name, ms, ms / base, 4.0 * ms / base); }
#define RUN64(OP, name) { double ms = time_ms([&] { hipLaunchKernelGGL(rate<OP>, dim3(grid), dim3(256), 0, 0, (double *)buf, iters, 1.0000001, 1e-9); }); \
            printf("wps %d %-16s %8.3f ms  %5.2f x v_add_u32  (%.2f cycles at 4/add)\n", wps, name, ms, ms / base, 4.0 * ms / base); }
#define RUNC(OP, name) { double ms = time_ms([&] { hipLaunchKernelGGL(rate_cvt<OP>, dim3(grid), dim3(256), 0, 0, (unsigned *)buf, iters, 1.5); }); \
            printf("wps %d %-16s %8.3f ms  %5.2f x v_add_u32  (%.2f cycles at 4/add)\n", wps, name, ms, ms / base, 4.0 * ms / base); }
        RUN32(0, "v_add_u32") RUN32(1, "v_mul_lo_u32") RUN32(2, "v_cndmask_b32") RUN32(3, "v_fma_f32") RUN32(4, "v_and_or_b32") RUN32(5, "v_mad_u32_u24")
        RUN64(0, "v_fma_f64") RUN64(1, "v_mul_f64") RUN64(2, "v_add_f64") RUN64(3, "v_min_f64") RUN64(4, "v_floor_f64") RUN64(5, "v_cmp_gt_f64")
        RUN64(6, "v_rcp_f64")
        RUNC(0, "v_cvt_i32_f64") RUNC(1, "v_cvt_f64_i32") RUNC(2, "v_cvt_f64_f32")
        (void)per_iter;
    }
    return 0;
}
