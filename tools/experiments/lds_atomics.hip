// lds_atomics.hip -- cost of LDS atomics on gfx950 as a function of same-address conflicts within a wave.
// One workgroup of 1024 threads per CU-ish (grid 256); every lane does ITER atomics to address (lane % distinct)
// (+ a per-wave offset when `spread` so that different waves do not collide).  Reports cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 256
template <int MODE>   // 0 u32 add noret, 1 u32 add ret, 2 u64 add, 3 u64 min, 4 u32 add to 16-bit packed (same as 0)
__global__ __launch_bounds__(1024) void k(int distinct, int spread, unsigned long long *out)
{
    __shared__ unsigned long long s[4096];
    for (int i = threadIdx.x; i < 4096; i += 1024) s[i] = MODE == 3 ? ~0ull : 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int idx = (lane % distinct) + (spread ? wave * 64 : 0);
    unsigned acc = 0;
    const long long t0 = clock64();
#pragma unroll 8
    for (int i = 0; i < ITER; ++i) {
        const int a = (idx + (i & 3) * 1024) & 4095;
        if (MODE == 0) atomicAdd(reinterpret_cast<unsigned *>(s) + a, 1u);
        if (MODE == 1) acc += atomicAdd(reinterpret_cast<unsigned *>(s) + a, 1u);
        if (MODE == 2) atomicAdd(&s[a], 3ull);
        if (MODE == 3) atomicMin(&s[a], (unsigned long long)(i * 64 + lane));
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned long long)(t1 - t0);
    if (acc == 0xdeadbeef) out[1] = s[lane];
}
int main()
{
    unsigned long long *d, h;
    hipMalloc(&d, 64);
    const char *names[] = {"u32 add", "u32 add ret", "u64 add", "u64 min"};
    for (int mode = 0; mode < 4; ++mode)
        for (int spread = 0; spread < 2; ++spread)
            for (int distinct : {1, 2, 4, 8, 16, 32, 64}) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, 0, distinct, spread, d);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, 0, distinct, spread, d);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, 0, distinct, spread, d);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(1024), 0, 0, distinct, spread, d);
                    hipDeviceSynchronize();
                }
                hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
                // 16 waves x ITER wave-instructions share one LDS: clocks are 100 MHz ticks? report raw and per wave-instr
                printf("%-12s spread=%d distinct=%2d  ticks=%8llu  ticks/(wave-instr)=%.3f\n", names[mode], spread, distinct, h,
                       (double)h / (16.0 * ITER));
            }
    return 0;
}
