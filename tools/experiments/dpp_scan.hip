// dpp_scan.hip -- checks the DPP wave64 inclusive scan / reductions used by the BEV kernels against a serial loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../pc-accumulation-lib_amd/csrc/pca_wave.h"
__global__ void k(const uint32_t *in, uint32_t *scan, uint32_t *red, uint64_t mask_bits)
{
    const int lane = threadIdx.x & 63;
    const uint32_t v = in[blockIdx.x * 64 + lane];
    scan[blockIdx.x * 64 + lane] = wave_incl_scan_add(v);
    const bool in_grp = (mask_bits >> lane) & 1ull;
    const uint32_t s = wave_reduce_add(in_grp ? v : 0u);
    const uint32_t mn = wave_reduce_min(in_grp ? v : 0xffffffffu);
    const uint32_t mx = wave_reduce_max(in_grp ? v : 0u);
    scan[4096 + blockIdx.x * 64 + lane] = wave_bit_transpose32(v * 2654435761u);
    if (lane == 0) { red[blockIdx.x * 3 + 0] = s; red[blockIdx.x * 3 + 1] = mn; red[blockIdx.x * 3 + 2] = mx; }
}
int main()
{
    const int B = 64;
    uint32_t h[B * 64], hs[2 * B * 64], hr[B * 3];
    srand(1);
    for (int i = 0; i < B * 64; ++i) h[i] = (uint32_t)rand() % 100000;
    uint32_t *d, *ds, *dr;
    hipMalloc(&d, sizeof h); hipMalloc(&ds, sizeof hs); hipMalloc(&dr, sizeof hr);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    int bad = 0;
    for (uint64_t mask : {~0ull, 0x00000000ffff0000ull, 0x8000000000000001ull, 0x5555555555555555ull, 1ull << 37}) {
        hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, d, ds, dr, mask);
        hipMemcpy(hs, ds, sizeof hs, hipMemcpyDeviceToHost);
        hipMemcpy(hr, dr, sizeof hr, hipMemcpyDeviceToHost);
        for (int b = 0; b < B; ++b) {
            uint32_t run = 0, s = 0, mn = 0xffffffffu, mx = 0;
            for (int l = 0; l < 64; ++l) {
                run += h[b * 64 + l];
                if (hs[b * 64 + l] != run) ++bad;
                if ((mask >> l) & 1) { s += h[b * 64 + l]; mn = h[b * 64 + l] < mn ? h[b * 64 + l] : mn; mx = h[b * 64 + l] > mx ? h[b * 64 + l] : mx; }
            }
            if (hr[b * 3] != s || hr[b * 3 + 1] != mn || hr[b * 3 + 2] != mx) ++bad;
            for (int l = 0; l < 64; ++l) {                   // bit j of lane l = bit (l & 31) of lane (l & 32) + j
                uint32_t want = 0;
                for (int j = 0; j < 32; ++j) want |= (((h[b * 64 + (l & 32) + j] * 2654435761u) >> (l & 31)) & 1u) << j;
                if (hs[4096 + b * 64 + l] != want) ++bad;
            }
        }
    }
    printf("dpp scan/reduce mismatches: %d\n", bad);
    return bad != 0;
}
