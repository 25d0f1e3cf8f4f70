// f16_cvt.hip -- is f64 -> (round-to-odd) f32 -> (RNE, hardware) f16 bit-identical to the direct round-to-nearest-even
// routine f64_to_f16_bits of csrc/pca_common.h?  Sweeps random doubles over the whole f16 range incl. subnormals,
// exact halfway cases and values around them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../pc-accumulation-lib_amd/csrc/pca_common.h"
__global__ void k(const double *in, int n, unsigned long long *bad, double *example)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint16_t a = f64_to_f16_bits_reference(in[i]), b = f64_to_f16_bits(in[i]);
        if (a != b) { if (atomicAdd(bad, 1ull) == 0) *example = in[i]; }
    }
}
int main()
{
    const int n = 1 << 24;
    double *h = (double *)malloc(sizeof(double) * n);
    srand48(3);
    for (int i = 0; i < n; ++i) {
        const int kind = i & 7;
        uint16_t hb = (uint16_t)(lrand48() & 0x7fff);
        if ((hb & 0x7c00) == 0x7c00) hb &= 0x3ff;                     // no inf / nan seeds
        // decode the random half to double exactly
        const int e = (hb >> 10) & 31, m = hb & 1023;
        double v = e ? (1.0 + m / 1024.0) * pow(2.0, e - 15) : (m / 1024.0) * pow(2.0, -14);
        const double ulp = e ? pow(2.0, e - 25) : pow(2.0, -24);
        if (kind == 0) v = v;                                         // exactly representable
        else if (kind == 1) v += 0.5 * ulp;                           // exact halfway
        else if (kind == 2) v += 0.5 * ulp * (1.0 + 1e-15);           // just above halfway
        else if (kind == 3) v += 0.5 * ulp * (1.0 - 1e-15);           // just below halfway
        else if (kind == 4) v += drand48() * ulp;
        else if (kind == 5) v = drand48();                            // probabilities
        else if (kind == 6) v = (drand48() - 0.5) * 40.0;             // elevations
        else v = drand48() * 1e-6;                                    // deep subnormal / underflow
        h[i] = (lrand48() & 1) ? v : -v;
    }
    double *d, *ex; unsigned long long *bad, hbad = 0; double hex = 0;
    hipMalloc(&d, sizeof(double) * n); hipMalloc(&bad, 8); hipMalloc(&ex, 8);
    hipMemcpy(d, h, sizeof(double) * n, hipMemcpyHostToDevice); hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, n, bad, ex);
    hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hex, ex, 8, hipMemcpyDeviceToHost);
    printf("f16 conversion mismatches: %llu of %d (first: %.17g)\n", hbad, n, hex);
    return hbad != 0;
}
