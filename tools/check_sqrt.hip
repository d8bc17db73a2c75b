// Dev tool (GPU box): exhaustive check of uavenv::sqrt_rn (device form) against the compiler's correctly rounded
// __builtin_sqrtf over every float in {0} U [2^-126, 2^127].   hipcc --offload-arch=gfx950 -ffp-contract=off -I<csrc> ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include "uavenv_noise.h"

__global__ void check(unsigned long long* bad, unsigned int* first_bad, unsigned int* by_exp) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i <= 0x7F000000ull; i += stride) {
        const unsigned int bits = i == 0 ? 0u : (unsigned int)(0x00800000ull + i - 1);
        if (bits > 0x7F000000u) break;
        const float x = __uint_as_float(bits);
        const float a = uavenv::sqrt_rn(x), b = __builtin_sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(b)) { local++; atomicMin(first_bad, bits); atomicAdd(&by_exp[bits >> 23], 1u); }
    }
    if (local) atomicAdd(bad, local);
}

int main() {
    unsigned long long* bad; unsigned int* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    unsigned int* by_exp; hipMalloc(&by_exp, 1024); hipMemset(by_exp, 0, 1024);
    unsigned long long z = 0; unsigned int f = 0xFFFFFFFFu;
    hipMemcpy(bad, &z, 8, hipMemcpyHostToDevice); hipMemcpy(first, &f, 4, hipMemcpyHostToDevice);
    check<<<4096, 256>>>(bad, first, by_exp);
    hipDeviceSynchronize();
    hipMemcpy(&z, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    printf("sqrt_rn vs __builtin_sqrtf: %llu mismatches (first bad bits 0x%08x)\n", z, f);
    unsigned int h[256]; hipMemcpy(h, by_exp, 1024, hipMemcpyDeviceToHost);
    int lo = 256, hi = -1;
    for (int e = 0; e < 256; e++) if (h[e]) { if (e < lo) lo = e; if (e > hi) hi = e; }
    if (hi >= 0) printf("mismatching biased exponents: %d .. %d (i.e. x in [2^%d, 2^%d))\n", lo, hi, lo - 127, hi - 126);
    // the environment takes roots of 0 and of values in [1e-8, 1e12] only: require a clean range 2^-64 .. 2^64
    unsigned long long in_domain = 0;
    for (int e = 127 - 64; e <= 127 + 64; e++) in_domain += h[e];
    printf("mismatches for x in {0} U [2^-64, 2^65): %llu\n", in_domain);
    return in_domain != 0;
}
