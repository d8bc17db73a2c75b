// GPU box: the operand / result lane maps of v_mfma_f32_4x4x1_16b_f32 (16 blocks of a 4x4 outer product), checked with exact
// integer data: D[reg r][lane l] must equal A[lane 4 * (l / 4) + r] * B[lane l].     hipcc --offload-arch=gfx950 -o /tmp/c tools/check_mfma4x4.hip && /tmp/c
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l] + 100.f, b[l], c, 0, 0, 0);      // accumulates: (2a + 100) * b
    for (int r = 0; r < 4; r++) d[r * 64 + l] = c[r];
}
int main() {
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int l = 0; l < 64; l++) { ha[l] = (float)(l + 1); hb[l] = (float)(3 * l + 7); }
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
        const float want = (2.f * ha[4 * (l / 4) + r] + 100.f) * hb[l];
        if (hd[r * 64 + l] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, hd[r * 64 + l], want); bad++; }
    }
    printf("mfma_f32_4x4x1 lane map D[r][l] = A[4*(l/4)+r] * B[l]: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    return bad != 0;
}
