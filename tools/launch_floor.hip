// Dev tool (GPU box): the floor under a launch -- back-to-back launches of kernels that do (almost) nothing, by grid shape.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(unsigned* p) { if (p == (unsigned*)1) p[0] = 1; }
__global__ void touch_kernel(unsigned* p) {           // one dword load + one dword store per wave, like a prologue that exits
    unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) p[w + 65536] = p[w] + 1;
}
template <typename F> float time_us(F f, int n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < n; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / n;
}
int main() {
    unsigned* p; hipMalloc(&p, 1 << 20); hipMemset(p, 0, 1 << 20);
    int shapes[][2] = {{256, 1024}, {512, 512}, {1024, 256}, {4096, 64}, {64, 256}, {256, 256}, {64, 1024}};
    for (auto& s : shapes) {
        float e = time_us([&] { empty_kernel<<<s[0], s[1]>>>(p); }, 2000);
        float t = time_us([&] { touch_kernel<<<s[0], s[1]>>>(p); }, 2000);
        printf("grid %5d x block %4d (%5d waves): empty %.2f us   load+store per wave %.2f us\n", s[0], s[1], s[0] * s[1] / 64, e, t);
    }
    return 0;
}
