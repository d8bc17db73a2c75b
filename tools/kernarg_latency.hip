// Dev tool (GPU box): what does a wave pay for reading its kernel arguments, and does it depend on HOW the kernel was launched?
//   hipcc --offload-arch=gfx950 -O3 tools/kernarg_latency.hip -o /tmp/kernarg_latency && /tmp/kernarg_latency
// 256 workgroups of one wave.  Each wave times (s_memtime) dependent scalar loads from its kernarg segment: the first touch of
// the line at +0x40, a second touch of it, the first touch of +0x100, and for comparison the first touch of a line of an
// ordinary device buffer.  The same kernel is launched (a) eagerly back to back and (b) as 50 nodes of a HIP graph.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { uint32_t w[96]; };                       // 384 bytes of arguments: six kernarg lines
typedef const __attribute__((address_space(4))) uint32_t* CU32;

__global__ void probe(Big big, const uint32_t* buf, unsigned long long* out, uint32_t* sink) {
    CU32 ka = (CU32)__builtin_amdgcn_kernarg_segment_ptr();
    CU32 b = (CU32)buf;
    unsigned long long t[6];
    uint32_t v = (uint32_t)blockIdx.x & 0u;
    t[0] = __builtin_amdgcn_s_memtime();
    v += ka[16 + (v & 0u)]; asm volatile("" : "+s"(v));
    t[1] = __builtin_amdgcn_s_memtime();
    v += ka[17 + (v & 0u)]; asm volatile("" : "+s"(v));
    t[2] = __builtin_amdgcn_s_memtime();
    v += ka[64 + (v & 0u)]; asm volatile("" : "+s"(v));
    t[3] = __builtin_amdgcn_s_memtime();
    v += b[64 + (v & 0u)]; asm volatile("" : "+s"(v));
    t[4] = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        for (int i = 0; i < 5; i++) out[blockIdx.x * 8 + i] = t[i];
        sink[blockIdx.x] = v;
    }
}

__global__ void empty_kernel(Big big, uint32_t* sink) { if (big.w[0] == 12345u) sink[0] = 1; }

static int report(const char* how, unsigned long long* out, int B) {
    std::vector<unsigned long long> h(B * 8);
    CHECK(hipMemcpy(h.data(), out, B * 8 * 8, hipMemcpyDeviceToHost));
    const char* names[4] = {"kernarg +0x40 first touch", "kernarg +0x44 same line", "kernarg +0x100 first touch", "device buffer line, first touch"};
    printf("%s\n", how);
    for (int k = 0; k < 4; k++) {
        std::vector<long> d(B);
        for (int i = 0; i < B; i++) d[i] = (long)(h[i * 8 + k + 1] - h[i * 8 + k]);
        std::sort(d.begin(), d.end());
        printf("   %-34s p10 %5ld  p50 %5ld  p90 %5ld  max %6ld cycles\n", names[k], d[B / 10], d[B / 2], d[B * 9 / 10], d[B - 1]);
    }
    return 0;
}

int main() {
    const int B = 256;
    uint32_t *buf, *sink;
    unsigned long long* out;
    CHECK(hipMalloc(&buf, 4096)); CHECK(hipMalloc(&sink, B * 4)); CHECK(hipMalloc(&out, B * 8 * 8));
    CHECK(hipMemset(buf, 0, 4096));
    Big big{};
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    for (int it = 0; it < 50; it++) probe<<<B, 64, 0, s>>>(big, buf, out, sink);
    CHECK(hipStreamSynchronize(s));
    if (report("eager launches (last of 50 back to back)", out, B)) return 1;
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int it = 0; it < 50; it++) probe<<<B, 64, 0, s>>>(big, buf, out, sink);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 3; r++) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    if (report("graph replay (last of 50 nodes, third replay)", out, B)) return 1;
    // the per-node floor: 200 EMPTY kernels of the step kernel's shape (256 workgroups x 1024 threads), as graph nodes and eagerly
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int shape = 0; shape < 2; shape++) {
        const int wg = shape == 0 ? 1024 : 256, blocks = shape == 0 ? 256 : 64;
        hipGraph_t g2; hipGraphExec_t ge2;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int it = 0; it < 200; it++) empty_kernel<<<blocks, wg, 0, s>>>(big, sink);
        CHECK(hipStreamEndCapture(s, &g2));
        CHECK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        for (int r = 0; r < 3; r++) CHECK(hipGraphLaunch(ge2, s));
        CHECK(hipStreamSynchronize(s));
        float ms = 0.f;
        CHECK(hipEventRecord(e0, s));
        for (int r = 0; r < 10; r++) CHECK(hipGraphLaunch(ge2, s));
        CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel %d x %d: %.2f us per graph node", blocks, wg, ms / 2000 * 1e3);
        CHECK(hipEventRecord(e0, s));
        for (int it = 0; it < 2000; it++) empty_kernel<<<blocks, wg, 0, s>>>(big, sink);
        CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1)); CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf(", %.2f us per eager launch\n", ms / 2000 * 1e3);
    }
    return 0;
}
