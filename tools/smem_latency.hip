// Dev tool (GPU box): what does one scalar load cost on the critical path of a short kernel?
//   hipcc --offload-arch=gfx950 -O3 tools/smem_latency.hip -o /tmp/smem_latency && /tmp/smem_latency
// One wavefront per workgroup, 256 workgroups (one per CU).  Each wave times, with s_memtime, a chain of DEPENDENT
// scalar loads from a constants buffer: the first touch of a 64-byte line after the kernel launch, a second touch of
// the same line, a first touch of the next line, ... and the same for a buffer the PREVIOUS launch wrote (the way the
// step kernel's records and action words are produced).  Printed: median cycles per load over the workgroups.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef const __attribute__((address_space(4))) uint32_t* CU32;

__device__ __forceinline__ uint32_t sload(CU32 p, uint32_t dep) {
    // the address depends on the previous result (always +0 at run time): a dependent chain
    return p[dep & 0u];
}

__global__ void probe(const uint32_t* consts, const uint32_t* written, uint32_t* sink, unsigned long long* out, uint32_t* next_written) {
    CU32 c = (CU32)consts, w = (CU32)written;
    unsigned long long t[12];
    uint32_t v = (uint32_t)blockIdx.x & 0u;
    t[0] = __builtin_amdgcn_s_memtime();
    v += sload(c + 0, v);                  // first touch, line 0
    asm volatile("" : "+s"(v));
    t[1] = __builtin_amdgcn_s_memtime();
    v += sload(c + 1, v);                  // same line again
    asm volatile("" : "+s"(v));
    t[2] = __builtin_amdgcn_s_memtime();
    v += sload(c + 16, v);                 // next 64-byte line, first touch
    asm volatile("" : "+s"(v));
    t[3] = __builtin_amdgcn_s_memtime();
    v += sload(c + 32, v);                 // third line
    asm volatile("" : "+s"(v));
    t[4] = __builtin_amdgcn_s_memtime();
    v += sload(c + 33, v);                 // third line again
    asm volatile("" : "+s"(v));
    t[5] = __builtin_amdgcn_s_memtime();
    v += sload(w + 32 * blockIdx.x, v);    // a line the previous launch wrote (this workgroup's own 128-byte record)
    asm volatile("" : "+s"(v));
    t[6] = __builtin_amdgcn_s_memtime();
    v += sload(w + 32 * blockIdx.x + 16, v);   // its second line
    asm volatile("" : "+s"(v));
    t[7] = __builtin_amdgcn_s_memtime();
    // four INDEPENDENT first-touch loads of four more lines of the constants (issued back to back, one wait)
    uint32_t a0 = c[64], a1 = c[80], a2 = c[96], a3 = c[112];
    v += a0 + a1 + a2 + a3;
    asm volatile("" : "+s"(v));
    t[8] = __builtin_amdgcn_s_memtime();
    // vector load of a line the previous launch wrote (sensor row): L2 hit latency through the vector path
    uint32_t x = written[32 * blockIdx.x + 8 * 1024 * 32 + (threadIdx.x & 63)];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x));
    t[9] = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        for (int i = 0; i < 10; i++) out[blockIdx.x * 16 + i] = t[i];
        sink[blockIdx.x] = v + x;
    }
    // produce next launch's "written" buffer (both regions)
    next_written[32 * blockIdx.x + (threadIdx.x & 31)] = v + threadIdx.x;
    next_written[32 * blockIdx.x + 8 * 1024 * 32 + (threadIdx.x & 63)] = x + 1;
}

int main() {
    const int B = 256;
    uint32_t *consts, *wa, *wb, *sink;
    unsigned long long* out;
    const size_t wn = 8 * 1024 * 32 + 32 * B + 64;
    CHECK(hipMalloc(&consts, 4096)); CHECK(hipMalloc(&wa, wn * 4)); CHECK(hipMalloc(&wb, wn * 4));
    CHECK(hipMalloc(&sink, B * 4)); CHECK(hipMalloc(&out, B * 16 * 8));
    CHECK(hipMemset(consts, 0, 4096)); CHECK(hipMemset(wa, 0, wn * 4)); CHECK(hipMemset(wb, 0, wn * 4));
    for (int it = 0; it < 50; it++) {
        probe<<<B, 64>>>(consts, it & 1 ? wb : wa, sink, out, it & 1 ? wa : wb);
    }
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(B * 16);
    CHECK(hipMemcpy(h.data(), out, B * 16 * 8, hipMemcpyDeviceToHost));
    const char* names[9] = {"consts line 0, first touch after launch", "consts line 0, second touch", "consts line 1, first touch",
                            "consts line 2, first touch", "consts line 2, second touch", "own record line 0 (written by previous launch)",
                            "own record line 1", "four independent first-touch lines, one wait", "vector load of a row the previous launch wrote"};
    for (int i = 0; i < 9; i++) {
        std::vector<long long> d;
        for (int b = 0; b < B; b++) d.push_back((long long)(h[b * 16 + i + 1] - h[b * 16 + i]));
        std::sort(d.begin(), d.end());
        printf("%-56s cycles: p10 %5lld  p50 %5lld  p90 %5lld\n", names[i], d[B / 10], d[B / 2], d[B * 9 / 10]);
    }
    printf("(each figure includes one s_memtime round trip of its own, the second-touch rows are that floor)\n");
    return 0;
}
