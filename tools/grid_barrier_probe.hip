// Probe: what does a device-wide barrier inside one launch cost on MI355X (256 workgroups x 1024 threads, one per CU), against
// the ~5 us a dependent kernel launch costs?  Decides whether the DQN update's 12 dependent launches are worth fusing into one
// persistent launch.  Every spin is bounded: a barrier that does not complete sets `fail` and all workgroups run to the end.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_exp/grid_barrier_probe tools/grid_barrier_probe.hip && tools/_exp/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Sync { unsigned arrived; unsigned launches; unsigned fail; };   // launches: barriers completed by earlier launches

__device__ __forceinline__ bool grid_barrier(Sync* s, unsigned target) {
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __threadfence();                                          // release this workgroup's writes at device scope
        __hip_atomic_fetch_add(&s->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 0;
        for (unsigned spin = 0; spin < (1u << 18); ++spin) {
            unsigned v = __hip_atomic_load(&s->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // an acquire here would invalidate the XCD's L2 on every spin
            if ((int)(v - target) >= 0) { good = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!good) __hip_atomic_store(&s->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// phase p: workgroup b writes row b of `buf` (work_floats per workgroup), then after the barrier reads row (b+17) % G and checks
__global__ __launch_bounds__(1024) void probe(Sync* s, float* buf, unsigned* errors, int phases, int work_floats) {
    const unsigned G = gridDim.x, b = blockIdx.x;
    const unsigned launch = __hip_atomic_load(&s->launches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned bad = 0;
    for (int p = 0; p < phases; ++p) {
        for (int i = threadIdx.x; i < work_floats; i += blockDim.x) buf[(size_t)b * work_floats + i] = (float)(launch * 131 + p * 7 + b + i);
        if (p + 1 < phases || true) {
            if (!grid_barrier(s, (launch + p + 1) * G)) return;
        }
        const unsigned o = (b + 17) % G;
        for (int i = threadIdx.x; i < work_floats; i += blockDim.x)
            bad += buf[(size_t)o * work_floats + i] != (float)(launch * 131 + p * 7 + o + i);
        // second barrier so that nobody overwrites a row somebody is still checking
        // (folded into the count: two barriers per phase would double the cost measured; instead alternate halves of buf)
        buf += (size_t)G * work_floats * ((p & 1) ? -1 : 1);
    }
    if (bad) atomicAdd(errors, bad);
    if (b == 0 && threadIdx.x == 0) {
        // all workgroups have passed the last barrier's arrival; bump the launch count for the next launch's targets
        __hip_atomic_store(&s->launches, launch + phases, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(1024) void phase_kernel(float* buf, int work_floats, int p) {
    const unsigned b = blockIdx.x;
    for (int i = threadIdx.x; i < work_floats; i += blockDim.x) buf[(size_t)b * work_floats + i] = (float)(p * 7 + b + i);
}

int main() {
    const int G = 256;
    Sync* s; float* buf; unsigned* err;
    CK(hipMalloc(&s, sizeof(Sync))); CK(hipMemset(s, 0, sizeof(Sync)));
    CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    const int max_work = 1 << 16;
    CK(hipMalloc(&buf, (size_t)2 * G * max_work * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int work : {1024, 16384, 65536}) {
        for (int phases : {1, 2, 12, 24}) {
            const int reps = 200;
            for (int w = 0; w < 20; ++w) probe<<<G, 1024, 0, st>>>(s, buf, err, phases, work);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) probe<<<G, 1024, 0, st>>>(s, buf, err, phases, work);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Sync h; unsigned herr; CK(hipMemcpy(&h, s, sizeof h, hipMemcpyDeviceToHost)); CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            // the same phases as separate dependent launches
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) for (int p = 0; p < phases; ++p) phase_kernel<<<G, 1024, 0, st>>>(buf, work, p);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
            printf("work %6d floats/wg  phases %2d : one launch %7.2f us (%.2f us/phase)   separate launches %7.2f us (%.2f us/phase)   fail=%u mismatches=%u\n",
                   work, phases, ms * 1e3 / reps, ms * 1e3 / reps / phases, ms2 * 1e3 / reps, ms2 * 1e3 / reps / phases, h.fail, herr);
            if (h.fail) { printf("barrier timed out: stopping\n"); return 2; }
        }
    }
    return 0;
}
