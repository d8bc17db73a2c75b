// uavenv_learner.hip -- the DQN update of the reference's MLP policy as a handful of HIP launches.
//
// What it replaces: one `DQN.train()` gradient step of stable-baselines3 as the reference configures it
// (agents/dqn/dqn.py:1077-1099: MlpPolicy net_arch [512, 512, 256], batch_size 256, gamma 0.99, Adam, max_grad_norm 10) --
// forward of the online and the target network, smooth-L1 TD loss, backward, gradient clipping, Adam.  In PyTorch that is
// ~60 launches of which a dozen are float32 library GEMMs with a batch of 256 rows: 14-29 us each, because a 256 x 512 output
// is 16 macro tiles for 256 CUs (profiles/r02p_learner_kernel_stats.md); 337 us per update when the GEMM choices are tuned.
// The arithmetic is 1.45 GFLOP -- 9 us at the f32 MFMA rate -- so the update is latency- and occupancy-bound, not compute-bound.
//
// Here: ONE small-batch GEMM kernel on v_mfma_f32_16x16x4_f32 for all three products of a Linear layer
//     forward   Y[b][n]  += sum_k X[b][k] W[n][k]      (both operands contiguous along k)
//     input     dX[b][k] += sum_n dZ[b][n] W[n][k]
//     weight    dW[n][k] += sum_b dZ[b][n] X[b][k]     (+ bias gradient = row sums of the transposed operand)
// a workgroup of 16 wavefronts owns a 16 x 64 (or 16 x 32) output tile and splits K among its wavefronts (128-640 workgroups per
// product: every CU busy, four wavefronts per SIMD hiding each other's loads), fragments come straight from global memory
// (everything is L2 resident: 2.8 MB of weights, < 1 MB of activations), the partial tiles meet in LDS and are stored once.
// ReLU is never a pass of its own: a layer stores its PRE-activation and the
// consumer applies max(., 0) -- or, backward, the (z > 0) mask -- while loading the operand.  Around it: the TD loss with its
// gradient in one workgroup, the global gradient norm, and clip + Adam over ONE flat parameter buffer.
// fp32 throughout; parity is against torch autograd + torch.optim.Adam on the same batch (tests/test_gpu_mlp_update.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uavenv.h"
#include "uavenv_noise.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef UavGemm GemmArgs;     // include/uavenv.h: operands, strides, flags of one product

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// four consecutive-k elements of an operand: `p` points at element (row, k); contiguous along k (KC) or strided by `sk`
__device__ __forceinline__ f32x4 load_k4(bool KC, const float* p, int64_t sk, int k, int K, bool row_ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!row_ok) return v;
    if (KC) {
        if (k + 3 < K && aligned16(p)) return *reinterpret_cast<const f32x4*>(p);
        if (k < K) v.x = p[0];
        if (k + 1 < K) v.y = p[1];
        if (k + 2 < K) v.z = p[2];
        if (k + 3 < K) v.w = p[3];
    } else {
        if (k < K) v.x = p[0];
        if (k + 1 < K) v.y = p[sk];
        if (k + 2 < K) v.z = p[2 * sk];
        if (k + 3 < K) v.w = p[3 * sk];
    }
    return v;
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); return v; }

// C[M x N] = A[M x K] . B[K x N] (+ bias); M, N and K are arbitrary (guards on the tails).  One WORKGROUP of 16 wavefronts owns
// one 16 x (16 * NSUB) output tile; wavefront w takes the k blocks w, w + 16, ... (split-K inside the workgroup), the 16 partial
// tiles meet in LDS and every thread adds up one element and stores it: no atomics (device-scope float atomics on 8 XCDs were
// what the first form of this kernel spent its 13 us on), no zeroed outputs.  MFMA operand map: lane l = (row / column
// r = l & 15, k group g = l >> 4) holds k = 16 * kb + 4 * g + t in step t -- any assignment works as long as A and B share it.
// Two independent products may share a launch (the online and the target network's forward of a layer; a layer's weight gradient and
// the gradient w.r.t. its input): workgroups [0, wgs0) belong to g0, the rest to g1 -- every launch saved is ~5 us of dispatch and
// cold-cache latency that a 2 us product cannot hide.
template <int NSUB>
__global__ __launch_bounds__(1024) void gemm_f32_kernel(GemmArgs g0, GemmArgs g1, int wgs0, int dual_mask) {
    constexpr int kCols = 16 * NSUB, kTile = 16 * kCols;
    __shared__ float red[16][kTile + 16];
    const bool second = (int)blockIdx.x >= wgs0;
    const GemmArgs& g = second ? g1 : g0;
    const int wg = second ? (int)blockIdx.x - wgs0 : (int)blockIdx.x;
    // "dual": the workgroup owns TWO tiles, 8 wavefronts each (a short reduction gives 16 wavefronts one k block apiece and
    // twice the workgroups; past 256 workgroups a launch runs in two rounds -- 6.0 us instead of 3.6 before any product)
    const bool dual = (dual_mask >> (second ? 1 : 0)) & 1;
    const bool AK = g.a_sk == 1, BK = g.b_sk == 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mt = (g.M + 15) >> 4, ntiles = mt * ((g.N + kCols - 1) / kCols);
    const int tile = dual ? 2 * wg + (wave >> 3) : wg;
    const bool live = tile < ntiles;                          // an odd tile count leaves the last workgroup's second half idle
    const int im = tile % mt, in = tile / mt;
    const int kblocks = (g.K + 15) >> 4;
    const int r = lane & 15, gq = lane >> 4;
    const int m = live ? im * 16 + r : g.M;
    const bool a_relu = (g.flags & UAVENV_GEMM_A_RELU) != 0, a_mask = (g.flags & UAVENV_GEMM_A_MASK) != 0;
    const bool b_relu = (g.flags & UAVENV_GEMM_B_RELU) != 0;
    const bool want_rowsum = (g.flags & UAVENV_GEMM_ROWSUM) != 0 && in == 0;
    f32x4 acc[NSUB];
#pragma unroll
    for (int j = 0; j < NSUB; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rowsum = 0.f;
    // Interior k blocks of interior tiles (every row, column and k in range, 16-byte aligned where four floats are read at once)
    // take a straight-line path: the A fragment, its mask and all NSUB B fragments are requested before the first wait -- the
    // guarded form (tails, odd strides) branches per element and the compiler waits after every fragment.
    const bool rows_in = live && im * 16 + 16 <= g.M && in * kCols + kCols <= g.N;
    const bool a_vec_ok = !AK || (aligned16(g.A) && (g.a_sm & 3) == 0 && (!a_mask || aligned16(g.a_mask)));
    const bool b_vec_ok = !BK || (aligned16(g.B) && (g.b_sn & 3) == 0);
    const bool fast_tile = rows_in && a_vec_ok && b_vec_ok;
    auto fetch4 = [](bool KC, const float* p, int64_t sk) -> f32x4 {
        if (KC) return *reinterpret_cast<const f32x4*>(p);
        return f32x4{p[0], p[sk], p[2 * sk], p[3 * sk]};
    };
    struct Frag { f32x4 a, z, b[NSUB]; };
    auto fetch = [&](Frag& f, int kb) {                       // all requests of a k block, no waits in between
        const int k = kb * 16 + 4 * gq;
        const int64_t a_off = (int64_t)m * g.a_sm + (int64_t)k * g.a_sk;
        f.a = fetch4(AK, g.A + a_off, g.a_sk);
        f.z = a_mask ? fetch4(AK, g.a_mask + a_off, g.a_sk) : f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int j = 0; j < NSUB; j++)
            f.b[j] = fetch4(BK, g.B + (int64_t)k * g.b_sk + (int64_t)(in * kCols + 16 * j + r) * g.b_sn, g.b_sk);
    };
    auto multiply = [&](Frag& f) {
        f32x4 a = f.a;
        if (a_relu) a = relu4(a);
        a.x = f.z.x > 0.f ? a.x : 0.f; a.y = f.z.y > 0.f ? a.y : 0.f; a.z = f.z.z > 0.f ? a.z : 0.f; a.w = f.z.w > 0.f ? a.w : 0.f;
        if (want_rowsum) rowsum += (a.x + a.y) + (a.z + a.w);
#pragma unroll
        for (int j = 0; j < NSUB; j++) {
            f32x4 bj = f.b[j];
            if (b_relu) bj = relu4(bj);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bj.x, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bj.y, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bj.z, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bj.w, acc[j], 0, 0, 0);
        }
    };
    // (Requesting the NEXT k block before this one's products -- software pipelining, two fragment sets in registers -- was
    //  measured twice, before and after this path existed: 112 vs 108 us and 58.3 vs 52.9 us over the update's eight launches.
    //  Slower both times; the plain request-all, wait, multiply order stays.)
    Frag f;
    for (int kb = dual ? (wave & 7) : wave; kb < kblocks; kb += dual ? 8 : 16) {
        if (fast_tile && kb * 16 + 16 <= g.K) {               // wave-uniform
            fetch(f, kb);
            multiply(f);
            continue;
        }
        const int k = kb * 16 + 4 * gq;
        const int64_t a_off = (int64_t)m * g.a_sm + (int64_t)k * g.a_sk;
        f32x4 a = load_k4(AK, g.A + a_off, g.a_sk, k, g.K, m < g.M);
        if (a_relu) a = relu4(a);
        if (a_mask) {
            const f32x4 z = load_k4(AK, g.a_mask + a_off, g.a_sk, k, g.K, m < g.M);
            a.x = z.x > 0.f ? a.x : 0.f; a.y = z.y > 0.f ? a.y : 0.f; a.z = z.z > 0.f ? a.z : 0.f; a.w = z.w > 0.f ? a.w : 0.f;
        }
        if (want_rowsum) rowsum += (a.x + a.y) + (a.z + a.w);
#pragma unroll
        for (int j = 0; j < NSUB; j++) {
            const int n = in * kCols + 16 * j + r;
            f32x4 b = load_k4(BK, g.B + (int64_t)k * g.b_sk + (int64_t)n * g.b_sn, g.b_sk, k, g.K, live && n < g.N);
            if (b_relu) b = relu4(b);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[j], 0, 0, 0);
        }
    }
    // partial tile -> LDS (lane: column 16 j + r, rows 4 gq + i)
#pragma unroll
    for (int j = 0; j < NSUB; j++) {
        float* p = &red[wave][(4 * gq) * kCols + 16 * j + r];
        p[0] = acc[j].x; p[kCols] = acc[j].y; p[2 * kCols] = acc[j].z; p[3 * kCols] = acc[j].w;
    }
    if (want_rowsum) {                                        // the four k groups of a row sit 16 lanes apart
        rowsum += __shfl_xor(rowsum, 16);
        rowsum += __shfl_xor(rowsum, 32);
        if (gq == 0) red[wave][kTile + r] = rowsum;
    }
    __syncthreads();
    // UAVENV_GEMM_SUMSQ: the squares of what this tile writes, summed per 32-column group (a 64-column tile is two groups: lanes
    // 0..31 | 32..63 of every wavefront; a 32-column tile one) -- the gradient norm's partial sums without a launch of their own
    const bool want_sq = (g.flags & UAVENV_GEMM_SUMSQ) != 0;
    __shared__ float sqr[16][2];
    for (int h = 0; h < (dual ? 2 : 1); h++) {
        const int t = dual ? 2 * wg + h : wg, w0 = 8 * h, w1 = dual ? w0 + 8 : 16;
        if (t >= ntiles) break;
        const int tm = t % mt, tn = t / mt;
        float sq = 0.f;
        for (int e = threadIdx.x; e < kTile; e += 1024) {
            float v = 0.f;
            if (dual) {
#pragma unroll
                for (int w = 0; w < 8; w++) v += red[w0 + w][e];
            } else {
#pragma unroll
                for (int w = 0; w < 16; w++) v += red[w][e];
            }
            const int mm = tm * 16 + e / kCols, n = tn * kCols + e % kCols;
            if (mm < g.M && n < g.N) {
                v += (g.flags & UAVENV_GEMM_BIAS) ? g.bias[n] : 0.f;
                g.C[(int64_t)mm * g.ldc + n] = v;
                sq += v * v;
            }
        }
        if ((g.flags & UAVENV_GEMM_ROWSUM) && tn == 0 && threadIdx.x < 16) {
            float v = 0.f;
            for (int w = w0; w < w1; w++) v += red[w][kTile + threadIdx.x];
            if (tm * 16 + (int)threadIdx.x < g.M) { g.row_sum[tm * 16 + threadIdx.x] = v; sq += v * v; }
        }
        if (want_sq) {                                        // (uniform over the workgroup)
            for (int o = 16; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
            if ((lane & 31) == 0) sqr[wave][lane >> 5] = sq;
            __syncthreads();
            if (threadIdx.x < 2) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 16; w++) a += sqr[w][threadIdx.x];
                const int groups = (g.N + 31) >> 5;
                if (NSUB == 4) {
                    const int gi = 2 * tn + (int)threadIdx.x;
                    if (gi < groups) g.sumsq[tm * groups + gi] = a;
                } else {                                      // 32-column tile: one group, both halves of the wavefronts belong to it
                    a += __shfl_xor(a, 1);
                    if (threadIdx.x == 0) g.sumsq[tm * groups + tn] = a;
                }
            }
            if (h == 0 && dual) __syncthreads();              // (the second tile's sums reuse sqr)
        }
    }
}

// SB3 DQN.train's loss on one batch and its gradient w.r.t. the online Q-values, one workgroup:
//   target = reward_scale * r + gamma * max_a' Q_target(s', a')      (no (1 - done): every episode end is a truncation)
//   loss = sum_b w_b * smooth_l1(Q(s_b, a_b) - target_b) / max(sum_b w_b, 1),  w = valid
//   dq[b][a_b] = w_b / max(sum w, 1) * clamp(Q - target, -1, 1), zero elsewhere.
// Also advances the optimiser's step count and writes Adam's two bias corrections for it.
__global__ __launch_bounds__(1024) void td_loss_kernel(const float* __restrict__ q, const float* __restrict__ q_next, const int64_t* __restrict__ action,
                                                     const float* __restrict__ reward, const uint8_t* __restrict__ valid, int32_t batch,
                                                     int32_t n_actions, float gamma, float reward_scale, float beta1, float beta2,
                                                     float* __restrict__ dq, float* __restrict__ scalars) {
    __shared__ float s_w[16], s_l[16];
    const int b = threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float w = 0.f, d = 0.f, l = 0.f;
    int a = 0;
    if (b < batch) {
        float mx = q_next[(size_t)b * n_actions];
        for (int j = 1; j < n_actions; j++) mx = fmaxf(mx, q_next[(size_t)b * n_actions + j]);
        a = (int)action[b];
        d = q[(size_t)b * n_actions + a] - (reward_scale * reward[b] + gamma * mx);
        w = valid[b] ? 1.f : 0.f;
        const float ad = fabsf(d);
        l = w * (ad < 1.f ? 0.5f * d * d : ad - 0.5f);
    }
    float ws = w, ls = l;
    for (int o = 32; o > 0; o >>= 1) { ws += __shfl_xor(ws, o); ls += __shfl_xor(ls, o); }
    if (lane == 0) { s_w[wv] = ws; s_l[wv] = ls; }
    __syncthreads();
    float wt = 0.f, lt = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) { wt += s_w[i]; lt += s_l[i]; }
    const float inv = 1.f / fmaxf(wt, 1.f);
    if (b < batch) {
        for (int j = 0; j < n_actions; j++) dq[(size_t)b * n_actions + j] = 0.f;
        dq[(size_t)b * n_actions + a] = w * inv * fminf(fmaxf(d, -1.f), 1.f);
    }
    if (threadIdx.x == 0) {
        scalars[UAVENV_UPD_LOSS] = lt * inv;
        const float t = scalars[UAVENV_UPD_STEP] + 1.f;
        scalars[UAVENV_UPD_STEP] = t;
        scalars[UAVENV_UPD_BC1] = 1.f - powf(beta1, t);
        scalars[UAVENV_UPD_BC2] = 1.f - powf(beta2, t);
    }
}

// gradient norm in two steps without atomics: every workgroup leaves the sum of squares of its share in partial[blockIdx.x] ...
constexpr int kNormBlocks = 256;
__global__ __launch_bounds__(256) void sum_squares_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float sw[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += g[i] * g[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

// ... and every workgroup of the optimiser kernel adds the kNormBlocks partials up again (1 KB from L2).
// torch.nn.utils.clip_grad_norm_(max_norm) followed by torch.optim.Adam.step() (no weight decay, no amsgrad) over flat buffers
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2,
                                                      int64_t n, float* __restrict__ scalars, const float* __restrict__ partial, int n_partials,
                                                      float max_norm, float beta1, float beta2, float eps) {
    __shared__ float sw[4];
    float s = 0.f;                                             // blockDim.x == 256
    for (int i0 = threadIdx.x; i0 < n_partials; i0 += 8 * 256) {   // eight loads in flight (a plain loop waits for each in turn: +2 us)
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int i = i0 + 256 * u; t[u] = partial[i < n_partials ? i : i0]; }
#pragma unroll
        for (int u = 0; u < 8; u++) s += (i0 + 256 * u < n_partials) ? t[u] : 0.f;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    const float norm2 = (sw[0] + sw[1]) + (sw[2] + sw[3]);
    if (blockIdx.x == 0 && threadIdx.x == 0) scalars[UAVENV_UPD_NORM2] = norm2;
    const float coef = fminf(max_norm / (sqrtf(norm2) + 1e-6f), 1.0f);
    const float lr = scalars[UAVENV_UPD_LR], bc1 = scalars[UAVENV_UPD_BC1], bc2s = sqrtf(scalars[UAVENV_UPD_BC2]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        const float a = beta1 * m1[i] + (1.f - beta1) * gi;
        const float v = beta2 * m2[i] + (1.f - beta2) * gi * gi;
        m1[i] = a; m2[i] = v;
        p[i] -= (lr / bc1) * (a / (sqrtf(v) / bc2s + eps));
    }
}

// SB3's epsilon-greedy action selection for a vector of environments in one launch (argmax, the coin, the random action):
// ONE workgroup walks all environments, so that it can also advance the draw counter it keys its Philox words with -- a captured
// launch then draws fresh numbers at every replay without a second kernel.  shared_coin: SB3's predict() flips ONE coin for the
// whole vector (the artefact DQNLearner(shared_exploration_coin=True) reproduces); otherwise every environment flips its own.
__global__ __launch_bounds__(1024) void epsilon_greedy_kernel(const float* __restrict__ q, int32_t n_envs, int32_t n_actions,
                                                            const float* __restrict__ eps_dev, float* __restrict__ counter_dev, uint64_t seed,
                                                            int32_t shared_coin, int32_t* __restrict__ actions_out) {
    const uint64_t cnt = (uint64_t)counter_dev[0];
    const float eps = eps_dev[0];
    const uavenv::Words4 w0 = uavenv::philox4x32<UAVENV_PHILOX_ROUNDS>(0xFFFFFFFFu, (uint32_t)cnt, (uint32_t)(cnt >> 32), 0x45505347u /* "EPSG" */,
                                                                      (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int e = threadIdx.x; e < n_envs; e += blockDim.x) {
        const float* row = q + (size_t)e * n_actions;
        int best = 0; float bv = row[0];
        for (int a = 1; a < n_actions; a++) { const float v = row[a]; if (v > bv) { bv = v; best = a; } }      // first maximum, like torch.argmax
        const uavenv::Words4 w = uavenv::philox4x32<UAVENV_PHILOX_ROUNDS>((uint32_t)e, (uint32_t)cnt, (uint32_t)(cnt >> 32), 0x45505347u,
                                                                         (uint32_t)seed, (uint32_t)(seed >> 32));
        const float coin = uavenv::u24(shared_coin ? w0.w0 : w.w0);
        const int rnd = (int)uavenv::mulhi32(w.w1, (uint32_t)n_actions);
        actions_out[e] = coin < eps ? rnd : best;
    }
    __syncthreads();
    if (threadIdx.x == 0) counter_dev[0] = (float)(cnt + 1);
}

// The LAST layer of the acting forward (a [n_envs x K] . [K x n_actions] product with n_actions = 5: 5 us as a library GEMM) and the
// epsilon-greedy selection above in ONE launch: a wavefront takes 16 environments, four lanes per environment split K in 16-byte
// pieces (a row's four lanes read 64 contiguous bytes per step), the weights come from LDS, the partial sums meet through two
// lane swaps, and the environment's first lane takes the argmax, flips the coin and writes the action -- the draws are the
// epsilon_greedy_kernel's (same Philox words per environment and call).  The draw counter is advanced by whichever workgroup
// finishes last (a ticket counter: every workgroup has read the counter long before the last one takes its ticket).
constexpr int kSelMaxActions = 8;
// NA: compile-time bound of n_actions (5 for this environment; 8 otherwise -- rows past n_actions repeat the last action's weights and
// are ignored).  Every load is unconditional with a clamped address: a load inside a branch makes the compiler wait for it at the
// branch's end, and the 16 steps over a 256-wide row then cost 16 trips to memory instead of one (9 us instead of 4).
template <int NA>
__global__ __launch_bounds__(256) void q_head_select_kernel(const float* __restrict__ h, const float* __restrict__ W, const float* __restrict__ b,
                                                          int32_t n_envs, int32_t K, int32_t n_actions, const float* __restrict__ eps_dev,
                                                          float* __restrict__ counter_dev, int32_t* __restrict__ ticket_dev, uint64_t seed,
                                                          int32_t shared_coin, int32_t* __restrict__ actions_out, float* __restrict__ q_out) {
    extern __shared__ float wl[];                              // [n_actions][K], K a multiple of 4 here (the launcher checks)
    const int total = n_actions * K;
    {
        f32x4 t[4];                                            // 4 x 256 threads x 4 floats = 4096 floats per pass
        for (int base = 0; base < total; base += 4096) {
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = base + 4 * ((int)threadIdx.x + 256 * u); t[u] = *reinterpret_cast<const f32x4*>(W + (i < total ? i : total - 4)); }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = base + 4 * ((int)threadIdx.x + 256 * u); if (i < total) *reinterpret_cast<f32x4*>(&wl[i]) = t[u]; }
        }
    }
    const uint64_t cnt = (uint64_t)counter_dev[0];
    const float eps = eps_dev[0];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int part = lane & 3;
    const int e = (blockIdx.x * 4 + wave) * 16 + (lane >> 2);
    const bool live = e < n_envs;
    const float* row = h + (size_t)(live ? e : n_envs - 1) * K;
    float acc[NA];
#pragma unroll
    for (int a = 0; a < NA; a++) acc[a] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 256) {                      // 16 steps of 16 floats per pass
        f32x4 x[16];
#pragma unroll
        for (int i = 0; i < 16; i++) { const int k = k0 + 16 * i + 4 * part; x[i] = *reinterpret_cast<const f32x4*>(row + (k < K ? k : K - 4)); }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int k = k0 + 16 * i + 4 * part;
            const bool in = k < K;
            const int kc = in ? k : K - 4;
#pragma unroll
            for (int a = 0; a < NA; a++) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(&wl[(a < n_actions ? a : n_actions - 1) * K + kc]);
                const float p = (x[i].x * w.x + x[i].y * w.y) + (x[i].z * w.z + x[i].w * w.w);
                acc[a] += in ? p : 0.f;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < NA; a++) {
        acc[a] += __shfl_xor(acc[a], 1);
        acc[a] += __shfl_xor(acc[a], 2);
    }
    if (live && part == 0) {
        int best = 0; float bv = 0.f;
#pragma unroll
        for (int a = 0; a < NA; a++) {
            if (a < n_actions) {
                const float v = acc[a] + b[a];
                if (q_out != nullptr) q_out[(size_t)e * n_actions + a] = v;
                if (a == 0 || v > bv) { bv = v; best = a; }      // first maximum, like torch.argmax
            }
        }
        const uavenv::Words4 w = uavenv::philox4x32<UAVENV_PHILOX_ROUNDS>((uint32_t)e, (uint32_t)cnt, (uint32_t)(cnt >> 32), 0x45505347u,
                                                                         (uint32_t)seed, (uint32_t)(seed >> 32));
        uint32_t coin_word = w.w0;
        if (shared_coin)
            coin_word = uavenv::philox4x32<UAVENV_PHILOX_ROUNDS>(0xFFFFFFFFu, (uint32_t)cnt, (uint32_t)(cnt >> 32), 0x45505347u,
                                                               (uint32_t)seed, (uint32_t)(seed >> 32)).w0;
        const int rnd = (int)uavenv::mulhi32(w.w1, (uint32_t)n_actions);
        actions_out[e] = uavenv::u24(coin_word) < eps ? rnd : best;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (relaxed: nothing but the ticket travels between workgroups -- this workgroup's read of the counter completed long ago,
        //  its value went into every draw above; an acquire / release here would write back and invalidate the XCD's L2 per workgroup)
        const int t = __hip_atomic_fetch_add(ticket_dev, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (int)gridDim.x - 1) {                         // the last workgroup of the launch
            __hip_atomic_store(ticket_dev, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            counter_dev[0] = (float)(cnt + 1);
        }
    }
}

// ---- the attention extractor's core for training: one query per sample over T <= 64 tokens of E = 64 channels, H <= 8 heads ---------
// scores[h][t] = qk[h] . kv[t]  (qk: the query with the key projection and the 1/sqrt(d) folded in, learner.py AttentionFeatures.forward),
// a = softmax over the unmasked t, mix[h] = sum_t a[h][t] kv[t] -- and its backward.  As PyTorch ops these are six batched products of
// 4 x 64 x 50 per sample, for which the GEMM library launches 128 x 256 macro tiles (20 us each), plus masked_fill / softmax and their
// backward: ~15 launches, ~130 us of a 450 us update.  Here: one wavefront per sample, the sample's 50 x 64 token tile in LDS (row
// stride 65: a column walk hits 32 different banks), lanes = tokens for the products over channels, lanes = channels for the
// products over tokens, the head-sized operands as LDS broadcasts.
constexpr int kAcE = 64, kAcTmax = 64, kAcHmax = 8, kAcRow = kAcE + 1, kAcWaves = 1;   // (one wavefront per workgroup: a batch of 256 lands on 256 CUs)
struct AcShared { float kv[kAcTmax * kAcRow]; float hq[kAcHmax * kAcE]; float ht[kAcHmax * kAcTmax]; float hs[kAcHmax * kAcTmax]; };

__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// the sample's token tile -> LDS: lane = channel, one coalesced row per step, all rows requested before the first store
__device__ __forceinline__ void ac_load_tile(AcShared& sh, const float* __restrict__ kv, int T, int lane) {
    float r[kAcTmax];
#pragma unroll
    for (int t = 0; t < kAcTmax; t++) r[t] = kv[(size_t)(t < T ? t : T - 1) * kAcE + lane];
#pragma unroll
    for (int t = 0; t < kAcTmax; t++) sh.kv[t * kAcRow + lane] = r[t];
}

template <int H>
__global__ __launch_bounds__(64 * kAcWaves) void attn_core_fwd_kernel(const float* __restrict__ qk, const float* __restrict__ kv, const uint8_t* __restrict__ mask,
                                                                     int B, int T, float* __restrict__ mix, float* __restrict__ attn) {
    __shared__ AcShared shs[kAcWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kAcWaves + wave);
    if (b >= B) return;                                        // (wave-uniform; no workgroup barrier below)
    AcShared& sh = shs[wave];
    ac_load_tile(sh, kv + (size_t)b * T * kAcE, T, lane);
#pragma unroll
    for (int h = 0; h < H; h++) sh.hq[h * kAcE + lane] = qk[((size_t)b * H + h) * kAcE + lane];
    const bool on = lane < T && mask[(size_t)b * T + (lane < T ? lane : 0)] == 0;
    __builtin_amdgcn_s_waitcnt(0);                            // (one wavefront per tile: LDS traffic is in order, no barrier needed)
    float s[H];
#pragma unroll
    for (int h = 0; h < H; h++) s[h] = 0.f;
    const int tr = (lane < T ? lane : T - 1) * kAcRow;
#pragma unroll 8
    for (int e = 0; e < kAcE; e++) {                           // lane = token: scores over the channels
        const float x = sh.kv[tr + e];
#pragma unroll
        for (int h = 0; h < H; h++) s[h] += sh.hq[h * kAcE + e] * x;
    }
#pragma unroll
    for (int h = 0; h < H; h++) {
        const float v = on ? s[h] : -3.0e38f;
        const float m = wave_max(v);
        const float p = on ? __expf(v - m) : 0.f;
        const float a = p / wave_sum(p);
        sh.ht[h * kAcTmax + lane] = a;
        if (lane < T) attn[((size_t)b * H + h) * T + lane] = a;
    }
    float o[H];
#pragma unroll
    for (int h = 0; h < H; h++) o[h] = 0.f;
    for (int t = 0; t < T; t++) {                              // lane = channel: the weighted mean of the tokens
        const float x = sh.kv[t * kAcRow + lane];
#pragma unroll
        for (int h = 0; h < H; h++) o[h] += sh.ht[h * kAcTmax + t] * x;
    }
#pragma unroll
    for (int h = 0; h < H; h++) mix[((size_t)b * H + h) * kAcE + lane] = o[h];
}

template <int H>
__global__ __launch_bounds__(64 * kAcWaves) void attn_core_bwd_kernel(const float* __restrict__ qk, const float* __restrict__ kv, const float* __restrict__ attn,
                                                                     const float* __restrict__ dmix, int B, int T, float* __restrict__ dqk,
                                                                     float* __restrict__ dkv) {
    __shared__ AcShared shs[kAcWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = __builtin_amdgcn_readfirstlane((int)blockIdx.x * kAcWaves + wave);
    if (b >= B) return;
    AcShared& sh = shs[wave];
    ac_load_tile(sh, kv + (size_t)b * T * kAcE, T, lane);
    float q[H], g[H], a[H];
#pragma unroll
    for (int h = 0; h < H; h++) {
        q[h] = qk[((size_t)b * H + h) * kAcE + lane];
        g[h] = dmix[((size_t)b * H + h) * kAcE + lane];
        a[h] = lane < T ? attn[((size_t)b * H + h) * T + lane] : 0.f;
        sh.hq[h * kAcE + lane] = g[h];                         // dmix, for the products over the channels
        sh.ht[h * kAcTmax + lane] = a[h];
    }
    __builtin_amdgcn_s_waitcnt(0);
    float da[H];
#pragma unroll
    for (int h = 0; h < H; h++) da[h] = 0.f;
    const int tr = (lane < T ? lane : T - 1) * kAcRow;
#pragma unroll 8
    for (int e = 0; e < kAcE; e++) {                           // lane = token: da[h][t] = dmix[h] . kv[t]
        const float x = sh.kv[tr + e];
#pragma unroll
        for (int h = 0; h < H; h++) da[h] += sh.hq[h * kAcE + e] * x;
    }
#pragma unroll
    for (int h = 0; h < H; h++) {                              // softmax backward: ds = a (da - sum_t a da)
        const float dot = wave_sum(a[h] * da[h]);
        sh.hs[h * kAcTmax + lane] = a[h] * (da[h] - dot);
    }
    float o[H];
#pragma unroll
    for (int h = 0; h < H; h++) o[h] = 0.f;
    for (int t = 0; t < T; t++) {                              // lane = channel: dqk[h] = sum_t ds[h][t] kv[t];  dkv[t] = sum_h a dmix + ds qk
        const float x = sh.kv[t * kAcRow + lane];
        float d = 0.f;
#pragma unroll
        for (int h = 0; h < H; h++) {
            const float ds = sh.hs[h * kAcTmax + t];
            o[h] += ds * x;
            d += sh.ht[h * kAcTmax + t] * g[h] + ds * q[h];
        }
        dkv[((size_t)b * T + t) * kAcE + lane] = d;
    }
#pragma unroll
    for (int h = 0; h < H; h++) dqk[((size_t)b * H + h) * kAcE + lane] = o[h];
}

}  // namespace

// replaces: the three matrix products of torch.nn.Linear's forward / backward at DQN batch sizes (dqn.py:1086 batch_size 256).
// C[M x N] = A . B (+ bias) with A(m, k) = A[m * a_sm + k * a_sk], B(k, n) = B[k * b_sk + n * b_sn]; C row-major with ldc.
// One of a_sm / a_sk (b_sk / b_sn) must be 1; flags UAVENV_GEMM_*.  `second` (nullable): an independent product in the same launch.
static bool gemm_ok(const UavGemm* g) {
    if (!g->A || !g->B || !g->C || g->M < 1 || g->N < 1 || g->K < 1 || g->ldc < g->N) return false;
    if ((g->a_sm != 1 && g->a_sk != 1) || (g->b_sk != 1 && g->b_sn != 1)) return false;
    return !(((g->flags & UAVENV_GEMM_BIAS) && !g->bias) || ((g->flags & UAVENV_GEMM_A_MASK) && !g->a_mask) ||
             ((g->flags & UAVENV_GEMM_ROWSUM) && !g->row_sum) || ((g->flags & UAVENV_GEMM_SUMSQ) && !g->sumsq));
}
extern "C" int uavenv_gemm_f32(const UavGemm* first, const UavGemm* second, void* stream) {
    if (!first || !gemm_ok(first) || (second && !gemm_ok(second))) return UAVENV_E_INVALID;
    auto tiles = [](const UavGemm* g, int cols) { return (long)((g->M + 15) / 16) * ((g->N + cols - 1) / cols); };
    // 16 x 64 tiles, or 16 x 32 when that is what it takes to give every CU a workgroup
    const long wide = tiles(first, 64) + (second ? tiles(second, 64) : 0);
    const bool narrow = wide < 256 && tiles(first, 32) + (second ? tiles(second, 32) : 0) <= 256;
    const int cols = narrow ? 32 : 64;
    long t0 = tiles(first, cols), t1 = second ? tiles(second, cols) : 0;
    // more workgroups than CUs: the product with the shorter reduction gets two tiles per workgroup, then the other one
    int dual = 0;
    if (t0 + t1 > 256) {
        const bool first_shorter = !second || first->K <= second->K;
        dual |= first_shorter ? 1 : 2;
        if ((first_shorter ? (t0 + 1) / 2 + t1 : t0 + (t1 + 1) / 2) > 256) dual = second ? 3 : 1;
    }
    if (dual & 1) t0 = (t0 + 1) / 2;
    if (dual & 2) t1 = (t1 + 1) / 2;
    const dim3 grid((unsigned)(t0 + t1)), block(1024);
    hipStream_t s = (hipStream_t)stream;
    const UavGemm& g1 = second ? *second : *first;
    if (narrow) gemm_f32_kernel<2><<<grid, block, 0, s>>>(*first, g1, (int)t0, dual);
    else gemm_f32_kernel<4><<<grid, block, 0, s>>>(*first, g1, (int)t0, dual);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}

// replaces: the loss of SB3's DQN.train (smooth-L1 TD error, dqn.py:1087 gamma) and its backward into the Q-values; scalars_dev
// float[UAVENV_UPD_COUNT]: receives the loss, holds / advances the optimiser step count and Adam's bias corrections.  batch <= 1024.
extern "C" int uavenv_td_loss(const float* q_dev, const float* q_next_dev, const int64_t* action_dev, const float* reward_dev,
                              const uint8_t* valid_dev, int32_t batch, int32_t n_actions, float gamma, float reward_scale, float beta1,
                              float beta2, float* dq_dev, float* scalars_dev, void* stream) {
    if (!q_dev || !q_next_dev || !action_dev || !reward_dev || !valid_dev || !dq_dev || !scalars_dev || batch < 1 || batch > 1024 || n_actions < 1)
        return UAVENV_E_INVALID;
    const int threads = ((batch + 63) / 64) * 64;
    td_loss_kernel<<<dim3(1), dim3(threads), 0, (hipStream_t)stream>>>(q_dev, q_next_dev, action_dev, reward_dev, valid_dev, batch, n_actions,
                                                                     gamma, reward_scale, beta1, beta2, dq_dev, scalars_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}

// replaces: clip_grad_norm_(parameters, max_norm) + Adam.step() of SB3's DQN.train over ONE flat buffer of n parameters
// (the caller keeps the module's parameters as views of it): clip + Adam with the learning rate scalars[UAVENV_UPD_LR] and the bias
// corrections uavenv_td_loss wrote, on the n_partials partial sums of squares the weight-gradient products left in workspace_dev
// (UAVENV_GEMM_SUMSQ); n_partials == 0: a launch of its own computes them first (float [UAVENV_UPD_WORKSPACE]) -- what a caller
// that changed the gradients after the products (an all-reduce over ranks) needs.  scalars[UAVENV_UPD_NORM2] receives the squared norm.
extern "C" int uavenv_clip_adam(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n,
                                float* scalars_dev, float* workspace_dev, int32_t n_partials, float max_norm, float beta1, float beta2,
                                float eps, void* stream) {
    if (!param_dev || !grad_dev || !exp_avg_dev || !exp_avg_sq_dev || !scalars_dev || !workspace_dev || n < 1 || n_partials < 0)
        return UAVENV_E_INVALID;
    if (n_partials == 0) {
        sum_squares_kernel<<<dim3(kNormBlocks), dim3(256), 0, (hipStream_t)stream>>>(grad_dev, n, workspace_dev);
        n_partials = kNormBlocks;
    }
    const unsigned blocks = (unsigned)((n + 256 * 4 - 1) / (256 * 4) < 2048 ? (n + 256 * 4 - 1) / (256 * 4) : 2048);
    clip_adam_kernel<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(param_dev, grad_dev, exp_avg_dev, exp_avg_sq_dev, n, scalars_dev,
                                                                       workspace_dev, n_partials, max_norm, beta1, beta2, eps);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}

// replaces: the action selection of SB3's DQN at collect time (predict(): argmax Q with probability 1 - epsilon, else a uniform
// action) for a vector of environments: q_dev float [n_envs][n_actions], *eps_dev the exploration rate, *counter_dev a float the
// kernel advances by one per call (the Philox counter of its draws, with `seed`), actions_out_dev int32 [n_envs].
extern "C" int uavenv_epsilon_greedy(const float* q_dev, int32_t n_envs, int32_t n_actions, const float* eps_dev, float* counter_dev,
                                     uint64_t seed, int32_t shared_coin, int32_t* actions_out_dev, void* stream) {
    if (!q_dev || !eps_dev || !counter_dev || !actions_out_dev || n_envs < 1 || n_actions < 1) return UAVENV_E_INVALID;
    epsilon_greedy_kernel<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>(q_dev, n_envs, n_actions, eps_dev, counter_dev, seed, shared_coin,
                                                                        actions_out_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}

// replaces: the last Linear of the acting forward + uavenv_epsilon_greedy in one launch: h_dev float [n_envs][K] (the last hidden
// layer's activations, ReLU applied), w_dev float [n_actions][K], b_dev float [n_actions] (torch.nn.Linear's layout), n_actions <= 8,
// K a multiple of 4, both 16-byte aligned, n_actions * K * 4 bytes of LDS (<= 64 KB); *counter_dev as in uavenv_epsilon_greedy (same draws for the same counter and seed);
// ticket_dev int32 [1], zero before the first call (the kernel leaves it zero); q_out_dev float [n_envs][n_actions], nullable.
extern "C" int uavenv_q_head_select(const float* h_dev, const float* w_dev, const float* b_dev, int32_t n_envs, int32_t k, int32_t n_actions,
                                    const float* eps_dev, float* counter_dev, int32_t* ticket_dev, uint64_t seed, int32_t shared_coin,
                                    int32_t* actions_out_dev, float* q_out_dev, void* stream) {
    if (!h_dev || !w_dev || !b_dev || !eps_dev || !counter_dev || !ticket_dev || !actions_out_dev || n_envs < 1 || k < 1 || n_actions < 1 ||
        n_actions > kSelMaxActions || (long)n_actions * k * 4 > 65536) return UAVENV_E_INVALID;
    if ((k & 3) != 0 || (reinterpret_cast<uintptr_t>(h_dev) & 15) || (reinterpret_cast<uintptr_t>(w_dev) & 15)) return UAVENV_E_INVALID;
    const unsigned wgs = (unsigned)((n_envs + 63) / 64);
    const size_t lds = (size_t)n_actions * k * sizeof(float);
    if (n_actions <= 5)
        q_head_select_kernel<5><<<dim3(wgs), dim3(256), lds, (hipStream_t)stream>>>(h_dev, w_dev, b_dev, n_envs, k, n_actions, eps_dev, counter_dev,
                                                                                ticket_dev, seed, shared_coin, actions_out_dev, q_out_dev);
    else
        q_head_select_kernel<kSelMaxActions><<<dim3(wgs), dim3(256), lds, (hipStream_t)stream>>>(h_dev, w_dev, b_dev, n_envs, k, n_actions, eps_dev,
                                                                                             counter_dev, ticket_dev, seed, shared_coin,
                                                                                             actions_out_dev, q_out_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}


// replaces: the batched products + masked softmax at the centre of nn.MultiheadAttention's forward / backward for ONE query per sample
// (dqn.py:633-640 cross_attn(query, keys, keys, key_padding_mask)), after the key / value projections have been folded out
// (learner.py AttentionFeatures.forward).  qk float [B][H][64] (scaled query times the key projection), kv float [B][T][64] (the
// tokens), mask uint8 [B][T] (1 = ignore; never a whole row), H in {1, 2, 4, 8}, T <= 64: mix float [B][H][64], attn float [B][H][T].
extern "C" int uavenv_attn_core_forward(const float* qk_dev, const float* kv_dev, const uint8_t* mask_dev, int32_t batch, int32_t heads,
                                        int32_t tokens, float* mix_out_dev, float* attn_out_dev, void* stream) {
    if (!qk_dev || !kv_dev || !mask_dev || !mix_out_dev || !attn_out_dev || batch < 1 || tokens < 1 || tokens > kAcTmax) return UAVENV_E_INVALID;
    const dim3 grid((unsigned)((batch + kAcWaves - 1) / kAcWaves)), block(64 * kAcWaves);
    hipStream_t s = (hipStream_t)stream;
    switch (heads) {
    case 1: attn_core_fwd_kernel<1><<<grid, block, 0, s>>>(qk_dev, kv_dev, mask_dev, batch, tokens, mix_out_dev, attn_out_dev); break;
    case 2: attn_core_fwd_kernel<2><<<grid, block, 0, s>>>(qk_dev, kv_dev, mask_dev, batch, tokens, mix_out_dev, attn_out_dev); break;
    case 4: attn_core_fwd_kernel<4><<<grid, block, 0, s>>>(qk_dev, kv_dev, mask_dev, batch, tokens, mix_out_dev, attn_out_dev); break;
    case 8: attn_core_fwd_kernel<8><<<grid, block, 0, s>>>(qk_dev, kv_dev, mask_dev, batch, tokens, mix_out_dev, attn_out_dev); break;
    default: return UAVENV_E_INVALID;
    }
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}

// its backward: dmix float [B][H][64] in; dqk float [B][H][64], dkv float [B][T][64] out (both written whole).
extern "C" int uavenv_attn_core_backward(const float* qk_dev, const float* kv_dev, const float* attn_dev, const float* dmix_dev, int32_t batch,
                                         int32_t heads, int32_t tokens, float* dqk_out_dev, float* dkv_out_dev, void* stream) {
    if (!qk_dev || !kv_dev || !attn_dev || !dmix_dev || !dqk_out_dev || !dkv_out_dev || batch < 1 || tokens < 1 || tokens > kAcTmax)
        return UAVENV_E_INVALID;
    const dim3 grid((unsigned)((batch + kAcWaves - 1) / kAcWaves)), block(64 * kAcWaves);
    hipStream_t s = (hipStream_t)stream;
    switch (heads) {
    case 1: attn_core_bwd_kernel<1><<<grid, block, 0, s>>>(qk_dev, kv_dev, attn_dev, dmix_dev, batch, tokens, dqk_out_dev, dkv_out_dev); break;
    case 2: attn_core_bwd_kernel<2><<<grid, block, 0, s>>>(qk_dev, kv_dev, attn_dev, dmix_dev, batch, tokens, dqk_out_dev, dkv_out_dev); break;
    case 4: attn_core_bwd_kernel<4><<<grid, block, 0, s>>>(qk_dev, kv_dev, attn_dev, dmix_dev, batch, tokens, dqk_out_dev, dkv_out_dev); break;
    case 8: attn_core_bwd_kernel<8><<<grid, block, 0, s>>>(qk_dev, kv_dev, attn_dev, dmix_dev, batch, tokens, dqk_out_dev, dkv_out_dev); break;
    default: return UAVENV_E_INVALID;
    }
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}
