// uavenv_attention.hip -- fused forward of the reference's UAVAttentionExtractor.
//
// Reference: agents/dqn/dqn.py:548-650.  The extractor consumes the frame-stacked observation this
// library produces ([frame_0 | ... | frame_{k-1}], 153 floats each) and is, in eager PyTorch, a chain of
// ~15 small launches (Linear 3k->64, LayerNorm, Linear 3->64 over 50 tokens, a 1-query 4-head
// cross-attention with key masking, LayerNorm, Linear 128->128): launch-bound for acting batches.
//
// Mapping: embed_dim = 64 = the wavefront width, so ONE WAVEFRONT owns one sample and lane j owns
// embedding channel j (or sensor token j in the score phase).  Nothing is materialised: the K/V
// projections are folded algebraically,
//     score[s,h] = (Wk_h^T q_h) . e_s + q_h . bk_h        context_h = Wv_h (sum_s a[s,h] e_s) + bv_h
// so the kernel does five 64x64 mat-vecs, one 128x128 mat-vec and two token sweeps per sample, all
// with v_readlane broadcasts and DPP/readlane reductions.  fp32 throughout (a floating-point kernel:
// parity is against the PyTorch fp32 module, tolerance in tests/test_gpu_attention.py).  No MFMA: M = 1
// per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uavenv.h"

namespace {

constexpr int kEmbed = 64, kHeads = 4, kHeadDim = 16, kSlots = 50, kFrame = 153, kFeat = 128;

__device__ __forceinline__ float rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum / max inside each 16-lane row (every lane of the row gets the result)
__device__ __forceinline__ float row_sum(float v) {
    v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, dpp<0xB1>(v)); v = fmaxf(v, dpp<0x4E>(v)); v = fmaxf(v, dpp<0x141>(v)); v = fmaxf(v, dpp<0x140>(v));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row_sum(v);
    return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row_max(v);
    return fmaxf(fmaxf(rl(v, 0), rl(v, 16)), fmaxf(rl(v, 32), rl(v, 48)));
}
__device__ __forceinline__ float layer_norm(float y, float g, float b) {      // nn.LayerNorm(64), eps 1e-5, biased variance
    const float mean = wave_sum(y) * (1.0f / kEmbed);
    const float d = y - mean;
    const float var = wave_sum(d * d) * (1.0f / kEmbed);
    return d * (1.0f / sqrtf(var + 1e-5f)) * g + b;
}
// y[j] = b[j] + sum_i Wt[i][j] * x[i]   (x held one element per lane; Wt is the TRANSPOSED weight, [in][64])
__device__ __forceinline__ float matvec64(const float* __restrict__ Wt, float bias, float x, int lane, int n_in) {
    float y = bias;
    for (int i = 0; i < n_in; i++) y += Wt[i * kEmbed + lane] * rl(x, i);
    return y;
}

// packed parameter block (floats), built by uavenv_amd/attention.py:pack_attention_weights
struct Offsets {
    int uavT, uav_b, ln1_g, ln1_b, sw0, sw1, sw2, sens_b, WqT, bq, Wk, bk, WvT, bv, WoT, bo, ln2_g, ln2_b, WfT, bf, total;
};
__host__ __device__ inline Offsets offsets(int n_stack) {
    Offsets o; int p = 0;
    o.uavT = p; p += 3 * n_stack * kEmbed;
    o.uav_b = p; p += kEmbed; o.ln1_g = p; p += kEmbed; o.ln1_b = p; p += kEmbed;
    o.sw0 = p; p += kEmbed; o.sw1 = p; p += kEmbed; o.sw2 = p; p += kEmbed; o.sens_b = p; p += kEmbed;
    o.WqT = p; p += kEmbed * kEmbed; o.bq = p; p += kEmbed;
    o.Wk = p; p += kEmbed * kEmbed; o.bk = p; p += kEmbed;
    o.WvT = p; p += kEmbed * kEmbed; o.bv = p; p += kEmbed;
    o.WoT = p; p += kEmbed * kEmbed; o.bo = p; p += kEmbed;
    o.ln2_g = p; p += kEmbed; o.ln2_b = p; p += kEmbed;
    o.WfT = p; p += kFeat * kFeat; o.bf = p; p += kFeat;
    o.total = p;
    return o;
}

__global__ __launch_bounds__(256) void uav_attention_kernel(const float* __restrict__ obs, const float* __restrict__ W,
                                                          float* __restrict__ out, int batch, int n_stack) {
    const int lane = threadIdx.x & 63;
    const int sample = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sample >= batch) return;
    const Offsets o = offsets(n_stack);
    const float* x = obs + (size_t)sample * (size_t)(n_stack * kFrame);
    const int n_uav = 3 * n_stack;

    // 1. temporal UAV context: Linear(3k -> 64) + LayerNorm + ReLU over the UAV triples of all frames (dqn.py:610-613)
    const float xin = lane < n_uav ? x[(lane / 3) * kFrame + lane % 3] : 0.0f;
    float q0 = matvec64(W + o.uavT, W[o.uav_b + lane], xin, lane, n_uav);
    q0 = fmaxf(layer_norm(q0, W[o.ln1_g + lane], W[o.ln1_b + lane]), 0.0f);

    // sensor tokens of the newest frame (dqn.py:616-619); lane s = token s for the score phase
    const float* cur = x + (size_t)(n_stack - 1) * kFrame + 3;
    const bool tok = lane < kSlots;
    const float t0 = tok ? cur[3 * lane] : 0.0f, t1 = tok ? cur[3 * lane + 1] : 0.0f, t2 = tok ? cur[3 * lane + 2] : 0.0f;
    // key padding mask: ghost slots and out-of-range sensors; unmask everything if all are masked (dqn.py:621-628)
    bool masked = !tok || (fabsf(t0) + fabsf(t1) + fabsf(t2) < 1e-6f) || (t2 < 1e-6f);
    const bool all_masked = __ballot(tok && !masked) == 0ull;
    masked = tok ? (masked && !all_masked) : true;
    // sensor_proj parameters: lane i holds row i of Linear(3 -> 64)
    const float sw0 = W[o.sw0 + lane], sw1 = W[o.sw1 + lane], sw2 = W[o.sw2 + lane], sb = W[o.sens_b + lane];

    // 2. query projection, scaled by 1/sqrt(head_dim)  (nn.MultiheadAttention)
    const float q = matvec64(W + o.WqT, W[o.bq + lane], q0, lane, kEmbed) * 0.25f;

    // 3. fold the key projection into the query: qk_h[i] = sum_{d in head h} q[d] Wk[d][i];  c_h = q_h . bk_h
    float qk[kHeads];
#pragma unroll
    for (int h = 0; h < kHeads; h++) {
        float acc = 0.0f;
#pragma unroll 8
        for (int d = 0; d < kHeadDim; d++) acc += rl(q, h * kHeadDim + d) * W[o.Wk + (h * kHeadDim + d) * kEmbed + lane];
        qk[h] = acc;
    }
    const float cb = row_sum(q * W[o.bk + lane]);                 // lanes of row h hold c_h

    // 4. scores: lane s accumulates over embedding channels i;  e_s[i] = relu(W_s[i] . token_s + b[i])
    float sc[kHeads] = {rl(cb, 0), rl(cb, 16), rl(cb, 32), rl(cb, 48)};
#pragma unroll 4
    for (int i = 0; i < kEmbed; i++) {
        const float e = fmaxf(rl(sb, i) + rl(sw0, i) * t0 + rl(sw1, i) * t1 + rl(sw2, i) * t2, 0.0f);
#pragma unroll
        for (int h = 0; h < kHeads; h++) sc[h] += rl(qk[h], i) * e;
    }
    float a[kHeads];
#pragma unroll
    for (int h = 0; h < kHeads; h++) {                             // masked softmax over the 50 tokens
        const float v = masked ? -__builtin_inff() : sc[h];
        const float m = wave_max(v);
        const float p = masked ? 0.0f : __expf(v - m);
        a[h] = p / wave_sum(p);
    }

    // 5. attention-weighted token mix per head (lane i = channel i):  mix_h[i] = sum_s a[s,h] e_s[i]
    float mix[kHeads] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 2
    for (int s = 0; s < kSlots; s++) {
        const float e = fmaxf(sb + sw0 * rl(t0, s) + sw1 * rl(t1, s) + sw2 * rl(t2, s), 0.0f);
#pragma unroll
        for (int h = 0; h < kHeads; h++) mix[h] += rl(a[h], s) * e;
    }
    // 6. value projection of the mix of this lane's head, then the output projection and LayerNorm
    const int my_head = lane >> 4;
    float ctx = W[o.bv + lane];
#pragma unroll 4
    for (int i = 0; i < kEmbed; i++) {
        const float m0 = rl(mix[0], i), m1 = rl(mix[1], i), m2 = rl(mix[2], i), m3 = rl(mix[3], i);
        const float mh = my_head == 0 ? m0 : (my_head == 1 ? m1 : (my_head == 2 ? m2 : m3));
        ctx += W[o.WvT + i * kEmbed + lane] * mh;
    }
    float ao = matvec64(W + o.WoT, W[o.bo + lane], ctx, lane, kEmbed);
    ao = layer_norm(ao, W[o.ln2_g + lane], W[o.ln2_b + lane]);       // dqn.py:641

    // 7. fusion: Linear(128 -> 128) + ReLU over [uav_embed | attention context]; lane j -> outputs j and j + 64
    float f0 = W[o.bf + lane], f1 = W[o.bf + 64 + lane];
    const float* Wf = W + o.WfT;
#pragma unroll 4
    for (int i = 0; i < kEmbed; i++) {
        const float u = rl(q0, i), c = rl(ao, i);
        f0 += Wf[i * kFeat + lane] * u + Wf[(kEmbed + i) * kFeat + lane] * c;
        f1 += Wf[i * kFeat + 64 + lane] * u + Wf[(kEmbed + i) * kFeat + 64 + lane] * c;
    }
    float* y = out + (size_t)sample * kFeat;
    y[lane] = fmaxf(f0, 0.0f);
    y[64 + lane] = fmaxf(f1, 0.0f);
}

}  // namespace

// replaces: UAVAttentionExtractor.forward (agents/dqn/dqn.py:604-650) for inference.
// obs_dev float [batch][n_stack*153]; weights_dev = the packed block (uavenv_attention_weight_floats(n_stack) floats,
// layout in this file's Offsets / uavenv_amd/attention.py); out_dev float [batch][128].
extern "C" int uavenv_attention_weight_floats(int32_t n_stack) { return n_stack >= 1 && 3 * n_stack <= 64 ? offsets(n_stack).total : UAVENV_E_INVALID; }

extern "C" int uavenv_attention_features(const float* obs_dev, const float* weights_dev, float* out_dev, int32_t batch,
                                         int32_t n_stack, void* stream) {
    if (!obs_dev || !weights_dev || !out_dev || batch <= 0 || n_stack < 1 || 3 * n_stack > 64) return UAVENV_E_INVALID;
    uav_attention_kernel<<<dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(obs_dev, weights_dev, out_dev,
                                                                                                    batch, n_stack);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}
