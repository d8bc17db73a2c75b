// uavenv_attention.hip -- fused forward of the reference's UAVAttentionExtractor.
//
// Reference: agents/dqn/dqn.py:548-650.  The extractor consumes the frame-stacked observation this
// library produces ([frame_0 | ... | frame_{k-1}], 153 floats each) and is, in eager PyTorch, a chain of
// ~15 small launches (Linear 3k->64, LayerNorm, Linear 3->64 over 50 tokens, a 1-query 4-head
// cross-attention with key masking, LayerNorm, Linear 128->128): launch-bound for acting batches.
//
// Mapping (round 3).  A workgroup of sixteen wavefronts owns SIXTEEN samples:
//   * the six projections whose weights are shared by all samples -- UAV encoder 3k->64, Wq, the key projection
//     folded into the query (per head [16 x 16].[16 x 64]), the value projection of the per-head token mix
//     (per head [16 x 64].[64 x 16]), Wo, and the fusion 128->128 -- are [16 x K].[K x N] products on
//     v_mfma_f32_16x16x4_f32 (exact f32 fma chains): the 16 samples are the M dimension, a wavefront owns one
//     16-column tile of N, the weights are read ONCE per workgroup from a block packed in fragment order (each lane
//     loads its own consecutive floats), activations travel between the products through LDS in fragment order too
//     (lane l's A fragments of a K = 64 product are 16 consecutive floats of chunk l);
//   * what is different per sample -- the 64 x 50 matrix e[i][s] = relu(W_s[i] . token_s + b[i]), the scores
//     sum_i qk_h[i] e[i][s], the masked softmax and the mix sum_s a_h[s] e[i][s] -- belongs to wavefront w = sample w
//     (lane = token, then lane = channel) on the vector ALU, as packed-f32 arithmetic over channel / token PAIRS with the
//     wave-uniform operands arriving through scalar loads (sensor projection) or broadcast LDS reads (folded query, attention
//     weights, tokens): 10 vector instructions per pair where round 2's readlane + fma form took ~31.
// Nothing is materialised: the K/V projections are folded algebraically,
//     score[s,h] = (Wk_h^T q_h) . e_s (+ q_h . bk_h, constant over s: it cancels in the softmax and is dropped)
//     context_h = Wv_h (sum_s a[s,h] e_s) + bv_h
// fp32 throughout (a floating-point kernel: parity is against the PyTorch fp32 module, tolerance in
// tests/test_gpu_attention.py).  Round 2's kernel (one wavefront per sample, every multiply-accumulate a v_readlane +
// fma, the 138 KB weight block streamed per sample) took 48-50 us for 4096 samples; numbers for this one: DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uavenv.h"

// timing-only ablation builds (tools/attn_exp.sh): -DATTN_EXIT=k ends the kernel after phase k (every thread at the same point)
#ifdef ATTN_EXIT
#define ATTN_PHASE(k) do { if (ATTN_EXIT == (k)) return; } while (0)
#else
#define ATTN_PHASE(k) do { } while (0)
#endif

// diagnostic builds (tools/attn_exp.sh): -DATTN_STAMPS records s_memtime per wavefront of workgroup 0 at every phase boundary
#ifdef ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[16 * 16];
#define ATTN_STAMP(k) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_attn_stamps[(threadIdx.x >> 6) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" int uavenv_debug_attn_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), sizeof(g_attn_stamps)) == hipSuccess ? 0 : -2;
}
#else
#define ATTN_STAMP(k) do { } while (0)
#endif

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kEmbed = 64, kHeads = 4, kSlots = 50, kFrame = 153, kFeat = 128;
constexpr int kM = 16;                 // samples per workgroup = rows of every MFMA tile
constexpr int kChunk = 20;             // floats per chunk of a fragment-order buffer: 16 used + 4 of padding (80 B: an odd number
                                       // of 16-byte slots, so the 16 lanes of a ds_read_b128 group hit 16 different slots)
constexpr int kBuf = 64 * kChunk;      // one [16 samples x 64 channels] activation buffer in fragment order
constexpr int kQkStride = 256;         // folded queries per sample: [channel pair 32][head 4][2]
constexpr int kPairRow = 8;            // floats per token pair / channel pair row of the sweep operands
constexpr int kScratch = 2 * 25 * kPairRow;   // per wavefront: attention weights [25 token pairs][head 4][2] + tokens [25][3 (+1)][2]

// LDS map (floats)
constexpr int L_X = 0, L_Q0 = L_X + kBuf, L_Q = L_Q0 + kBuf, L_CTX = L_Q + kBuf, L_AO = L_CTX + kBuf;
constexpr int L_QK = L_AO + kBuf, L_MIX = L_QK + kM * kQkStride;
constexpr int kThreads = 1024;        // 16 wavefronts
constexpr int L_SWP = L_MIX + kHeads * kBuf;                      // the sensor projection as channel-pair rows (mix sweep: lane = channel)
constexpr int L_SCR = L_SWP + 32 * kPairRow, L_TOTAL = L_SCR + 16 * kScratch;

template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
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
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat(float v) { f32x2 r = {v, v}; return r; }
__device__ __forceinline__ f32x2 lo(f32x4 v) { f32x2 r = {v.x, v.y}; return r; }
__device__ __forceinline__ f32x2 hi(f32x4 v) { f32x2 r = {v.z, v.w}; return r; }
// what one wavefront wrote to its own LDS scratch becomes visible to its other lanes (no workgroup barrier: the scratch is private)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// packed parameter block (floats), built by uavenv_amd/attention.py:pack_attention_weights.  "P" blocks are in MFMA fragment
// order: [column tile][lane 64][k-step t], lane l = (k-group g = l >> 4, column c = l & 15) holding W^T[k = g * KT + t][16 * tile + c].
struct Offsets {
    int kt0;                                  // k-steps per lane of the UAV encoder product: 4 * ceil(3 * n_stack / 16)
    int uavP, uav_b, ln1_g, ln1_b, swp, WqP, bq, WkP, WvP, bv, WoP, bo, ln2_g, ln2_b, WfTopP, WfBotP, bf, total;
};
__host__ __device__ inline Offsets offsets(int n_stack) {
    Offsets o; int p = 0;
    o.kt0 = 4 * ((3 * n_stack + 15) / 16);
    o.uavP = p; p += 4 * 64 * o.kt0;
    o.uav_b = p; p += kEmbed; o.ln1_g = p; p += kEmbed; o.ln1_b = p; p += kEmbed;
    o.swp = p; p += 32 * kPairRow;            // [channel pair][sw0 pair, sw1 pair, sw2 pair, bias pair]
    o.WqP = p; p += 4 * 64 * 16; o.bq = p; p += kEmbed;
    o.WkP = p; p += kHeads * 4 * 64 * 4;      // [head][tile][lane][4]: rows 16h .. 16h+15 of Wk as a K = 16 product
    o.WvP = p; p += 4 * 64 * 16; o.bv = p; p += kEmbed;
    o.WoP = p; p += 4 * 64 * 16; o.bo = p; p += kEmbed;
    o.ln2_g = p; p += kEmbed; o.ln2_b = p; p += kEmbed;
    o.WfTopP = p; p += 8 * 64 * 16; o.WfBotP = p; p += 8 * 64 * 16; o.bf = p; p += kFeat;
    o.total = p;
    return o;
}

// The lane's B fragments of one 16 x 16 tile of a K = 64 product (16 consecutive floats of the packed block): fetched BEFORE
// the barrier that precedes the product, so that the L2 round trip overlaps the phase before it.
struct BFrag { f32x4 v[4]; };
__device__ __forceinline__ BFrag load_b(const float* __restrict__ b) {
    BFrag r;
#pragma unroll
    for (int t = 0; t < 4; t++) r.v[t] = *reinterpret_cast<const f32x4*>(b + 4 * t);
    return r;
}
// One 16 x 16 output tile: 16 k-steps of 4.  `a` = the lane's chunk of a fragment-order LDS buffer.  Two accumulators: a
// v_mfma_f32_16x16x4_f32 issues every 32 cycles but its result is ready for a dependent one after 40.
__device__ __forceinline__ f32x4 mfma_tile64(const float* a, const BFrag& b, f32x4 acc0) {
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + 4 * t);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b.v[t].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b.v[t].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b.v[t].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b.v[t].w, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}
// same for a runtime number of k-steps (the UAV encoder: K = 16 .. 64 by n_stack), fragments fetched in the loop
__device__ __forceinline__ f32x4 mfma_tile(const float* a, const float* __restrict__ b, int kt4, f32x4 acc0) {
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < kt4; t++) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + 4 * t);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + 4 * t);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}
// the result tile (lane: column c = l & 15, rows 4 * (l >> 4) + r) as the K-slice `tile` of the NEXT product's A operand
__device__ __forceinline__ void store_frag(float* buf, int tile, int lane, f32x4 v) {
    float* p = buf + (tile * 16 + 4 * (lane >> 4)) * kChunk + (lane & 15);
    p[0] = v.x; p[kChunk] = v.y; p[2 * kChunk] = v.z; p[3 * kChunk] = v.w;
}
// nn.LayerNorm(64) (eps 1e-5, biased variance) in place on a fragment-order buffer: wavefront w (< 4) takes samples 4w .. 4w+3,
// one 16-lane row per sample, four channels per lane
__device__ __forceinline__ void layer_norm_frag(float* buf, int w, int lane, const float* __restrict__ g, const float* __restrict__ b, bool relu) {
    const int s = 4 * w + (lane >> 4), cq = lane & 15;
    float* p = buf + ((cq >> 2) * 16 + s) * kChunk + 4 * (cq & 3);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(g + 4 * cq), bt = *reinterpret_cast<const f32x4*>(b + 4 * cq);
    f32x4 v = *reinterpret_cast<f32x4*>(p);
    const float mean = row_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / kEmbed);
    v -= mean;
    const float var = row_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / kEmbed);
    const float inv = __builtin_amdgcn_rsqf(var + 1e-5f);
    v = v * inv * gm + bt;
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<f32x4*>(p) = v;
}

// 16 wavefronts: wavefront w owns sample w in the per-sample sweeps (four wavefronts per SIMD hide each other's LDS and
// dependency latencies); the shared-weight products need 4 (one per 16-column tile), 8 (fusion) or 16 (key fold) of them.
__global__ __launch_bounds__(kThreads) void uav_attention_kernel(const float* __restrict__ obs, const float* __restrict__ W,
                                                               float* __restrict__ out, int batch, int n_stack) {
    __shared__ __attribute__((aligned(16))) float lds[L_TOTAL];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = blockIdx.x * kM;
    const Offsets o = offsets(n_stack);
    const size_t row = (size_t)n_stack * kFrame;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int c16 = lane & 15;
    ATTN_STAMP(0);

    // ---- 0. inputs.  Wavefront w = sample w: its UAV triples of all frames go to LDS in fragment order (k = 3 * frame +
    //         component, zero padded to 4 * kt0); its sensor tokens of the newest frame stay in registers until the sweeps
    //         (dqn.py:616-619; lane = token).
    const int sg_own = base + w;
    const bool own = sg_own < batch;                        // wave-uniform: the tail workgroup has rows that are no sample
    const bool tok = lane < kSlots;
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    {
        const float* x = obs + (size_t)(own ? sg_own : 0) * row;
        const int n_uav = 3 * n_stack, k0 = 4 * o.kt0;
        if (lane < k0) lds[L_X + ((lane / o.kt0) * 16 + w) * kChunk + lane % o.kt0] = (own && lane < n_uav) ? x[(lane / 3) * kFrame + lane % 3] : 0.0f;
        if (own && tok) { const float* cur = x + (size_t)(n_stack - 1) * kFrame + 3 + 3 * lane; t0 = cur[0]; t1 = cur[1]; t2 = cur[2]; }
        if (tid < 32 * kPairRow) lds[L_SWP + tid] = W[o.swp + tid];
    }
    __syncthreads();
    ATTN_PHASE(0);
    ATTN_STAMP(1);

    // ---- 1. temporal UAV context: Linear(3k -> 64) + LayerNorm + ReLU (dqn.py:610-613)
    BFrag bq_w;
    if (w < 4) {
        const float bias = W[o.uav_b + 16 * w + c16];
        const f32x4 acc = mfma_tile(lds + L_X + lane * kChunk, W + o.uavP + (w * 64 + lane) * o.kt0, o.kt0 >> 2, zero4);
        store_frag(lds + L_Q0, w, lane, acc + bias);
        bq_w = load_b(W + o.WqP + (w * 64 + lane) * 16);
    }
    __syncthreads();
    if (w < 4) layer_norm_frag(lds + L_Q0, w, lane, W + o.ln1_g, W + o.ln1_b, true);
    __syncthreads();
    ATTN_PHASE(1);
    ATTN_STAMP(2);

    // ---- 2. query projection, scaled by 1/sqrt(head_dim) (nn.MultiheadAttention)
    if (w < 4) {
        const float bias = W[o.bq + 16 * w + c16];
        const f32x4 acc = mfma_tile64(lds + L_Q0 + lane * kChunk, bq_w, zero4);
        store_frag(lds + L_Q, w, lane, (acc + bias) * 0.25f);
    }
    // ---- 3. the key projection folded into the query: qk_h[i] = sum_{d in head h} q[d] Wk[d][i]; wavefront w: head w >> 2,
    //         channels i = 16 * (w & 3) + c.  (q_h . bk_h would shift every score of head h by the same amount: no effect on the
    //         softmax, dropped.)
    const int h3 = w >> 2, tile3 = w & 3;
    const f32x4 bk = *reinterpret_cast<const f32x4*>(W + o.WkP + ((h3 * 4 + tile3) * 64 + lane) * 4);
    __syncthreads();
    {
        const int g = lane >> 4, i = 16 * tile3 + c16;
        const f32x4 av = *reinterpret_cast<const f32x4*>(lds + L_Q + (h3 * 16 + c16) * kChunk + 4 * g);        // q[sample c16][16h + 4g + t]
        f32x4 a0 = zero4, a1 = zero4;
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bk.x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bk.y, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bk.z, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bk.w, a1, 0, 0, 0);
        const f32x4 r = a0 + a1;
        float* p = lds + L_QK + (4 * g) * kQkStride + (i >> 1) * kPairRow + 2 * h3 + (i & 1);                   // [sample][pair][head][2]
        p[0] = r.x; p[kQkStride] = r.y; p[2 * kQkStride] = r.z; p[3 * kQkStride] = r.w;
    }
    // the value projection's fragments for after the sweeps
    BFrag bv_w;
    if (w < 4) bv_w = load_b(W + o.WvP + (w * 64 + lane) * 16);
    __syncthreads();
    ATTN_PHASE(3);
    ATTN_STAMP(3);

    // ---- 4. this wavefront's sample: scores over the 50 tokens, masked softmax, attention-weighted token mix (dqn.py:616-637),
    //         as packed-f32 arithmetic over channel pairs (scores: lane = token) and token pairs (mix: lane = channel): 6 vector
    //         instructions for a pair of e[i][s] = relu(W_s[i] . token_s + b[i]) and 4 for its contribution to the four heads.
    //         Every operand that is uniform over the wavefront has to be broadcast: the sensor projection of the score sweep comes
    //         through SCALAR loads (the packed block is read-only), the folded query / the attention weights / the token pairs of
    //         the mix sweep through broadcast ds_read_b128 -- which occupy the LDS as long as full-width reads, so they are what
    //         this phase is bound by together with the vector ALU (v_mfma_f32_4x4x1_16b_f32 with rows = heads
    //         was tried for the two contractions: correct, no LDS traffic, but the instruction occupies the matrix pipe for ~64
    //         cycles on gfx950 -- an eighth of the f32 MFMA rate -- and the phase took the same 8 us; e itself as K = 4 products on
    //         v_mfma_f32_16x16x4_f32 -- 16 x 16 tiles, contraction as 4 fma per head and tile row block, token tiles without a live
    //         token skipped in both sweeps -- was also built: correct, and 17.9 us instead of 15.2 on dense inputs, 14.0 instead of
    //         13.2 on typical masks: the 32 cross-lane-group additions per sweep cost more than the packed arithmetic they replace).
    {
        typedef const __attribute__((address_space(4))) float* cptr;        // constant address space: uniform loads become s_load
        float* As = lds + L_SCR + w * kScratch;            // [token pair 25][head 4][2]
        float* Ts = As + 25 * kPairRow;                    // [token pair 25][t0 pair, t1 pair, t2 pair, -]
        float* mixp = lds + L_MIX + ((lane >> 4) * 16 + w) * kChunk + c16;
        if (!own) {
#pragma unroll
            for (int h = 0; h < kHeads; h++) mixp[h * kBuf] = 0.0f;
        } else {
            // key padding mask: ghost slots and out-of-range sensors; unmask everything if all are masked (dqn.py:621-628)
            bool masked = !tok || (fabsf(t0) + fabsf(t1) + fabsf(t2) < 1e-6f) || (t2 < 1e-6f);
            const bool all_masked = __ballot(tok && !masked) == 0ull;
            masked = tok ? (masked && !all_masked) : true;
            float* tp = Ts + (lane >> 1) * kPairRow + (lane & 1);
            if (tok) { tp[0] = t0; tp[2] = t1; tp[4] = t2; }

            // scores: lane = token s, channel pairs (i, i+1)
            const f32x2 t0s = splat(t0), t1s = splat(t1), t2s = splat(t2), z2 = splat(0.0f);
            const cptr cw = (cptr)(W + o.swp);
            const f32x4* qk = reinterpret_cast<const f32x4*>(lds + L_QK + w * kQkStride);
            f32x2 sc0 = z2, sc1 = z2, sc2 = z2, sc3 = z2;
#pragma unroll 8
            for (int ip = 0; ip < 32; ip++) {
                const f32x2 w0 = {cw[8 * ip], cw[8 * ip + 1]}, w1 = {cw[8 * ip + 2], cw[8 * ip + 3]};
                const f32x2 w2 = {cw[8 * ip + 4], cw[8 * ip + 5]}, wb = {cw[8 * ip + 6], cw[8 * ip + 7]};
                const f32x4 q01 = qk[2 * ip], q23 = qk[2 * ip + 1];
                f32x2 e = w0 * t0s;
                e = pk_fma(w1, t1s, e);
                e = pk_fma(w2, t2s, e);
                e = e + wb;
                e.x = fmaxf(e.x, 0.0f); e.y = fmaxf(e.y, 0.0f);
                sc0 = pk_fma(lo(q01), e, sc0); sc1 = pk_fma(hi(q01), e, sc1);
                sc2 = pk_fma(lo(q23), e, sc2); sc3 = pk_fma(hi(q23), e, sc3);
            }
            ATTN_STAMP(4);
            const float sc[kHeads] = {sc0.x + sc0.y, sc1.x + sc1.y, sc2.x + sc2.y, sc3.x + sc3.y};
            float* ap = As + (lane >> 1) * kPairRow + (lane & 1);
#pragma unroll
            for (int h = 0; h < kHeads; h++) {                 // masked softmax over the tokens
                const float v = masked ? -__builtin_inff() : sc[h];
                const float m = wave_max(v);
                const float p = masked ? 0.0f : __expf(v - m);
                const float a = p * __builtin_amdgcn_rcpf(wave_sum(p));
                if (tok) ap[2 * h] = a;
            }
            ATTN_STAMP(5);
            wave_lds_sync();                                   // As / Ts are this wavefront's own: no workgroup barrier

            // mix: lane = channel i; mix_h[i] = sum_s a_h[s] e[i][s] over token pairs (s, s+1)
            const float* my = lds + L_SWP + (lane >> 1) * kPairRow + (lane & 1);
            const f32x2 c_sw0 = splat(my[0]), c_sw1 = splat(my[2]), c_sw2 = splat(my[4]), c_sb = splat(my[6]);
            const f32x4* a4 = reinterpret_cast<const f32x4*>(As);
            const f32x4* t4 = reinterpret_cast<const f32x4*>(Ts);
            f32x2 m0 = z2, m1 = z2, m2 = z2, m3 = z2;
            // (a pair of masked tokens has zero attention weight in every head: ghost slots and out-of-range sensors are most of a
            //  typical observation, and the pair mask is wave-uniform)
            const unsigned long long live = __ballot(!masked);
#pragma unroll 5
            for (int sp = 0; sp < 25; sp++) {
#ifndef ATTN_NO_SKIP
                if (((live >> (2 * sp)) & 3ull) == 0ull) continue;
#endif
                const f32x4 ta = t4[2 * sp], a01 = a4[2 * sp], a23 = a4[2 * sp + 1];
                const f32x2 tb = *reinterpret_cast<const f32x2*>(Ts + sp * kPairRow + 4);
                f32x2 e = c_sw0 * lo(ta);
                e = pk_fma(c_sw1, hi(ta), e);
                e = pk_fma(c_sw2, tb, e);
                e = e + c_sb;
                e.x = fmaxf(e.x, 0.0f); e.y = fmaxf(e.y, 0.0f);
                m0 = pk_fma(lo(a01), e, m0); m1 = pk_fma(hi(a01), e, m1);
                m2 = pk_fma(lo(a23), e, m2); m3 = pk_fma(hi(a23), e, m3);
            }
            mixp[0] = m0.x + m0.y; mixp[kBuf] = m1.x + m1.y; mixp[2 * kBuf] = m2.x + m2.y; mixp[3 * kBuf] = m3.x + m3.y;
        }
    }
    ATTN_STAMP(6);
    BFrag bo_w;
    if (w < 4) bo_w = load_b(W + o.WoP + (w * 64 + lane) * 16);
    __syncthreads();
    ATTN_PHASE(4);
    ATTN_STAMP(7);

    // ---- 5. value projection: wavefront h = head h: ctx[16h + c] = bv + sum_i Wv[16h + c][i] mix_h[i]
    if (w < 4) {
        const float bias = W[o.bv + 16 * w + c16];
        const f32x4 acc = mfma_tile64(lds + L_MIX + w * kBuf + lane * kChunk, bv_w, zero4);
        store_frag(lds + L_CTX, w, lane, acc + bias);
    }
    __syncthreads();
    // ---- 6. output projection + LayerNorm (dqn.py:641)
    if (w < 4) {
        const float bias = W[o.bo + 16 * w + c16];
        const f32x4 acc = mfma_tile64(lds + L_CTX + lane * kChunk, bo_w, zero4);
        store_frag(lds + L_AO, w, lane, acc + bias);
    }
    // ---- 7. fusion, first half: Linear(128 -> 128) over [uav_embed | attention context] = two K = 64 products into one tile;
    //         wavefront w < 8: output columns 16w ..; the uav_embed half does not wait for the attention branch
    f32x4 facc = zero4;
    BFrag bf_bot;
    if (w < 8) {
        const BFrag bf_top = load_b(W + o.WfTopP + (w * 64 + lane) * 16);
        bf_bot = load_b(W + o.WfBotP + (w * 64 + lane) * 16);
        facc = mfma_tile64(lds + L_Q0 + lane * kChunk, bf_top, zero4);
    }
    __syncthreads();
    if (w < 4) layer_norm_frag(lds + L_AO, w, lane, W + o.ln2_g, W + o.ln2_b, false);
    __syncthreads();
    ATTN_PHASE(6);
    ATTN_STAMP(8);
    if (w < 8) {
        f32x4 acc = mfma_tile64(lds + L_AO + lane * kChunk, bf_bot, facc);
        acc += W[o.bf + 16 * w + c16];
        const float r[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int sg = base + 4 * (lane >> 4) + q;
            if (sg < batch) out[(size_t)sg * kFeat + 16 * w + c16] = fmaxf(r[q], 0.0f);
        }
    }
    ATTN_STAMP(9);
}

}  // namespace

// replaces: UAVAttentionExtractor.forward (agents/dqn/dqn.py:604-650) for inference.
// obs_dev float [batch][n_stack*153]; weights_dev = the packed block (uavenv_attention_weight_floats(n_stack) floats,
// layout in this file's Offsets / uavenv_amd/attention.py); out_dev float [batch][128].
extern "C" int uavenv_attention_weight_floats(int32_t n_stack) { return n_stack >= 1 && 3 * n_stack <= 64 ? offsets(n_stack).total : UAVENV_E_INVALID; }

extern "C" int uavenv_attention_features(const float* obs_dev, const float* weights_dev, float* out_dev, int32_t batch,
                                         int32_t n_stack, void* stream) {
    if (!obs_dev || !weights_dev || !out_dev || batch <= 0 || n_stack < 1 || 3 * n_stack > 64) return UAVENV_E_INVALID;
    if ((reinterpret_cast<uintptr_t>(weights_dev) & 15u) != 0) return UAVENV_E_INVALID;      // the packed block is read in 16-byte pieces
    uav_attention_kernel<<<dim3((unsigned)((batch + kM - 1) / kM)), dim3(kThreads), 0, (hipStream_t)stream>>>(obs_dev, weights_dev, out_dev,
                                                                                                        batch, n_stack);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}
