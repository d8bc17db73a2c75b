// uavenv_replay.hip -- the consumer side of the transition ring (replay.py): one launch builds a sampled batch of stacked
// transitions.  What it replaces is the data path SB3's ReplayBuffer.sample + VecFrameStack bookkeeping stand for in the
// reference's trainer (agents/dqn/dqn.py:1083-1089, :1278): there every transition stores its 2 x n_stack frames; here a frame
// is stored once and the stacks are gathered on the fly -- in PyTorch that is ~40 small indexing launches (174 us of GPU time per
// batch of 256 under graph replay), here one kernel that reads (n_stack + 1) rows and writes 2 x n_stack rows per sample.
//
// Ring layout (replay.py): store[chunk][rank][ L blocks | T terminal rows | count ], block = [ E x D obs (padded to 4) | E x 4 aux ],
// aux = (action, reward, done, terminal ticket as int32 bits); slot s lives in chunk s / L, block s % L.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uavenv.h"
#include "uavenv_noise.h"

namespace uavenv {

struct RingView {
    const float* store; UavRingLayout g;
    __device__ __forceinline__ const float* part(long chunk, long rank) const { return store + (chunk * g.world + rank) * g.section; }
    __device__ __forceinline__ const float* obs(long slot, long rank, long env) const {
        return part(slot / g.slots_per_chunk, rank) + (slot % g.slots_per_chunk) * (long)g.block + env * g.obs_dim;
    }
    __device__ __forceinline__ const float* aux(long slot, long rank, long env) const {
        return part(slot / g.slots_per_chunk, rank) + (slot % g.slots_per_chunk) * (long)g.block + g.obs_floats + env * 4;
    }
};

constexpr int kMaxStack = 16;

// The draw of a batch, when the kernel makes it itself (uavenv_ring_sample_stacked): sample b of draw number `counter` gets the
// Philox words of counter (b, counter lo, counter hi, tag) under `seed`: j uniform on 0 .. n-2 (the transition out of the newest
// sampleable slot has no successor yet), rank uniform on the ranks, environment uniform on a rank's environments.
struct RingDraw { const int64_t* window; const float* counter; uint64_t seed; int64_t* index_out; };

// a position in the ring as (chunk, block within the chunk): stepping to the neighbouring slot is an add and a wrap, where
// slot / L and slot % L per frame were a 64-bit division each (~130 instructions; the kernel had 2 (k + 2) of them)
struct RingPos {
    int chunk, blk;
    __device__ __forceinline__ void back(const UavRingLayout& g) { if (--blk < 0) { blk = g.slots_per_chunk - 1; chunk = chunk == 0 ? g.num_chunks - 1 : chunk - 1; } }
    __device__ __forceinline__ void forward(const UavRingLayout& g) { if (++blk == g.slots_per_chunk) { blk = 0; chunk = chunk + 1 == g.num_chunks ? 0 : chunk + 1; } }
};

// one wavefront per sample; everything about the sample is wave-uniform, the lanes copy row elements.  KB: compile-time bound of
// the stack depth k, so that the rows of a sample live in registers: EVERY load of the sample (k episode-start flags, k + 1 rows,
// the successor's aux words, the chunk's terminal count) depends on (slot, rank, env) only and is issued before the first wait --
// one trip to HBM per sample (the ring is far larger than the L2s: every row is a miss) where the loop form the compiler made of
// `for f: load, multiply, store` paid k + 3 of them in a row (11 us for 256 samples x 4 frames).
template <bool kDraw, int KB>
__global__ __launch_bounds__(256) void uav_ring_gather_kernel(RingView rv, const int64_t* __restrict__ age, const int64_t* __restrict__ slot_,
                                                              const int64_t* __restrict__ rank_, const int64_t* __restrict__ env_, RingDraw dr,
                                                              int32_t batch,
                                                              int32_t k, float* __restrict__ obs_out, float* __restrict__ next_out,
                                                              int64_t* __restrict__ action_out, float* __restrict__ reward_out,
                                                              uint8_t* __restrict__ done_out, uint8_t* __restrict__ valid_out) {
    const int lane = threadIdx.x & 63;
    const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));   // (tells the compiler it is wave-uniform: addresses in SGPRs)
    if (b >= batch) return;
    const UavRingLayout& g = rv.g;
    const long cap = (long)g.num_chunks * g.slots_per_chunk;
    long j, slot, r, e;
    if (kDraw) {
        const long n = dr.window[0], oldest = dr.window[1];
        const uint64_t cnt = (uint64_t)dr.counter[0];
        const Words4 w = philox4x32<UAVENV_PHILOX_ROUNDS>((uint32_t)b, (uint32_t)cnt, (uint32_t)(cnt >> 32), 0x52494E47u /* "RING" */,
                                                          (uint32_t)dr.seed, (uint32_t)(dr.seed >> 32));
        j = (long)(((uint64_t)w.w0 * (uint64_t)(n - 1)) >> 32);
        r = (long)mulhi32(w.w1, (uint32_t)g.world);
        e = (long)mulhi32(w.w2, (uint32_t)g.envs);
        slot = oldest + j;                                     // oldest < cap, j < cap
        if (slot >= cap) slot -= cap;
        if (lane == 0 && dr.index_out != nullptr) {
            dr.index_out[b] = j; dr.index_out[batch + b] = slot; dr.index_out[2 * (long)batch + b] = r; dr.index_out[3 * (long)batch + b] = e;
        }
    } else {
        j = age[b]; slot = slot_[b]; r = rank_[b]; e = env_[b];
    }
    const int D = g.obs_dim;
    const long row_off = e * D, aux_off = g.obs_floats + e * 4;
    auto block_of = [&](const RingPos& p) { return rv.store + ((long)p.chunk * g.world + r) * g.section + (long)p.blk * g.block; };
    RingPos at{(int)((unsigned)slot / (unsigned)g.slots_per_chunk), 0};
    at.blk = (int)slot - at.chunk * g.slots_per_chunk;
    // frame f (0 = oldest) sits k-1-f slots behind `slot`; it belongs to the transition's episode iff no LATER frame of the
    // stack is the first observation of an episode (aux.done of a slot: the step INTO it ended one) and the ring reaches back
    // that far (age j counts slots from the oldest sampleable one).
    const float* blk[KB];                                     // blk[i]: the block of the slot i behind `slot` (frame k-1-i)
    {
        RingPos p = at;
#pragma unroll
        for (int i = 0; i < KB; i++) { blk[i] = block_of(p); if (i + 1 < k) p.back(g); }     // (i >= k: repeats the oldest frame's block, never stored)
    }
    RingPos nx = at;
    nx.forward(g);
    const float* nblk = block_of(nx);
    // ---- all loads -------------------------------------------------------------------------------------------------------
    // the sample's small words in ONE vector load (a scalar load per word would be waited for one at a time, ahead of the rows):
    // lane i < KB: aux.done of the slot i behind `slot` (is it the first observation of an episode?); lanes 16..19: the successor's
    // aux words -- the step OUT of `slot` keeps its action / reward / done / ticket with the NEXT slot; lane 20: the terminal count
    // of the successor's chunk
    const float* pt = rv.store + ((long)nx.chunk * g.world + r) * g.section;
    const float* q = blk[0] + aux_off + 2;
#pragma unroll
    for (int i = 1; i < KB; i++) if (lane == i) q = blk[i] + aux_off + 2;
    if (lane >= 16 && lane < 20) q = nblk + aux_off + (lane - 16);
    if (lane == 20) q = pt + g.count_off;
    float v[KB + 1][4];
    auto load_rows = [&](int c0) {
        int cc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int c = c0 + lane + 64 * u; cc[u] = c < D ? c : D - 1; }      // clamped: loads need no branch
#pragma unroll
        for (int i = 0; i < KB; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) v[i][u] = blk[i][row_off + cc[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) v[KB][u] = nblk[row_off + cc[u]];
    };
    auto store_rows = [&](int c0, unsigned long long starts) {
#pragma unroll
        for (int i = 0; i < KB; i++) {
            if (i < k) {
                const int f = k - 1 - i;
                // a LATER frame (a smaller i) starting an episode cuts frame f off; frame f's own flag only matters to older frames
                const bool later_start = (starts & ((1ull << i) - 1ull)) != 0ull;
                const float keep = (!later_start && (j - i) >= 0) ? 1.0f : 0.0f;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int c = c0 + lane + 64 * u;
                    if (c < D) {
                        const float val = v[i][u] * keep;          // (a product like the tensor expression: keeps the sign of a zero)
                        obs_out[((size_t)b * k + f) * D + c] = val;
                        if (f >= 1) next_out[((size_t)b * k + f - 1) * D + c] = val;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int c = c0 + lane + 64 * u;
            if (c < D) next_out[((size_t)b * k + k - 1) * D + c] = v[KB][u];
        }
    };
    // the first 256 floats of every row (all of it for the reference's 153 or 253), THEN the small words, in program order: the
    // first wait is for everything at once.  (With the rows inside a loop over passes, the compiler hoists the episode-boundary
    // ballot -- and the wait for the small words -- in front of the loop, i.e. in front of the row loads.)
    load_rows(0);
    const float meta = *q;
    // (an unconditional use of every row word here: without it the compiler sinks the successor row's loads into the guarded
    //  stores at the end, behind the wait for the small words -- a second trip to memory)
#pragma unroll
    for (int i = 0; i <= KB; i++) {
#pragma unroll
        for (int u = 0; u < 4; u++) asm volatile("" : "+v"(v[i][u]));
    }
    const unsigned long long starts = __ballot(meta > 0.5f) & ((1ull << k) - 1ull);       // bit i: the slot i behind is an episode's first
    store_rows(0, starts);
    for (int c0 = 256; c0 < D; c0 += 256) { load_rows(c0); store_rows(c0, starts); }
    // the newest frame of next_obs is the successor slot's row -- written above unconditionally -- except after an auto-reset (about
    // one sample in an episode's length): then it is the terminal row the step kernel put into the chunk's terminal section
    // (row = ticket mod T), copied over it, unless later episode ends of that chunk have overwritten that (count - ticket > T):
    // then the transition is reported invalid
    const int meta_bits = __float_as_int(meta);
    const float ax_action = __int_as_float(__builtin_amdgcn_readlane(meta_bits, 16)), ax_reward = __int_as_float(__builtin_amdgcn_readlane(meta_bits, 17));
    const bool done = __int_as_float(__builtin_amdgcn_readlane(meta_bits, 18)) > 0.5f;
    const int32_t ticket = __builtin_amdgcn_readlane(meta_bits, 19);
    const long count = (long)__builtin_amdgcn_readlane(meta_bits, 20);
    const bool have_term = done && ticket >= 0 && (count - (long)ticket) <= (long)g.terminal_rows;
    if (have_term) {                                          // wave-uniform
        const float* last = pt + g.term_off + (long)(ticket % g.terminal_rows) * D;
        for (int c = lane; c < D; c += 64) next_out[((size_t)b * k + k - 1) * D + c] = last[c];
    }
    if (lane == 0) {
        action_out[b] = (int64_t)ax_action;
        reward_out[b] = ax_reward;
        done_out[b] = done ? 1 : 0;
        valid_out[b] = (!done || have_term) ? 1 : 0;
    }
}

template <bool kDraw, typename... A>
void launch_gather(int k, dim3 grid, hipStream_t s, A... a) {
    if (k <= 4) uav_ring_gather_kernel<kDraw, 4><<<grid, dim3(256), 0, s>>>(a...);
    else if (k <= 10) uav_ring_gather_kernel<kDraw, 10><<<grid, dim3(256), 0, s>>>(a...);
    else uav_ring_gather_kernel<kDraw, kMaxStack><<<grid, dim3(256), 0, s>>>(a...);
}

}  // namespace uavenv

extern "C" int uavenv_ring_gather_stacked(const float* store_dev, const UavRingLayout* layout, const int64_t* age_dev,
                                          const int64_t* slot_dev, const int64_t* rank_dev, const int64_t* env_dev, int32_t batch,
                                          int32_t num_frames, float* obs_out_dev, float* next_obs_out_dev, int64_t* action_out_dev,
                                          float* reward_out_dev, uint8_t* done_out_dev, uint8_t* valid_out_dev, void* stream) {
    if (!store_dev || !layout || !age_dev || !slot_dev || !rank_dev || !env_dev || !obs_out_dev || !next_obs_out_dev ||
        !action_out_dev || !reward_out_dev || !done_out_dev || !valid_out_dev) return UAVENV_E_INVALID;
    const UavRingLayout& g = *layout;
    if (batch < 1 || num_frames < 1 || num_frames > uavenv::kMaxStack || g.num_chunks < 1 || g.world < 1 || g.slots_per_chunk < 1 ||
        g.envs < 1 || g.obs_dim < 1 || g.terminal_rows < 1 || g.obs_floats < g.envs * g.obs_dim || g.block < g.obs_floats + 4 * g.envs ||
        g.term_off < g.slots_per_chunk * g.block || g.count_off < g.term_off + g.terminal_rows * g.obs_dim || g.section <= g.count_off)
        return UAVENV_E_INVALID;
    uavenv::RingView rv{store_dev, g};
    uavenv::launch_gather<false>(num_frames, dim3((unsigned)((batch + 3) / 4)), (hipStream_t)stream,
        rv, age_dev, slot_dev, rank_dev, env_dev, uavenv::RingDraw{nullptr, nullptr, 0, nullptr}, batch, num_frames, obs_out_dev, next_obs_out_dev,
        action_out_dev, reward_out_dev, done_out_dev, valid_out_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}

// replaces: ReplayBuffer.sample INCLUDING the draw -- the ~10 small launches PyTorch spends on drawing (slot, rank, environment) for a
// batch (rand / randint / scaling / modulo) fold into the gather.  window_dev int64 [2] = (number of sampleable slots, ring position
// of the oldest one) and counter_dev float [1] (a number that differs from draw to draw, e.g. the optimiser's step count) live in
// device memory, so that a captured launch draws a fresh batch at every replay; index_out_dev int64 [4][batch] (nullable) receives
// the draw (age, slot, rank, environment).
extern "C" int uavenv_ring_sample_stacked(const float* store_dev, const UavRingLayout* layout, const int64_t* window_dev,
                                          const float* counter_dev, uint64_t seed, int32_t batch, int32_t num_frames, float* obs_out_dev,
                                          float* next_obs_out_dev, int64_t* action_out_dev, float* reward_out_dev, uint8_t* done_out_dev,
                                          uint8_t* valid_out_dev, int64_t* index_out_dev, void* stream) {
    if (!store_dev || !layout || !window_dev || !counter_dev || !obs_out_dev || !next_obs_out_dev || !action_out_dev || !reward_out_dev ||
        !done_out_dev || !valid_out_dev) return UAVENV_E_INVALID;
    const UavRingLayout& g = *layout;
    if (batch < 1 || num_frames < 1 || num_frames > uavenv::kMaxStack || g.num_chunks < 1 || g.world < 1 || g.slots_per_chunk < 1 ||
        g.envs < 1 || g.obs_dim < 1 || g.terminal_rows < 1 || g.obs_floats < g.envs * g.obs_dim || g.block < g.obs_floats + 4 * g.envs ||
        g.term_off < g.slots_per_chunk * g.block || g.count_off < g.term_off + g.terminal_rows * g.obs_dim || g.section <= g.count_off)
        return UAVENV_E_INVALID;
    uavenv::RingView rv{store_dev, g};
    uavenv::launch_gather<true>(num_frames, dim3((unsigned)((batch + 3) / 4)), (hipStream_t)stream,
        rv, (const int64_t*)nullptr, (const int64_t*)nullptr, (const int64_t*)nullptr, (const int64_t*)nullptr, uavenv::RingDraw{window_dev, counter_dev, seed, index_out_dev}, batch, num_frames, obs_out_dev,
        next_obs_out_dev, action_out_dev, reward_out_dev, done_out_dev, valid_out_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}
