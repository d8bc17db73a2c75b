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

// one wavefront per sample; everything about the sample is wave-uniform, the lanes copy row elements
template <bool kDraw>
__global__ __launch_bounds__(256) void uav_ring_gather_kernel(RingView rv, const int64_t* __restrict__ age, const int64_t* __restrict__ slot_,
                                                              const int64_t* __restrict__ rank_, const int64_t* __restrict__ env_, RingDraw dr,
                                                              int32_t batch,
                                                              int32_t k, float* __restrict__ obs_out, float* __restrict__ next_out,
                                                              int64_t* __restrict__ action_out, float* __restrict__ reward_out,
                                                              uint8_t* __restrict__ done_out, uint8_t* __restrict__ valid_out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
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
        slot = (oldest + j) % cap;
        if (lane == 0 && dr.index_out != nullptr) {
            dr.index_out[b] = j; dr.index_out[batch + b] = slot; dr.index_out[2 * (long)batch + b] = r; dr.index_out[3 * (long)batch + b] = e;
        }
    } else {
        j = age[b]; slot = slot_[b]; r = rank_[b]; e = env_[b];
    }
    const int D = g.obs_dim;
    // frame f (0 = oldest) sits k-1-f slots behind `slot`; it belongs to the transition's episode iff no LATER frame of the
    // stack is the first observation of an episode (aux.done of a slot: the step INTO it ended one) and the ring reaches back
    // that far (age j counts slots from the oldest sampleable one).  All k episode-start flags are fetched first (lane f reads
    // frame f's), so that the row copies below are independent loads instead of a chain of k dependent ones.
    float* orow = obs_out + (size_t)b * k * D;
    float* nrow = next_out + (size_t)b * k * D;
    float start_flag = 0.0f;
    if (lane < k) start_flag = rv.aux(((((slot - (k - 1 - lane)) % cap) + cap) % cap), r, e)[2];
    const unsigned long long starts = __ballot(start_flag > 0.5f);            // bit f: frame f is the first of an episode
#pragma unroll 4
    for (int f = 0; f < k; f++) {
        const long back = k - 1 - f;
        const long fs = (((slot - back) % cap) + cap) % cap;
        const bool later_start = (starts >> (f + 1)) != 0ull;                 // (frame f's own flag only matters to the frames older than f)
        const float keep = (!later_start && (j - back) >= 0) ? 1.0f : 0.0f;
        const float* src = rv.obs(fs, r, e);
        for (int c = lane; c < D; c += 64) {
            const float v = src[c] * keep;                     // (a product like the tensor expression: keeps the sign of a zero)
            orow[f * D + c] = v;
            if (f >= 1) nrow[(f - 1) * D + c] = v;
        }
    }
    // the step OUT of `slot`: its action / reward / done are stored with the NEXT slot; after an auto-reset the true next
    // observation is the terminal row the step kernel put into the chunk's terminal section (row = ticket mod T), unless later
    // episode ends of that chunk have overwritten it (count - ticket > T): then the transition is reported invalid
    const long nxt = (slot + 1) % cap;
    const float* ax = rv.aux(nxt, r, e);
    const bool done = ax[2] > 0.5f;
    const int32_t ticket = __float_as_int(ax[3]);
    const float* pt = rv.part(nxt / g.slots_per_chunk, r);
    const long count = (long)reinterpret_cast<const int32_t*>(pt)[g.count_off];
    const bool have_term = done && ticket >= 0 && (count - (long)ticket) <= (long)g.terminal_rows;
    const long row = (long)(ticket < 0 ? 0 : ticket) % g.terminal_rows;
    const float* last = have_term ? pt + g.term_off + row * D : rv.obs(nxt, r, e);
    for (int c = lane; c < D; c += 64) nrow[(k - 1) * D + c] = last[c];
    if (lane == 0) {
        action_out[b] = (int64_t)ax[0];
        reward_out[b] = ax[1];
        done_out[b] = done ? 1 : 0;
        valid_out[b] = (!done || have_term) ? 1 : 0;
    }
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
    uavenv::uav_ring_gather_kernel<false><<<dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
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
    uavenv::uav_ring_gather_kernel<true><<<dim3((unsigned)((batch + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        rv, nullptr, nullptr, nullptr, nullptr, uavenv::RingDraw{window_dev, counter_dev, seed, index_out_dev}, batch, num_frames, obs_out_dev,
        next_obs_out_dev, action_out_dev, reward_out_dev, done_out_dev, valid_out_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_E_HIP;
}
