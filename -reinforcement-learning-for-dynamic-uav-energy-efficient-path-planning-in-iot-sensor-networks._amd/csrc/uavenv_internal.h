// uavenv_internal.h -- structures shared by the kernels (uavenv_kernels.hip) and the C ABI
// (uavenv_capi.hip).  Not part of the public boundary (that is include/uavenv.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uavenv.h"

namespace uavenv {

#ifndef UAV_BLOCK
#define UAV_BLOCK 1024
#endif
#ifndef UAV_LDS_PAD       // dev-only occupancy cap for launch-shape experiments (tools/ablate.py)
#define UAV_LDS_PAD 0
#endif
// 16 wavefronts per workgroup = one workgroup per CU at the headline size, wavefront w on SIMD (w + const) mod 4
// (tools/wave_map.py): the step kernel deals the expensive collect-action environments of a workgroup round-robin over
// the four SIMDs (StepArgs::balance).
constexpr int kBlockThreads = UAV_BLOCK;
// Every other kernel, and the step kernel on batches too small to give each CU 16 waves, uses 4-wave workgroups (they
// spread over more CUs); the padding granule stays kBlockThreads / G environments, which both sizes divide.
constexpr int kSmallBlockThreads = 256;

// flag word per sensor (UAVENV_F_FLAGS)
constexpr uint32_t kSfMask = 15u, kAvgValid = 16u, kVisited = 32u, kDataCollected = 64u;

// Host-precomputed constants, passed to every kernel BY VALUE (kernarg segment -> scalar loads).
// Each derived value is computed on the host with the reference's own expression, cited.
// Field order = order of use in the step kernel (every-step fields first, then collect-only, then
// truncation/reset-only), so the scalar loads that fetch them do not drag cold fields into SGPRs.
struct Consts {
    // ---- every step ------------------------------------------------------------------------
    uint64_t seed;
    double rate, bmax, thr;
    double inv_bmax, inv_maxb, inv_rate;   // correctly rounded reciprocals for div_const() (0 if the divisor is 0)
    double sigma, lambda, one_minus_lambda, tx_power;
    double d_break;        // iot_sensors.py:174  (4*pi*ht*hr)/0.345
    double c_fs;           // iot_sensors.py:179  20*log10(868)
    double fspl_off;       //                     28
    double c_ht, c_hr;     // iot_sensors.py:183  20*log10(ht), 20*log10(hr)
    double sf_thr[4];
    double maxb;
    double alive_level;    // uav.py:224          0.02*max_battery
    double e_move, e_coll; // uav.py:176,180      (P*t)/3600
    double p_step, r_move, p_boundary, p_battery, p_loss;
    float  alt2;           // iot_sensors.py:164  altitude**2 as float32
    int32_t max_steps, fps, obs_dim, obs_slots, use_ema;
    uint32_t flags;
    // ---- collect steps -----------------------------------------------------------------------
    double coll_dur, p_cycle, noise_floor, cap_thr;
    double e_hover;        // uav.py:204          (P_hover*duration)/3600
    double used_hover;     // uav_env.py:530      (P_hover/3600)*duration
    double r_byte, r_new, r_done, r_urg, p_revisit, p_collision, p_hover, p_starvation;
    // ---- truncation / reset / DomainRand extras --------------------------------------------------
    double p_unvisited, p_starved, cr_thr;
    double fill_lo, fill_span;
    double min_start_dist, prox_eta, jain_weight;
    int32_t max_tries, n_grid_choices;
    int32_t gw[8], gh[8];
    int32_t n_max, pad0;   // the handle's sensor count (no environment has more): lanes >= n_max never hold a sensor; host side only
    double inv_small[65];  // RN(1/k), k = 1..64: divisions by a sensor / winner count go through div_const()
};

// All per-sensor arrays live in ONE allocation, array k at byte offset kOff_k * S where
// S = padded_envs * G lanes:  pos_x 0, pos_y 4, buffer 8, gen 16, tx 24, lost 32, avg 40, flags 48 (x S bytes).
// One base pointer (+ S) instead of eight pointers keeps the kernel's SGPR budget for constants.
struct Ptrs {
    char* sensor_base;
    uint64_t lanes;             // S
    UavEnvRecord* rec;
    UavEnvEpisodeStats* stats;
    const float* step_tape;     // [E][6][G] or nullptr
    const float* reset_tape;    // [E][4][G] or nullptr
    uint32_t* status;           // device word: OR of per-env status bits
    unsigned long long* stamps; // diagnostic build only (-DUAVENV_STAMPS): 8 words per wavefront, else unused
    // optional terminal snapshot (uavenv_enable_terminal_snapshot): what `_get_info()` of the episode's LAST step is made of
    // (uav_env.py:676-700) -- the record as it stood before the auto-reset, and buffer / generated / transmitted per sensor
    UavEnvRecord* term_rec;     // [E] or nullptr
    double* term_sensors;       // [E][3][G] = (data_buffer, total_data_generated, total_data_transmitted) or nullptr
};
constexpr uint64_t kOffPosX = 0, kOffPosY = 4, kOffBuffer = 8, kOffGen = 16, kOffTx = 24, kOffLost = 32, kOffAvg = 40,
                   kOffFlags = 48, kSensorBytesPerLane = 52;

// Where a step puts its results: ONE contiguous block of the argument struct, so that the kernels fetch all of it with a
// single wide scalar load (load_out_args) instead of one scalar load + wait per pointer at the point of use.
struct OutArgs {
    int32_t* actions_out;
    float* obs;
    double* reward;
    float* reward32;
    uint8_t* done;
    float* term_obs;
    // optional terminal-observation pool (uavenv_set_terminal_pool): a truncating env takes the ticket
    // t = atomicAdd(term_counter, 1), writes its terminal row to term_pool[t % term_rows] and the row index to
    // term_index[env] (-1 if not done); the ticket itself goes into the aux block
    float* term_pool;
    uint32_t* term_counter;
    int32_t* term_index;
    float* aux;                 // optional [E][4] = (action, reward, done as float; terminal ticket or -1 as int32 bits):
                                // the packed remainder of a transition block, so that replay insertion needs no pack kernel
    uint32_t* hint_out;         // null unless the random policy writes next-step hints
    int32_t term_rows;
    int32_t write_through;      // observation and sensor-state stores as `sc1` write-through stores (batches that fill the chip: the
                                // kernel boundary then has no dirty L2 lines to write back); any value gives the same results
};
static_assert(sizeof(OutArgs) == 96, "OutArgs is loaded as 24 dwords");

struct StepArgs {
    const int32_t* actions;     // nullptr => in-kernel random policy
    int32_t num_envs;           // E (arrays are padded to a whole number of workgroups)
    int32_t policy;             // UAVENV_POLICY_*
    // SIMD load balancing (a pure scheduling hint: any value gives the same results).  balance != 0: the wavefronts of
    // a workgroup take its environments collect-actions-first, so collect steps (1.6x the work of a move) spread evenly
    // over the CU's SIMDs.  The "is a collect" bits come from `actions`, or for the in-kernel random policy from
    // hint_in[wave unit] -- written by the PREVIOUS launch into its hint_out (double-buffered by the host: a launch never
    // writes the buffer its own wavefronts read).
    const uint32_t* hint_in;    // never null (one word per padded environment)
    int32_t balance;
    int32_t reserved;
    OutArgs out;
};

struct ResetArgs {
    const uint8_t* mask;
    float* obs;
    int32_t num_envs;
};

// launchers (uavenv_kernels.hip)
hipError_t launch_init(int G, int padded_envs, const Consts& c, const Consts* dev_consts, const Ptrs& p, uint32_t env_index_base,
                       int32_t grid_w, int32_t grid_h, int32_t n, float start_x, float start_y, hipStream_t s);
hipError_t launch_reset(int G, int padded_envs, const Consts& c, const Consts* dev_consts, const Ptrs& p, const ResetArgs& a, hipStream_t s);
bool step_uses_big_workgroups(int G, int padded_envs);   // 16-wave workgroups + SIMD load balancing, else 4-wave
// default_consts: the handle's constants equal uavenv_default_config()'s bit for bit (consts_are_default) -> the variant
// that reads them as instruction literals may run (it is also restricted to the lean configuration)
hipError_t launch_step(int G, int padded_envs, const Consts& c, const Consts* dev_consts, const Ptrs& p, const StepArgs& a,
                       bool default_consts, hipStream_t s);
hipError_t launch_rollout(int G, int padded_envs, const Consts& c, const Consts* dev_consts, const Ptrs& p, const StepArgs& a,
                          int32_t num_steps, bool default_consts, hipStream_t s);
hipError_t launch_frame_stack(float* stacked, const float* obs, const uint8_t* done, const float* terminal_obs,
                              float* terminal_stacked, int32_t num_envs, int32_t k, int32_t D, hipStream_t s);
hipError_t launch_dump_noise(int G, int padded_envs, const Consts& c, const Consts* dev_consts, const Ptrs& p, float* step_tape,
                             float* reset_tape, int32_t num_envs, hipStream_t s);

}  // namespace uavenv
