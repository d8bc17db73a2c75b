/*
 * uavenv.h -- C ABI of the MI355X-native batched UAV-IoT environment (libuavenv_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of the reference: the Gymnasium environment
 * reset()/step() chain of
 *     /root/reference/src/environment/uav_env.py      (UAVEnvironment, :240)
 *     /root/reference/src/environment/iot_sensors.py  (IoTSensor, :10)
 *     /root/reference/src/environment/uav.py          (UAV, :63)
 *     /root/reference/src/rewards/reward_function.py  (RewardFunction, :4)
 * batched over E independent environment instances and executed by hand-written HIP kernels for
 * gfx950 (one 16/32/64-lane group of a wavefront per environment, one lane per sensor).
 *
 * Conventions
 *   - plain C: opaque handle, plain pointers and sizes, no C++/torch types;
 *   - every function returns 0 on success or a negative UAVENV_E_* code and never throws or aborts;
 *     uavenv_last_error() gives the text;
 *   - `*_dev` pointers are device (HBM) pointers valid on the handle's device; `stream` is a
 *     hipStream_t passed as void* (NULL = the null stream); calls are asynchronous on that stream
 *     unless stated otherwise;
 *   - per-sensor arrays are laid out [num_envs][lane_stride] where lane_stride =
 *     uavenv_lane_stride() (16, 32 or 64: the lane-group width that holds max_sensors; the environment variable
 *     UAVENV_LANE_GROUP=32|64, read by uavenv_create, asks for a wider group than the sensor count needs -- at 4096
 *     environments x 20 sensors one environment per wavefront steps in 7.2 us instead of 7.8 (tools/lane_group_probe.py),
 *     while fused rollouts and larger batches are faster with the narrow group, which therefore stays the default).
 *
 * The reference interface each entry point replaces is cited next to it.
 */
#ifndef UAVENV_H
#define UAVENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVENV_ABI_VERSION 2

/* error codes */
#define UAVENV_OK              0
#define UAVENV_E_INVALID      -1   /* bad argument / config                                   */
#define UAVENV_E_HIP          -2   /* a HIP runtime call failed                               */
#define UAVENV_E_ACTION       -3   /* an action outside 0..4 was seen (uav_env.py:468)        */
#define UAVENV_E_ALLOC        -4

/* UavEnvConfig.flags */
#define UAVENV_FLAG_RANDOM_LAYOUT  1u   /* dqn.py:340-360 fresh uniform layout, empty buffers, SF inherited from the old sensor 0 */
#define UAVENV_FLAG_FAR_START      2u   /* dqn.py:364-365,375-403 rejection-sampled UAV start            */
#define UAVENV_FLAG_PROX_SHAPING   4u   /* dqn.py:417-425 proximity shaping reward                       */
#define UAVENV_FLAG_JAIN_BONUS     8u   /* dqn.py:434-442 per-step Jain's fairness bonus (once a sensor has generated data) */
#define UAVENV_FLAG_AUTO_RESET    16u   /* SB3 VecEnv semantics: reset inside step() on truncation        */

/* Every constant of the path is a runtime parameter.  Defaults (uavenv_default_config)
 * are the reference's training configuration: BASE_ENV_CONFIG (agents/dqn/dqn.py:1068-1075) over the
 * UAVEnvironment kwargs (uav_env.py:266-287), IoTSensor (iot_sensors.py:39-57), UAV (uav.py:93-94,125)
 * and RewardFunction (reward_function.py:7-27 as overridden at uav_env.py:339-344). */
typedef struct UavEnvConfig {
    uint32_t struct_size;              /* = sizeof(UavEnvConfig); checked by uavenv_create           */
    int32_t  grid_w, grid_h;           /* uav_env.py:268 grid_size                                    */
    int32_t  num_sensors;              /* uav_env.py:270 (<= 64)                                      */
    int32_t  max_steps;                /* uav_env.py:280                                              */
    int32_t  include_sensor_positions; /* uav_env.py:286 -> 5 features per sensor instead of 3        */
    int32_t  pad_sensors;              /* dqn.py:286-298: zero-pad the observation to this many slots */
    uint32_t flags;                    /* UAVENV_FLAG_*                                               */
    int32_t  max_start_tries;          /* dqn.py NAV_CONFIG["max_start_tries"]                        */
    int32_t  use_ema_adr;              /* iot_sensors.py:54                                           */
    int32_t  num_grid_choices;         /* dqn.py:280-283 curriculum grid list; 0 = keep grid_w/grid_h */
    int32_t  grid_choices_w[8], grid_choices_h[8];
    uint64_t seed;                     /* Philox key: replaces the reference's three global RNG streams */
    double data_generation_rate, max_buffer_size, rssi_threshold, duty_cycle;      /* uav_env.py:271-276 */
    double start_x, start_y, max_battery, collection_duration;                     /* uav_env.py:277-279 */
    double tx_power_dbm, noise_floor_dbm, uav_altitude, sensor_height, wavelength, freq_mhz,
           fspl_offset_db, adr_lambda, shadowing_std_db, capture_threshold_db;     /* iot_sensors.py:39-57,147-197; uav_env.py:567 */
    double sf_thresholds[4];           /* iot_sensors.py:32-37 -> SF 7, 9, 11, 12                      */
    double fill_lo, fill_hi;           /* uav_env.py:410 initial buffer fill U(lo, hi)                 */
    double power_move, power_hover, alive_fraction;                                /* uav.py:93-94, :224 */
    double reward_per_byte, reward_new_sensor, reward_completion, reward_urgency_reduction,
           reward_movement, penalty_revisit, penalty_boundary, penalty_collision, penalty_battery,
           penalty_hover, penalty_step, penalty_data_loss, penalty_starvation, penalty_unvisited,
           penalty_starved, starvation_cr_threshold;                               /* reward_function.py:7-27 */
    double min_start_dist, prox_eta, jain_weight;                                  /* dqn.py NAV_CONFIG, :442 */
} UavEnvConfig;

/* Per-environment scalar state as stored in HBM (128 bytes, one record per environment).
 * Mirrors the attributes callers of the reference reach into: env.uav.position / .battery,
 * env.current_step, env.total_reward, env.total_data_collected, env.capture_effect_triggers,
 * env.boundary_hits, env.edge_steps (uav_env.py:293-303, 676-700). */
typedef struct UavEnvRecord {
    double  battery;               /* uav.py:120 (Wh)                                     */
    double  total_reward;          /* uav_env.py:487 (unshaped)                           */
    double  total_data_collected;  /* uav_env.py:588                                      */
    double  last_step_bytes;       /* uav_env.py:607                                      */
    double  prev_dist_nearest;     /* dqn.py:368, :425                                    */
    double  episode_return;        /* sum of RETURNED (shaped) rewards: Monitor's info["episode"]["r"] */
    float   uav_x, uav_y;          /* uav.py:113 (float32 grid units)                     */
    float   start_x, start_y;      /* uav.py:112                                          */
    int32_t current_step;          /* uav_env.py:439                                      */
    uint32_t episode;              /* number of resets - 1 (Philox counter word 1)        */
    int32_t capture_triggers, boundary_hits, edge_steps, collisions_total;
    int32_t first_full_coverage_step;   /* dqn.py:428-431; -1 = not yet                   */
    int32_t grid_w, grid_h, num_sensors;
    uint32_t env_index;            /* GLOBAL environment index (Philox counter word 0)    */
    uint32_t status;               /* bit 0: an invalid action was seen                   */
    double  inv_grid_w, inv_grid_h;/* 1/grid_w, 1/grid_h (kept by the library: lets the kernel divide by multiplying) */
} UavEnvRecord;

/* Written once per finished episode when UAVENV_FLAG_AUTO_RESET is set (what DomainRandEnv.reset
 * snapshots into last_episode_stats, dqn.py:305-331, plus Monitor's r/l). */
typedef struct UavEnvEpisodeStats {
    double  episode_return, total_reward, total_generated, total_collected, total_lost,
            battery_remaining, jains_index, fairness_std;
    int32_t length, sensors_visited, num_sensors, grid_w, grid_h, first_full_coverage_step;
    uint32_t episode, valid;
} UavEnvEpisodeStats;

/* fields for uavenv_get_state / uavenv_set_state */
enum {
    UAVENV_F_POS_X = 0,    /* float  [E][stride]  iot_sensors.py:66 position[0]          */
    UAVENV_F_POS_Y = 1,    /* float  [E][stride]                                        */
    UAVENV_F_BUFFER = 2,   /* double [E][stride]  data_buffer                           */
    UAVENV_F_GEN = 3,      /* double [E][stride]  total_data_generated                  */
    UAVENV_F_TX = 4,       /* double [E][stride]  total_data_transmitted                */
    UAVENV_F_LOST = 5,     /* double [E][stride]  total_data_lost                       */
    UAVENV_F_AVG_RSSI = 6, /* double [E][stride]  avg_rssi (valid iff flag bit 4)       */
    UAVENV_F_FLAGS = 7,    /* uint32 [E][stride]  bits 0-3 SF, 4 avg_valid, 5 visited, 6 data_collected */
    UAVENV_F_RECORD = 8,   /* UavEnvRecord [E]                                          */
    UAVENV_F_EPISODE_STATS = 9, /* UavEnvEpisodeStats [E]                              */
    UAVENV_F_TERM_RECORD = 10,  /* UavEnvRecord [E]: the record of each env's last TERMINAL step (uavenv_enable_terminal_snapshot) */
    UAVENV_F_TERM_SENSORS = 11, /* double [E][3][stride]: data_buffer, total_data_generated, total_data_transmitted at that step  */
    UAVENV_F_COUNT = 12
};

/* The in-kernel noise generator: Philox4x32 (Salmon et al., SC'11) with this many rounds.  7 is the smallest round count the
 * authors report as passing BigCrush ("Crush-resistant"); 10 is Random123's default safety margin.  The three rounds cost
 * 0.4 us of an 8 us step launch (24 multiply / xor instructions per lane and call), so the library uses 7 -- the oracle
 * (oracle/uavenv_oracle.c) reads the same constant, tests/test_noise_spec.py pins both round counts to Random123's vectors. */
#define UAVENV_PHILOX_ROUNDS 7

/* noise-tape slots (float [E][slots][stride]); NULL tape = in-kernel Philox4x32-7 */
enum { UAVENV_TAPE_ZA = 0, UAVENV_TAPE_ZB, UAVENV_TAPE_U, UAVENV_TAPE_ZC, UAVENV_TAPE_ZD, UAVENV_TAPE_ZE,
       UAVENV_TAPE_ZP,          /* in-range sample drawn by a heuristic policy before the step */
       UAVENV_TAPE_STEP_SLOTS };

/* action sources for uavenv_step_policy / uavenv_rollout */
#define UAVENV_POLICY_ACTIONS            0   /* actions_dev                                                   */
#define UAVENV_POLICY_RANDOM             1   /* uniform random (lane 0's spare Philox word), BASELINE.md sec. 4 */
#define UAVENV_POLICY_NEAREST            2   /* greedy_agents.py:73-100  NearestSensorGreedy, on device       */
#define UAVENV_POLICY_MAX_THROUGHPUT_V2  3   /* greedy_agents.py:105-216 MaxThroughputGreedyV2, on device     */
/* reset tape: buffer-fill uniform, (zD, zE) of the reset observation, and zS (element 0 only): the ADR sample the DISCARDED
 * reset observation of DomainRandEnv.reset draws for the OLD sensor 0, whose SF the fresh sensors inherit (dqn.py:340-351) */
enum { UAVENV_RTAPE_FILL = 0, UAVENV_RTAPE_ZD, UAVENV_RTAPE_ZE, UAVENV_RTAPE_ZS, UAVENV_RTAPE_SLOTS };

typedef struct UavEnv UavEnv;

/* ---- configuration ------------------------------------------------------------------------- */
int uavenv_abi_version(void);
/* replaces: the defaults of UAVEnvironment.__init__ (uav_env.py:266-287) + BASE_ENV_CONFIG (dqn.py:1068-1075) */
int uavenv_default_config(UavEnvConfig* cfg);
/* replaces: observation_space.shape[0] (uav_env.py:348-355; padded form dqn.py:249-254) */
int uavenv_obs_dim(const UavEnvConfig* cfg);

/* ---- lifetime ------------------------------------------------------------------------------ */
/* replaces: UAVEnvironment.__init__ (uav_env.py:266-361) for `num_envs` instances.  env_index_base is
 * the GLOBAL index of this shard's first environment (multi-GPU: rank r owns [base, base+num_envs)), so
 * results do not depend on how environments are sharded.  Sensor layouts are drawn per environment from
 * the Philox key (replaces _generate_uniform_sensor_positions, uav_env.py:366-374). */
int uavenv_create(const UavEnvConfig* cfg, int32_t num_envs, uint32_t env_index_base, int32_t device, UavEnv** out);
int uavenv_destroy(UavEnv* env);                       /* replaces: close() (uav_env.py:893-895) */
const char* uavenv_last_error(const UavEnv* env);      /* env may be NULL: error of a failed create */

int uavenv_num_envs(const UavEnv* env);
int uavenv_lane_stride(const UavEnv* env);
int uavenv_env_obs_dim(const UavEnv* env);

/* ---- per-environment parameters (BASELINE config 5: mixed grid / sensor-count sweeps) ------- */
/* host arrays of num_envs entries, each nullable; take effect immediately (call before uavenv_reset);
 * num_sensors[i] must be in 1..cfg.num_sensors (the observation keeps cfg.num_sensors slots, zero padded) */
int uavenv_set_env_params(UavEnv* env, const int32_t* grid_w, const int32_t* grid_h, const int32_t* num_sensors);
/* replaces: the sensor_positions kwarg (uav_env.py:269); host float [E][stride] */
int uavenv_set_positions(UavEnv* env, const float* pos_x, const float* pos_y);
/* replaces: reset(seed=...) / VecEnv.seed(): re-keys all randomness.
 * The seed travels to the step kernels as a launch ARGUMENT (it is preloaded into SGPRs with the wave launch), and a captured
 * HIP graph keeps the arguments it was captured with: graphs of step / rollout launches captured before this call go on drawing
 * with the OLD seed and must be captured again.  The same holds for everything else a launch takes by value: the tape pointers
 * (uavenv_set_noise_tape), the terminal pool, the aux output and the terminal snapshot.  (uavenv_amd: BatchedUAVEnv.launch_epoch
 * counts these calls; TransitionRing.replay_chunk and DQNLearner refuse / re-capture graphs of an older epoch.) */
int uavenv_set_seed(UavEnv* env, uint64_t seed);
/* replaces: attribute writes on a LIVE environment's sensors / UAV / reward function, e.g. `s.shadowing_std_db = sigma` for
 * every sensor in agents/dqn/dqn_evaluation_results/sim_to_real_sweep.py:109-117: read the handle's configuration, change
 * fields, hand it back.  The constants are re-derived and take effect with the next launch (the call synchronises the
 * device).  num_sensors, pad_sensors and include_sensor_positions (buffer and observation sizes) must stay as created
 * (UAVENV_E_INVALID otherwise).  The state is left alone: per-environment grids / sensor counts (uavenv_set_env_params)
 * are kept, a smaller max_buffer_size is the caller's to reconcile with the buffers (see uavenv_set_state).  Like
 * uavenv_set_seed it changes launch ARGUMENTS (the literal-constants kernel variant is chosen per launch): re-capture graphs. */
int uavenv_get_config(const UavEnv* env, UavEnvConfig* out);
int uavenv_set_config(UavEnv* env, const UavEnvConfig* cfg);
/* replaces: DomainRandEnv.set_curriculum_stage (dqn.py:258-277): grid list sampled at each reset */
int uavenv_set_grid_choices(UavEnv* env, int32_t count, const int32_t* w, const int32_t* h);

/* ---- noise tape (parity testing) ------------------------------------------------------------ */
/* step_tape_dev: float [E][7][stride] consumed by the next uavenv_step; reset_tape_dev:
 * float [E][4][stride] consumed by uavenv_reset and by auto-resets.  NULL restores Philox. */
int uavenv_set_noise_tape(UavEnv* env, const float* step_tape_dev, const float* reset_tape_dev);
/* writes the tapes the NEXT step (and a reset opening the next episode) would draw from Philox */
int uavenv_dump_noise(UavEnv* env, float* step_tape_out_dev, float* reset_tape_out_dev, void* stream);

/* ---- the hot path --------------------------------------------------------------------------- */
/* replaces: UAVEnvironment.reset (uav_env.py:400-427; DomainRandEnv.reset dqn.py:301-373 under the
 * RANDOM_LAYOUT / FAR_START flags).  mask_dev: uint8 [E], nullable = reset all.
 * obs_out_dev: float [E][obs_dim], rows of unmasked envs are left untouched. */
int uavenv_reset(UavEnv* env, const uint8_t* mask_dev, float* obs_out_dev, void* stream);

/* replaces: UAVEnvironment.step (uav_env.py:429-488; DomainRandEnv.step dqn.py:415-444 under the
 * shaping flags), for all E environments in ONE kernel launch.
 *   actions_dev       int32 [E]           0 UP(+y) 1 DOWN(-y) 2 LEFT(-x) 3 RIGHT(+x) 4 COLLECT (uav_env.py:497)
 *   obs_out_dev       float [E][obs_dim]  observation after the step (after the auto-reset when one happened)
 *   reward_out_dev    double [E]          nullable
 *   reward32_out_dev  float [E]           nullable (what SB3's VecEnv hands the agent)
 *   done_out_dev      uint8 [E]           truncated flag (`terminated` is always False: uav_env.py:471)
 *   terminal_obs_dev  float [E][obs_dim]  nullable; rows written only where done (SB3 "terminal_observation")
 * The action array must stay unmodified until the launch has finished (it is read through the scalar cache).
 * Scheduling note: the wavefronts of a workgroup pick their environments collect-actions-first so that the
 * expensive collect steps spread over a CU's SIMDs; results do not depend on it (UAVENV_NO_BALANCE=1 in the
 * process environment at uavenv_create time keeps the plain mapping, for A/B timing). */
int uavenv_step(UavEnv* env, const int32_t* actions_dev, float* obs_out_dev, double* reward_out_dev,
                float* reward32_out_dev, uint8_t* done_out_dev, float* terminal_obs_dev, void* stream);

/* same, with the uniform-random policy of BASELINE.md section 4 drawn in-kernel (word 3 of lane 0's
 * observation-noise call of the previous step);
 * actions_out_dev (int32 [E], nullable) receives the actions taken. */
int uavenv_step_random(UavEnv* env, int32_t* actions_out_dev, float* obs_out_dev, double* reward_out_dev,
                       float* reward32_out_dev, uint8_t* done_out_dev, float* terminal_obs_dev, void* stream);

/* num_steps consecutive uavenv_step_random launches issued by ONE call (one kernel launch per step, like calling it in a loop:
 * the results are the same bit for bit).  Step k writes its observations to obs_out_dev + k * obs_stride (floats) and, when
 * aux_out_dev is given, its (action, reward, done, ticket) block to aux_out_dev + k * aux_stride instead of the buffer set by
 * uavenv_set_aux_output -- e.g. the slots of one replay-ring chunk; reward32_out_dev / done_out_dev (nullable) are rewritten
 * by every step.  Terminal observations go to the terminal pool if one is set (uavenv_set_terminal_pool), else nowhere.
 * replaces: the `for _ in range(n): obs, r, d, info = env.step(env.action_space.sample())` loop of uav_env.py:935-960 for
 * callers that are launch-bound from Python (12 us per call against a 9 us kernel at 4096 environments). */
int uavenv_step_random_n(UavEnv* env, int32_t num_steps, float* obs_out_dev, int64_t obs_stride, float* aux_out_dev,
                         int64_t aux_stride, float* reward32_out_dev, uint8_t* done_out_dev, void* stream);

/* same, with the action chosen IN the kernel by `policy` (UAVENV_POLICY_RANDOM / _NEAREST / _MAX_THROUGHPUT_V2):
 * replaces agent.select_action(obs) + env.step(action) of the heuristic baselines (greedy_agents.py; used by the
 * curriculum gate dqn.py:456-543), whose is_in_range() samples come from Philox call 5 / tape slot zP. */
int uavenv_step_policy(UavEnv* env, int32_t policy, int32_t* actions_out_dev, float* obs_out_dev, double* reward_out_dev,
                       float* reward32_out_dev, uint8_t* done_out_dev, float* terminal_obs_dev, void* stream);

/* K consecutive steps in ONE launch (SURVEY 8b "uavenv_step_k"): policy UAVENV_POLICY_ACTIONS = open-loop
 * actions (actions_dev int32 [K][E]), otherwise an in-kernel policy (random or heuristic).  Sensor state and the
 * per-environment record stay in registers for the whole launch, but EVERY step still writes its block:
 * obs_out_dev [K][E][obs_dim], reward [K][E], done [K][E], actions_out [K][E] (each nullable), e.g. K
 * consecutive slots of a replay ring.  Bit-identical to K uavenv_step / uavenv_step_random launches.
 * replaces: the `for _ in range(K): env.step(policy(obs))` loop of uav_env.py:935-960 / SB3
 * collect_rollouts for policies that do not read the observation (random warm-up, action replay). */
int uavenv_rollout(UavEnv* env, int32_t num_steps, int32_t policy, const int32_t* actions_dev, int32_t* actions_out_dev,
                   float* obs_out_dev, double* reward_out_dev, float* reward32_out_dev, uint8_t* done_out_dev,
                   float* terminal_obs_dev, void* stream);

/* Optional compact pool for terminal observations (replay buffers need next_obs = terminal observation on
 * truncated transitions, SB3 "terminal_observation"; ~1 in 1500 steps per env truncates).  When set, a
 * truncating env takes the ticket t = atomicAdd(*counter_dev, 1) and writes its terminal row to
 * pool_dev[t % rows][obs_dim] instead of terminal_obs_dev[env]; index_out_dev[env] (int32 [E], nullable) receives
 * that row or -1, and the aux block (below) the ticket itself.  Rows recycle: the row of ticket t is intact while
 * *counter_dev - t <= rows, which lets a consumer DETECT an overwritten row instead of reading another environment's
 * observation.  The caller may move the pool / restart the counter between launches (e.g. one pool section per
 * replay-ring chunk).  pool_dev == NULL restores the per-env terminal_obs_dev rows. */
int uavenv_set_terminal_pool(UavEnv* env, float* pool_dev, int32_t rows, uint32_t* counter_dev, int32_t* index_out_dev);

/* Optional packed remainder of the transition block: when set, every step also writes
 * aux_out_dev [E][4] = (action, reward, done as float32; the terminal-pool ticket, or -1, as int32 bits) -- with
 * obs_out_dev this is the whole (obs, action, reward, done) transition, so inserting into a replay buffer needs no
 * extra pack kernel and the multi-GPU exchange is one all-gather of one contiguous block.  capacity_steps = how many
 * [E][4] blocks the buffer holds: uavenv_rollout writes [K][E][4] and refuses K > capacity_steps (UAVENV_E_INVALID)
 * instead of running past the buffer.  NULL disables. */
int uavenv_set_aux_output(UavEnv* env, float* aux_out_dev, int32_t capacity_steps);

/* Terminal snapshot: what `info` of an episode's LAST step is made of.  With UAVENV_FLAG_AUTO_RESET an environment that
 * truncates is reset inside the same launch, so the state `_get_info()` (uav_env.py:676-700) describes -- read by
 * BestByMetricCallback through infos[0] at dqn.py:1150-1155: total_data_collected, battery, sensor_collection_ratios --
 * is gone when the step returns.  When enabled, the step kernels store, for every environment that ends an episode, its
 * record as it stood before the reset (UAVENV_F_TERM_RECORD) and data_buffer / total_data_generated /
 * total_data_transmitted of every sensor (UAVENV_F_TERM_SENSORS); rows of environments that have not ended an episode yet
 * are zero.  Costs nothing on steps that end no episode.  enable = 0 releases the buffers. */
int uavenv_enable_terminal_snapshot(UavEnv* env, int32_t enable);

/* ---- frame stack (the caller directly above the path in the trainer: dqn.py:1278) ------------------ */
/* replaces: SB3 VecFrameStack(n_stack=k).step_wait on device, in place.  stacked_dev float [E][k*obs_dim]
 * (oldest frame first) is shifted left by one frame and obs_dev [E][obs_dim] appended; where done_dev[e]
 * the older frames are zeroed first.  terminal_stacked_dev (nullable, with terminal_obs_dev) receives, for
 * done envs, SB3's stacked "terminal_observation" = [old frames shifted | terminal_obs].  k*obs_dim <= 2560.
 * Needs no UavEnv handle. */
int uavenv_frame_stack(float* stacked_dev, const float* obs_dev, const uint8_t* done_dev, const float* terminal_obs_dev,
                       float* terminal_stacked_dev, int32_t num_envs, int32_t num_frames, int32_t obs_dim, void* stream);

/* ---- replay sampling (the consumer of the transitions the step kernel writes: dqn.py:1083-1089 ReplayBuffer) -------- */
/* The transition ring of replay.py: store[chunk][rank][ slots_per_chunk blocks | terminal rows | count ], all in floats;
 * block = [ envs x obs_dim observations, padded to obs_floats | envs x 4 aux = (action, reward, done, ticket bits) ]. */
typedef struct UavRingLayout {
    int64_t section;            /* floats per (chunk, rank) part                                        */
    int32_t num_chunks, world, slots_per_chunk, envs, obs_dim, terminal_rows;
    int32_t block, obs_floats;  /* floats per block; offset of the aux rows inside a block              */
    int32_t term_off, count_off;/* float offsets of the terminal rows / the int32 counter inside a part */
} UavRingLayout;
/* replaces: ReplayBuffer.sample + VecFrameStack's stacking for a drawn batch, in one launch.  For sample b with ring slot
 * slot[b], rank[b], env[b] and age[b] (slots between the oldest sampleable slot and slot[b]): obs_out[b] = the num_frames
 * frames ending at the slot (oldest first; frames from before the episode start or before the ring start zeroed),
 * next_obs_out[b] = frames 1.. + the next observation (the terminal row after an auto-reset), action / reward / done of the
 * step out of the slot, valid = 0 where the terminal row has been overwritten.  num_frames <= 16.  Needs no UavEnv handle. */
int uavenv_ring_gather_stacked(const float* store_dev, const UavRingLayout* layout, const int64_t* age_dev, const int64_t* slot_dev,
                               const int64_t* rank_dev, const int64_t* env_dev, int32_t batch, int32_t num_frames,
                               float* obs_out_dev, float* next_obs_out_dev, int64_t* action_out_dev, float* reward_out_dev,
                               uint8_t* done_out_dev, uint8_t* valid_out_dev, void* stream);

/* same with the draw made IN the kernel (replaces the ~10 PyTorch launches that draw slot / rank / environment of a batch): sample b of
 * draw number *counter_dev gets Philox4x32 words keyed by (seed; b, counter): age uniform on 0 .. n-2, rank uniform on the ranks,
 * environment uniform on a rank's environments.  window_dev int64 [2] = (n = number of sampleable slots, ring position of the oldest);
 * counter_dev float [1] must differ from draw to draw (e.g. the optimiser's step count) -- both in device memory, so that a captured
 * launch draws a fresh batch at every replay.  index_out_dev int64 [4][batch] (nullable): the draw (age, slot, rank, environment). */
int uavenv_ring_sample_stacked(const float* store_dev, const UavRingLayout* layout, const int64_t* window_dev, const float* counter_dev,
                               uint64_t seed, int32_t batch, int32_t num_frames, float* obs_out_dev, float* next_obs_out_dev,
                               int64_t* action_out_dev, float* reward_out_dev, uint8_t* done_out_dev, uint8_t* valid_out_dev,
                               int64_t* index_out_dev, void* stream);

/* ---- the consumer of the observation layout: UAVAttentionExtractor forward (dqn.py:548-650) ------------- */
/* replaces: UAVAttentionExtractor.forward for inference, fused into one launch.  obs_dev float
 * [batch][n_stack*153] (the frame-stacked, 50-slot padded observation), weights_dev = the extractor's parameters
 * packed as uavenv_attention_weight_floats(n_stack) floats (order and transposes: csrc/uavenv_attention.hip
 * `Offsets`; uavenv_amd/attention.py packs a torch state_dict), out_dev float [batch][128].  Needs no UavEnv. */
int uavenv_attention_weight_floats(int32_t n_stack);
int uavenv_attention_features(const float* obs_dev, const float* weights_dev, float* out_dev, int32_t batch,
                              int32_t n_stack, void* stream);

/* ---- the learner directly above the path: one DQN gradient step of the MLP policy (dqn.py:1077-1099) ------------------ */
/* operand transforms / epilogues of uavenv_gemm_f32 */
#define UAVENV_GEMM_A_RELU   1   /* A is used as max(A, 0): the stored pre-activation of a ReLU layer as the next product's input */
#define UAVENV_GEMM_A_MASK   2   /* A is used where a_mask (same indexing) > 0, else 0: ReLU's backward                          */
#define UAVENV_GEMM_B_RELU   4   /* B is used as max(B, 0)                                                                        */
#define UAVENV_GEMM_BIAS     8   /* bias[n] is added to every row (once)                                                          */
#define UAVENV_GEMM_ROWSUM  16   /* row_sum[m] += sum_k A(m, k) after the transform: the bias gradient of a weight-gradient product */
#define UAVENV_GEMM_SUMSQ   32   /* sumsq[..] receives sums of squares of everything the product wrote (C and row_sum): the gradient
                                  * norm's partial sums, one per (16-row tile, 32-column group): float [UAVENV_GEMM_SUMSQ_COUNT(M, N)],
                                  * every entry written exactly once                                                                */
#define UAVENV_GEMM_SUMSQ_COUNT(M, N) ((((M) + 15) / 16) * (((N) + 31) / 32))
/* scalars block of the update (float [UAVENV_UPD_COUNT]) */
enum { UAVENV_UPD_LOSS = 0, UAVENV_UPD_NORM2, UAVENV_UPD_STEP, UAVENV_UPD_BC1, UAVENV_UPD_BC2, UAVENV_UPD_LR, UAVENV_UPD_COUNT = 8 };
/* one product C[M x N] = A . B (+ bias): A(m, k) = A[m * a_sm + k * a_sk], B(k, n) = B[k * b_sk + n * b_sn] (one stride of each
 * pair must be 1), C row-major with ldc; all device pointers */
typedef struct UavGemm {
    const float* A; const float* B; float* C;
    const float* bias;        /* UAVENV_GEMM_BIAS   */
    const float* a_mask;      /* UAVENV_GEMM_A_MASK: indexed like A */
    float* row_sum;           /* UAVENV_GEMM_ROWSUM: float [M]      */
    int32_t M, N, K, flags;
    int64_t a_sm, a_sk, b_sk, b_sn, ldc;
    float* sumsq;             /* UAVENV_GEMM_SUMSQ  */
} UavGemm;
/* replaces: torch.nn.Linear's three matrix products (forward, input gradient, weight gradient) at DQN batch sizes, where the
 * library GEMMs fill 16 of 256 CUs: on the f32 MFMA, one workgroup per 16 x 64 (or 16 x 32) output tile with K split over its 16
 * wavefronts.  `second` (nullable) is an independent product that shares the launch.  Needs no UavEnv handle. */
int uavenv_gemm_f32(const UavGemm* first, const UavGemm* second, void* stream);
/* replaces: the loss of SB3's DQN.train -- smooth-L1 between Q(s, a) and reward_scale * r + gamma * max_a' Q_target(s', a'),
 * averaged over the valid transitions -- and its gradient dq [batch][n_actions]; also advances scalars[UAVENV_UPD_STEP] and writes
 * Adam's bias corrections for that step.  batch <= 1024. */
int uavenv_td_loss(const float* q_dev, const float* q_next_dev, const int64_t* action_dev, const float* reward_dev,
                   const uint8_t* valid_dev, int32_t batch, int32_t n_actions, float gamma, float reward_scale, float beta1, float beta2,
                   float* dq_dev, float* scalars_dev, void* stream);
/* replaces: torch.nn.utils.clip_grad_norm_ + torch.optim.Adam.step (SB3 DQN: max_grad_norm 10, Adam) over ONE flat buffer of
 * n parameters: the gradient is scaled by min(1, max_norm / (norm + 1e-6)) and Adam applied with the learning rate in
 * scalars[UAVENV_UPD_LR] and the bias corrections uavenv_td_loss left; scalars[UAVENV_UPD_NORM2] receives the squared norm.
 * n_partials == 0: the norm is computed here (one more launch), workspace_dev: float [UAVENV_UPD_WORKSPACE] of scratch.
 * n_partials > 0: workspace_dev holds that many partial sums of squares of the gradient already (the weight-gradient products
 * left them: UAVENV_GEMM_SUMSQ) -- valid only while grad_dev is what those products wrote (not after an all-reduce). */
#define UAVENV_UPD_WORKSPACE 256
int uavenv_clip_adam(float* param_dev, const float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n,
                     float* scalars_dev, float* workspace_dev, int32_t n_partials, float max_norm, float beta1, float beta2, float eps,
                     void* stream);

/* replaces: the action selection of SB3's DQN while collecting (argmax Q with probability 1 - epsilon, else a uniform action) for
 * a vector of environments, one launch instead of six: q_dev float [n_envs][n_actions], *eps_dev the exploration rate, *counter_dev
 * a float that the kernel advances by one per call (it keys the Philox draws together with `seed`, so a captured launch draws fresh
 * numbers at every replay), actions_out_dev int32 [n_envs].  shared_coin != 0: ONE coin for the whole vector, as SB3's predict(). */
int uavenv_epsilon_greedy(const float* q_dev, int32_t n_envs, int32_t n_actions, const float* eps_dev, float* counter_dev,
                          uint64_t seed, int32_t shared_coin, int32_t* actions_out_dev, void* stream);

/* replaces: the LAST Linear layer of the acting forward (dqn.py:1078 policy "MlpPolicy": n_actions outputs) and
 * uavenv_epsilon_greedy in one launch.  h_dev float [n_envs][k]: the last hidden layer's activations; w_dev float [n_actions][k],
 * b_dev float [n_actions] (torch.nn.Linear's layout); n_actions <= 8, k a multiple of 4 (h_dev, w_dev 16-byte aligned),
 * n_actions * k <= 16384.  Draws and *counter_dev as in
 * uavenv_epsilon_greedy (the same actions for the same Q-values, counter and seed); ticket_dev int32 [1]: zero before the first call,
 * left zero by every call; q_out_dev float [n_envs][n_actions] (nullable) receives the Q-values. */
int uavenv_q_head_select(const float* h_dev, const float* w_dev, const float* b_dev, int32_t n_envs, int32_t k, int32_t n_actions,
                         const float* eps_dev, float* counter_dev, int32_t* ticket_dev, uint64_t seed, int32_t shared_coin,
                         int32_t* actions_out_dev, float* q_out_dev, void* stream);

/* replaces: the batched products and the masked softmax at the centre of nn.MultiheadAttention for ONE query per sample
 * (dqn.py:633-640: cross_attn(query, keys, keys, key_padding_mask)) in the TRAINING step, forward and backward, once the key and
 * value projections are folded out of the token dimension (uavenv_amd/learner.py AttentionFeatures.forward):
 *   scores[h][t] = qk[h] . kv[t],  a = softmax over the tokens with mask == 0,  mix[h] = sum_t a[h][t] kv[t].
 * qk_dev float [batch][heads][64], kv_dev float [batch][tokens][64], mask_dev uint8 [batch][tokens] (1 = ignore; never a whole row),
 * heads in {1, 2, 4, 8}, tokens <= 64.  Forward writes mix float [batch][heads][64] and attn float [batch][heads][tokens] (kept for
 * the backward); backward takes dmix and writes dqk float [batch][heads][64], dkv float [batch][tokens][64] (whole). */
int uavenv_attn_core_forward(const float* qk_dev, const float* kv_dev, const uint8_t* mask_dev, int32_t batch, int32_t heads,
                             int32_t tokens, float* mix_out_dev, float* attn_out_dev, void* stream);
int uavenv_attn_core_backward(const float* qk_dev, const float* kv_dev, const float* attn_dev, const float* dmix_dev, int32_t batch,
                              int32_t heads, int32_t tokens, float* dqk_out_dev, float* dkv_out_dev, void* stream);

/* ---- state access (checkpoint / parity / the attribute reads of SURVEY 1) -------------------- */
/* Copies one whole field.  `bytes` must equal the field size; dst/src may be host or device.
 * A state handed to uavenv_set_state must be one the library could have produced: 0 <= data_buffer <= max_buffer_size in every
 * lane, INCLUDING the lanes beyond an environment's sensor count (the kernels leave those rows alone by arithmetic -- they add 0
 * bytes and clamp with min / max -- not by masking them out; what uavenv_get_state returned always qualifies). */
int uavenv_get_state(UavEnv* env, int32_t field, void* dst, size_t bytes, int32_t dst_on_device, void* stream);
int uavenv_set_state(UavEnv* env, int32_t field, const void* src, size_t bytes, int32_t src_on_device, void* stream);
size_t uavenv_state_bytes(const UavEnv* env, int32_t field);

/* ---- synchronous host-buffer convenience (the reference hands numpy arrays across the boundary) -- */
int uavenv_reset_host(UavEnv* env, const uint8_t* mask, float* obs_out);
int uavenv_step_host(UavEnv* env, const int32_t* actions, float* obs_out, double* reward_out,
                     uint8_t* done_out, float* terminal_obs_out);

/* ---- measurement ---------------------------------------------------------------------------- */
/* Runs `steps` uavenv_step_random launches back to back on `stream`, bracketed by HIP events on that
 * same stream; returns the average milliseconds per launch (bench.py's roofline.achieved comes from it). */
int uavenv_time_steps(UavEnv* env, int32_t steps, float* obs_out_dev, double* reward_out_dev,
                      uint8_t* done_out_dev, void* stream, float* avg_ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* UAVENV_H */
