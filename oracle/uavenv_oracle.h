/*
 * uavenv_oracle.h -- CPU ORACLE for the UAV-IoT environment hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a scalar C restatement of the reference's reset()/step() chain
 * (/root/reference/src/environment/{uav_env.py,iot_sensors.py,uav.py},
 *  /root/reference/src/rewards/reward_function.py).  It is the checker for the HIP path; it is
 * never shipped as, linked into, or called from the product package.  Only tests/,
 * __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load it.
 *
 * PARITY PINNING: validated step-for-step against the real reference (imported from
 * /root/reference in the build container) by tests/golden/make_golden.py, which also wrote the
 * committed fixtures tests/golden/ (npz files) that tests/test_oracle_golden.py replays on every run.
 *
 * Arithmetic contract (why the oracle is "idealised" in exactly two places): the reference
 * evaluates `dx**2` on numpy float32 scalars through libm powf() and `np.log10(float32)` through
 * a platform-dependent SIMD/libm routine; both are NOT correctly rounded and differ between
 * machines by 1 ulp (measured here: 0.07 % resp. 47 % of inputs).  The oracle uses the correctly
 * rounded values (dx*dx, and orc_log10_f32: float64 evaluation rounded once to float32).  Everything else follows the reference's own
 * operation order and its float32/float64 mix.
 */
#ifndef UAVENV_ORACLE_H
#define UAVENV_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_FLAG_RANDOM_LAYOUT   1u  /* dqn.py:342-360  fresh uniform sensor layout + empty buffers on every reset */
#define ORC_FLAG_FAR_START       2u  /* dqn.py:364-365, 375-403 rejection-sampled UAV start                      */
#define ORC_FLAG_PROX_SHAPING    4u  /* dqn.py:417-425                                                            */
#define ORC_FLAG_JAIN_BONUS      8u  /* dqn.py:434-442                                                            */

/* Field-for-field the same layout as UavEnvConfig in include/uavenv.h (kept separate on purpose:
 * the oracle must not depend on product headers). */
typedef struct OrcConfig {
    uint32_t struct_size;
    int32_t  grid_w, grid_h;
    int32_t  num_sensors;
    int32_t  max_steps;
    int32_t  include_sensor_positions;
    int32_t  pad_sensors;
    uint32_t flags;
    int32_t  max_start_tries;
    int32_t  use_ema_adr;
    int32_t  num_grid_choices;               /* dqn.py:280-283 curriculum grid list; 0 = keep grid_w/h */
    int32_t  grid_choices_w[8], grid_choices_h[8];
    uint64_t seed;
    double data_generation_rate, max_buffer_size, rssi_threshold, duty_cycle;
    double start_x, start_y, max_battery, collection_duration;
    double tx_power_dbm, noise_floor_dbm, uav_altitude, sensor_height, wavelength, freq_mhz,
           fspl_offset_db, adr_lambda, shadowing_std_db, capture_threshold_db;
    double sf_thresholds[4];
    double fill_lo, fill_hi;
    double power_move, power_hover, alive_fraction;
    double reward_per_byte, reward_new_sensor, reward_completion, reward_urgency_reduction,
           reward_movement, penalty_revisit, penalty_boundary, penalty_collision, penalty_battery,
           penalty_hover, penalty_step, penalty_data_loss, penalty_starvation, penalty_unvisited,
           penalty_starved, starvation_cr_threshold;
    double min_start_dist, prox_eta, jain_weight;
} OrcConfig;

#define ORC_MAX_SENSORS 64

#define ORC_POLICY_NEAREST            2   /* greedy_agents.py:73  NearestSensorGreedy   */
#define ORC_POLICY_MAX_THROUGHPUT_V2  3   /* greedy_agents.py:105 MaxThroughputGreedyV2 */

/* One environment instance; plain data so tests can poke it through ctypes. */
typedef struct OrcEnv {
    OrcConfig cfg;
    int32_t n;                 /* active sensors                       */
    int32_t grid_w, grid_h;    /* may differ from cfg under layout randomisation */
    uint32_t env_index;        /* global env index = Philox counter word 0 */
    uint32_t episode;          /* number of resets performed so far     */
    /* per sensor (IoTSensor attributes, iot_sensors.py:66-103) */
    float  pos_x[ORC_MAX_SENSORS], pos_y[ORC_MAX_SENSORS];
    double buffer[ORC_MAX_SENSORS], gen[ORC_MAX_SENSORS], tx[ORC_MAX_SENSORS], lost[ORC_MAX_SENSORS];
    double avg_rssi[ORC_MAX_SENSORS], cur_rssi[ORC_MAX_SENSORS];
    int32_t sf[ORC_MAX_SENSORS];
    uint8_t avg_valid[ORC_MAX_SENSORS], visited[ORC_MAX_SENSORS], data_collected[ORC_MAX_SENSORS];
    /* per env (UAV + UAVEnvironment attributes) */
    float  uav_x, uav_y, start_x, start_y;
    double battery, previous_data_loss, total_reward, total_data_collected, last_step_bytes;
    double prev_dist_nearest;
    int32_t current_step, capture_triggers, boundary_hits, edge_steps, collisions_total;
    int32_t first_full_coverage_step;
} OrcEnv;

void orc_default_config(OrcConfig* c);
int  orc_obs_dim(const OrcConfig* c, int n);
void orc_init(OrcEnv* e, const OrcConfig* c, uint32_t env_index, const float* pos_x, const float* pos_y);

/* Tape-driven entry points.  reset_tape: float[4][n] = (u_fill, zD, zE, zS -- element 0 only: the sample behind the
 * SF that fresh DomainRandEnv sensors inherit, dqn.py:340-351; unused without ORC_FLAG_RANDOM_LAYOUT); step_tape: float[6][n] =
 * (zA, zB, u, zC, zD, zE).  Returns 0, or -1 on an invalid action (after ageing the sensors, like
 * the reference: uav_env.py:439-468). */
void orc_reset_tape(OrcEnv* e, const float* reset_tape, float* obs_out);
int  orc_step_tape(OrcEnv* e, int action, const float* step_tape, float* obs_out, double* reward_out,
                   int* truncated_out);

/* Keyed entry points: the tape is generated by the counter-based noise specification shared with
 * the HIP kernel (Philox4x32 with ORC_PHILOX_ROUNDS rounds + transcendental-free Box-Muller, DESIGN.md "Noise"). */
#define ORC_PHILOX_ROUNDS 7   /* the oracle's own statement of include/uavenv.h:UAVENV_PHILOX_ROUNDS (tests compare the two) */
void orc_noise_step_tape(uint64_t seed, uint32_t env_index, uint32_t episode, uint32_t step, int n, float* tape7);
void orc_noise_reset_tape(uint64_t seed, uint32_t env_index, uint32_t episode, int n, float* tape4);
void orc_noise_positions(uint64_t seed, uint32_t env_index, uint32_t episode, int n, int w, int h, float* px, float* py);
int  orc_noise_action(uint64_t seed, uint32_t env_index, uint32_t episode, uint32_t step);
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);   /* Random123's default, for its vectors */
int  orc_philox_rounds(void);
void orc_normal_pair(uint32_t a, uint32_t b, float* z0, float* z1);

void orc_reset_keyed(OrcEnv* e, float* obs_out);
int  orc_step_keyed(OrcEnv* e, int action, float* obs_out, double* reward_out, int* truncated_out);

/* Run `steps` vector steps of `num_envs` envs with the random policy and auto-reset; returns the
 * number of env-steps executed.  Used by bench.py's cpu_baseline leg (kind "port"). */
long orc_run_random_policy(const OrcConfig* c, int num_envs, uint32_t env_index_base, int steps,
                           double* reward_checksum);

int  orc_policy_action(const OrcEnv* e, int policy, const float* zP);
int  orc_step_policy_tape(OrcEnv* e, int policy, const float* step_tape7, float* obs_out, double* reward_out,
                          int* truncated_out, int* action_out);
int  orc_step_policy_keyed(OrcEnv* e, int policy, float* obs_out, double* reward_out, int* truncated_out, int* action_out);
long orc_trace_keyed(const OrcConfig* c, int num_envs, uint32_t base, int steps, const int32_t* actions, int policy,
                     int auto_reset, float* obs_out, double* rew_out, uint8_t* done_out, float* term_out,
                     int32_t* actions_out, float* reset_obs_out, OrcEnv* final_envs);

/* dqn.py:305-331: what DomainRandEnv.reset snapshots into `last_episode_stats` from the state the finished episode left
 * behind (call it BEFORE the reset that opens the next episode).  `rated` = number of sensors with generated data
 * (len(sensor_rates)); fairness_std = np.std(sensor_rates) (numpy's two-pass form over pairwise sums), 0 when none. */
typedef struct OrcEpisodeStats {
    double total_generated, total_collected, total_lost, battery_remaining, ndr, fairness_std, jains_index,
           data_efficiency, bytes_per_wh;
    int32_t grid_w, grid_h, num_sensors, rated, length, first_full_coverage_step;
} OrcEpisodeStats;
void orc_episode_stats(const OrcEnv* e, OrcEpisodeStats* out);

/* the four Philox words of one call of the noise specification (counter layout: DESIGN.md "Noise") */
void orc_noise_words(uint64_t seed, uint32_t env_index, uint32_t episode, uint32_t step, uint32_t lane, uint32_t call,
                     uint32_t out[4]);

/* Scalar known-answer helpers. */
float  orc_log10_f32(float d);
double orc_rssi_deterministic(const OrcConfig* c, float ux, float uy, float sx, float sy);

#ifdef __cplusplus
}
#endif
#endif
