/*
 * uavenv_oracle.c -- CPU ORACLE (test infrastructure, see uavenv_oracle.h).
 *
 * Scalar restatement of the reference UAV-IoT environment hot path.  Every function cites the
 * reference file:line it follows.  Paths are relative to /root/reference/src/.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math [-mfma] -shared -fPIC  (see oracle/Makefile).
 * The only fused operations are the explicit fma()/fmaf() calls of the noise and log10 specifications.
 * -ffp-contract=off matters: numpy never fuses a*b+c, so neither may we.
 */
#define _GNU_SOURCE
#include "uavenv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Configuration defaults                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* Defaults = the reference's training configuration: BASE_ENV_CONFIG (agents/dqn/dqn.py:1068-1075)
 * over the env kwargs (environment/uav_env.py:266-287), IoTSensor defaults
 * (environment/iot_sensors.py:39-57), UAV defaults (environment/uav.py:93-94,125) and
 * RewardFunction defaults as overridden by the env (rewards/reward_function.py:7-27,
 * uav_env.py:339-344). */
void orc_default_config(OrcConfig* c) {
    memset(c, 0, sizeof(*c));
    c->struct_size = (uint32_t)sizeof(*c);
    c->grid_w = 500; c->grid_h = 500;
    c->num_sensors = 20;
    c->max_steps = 2100;
    c->include_sensor_positions = 0;
    c->pad_sensors = 0;
    c->flags = 0;
    c->max_start_tries = 200;
    c->use_ema_adr = 1;
    c->seed = 0;
    c->data_generation_rate = 22.0 / 10;
    c->max_buffer_size = 1000.0;
    c->rssi_threshold = -85.0;
    c->duty_cycle = 10.0;
    c->start_x = 0.0; c->start_y = 0.0;
    c->max_battery = 274.0;
    c->collection_duration = 1.0;
    c->tx_power_dbm = 14.0;
    c->noise_floor_dbm = -105.0;
    c->uav_altitude = 100.0;
    c->sensor_height = 0.5;
    c->wavelength = 0.345;
    c->freq_mhz = 868.0;
    c->fspl_offset_db = 28.0;
    c->adr_lambda = 0.1;
    c->shadowing_std_db = 4.0;
    c->capture_threshold_db = 6.0;
    c->sf_thresholds[0] = -60.0; c->sf_thresholds[1] = -70.0;
    c->sf_thresholds[2] = -78.0; c->sf_thresholds[3] = -85.0;
    c->fill_lo = 0.20; c->fill_hi = 0.60;
    c->power_move = 500.0; c->power_hover = 700.0; c->alive_fraction = 0.02;
    c->reward_per_byte = 100.0; c->reward_new_sensor = 5000.0; c->reward_completion = 100.0;
    c->reward_urgency_reduction = 20.0; c->reward_movement = 10.0;
    c->penalty_revisit = -2.0; c->penalty_boundary = -50.0; c->penalty_collision = -10.0;
    c->penalty_battery = -0.5; c->penalty_hover = -5.0; c->penalty_step = -0.5;
    c->penalty_data_loss = -1.0; c->penalty_starvation = -1000.0; c->penalty_unvisited = -5000.0;
    c->penalty_starved = -1000.0; c->starvation_cr_threshold = 0.20;
    c->min_start_dist = 50.0; c->prox_eta = 2.0; c->jain_weight = 0.5;
}

/* uav_env.py:348-355 (3 + fps*N) and dqn.py:249-254 (padding to max_sensors_limit slots). */
int orc_obs_dim(const OrcConfig* c, int n) {
    int fps = c->include_sensor_positions ? 5 : 3;
    int slots = (c->pad_sensors > n) ? c->pad_sensors : n;
    return 3 + fps * slots;
}

/* iot_sensors.py:13-20 LORA_DATA_RATES (bytes/s), :22-29 REQUIRED_SNR_DB; uav_env.py:647 sf_quality */
static double sf_data_rate(int sf) {
    switch (sf) {
        case 7: return 5470 / 8.0;  case 8: return 3125 / 8.0;  case 9: return 1760 / 8.0;
        case 10: return 980 / 8.0;  case 11: return 440 / 8.0;  default: return 250 / 8.0;
    }
}
static double sf_required_snr(int sf) {
    switch (sf) {
        case 7: return -6.0;  case 8: return -9.0;  case 9: return -12.0;
        case 10: return -15.0; case 11: return -17.5; case 12: return -20.0;
        default: return 7.5;   /* iot_sensors.py:200 .get(sf, 7.5) */
    }
}
static double sf_link_quality(int sf) {
    switch (sf) {
        case 7: return 1.0; case 8: return 0.8; case 9: return 0.6;
        case 10: return 0.4; case 11: return 0.2; case 12: return 0.1;
        default: return 0.1;   /* uav_env.py:657 .get(sf, 0.1) */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Physics                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* iot_sensors.py:147-189  deterministic part of calculate_rssi (everything before the shadowing
 * draw).  Distances in float32, 20*log10f(d) in float32, the rest in float64. */
/* log10 of a positive normal float32 in float64, rounded once: the correctly rounded float32 log10
 * except with probability ~1e-8 (absolute error ~1e-15).  A fixed IEEE + - * / fma sequence (no libm transcendentals) so
 * that the HIP kernel, which evaluates the same specification, agrees bit for bit:
 *   x = m * 2^e, m folded into [sqrt(1/2), sqrt(2));  ln m = 2 atanh(s), s = (m-1)/(m+1), odd series to s^17;
 *   result = e*log10(2) + ln(m)*log10(e).
 * (The reference's own np.log10(float32) is a platform routine that is 1 ulp off on 47 % of inputs.) */
float orc_log10_f32(float d) {
    double x = (double)d;
    uint64_t bits; memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7FF) - 1023;
    uint64_t mb = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m; memcpy(&m, &mb, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 1.0 / 17;
    p = fma(p, s2, 1.0 / 15);
    p = fma(p, s2, 1.0 / 13);
    p = fma(p, s2, 1.0 / 11);
    p = fma(p, s2, 1.0 / 9);
    p = fma(p, s2, 1.0 / 7);
    p = fma(p, s2, 1.0 / 5);
    p = fma(p, s2, 1.0 / 3);
    double t = 2.0 * s;
    double ln_m = fma(t, s2 * p, t);
    double r = fma((double)e, 0.30102999566398120, ln_m * 0.43429448190325182);
    return (float)r;
}

double orc_rssi_deterministic(const OrcConfig* c, float ux, float uy, float sx, float sy) {
    float dx = (ux - sx) * 10.0f;                         /* :161 */
    float dy = (uy - sy) * 10.0f;                         /* :162 */
    float ground = sqrtf(dx * dx + dy * dy);              /* :163 */
    float alt2 = (float)(c->uav_altitude * c->uav_altitude);   /* altitude**2 is a Python scalar -> weak float32 */
    float d = sqrtf(ground * ground + alt2);              /* :164 */
    double ht = c->sensor_height, hr = c->uav_altitude;
    double d_break = (4 * M_PI * ht * hr) / c->wavelength;    /* :174 */
    float l10 = orc_log10_f32(d);                         /* correctly rounded float32 log10 (w.p. 1 - 1e-8) */
    double path_loss;
    if ((double)d < d_break) {
        float t = 20.0f * l10;                            /* :179 float32 product */
        path_loss = ((double)t + (20 * log10(c->freq_mhz))) - c->fspl_offset_db;
    } else {
        float t = 40.0f * l10;                            /* :183 */
        path_loss = ((double)t - (20 * log10(ht))) - (20 * log10(hr));
    }
    return c->tx_power_dbm - path_loss;                   /* :186 */
}

/* iot_sensors.py:147-197 calculate_rssi with the shadowing normal injected: np.random.normal(0, s)
 * evaluates loc + scale*z (:192). */
static double sensor_rssi(const OrcEnv* e, int i, double z) {
    double det = orc_rssi_deterministic(&e->cfg, e->uav_x, e->uav_y, e->pos_x[i], e->pos_y[i]);
    double shadowing = 0.0 + e->cfg.shadowing_std_db * z;
    return det + shadowing;                                /* :195 */
}

/* iot_sensors.py:223-259 update_spreading_factor (history lists are diagnostics, not restated). */
static void sensor_update_sf(OrcEnv* e, int i, double z) {
    const OrcConfig* c = &e->cfg;
    double cur = sensor_rssi(e, i, z);
    e->cur_rssi[i] = cur;                                  /* :235 */
    if (!e->avg_valid[i]) { e->avg_rssi[i] = cur; e->avg_valid[i] = 1; }          /* :239-240 */
    else if (c->use_ema_adr)
        e->avg_rssi[i] = (c->adr_lambda * cur) + ((1 - c->adr_lambda) * e->avg_rssi[i]);   /* :242-244 */
    else e->avg_rssi[i] = cur;
    static const int sf_of[4] = {7, 9, 11, 12};            /* :32-37 RSSI_SF_MAPPING */
    for (int k = 0; k < 4; k++)
        if (e->avg_rssi[i] > c->sf_thresholds[k]) { e->sf[i] = sf_of[k]; break; }   /* :252-255 sticky otherwise */
}

/* iot_sensors.py:202-212 get_success_probability */
static double sensor_success_probability(const OrcEnv* e, int i, double z, int advanced) {
    double rssi = sensor_rssi(e, i, z);
    if (rssi < e->cfg.rssi_threshold) return 0.0;
    if (advanced) {
        double snr_db = rssi - e->cfg.noise_floor_dbm;
        double req = sf_required_snr(e->sf[i]);
        return 1.0 / (1.0 + exp(-(snr_db - req)));
    }
    return 1.0;
}

/* uav_env.py:376-384 _calculate_urgency */
static double calc_urgency(const OrcEnv* e, int i) {
    double util = e->buffer[i] / e->cfg.max_buffer_size;
    double loss_rate = (e->gen[i] > 0) ? e->lost[i] / e->gen[i] : 0.0;
    double u = util * (1.0 + loss_rate * 10.0);
    return u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
}

/* uav_env.py:386-394 _get_sensor_urgencies: AoI approximation stored as FLOAT32 */
static float aoi_urgency(const OrcEnv* e, int i) {
    if (e->cfg.data_generation_rate > 0) return (float)(e->buffer[i] / e->cfg.data_generation_rate);
    return 0.0f;
}

/* numpy's float32 add.reduce (pairwise sum, PW_BLOCKSIZE 128): for n < 8 a plain loop, else 8
 * strided accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and a sequential tail.
 * n <= 64 < 128 here so there is no recursion.  Follows np.sum at uav_env.py:601. */
static float np_sum_f32(const float* a, int n) {
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    float r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}
static double np_sum_f64(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* rewards/reward_function.py:46-57 calculate_starvation_penalty (np.var = two-pass, ddof 0) */
static double starvation_penalty(const OrcEnv* e) {
    int n = e->n;
    if (n <= 1) return 0.0;
    double mx = e->buffer[0];
    for (int i = 1; i < n; i++) if (e->buffer[i] > mx) mx = e->buffer[i];
    if (mx == 0) return 0.0;
    double nb[ORC_MAX_SENSORS], dv[ORC_MAX_SENSORS];
    for (int i = 0; i < n; i++) nb[i] = e->buffer[i] / mx;
    double mean = np_sum_f64(nb, n) / n;
    for (int i = 0; i < n; i++) { double x = nb[i] - mean; dv[i] = x * x; }
    double var = np_sum_f64(dv, n) / n;
    return e->cfg.penalty_starvation * var;
}

/* ------------------------------------------------------------------------------------------ */
/* Observation                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* uav_env.py:638-674 _get_observation.  SIDE EFFECT: advances the ADR EMA of every sensor
 * (slot zD) and draws a fresh in-range sample (slot zE).  zd/ze: float[n]. */
static void build_observation(OrcEnv* e, const float* zd, const float* ze, float* obs) {
    const OrcConfig* c = &e->cfg;
    double W = (double)e->grid_w, H = (double)e->grid_h;
    double ux = (double)e->uav_x, uy = (double)e->uav_y;
    int fps = c->include_sensor_positions ? 5 : 3;
    int k = 0;
    obs[k++] = (float)(ux / W);
    obs[k++] = (float)(uy / H);
    obs[k++] = (float)(e->battery / c->max_battery);
    for (int i = 0; i < e->n; i++) {
        double urgency = calc_urgency(e, i);                        /* :652 (before the SF update) */
        sensor_update_sf(e, i, (double)zd[i]);                      /* :654 */
        int in_range = sensor_rssi(e, i, (double)ze[i]) >= c->rssi_threshold;   /* :658, iot_sensors.py:214-219 */
        double lq = in_range ? sf_link_quality(e->sf[i]) : 0.0;
        obs[k++] = (float)(e->buffer[i] / c->max_buffer_size);
        obs[k++] = (float)urgency;
        obs[k++] = (float)lq;
        if (c->include_sensor_positions) {                          /* :668-672 */
            obs[k++] = (float)(((double)e->pos_x[i] - ux) / W);
            obs[k++] = (float)(((double)e->pos_y[i] - uy) / H);
        }
    }
    int total = orc_obs_dim(c, e->n);                               /* dqn.py:286-298 zero padding */
    (void)fps;
    while (k < total) obs[k++] = 0.0f;
}

/* dqn.py:406-412 _dist_to_nearest_with_data (np.linalg.norm on float32 vectors) */
static double dist_nearest_with_data(const OrcEnv* e) {
    float best = -1.0f;
    for (int i = 0; i < e->n; i++) {
        if (!(e->buffer[i] > 0)) continue;
        float dx = e->pos_x[i] - e->uav_x, dy = e->pos_y[i] - e->uav_y;
        float d = sqrtf(dx * dx + dy * dy);
        if (best < 0.0f || d < best) best = d;
    }
    return best < 0.0f ? 0.0 : (double)best;
}

/* ------------------------------------------------------------------------------------------ */
/* init / reset                                                                               */
/* ------------------------------------------------------------------------------------------ */

void orc_init(OrcEnv* e, const OrcConfig* c, uint32_t env_index, const float* pos_x, const float* pos_y) {
    memset(e, 0, sizeof(*e));
    e->cfg = *c;
    e->n = c->num_sensors;
    e->grid_w = c->grid_w; e->grid_h = c->grid_h;
    e->env_index = env_index;
    e->episode = 0xFFFFFFFFu;                /* first reset makes it 0 */
    for (int i = 0; i < e->n; i++) {
        e->pos_x[i] = pos_x ? pos_x[i] : 0.0f;
        e->pos_y[i] = pos_y ? pos_y[i] : 0.0f;
        e->sf[i] = 12;
    }
    e->start_x = (float)c->start_x; e->start_y = (float)c->start_y;   /* uav.py:112 float32 */
    e->uav_x = e->start_x; e->uav_y = e->start_y;
    e->battery = c->max_battery;
    e->first_full_coverage_step = -1;
}

/* dqn.py:340-351: DomainRandEnv.reset builds its fresh sensors with `spreading_factor = s0.spreading_factor`,
 * where s0 is the OLD sensor 0 AFTER the discarded `super().reset()` (dqn.py:340): IoTSensor.reset put it
 * back to SF 12 with no EMA sample (iot_sensors.py:309-311), uav.reset() moved the UAV to uav.start_position
 * = the PREVIOUS episode's start (uav.py:256), and the reset observation (uav_env.py:427 -> :654) ran
 * update_spreading_factor once on it: avg = cur (first sample, iot_sensors.py:239-240), SF by the threshold
 * list, 12 kept when none fires (:251-255).  zS is that call's shadowing sample.  Must be evaluated BEFORE the
 * layout / start position of the new episode replace the old ones. */
static int inherited_sf(const OrcEnv* e, double zS) {
    const OrcConfig* c = &e->cfg;
    double det = orc_rssi_deterministic(c, e->start_x, e->start_y, e->pos_x[0], e->pos_y[0]);
    double avg = det + (0.0 + c->shadowing_std_db * zS);
    static const int sf_of[4] = {7, 9, 11, 12};
    int sf = 12;
    for (int k = 0; k < 4; k++)
        if (avg > c->sf_thresholds[k]) { sf = sf_of[k]; break; }
    return sf;
}

/* uav_env.py:400-427 reset + iot_sensors.py:305-321 IoTSensor.reset + uav.py:241-258 UAV.reset.
 * `fill_u`: the uniform behind np_random.uniform(0.20, 0.60) = lo + (hi-lo)*u (:410).
 * Under ORC_FLAG_RANDOM_LAYOUT (dqn.py:342-360) the sensors are REPLACED by fresh objects:
 * buffer 0, generated 0, no EMA sample, SF = `fresh_sf` (inherited_sf above) -- the prefill is discarded. */
static void reset_common(OrcEnv* e, const float* fill_u, int fresh_sf) {
    const OrcConfig* c = &e->cfg;
    e->episode += 1u;
    e->uav_x = e->start_x; e->uav_y = e->start_y;                   /* uav.py:256 */
    e->battery = c->max_battery;                                    /* uav.py:257 */
    for (int i = 0; i < e->n; i++) {
        double fill = c->fill_lo + (c->fill_hi - c->fill_lo) * (double)fill_u[i];
        double clipped = fill < 0.0 ? 0.0 : (fill > 1.0 ? 1.0 : fill);
        e->buffer[i] = c->max_buffer_size * clipped;                /* iot_sensors.py:308 */
        e->sf[i] = 12;                                              /* :309 */
        e->avg_valid[i] = 0; e->avg_rssi[i] = 0.0; e->cur_rssi[i] = 0.0;   /* :311-312 */
        e->gen[i] = e->buffer[i];                                   /* :316 */
        e->tx[i] = 0.0; e->lost[i] = 0.0;                           /* :317-318 */
        e->visited[i] = 0;                                          /* uav_env.py:416 (set()) */
        /* data_collected is NOT cleared by IoTSensor.reset (it is by object replacement) */
        if (c->flags & ORC_FLAG_RANDOM_LAYOUT) {
            e->buffer[i] = 0.0; e->gen[i] = 0.0; e->data_collected[i] = 0;
            e->sf[i] = fresh_sf;                                    /* dqn.py:351 */
        }
    }
    e->current_step = 0; e->total_reward = 0.0; e->total_data_collected = 0.0;    /* :413-415 */
    e->previous_data_loss = 0.0; e->capture_triggers = 0; e->boundary_hits = 0;   /* :418-420 */
    e->edge_steps = 0; e->last_step_bytes = 0.0; e->collisions_total = 0;         /* :421-422 */
    e->first_full_coverage_step = -1;                                             /* dqn.py:302 */
}

void orc_reset_tape(OrcEnv* e, const float* rt, float* obs_out) {
    int n = e->n;
    reset_common(e, rt + 0 * n, inherited_sf(e, (double)rt[3 * n + 0]));
    build_observation(e, rt + 1 * n, rt + 2 * n, obs_out);          /* :427 */
    e->prev_dist_nearest = dist_nearest_with_data(e);               /* dqn.py:368 */
}

/* ------------------------------------------------------------------------------------------ */
/* step                                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* uav_env.py:494-516 _execute_move_action + uav.py:127-185 UAV.move +
 * reward_function.py:69-79 calculate_movement_reward */
static double execute_move(OrcEnv* e, int action, double step_data_loss) {
    const OrcConfig* c = &e->cfg;
    double battery_before = e->battery;
    float nx = e->uav_x, ny = e->uav_y;
    if (action == 0) ny += 1.0f;            /* UP    */
    else if (action == 1) ny -= 1.0f;       /* DOWN  */
    else if (action == 2) nx -= 1.0f;       /* LEFT  */
    else nx += 1.0f;                        /* RIGHT */
    int ok = (0 <= nx && nx < (float)e->grid_w && 0 <= ny && ny < (float)e->grid_h);   /* uav.py:174 */
    double time_step = 1.0;
    if (ok) {
        e->uav_x = nx; e->uav_y = ny;
        e->battery -= (c->power_move * time_step) / 3600;           /* uav.py:176-177 */
    } else {
        double power_collision = c->power_move * 0.5;               /* uav.py:125 */
        e->battery -= (power_collision * time_step) / 3600;         /* uav.py:180-181 */
        e->boundary_hits += 1;                                      /* uav_env.py:503-504 */
    }
    double battery_used = battery_before - e->battery;              /* :501 */
    double reward = c->penalty_step;
    reward += ok ? c->reward_movement : c->penalty_boundary;
    reward += c->penalty_battery * battery_used;
    reward += c->penalty_data_loss * step_data_loss;                /* uav_env.py:514 */
    e->last_step_bytes = 0.0;                                       /* :463 */
    return reward;
}

/* uav_env.py:518-632 _execute_collect_action + reward_function.py:81-128 */
static double execute_collect(OrcEnv* e, const float* zA, const float* zB, const float* uL,
                              const float* zC, double step_data_loss) {
    const OrcConfig* c = &e->cfg;
    int n = e->n;
    float before[ORC_MAX_SENSORS], after[ORC_MAX_SENSORS], diff[ORC_MAX_SENSORS];
    for (int i = 0; i < n; i++) before[i] = aoi_urgency(e, i);      /* P0 :526 */

    e->battery -= (c->power_hover * c->collection_duration) / 3600; /* P1 :529, uav.py:204-205 */
    double battery_used = (c->power_hover / (60 * 60)) * c->collection_duration;   /* :530, uav.py:260-263 */

    /* P2 :535-551.  Buckets keep SF first-seen order like the dict at :533. */
    int bucket_sf[6], bucket_cnt[6], bucket_members[6][ORC_MAX_SENSORS], nb = 0;
    for (int i = 0; i < n; i++) {
        if (e->buffer[i] <= 0) continue;                            /* :536 */
        sensor_update_sf(e, i, (double)zA[i]);                      /* :539 */
        double p_link = sensor_success_probability(e, i, (double)zB[i], 1);   /* :543 */
        double p_cycle = c->duty_cycle / 100.0;                     /* iot_sensors.py:105-107 */
        double p_overall = p_link * p_cycle;
        if (p_overall > (double)uL[i]) {                            /* :549 */
            int b = -1;
            for (int k = 0; k < nb; k++) if (bucket_sf[k] == e->sf[i]) { b = k; break; }
            if (b < 0) { b = nb++; bucket_sf[b] = e->sf[i]; bucket_cnt[b] = 0; }
            bucket_members[b][bucket_cnt[b]++] = i;
        }
    }

    /* P3 :554-572 Capture Effect */
    int winners[6], nw = 0, collision_count = 0;
    for (int b = 0; b < nb; b++) {
        if (bucket_cnt[b] == 1) { winners[nw++] = bucket_members[b][0]; continue; }
        collision_count += bucket_cnt[b] - 1;
        /* sorted(..., key=current_rssi, reverse=True) is stable: the first of equal keys stays first */
        int top = -1, second = -1;
        for (int k = 0; k < bucket_cnt[b]; k++) {
            int i = bucket_members[b][k];
            if (top < 0 || e->cur_rssi[i] > e->cur_rssi[top]) { second = top; top = i; }
            else if (second < 0 || e->cur_rssi[i] > e->cur_rssi[second]) second = i;
        }
        if (e->cur_rssi[top] > (e->cur_rssi[second] + c->capture_threshold_db)) {
            winners[nw++] = top;
            e->capture_triggers += 1;
        }
    }
    e->collisions_total += collision_count;

    /* P4 :575-594 + iot_sensors.py:127-145 collect_data */
    double total_bytes = 0.0;
    int any_new = 0;
    for (int w = 0; w < nw; w++) {
        int i = winners[w];
        double prob = sensor_success_probability(e, i, (double)zC[i], 0);
        double bytes = 0.0; int success;
        if (prob <= 0.5 || e->buffer[i] <= 0) { success = prob > 0.5; }
        else {
            double max_collectible = sf_data_rate(e->sf[i]) * c->collection_duration;
            bytes = e->buffer[i] < max_collectible ? e->buffer[i] : max_collectible;
            e->buffer[i] -= bytes;
            e->tx[i] += bytes;
            if (bytes > 0) e->data_collected[i] = 1;
            success = 1;
        }
        if (success && bytes > 0) {
            total_bytes += bytes;
            e->total_data_collected += bytes;
            if (!e->visited[i]) { any_new = 1; e->visited[i] = 1; }
        }
    }
    int attempted_empty = 0, all_collected = 1;
    for (int i = 0; i < n; i++) {
        if (e->buffer[i] <= 0) attempted_empty = 1; else all_collected = 0;     /* :596, :604 */
    }
    /* P5 :599-602 float32 arithmetic */
    for (int i = 0; i < n; i++) {
        after[i] = aoi_urgency(e, i);
        float d = before[i] - after[i];
        diff[i] = d > 0.0f ? d : 0.0f;
    }
    double urgency_reduced = (double)np_sum_f32(diff, n);

    /* P6 :607-630 */
    e->last_step_bytes = total_bytes;
    double mean_urgency = 0.0;
    if (nw > 0) {
        double s = 0.0;                     /* np.mean over <=6 values: plain sequential sum */
        for (int w = 0; w < nw; w++) s += calc_urgency(e, winners[w]);
        mean_urgency = s / nw;
    }
    double reward = c->penalty_step + c->penalty_hover;             /* reward_function.py:97 */
    if (total_bytes > 0) {
        reward += c->reward_per_byte * total_bytes * mean_urgency;
        if (any_new) reward += c->reward_new_sensor;
    }
    if (urgency_reduced > 0) reward += c->reward_urgency_reduction * urgency_reduced;
    if (attempted_empty && total_bytes == 0) reward += c->penalty_revisit;
    reward += c->penalty_battery * battery_used;
    if (collision_count > 0) reward += c->penalty_collision * collision_count;
    if (step_data_loss > 0) reward += c->penalty_data_loss * step_data_loss;
    reward += starvation_penalty(e);
    if (all_collected) reward += c->reward_completion;
    return reward;
}

/* dqn.py:446-451 _jains over r_i = 100*tx_i/gen_i for gen_i > 0 */
static double jains_index(const OrcEnv* e, int* count_out) {
    double s1 = 0.0, s2 = 0.0; int cnt = 0;
    for (int i = 0; i < e->n; i++) {
        double g = e->gen[i];
        if (g > 0) { double r = (e->tx[i] / g) * 100; s1 += r; s2 += r * r; cnt++; }
    }
    if (count_out) *count_out = cnt;
    if (cnt > 0 && s2 > 0) return (s1 * s1) / (cnt * s2);
    return 1.0;
}

/* dqn.py:305-331 last_episode_stats: python sum() = sequential float64 adds, np.std = sqrt(mean(|x - mean(x)|^2)) with
 * numpy's pairwise sums, `_jains` over the percentage rates (:446-451). */
void orc_episode_stats(const OrcEnv* e, OrcEpisodeStats* o) {
    const OrcConfig* c = &e->cfg;
    int n = e->n, cnt = 0, visited = 0;
    double rates[ORC_MAX_SENSORS], dv[ORC_MAX_SENSORS];
    double tg = 0.0, tc = 0.0, tl = 0.0;
    for (int i = 0; i < n; i++) {
        if (e->gen[i] > 0) rates[cnt++] = e->tx[i] / e->gen[i] * 100;              /* :308-312 */
        tg += e->gen[i]; tc += e->tx[i]; tl += e->lost[i];                          /* :313-314, :319 */
        visited += e->visited[i];
    }
    double battery_used = c->max_battery - e->battery;                             /* :315 */
    o->total_generated = tg; o->total_collected = tc; o->total_lost = tl;
    o->battery_remaining = e->battery;
    o->ndr = (double)visited / n * 100;                                            /* :321 */
    o->fairness_std = 0.0;
    if (cnt > 0) {                                                                 /* :322 np.std */
        double mean = np_sum_f64(rates, cnt) / cnt;
        for (int i = 0; i < cnt; i++) { double d = rates[i] - mean; dv[i] = d * d; }
        o->fairness_std = sqrt(np_sum_f64(dv, cnt) / cnt);
    }
    o->jains_index = jains_index(e, NULL);                                         /* :323 */
    o->grid_w = e->grid_w; o->grid_h = e->grid_h; o->num_sensors = n; o->rated = cnt;
    o->data_efficiency = tg > 0 ? tc / tg * 100 : 0.0;                             /* :326-327 */
    o->bytes_per_wh = battery_used > 0 ? tc / battery_used : 0.0;                  /* :328-329 */
    o->length = e->current_step;
    o->first_full_coverage_step = e->first_full_coverage_step;
}

/* uav_env.py:429-488 step (+ dqn.py:415-444 DomainRandEnv.step when shaping flags are set) */
int orc_step_tape(OrcEnv* e, int action, const float* tp, float* obs_out, double* reward_out,
                  int* truncated_out) {
    const OrcConfig* c = &e->cfg;
    int n = e->n;
    const float *zA = tp, *zB = tp + n, *uL = tp + 2 * n, *zC = tp + 3 * n, *zD = tp + 4 * n, *zE = tp + 5 * n;
    double prev_dist = e->prev_dist_nearest;                        /* dqn.py:417 */

    e->current_step += 1;                                           /* :439 */
    {   /* :443-447 edge-cell bookkeeping on the PRE-move position */
        double W = (double)e->grid_w, H = (double)e->grid_h, ux = (double)e->uav_x, uy = (double)e->uav_y;
        double eps = 1e-6;
        if (ux <= eps || uy <= eps || ux >= W - 1 - eps || uy >= H - 1 - eps) e->edge_steps += 1;
    }
    double step_duration = (action == 4) ? c->collection_duration : 1.0;   /* :450 */
    for (int i = 0; i < n; i++) {                                   /* :453-454, iot_sensors.py:114-125 */
        double new_data = c->data_generation_rate * step_duration;
        e->gen[i] += new_data;
        double potential = e->buffer[i] + new_data;
        if (potential > c->max_buffer_size) {
            e->lost[i] += potential - c->max_buffer_size;
            e->buffer[i] = c->max_buffer_size;
        } else e->buffer[i] = potential;
    }
    double current_loss = 0.0;                                      /* :457 python sum(), sequential */
    for (int i = 0; i < n; i++) current_loss += e->lost[i];
    double step_data_loss = current_loss - e->previous_data_loss;
    e->previous_data_loss = current_loss;

    double reward;
    if (action >= 0 && action <= 3) reward = execute_move(e, action, step_data_loss);
    else if (action == 4) reward = execute_collect(e, zA, zB, uL, zC, step_data_loss);
    else return -1;                                                 /* :468 ValueError after ageing */

    int truncated = 0;                                              /* :471-478; terminated is always False */
    if (!(e->battery > (c->alive_fraction * c->max_battery))) truncated = 1;      /* uav.py:224 */
    if (e->current_step >= c->max_steps) truncated = 1;
    if (truncated) {                                                /* :481-485 */
        int visited = 0;
        for (int i = 0; i < n; i++) visited += e->visited[i];
        int unvisited = n - visited;
        if (unvisited > 0) reward += c->penalty_unvisited * unvisited;
        double pen = 0.0;                                           /* reward_function.py:59-67 */
        for (int i = 0; i < n; i++)
            if (e->gen[i] > 0) {
                double cr = e->tx[i] / e->gen[i];
                if (cr < c->starvation_cr_threshold) pen += c->penalty_starved;
            }
        reward += pen;
    }
    e->total_reward += reward;                                      /* :487 */
    build_observation(e, zD, zE, obs_out);                          /* :488 */

    if (c->flags & ORC_FLAG_PROX_SHAPING) {                         /* dqn.py:419-425 */
        double curr = dist_nearest_with_data(e);
        if (prev_dist > 0) reward += c->prox_eta * (prev_dist - curr);
        e->prev_dist_nearest = curr;
    }
    if (e->first_full_coverage_step < 0) {                          /* dqn.py:428-431 */
        int visited = 0;
        for (int i = 0; i < n; i++) visited += e->visited[i];
        if (visited == n) e->first_full_coverage_step = e->current_step;
    }
    if (c->flags & ORC_FLAG_JAIN_BONUS) {                           /* dqn.py:434-442 */
        int rated;
        double j = jains_index(e, &rated);
        if (rated > 0) reward += c->jain_weight * (j - 0.5) / n;    /* `if rates:` -- no bonus before any sensor generated data */
    }

    *reward_out = reward;
    *truncated_out = truncated;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Counter-based noise specification (shared, by specification, with the HIP kernel)          */
/* ------------------------------------------------------------------------------------------ */

/* Philox4x32-R (Salmon et al., SC'11).  Integer-only, so bit-exact on every machine.  The noise specification uses
 * R = ORC_PHILOX_ROUNDS = 7, the smallest count the paper reports as Crush-resistant; R = 10 is Random123's default. */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < rounds; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc_philox4x32(ctr, key, 10, out); }
int orc_philox_rounds(void) { return ORC_PHILOX_ROUNDS; }

static inline float u32_as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* Box-Muller on two 32-bit words WITHOUT libm transcendentals: only IEEE float32 +,-,*,fma,sqrt and
 * integer ops, in a fixed order, so CPU and GPU produce bit-identical normals.
 *   radius: u1 = (a>>8 + 1) * 2^-24 in (0,1];  -ln(u1) by exponent split + degree-9 polynomial
 *           (Cephes logf coefficients) on m in [sqrt(1/2), sqrt(2));
 *   angle:  top 2 bits of (b>>8) pick the quadrant, the low 22 bits the angle in [-pi/4, pi/4);
 *           sin/cos by the Cephes sinf/cosf minimax polynomials. */
void orc_normal_pair(uint32_t a, uint32_t b, float* z0, float* z1) {
    uint32_t k = (a >> 8) + 1u;
    float u1 = (float)k * 0x1p-24f;
    uint32_t bits = f32_as_u32(u1);
    int ex = (int)(bits >> 23) - 127;
    float m = u32_as_f32((bits & 0x007FFFFFu) | 0x3F800000u);       /* [1,2) */
    if (m > 1.41421354f) { m = m * 0.5f; ex += 1; }
    float t = m - 1.0f;
    float z = t * t;
    float p = 7.0376836292E-2f;
    p = fmaf(p, t, -1.1514610310E-1f);
    p = fmaf(p, t, 1.1676998740E-1f);
    p = fmaf(p, t, -1.2420140846E-1f);
    p = fmaf(p, t, 1.4249322787E-1f);
    p = fmaf(p, t, -1.6668057665E-1f);
    p = fmaf(p, t, 2.0000714765E-1f);
    p = fmaf(p, t, -2.4999993993E-1f);
    p = fmaf(p, t, 3.3333331174E-1f);
    float y = (t * z) * p;
    y = fmaf(-0.5f, z, y);
    float ln = fmaf((float)ex, 0.693147182f, t + y);
    float r2 = -2.0f * ln;
    if (!(r2 > 0.0f)) r2 = 0.0f;
    float r = sqrtf(r2);

    uint32_t kb = b >> 8;
    uint32_t q = kb >> 22;
    float f = (float)(kb & 0x003FFFFFu) * 0x1p-22f;                 /* [0,1) */
    float phi = (f - 0.5f) * 1.57079637f;                           /* [-pi/4, pi/4) */
    float zz = phi * phi;
    float s = -1.9515295891E-4f;
    s = fmaf(s, zz, 8.3321608736E-3f);
    s = fmaf(s, zz, -1.6666654611E-1f);
    s = fmaf(s * zz, phi, phi);
    float c = 2.443315711809948E-5f;
    c = fmaf(c, zz, -1.388731625493765E-3f);
    c = fmaf(c, zz, 4.166664568298827E-2f);
    c = fmaf(c * zz, zz, fmaf(-0.5f, zz, 1.0f));
    float cs, sn;
    switch (q) {
        case 0: cs = c; sn = s; break;
        case 1: cs = -s; sn = c; break;
        case 2: cs = -c; sn = -s; break;
        default: cs = s; sn = -c; break;
    }
    *z0 = r * cs;
    *z1 = r * sn;
}

/* Counter layout: (env_index, episode, step, lane | call<<16); key = 64-bit seed.
 *   call 0 (every step incl. step 0 = reset observation): w0,w1 -> (zD,zE); w2 -> lottery u
 *   call 1 (collect steps):                               w0,w1 -> (zA,zB); w2,w3 -> (zC,-)
 *   call 2 (reset, step 0): w0 -> u_fill; w1,w2 -> layout x,y; lane 0's w3 -> curriculum grid choice
 *   (no call 3: the random policy's action of step s is lane 0's w3 of call 0 at step s-1)
 *   call 4 (lane = try):    w0,w1 -> far-start candidate
 *   call 5 (policy steps):  w0,w1 -> (zP, -) in-range sample drawn by a heuristic policy before the step
 *   call 6 (reset, lane 0): w0,w1 -> (zS, -) shadowing sample of the discarded reset observation's ADR update of the
 *                           OLD sensor 0, which decides the SF the fresh DomainRandEnv sensors inherit (dqn.py:340-351) */
static void noise_words(uint64_t seed, uint32_t env, uint32_t ep, uint32_t step, uint32_t lane,
                        uint32_t call, uint32_t w[4]) {
    uint32_t ctr[4] = {env, ep, step, lane | (call << 16)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    orc_philox4x32(ctr, key, ORC_PHILOX_ROUNDS, w);
}
static inline float u24(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }
void orc_noise_words(uint64_t seed, uint32_t env, uint32_t ep, uint32_t step, uint32_t lane, uint32_t call, uint32_t w[4]) {
    noise_words(seed, env, ep, step, lane, call, w);
}

void orc_noise_step_tape(uint64_t seed, uint32_t env, uint32_t ep, uint32_t step, int n, float* tp) {
    for (int i = 0; i < n; i++) {
        uint32_t w[4]; float a, b;
        noise_words(seed, env, ep, step, (uint32_t)i, 5, w);
        orc_normal_pair(w[0], w[1], &a, &b); tp[6 * n + i] = a;
        noise_words(seed, env, ep, step, (uint32_t)i, 1, w);
        orc_normal_pair(w[0], w[1], &a, &b); tp[0 * n + i] = a; tp[1 * n + i] = b;
        orc_normal_pair(w[2], w[3], &a, &b); tp[3 * n + i] = a;
        noise_words(seed, env, ep, step, (uint32_t)i, 0, w);
        orc_normal_pair(w[0], w[1], &a, &b); tp[4 * n + i] = a; tp[5 * n + i] = b;
        tp[2 * n + i] = u24(w[2]);
    }
}
void orc_noise_reset_tape(uint64_t seed, uint32_t env, uint32_t ep, int n, float* tp) {
    for (int i = 0; i < n; i++) {
        uint32_t w[4]; float a, b;
        noise_words(seed, env, ep, 0, (uint32_t)i, 2, w);
        tp[0 * n + i] = u24(w[0]);
        noise_words(seed, env, ep, 0, (uint32_t)i, 0, w);
        orc_normal_pair(w[0], w[1], &a, &b); tp[1 * n + i] = a; tp[2 * n + i] = b;
        tp[3 * n + i] = 0.0f;
    }
    {   /* zS: lane 0 of call 6 */
        uint32_t w[4]; float a, b;
        noise_words(seed, env, ep, 0, 0u, 6, w);
        orc_normal_pair(w[0], w[1], &a, &b); tp[3 * n + 0] = a;
    }
}
void orc_noise_positions(uint64_t seed, uint32_t env, uint32_t ep, int n, int gw, int gh, float* px, float* py) {
    for (int i = 0; i < n; i++) {
        uint32_t w[4];
        noise_words(seed, env, ep, 0, (uint32_t)i, 2, w);
        px[i] = u24(w[1]) * (float)gw;
        py[i] = u24(w[2]) * (float)gh;
    }
}
/* uniform-random policy: the action of step s (s >= 1) is word 3 of the call that lane 0 makes for the observation
 * noise of step s-1 (call 0; for s = 1 that is the call of the reset observation), scaled to 0..4 */
int orc_noise_action(uint64_t seed, uint32_t env, uint32_t ep, uint32_t step) {
    uint32_t w[4];
    noise_words(seed, env, ep, step - 1u, 0, 0, w);
    return (int)(((uint64_t)w[3] * 5u) >> 32);
}

/* dqn.py:375-403 _sample_far_start with Philox candidates */
static void sample_far_start(OrcEnv* e) {
    const OrcConfig* c = &e->cfg;
    double W = (double)e->grid_w, H = (double)e->grid_h;
    float best_x = 0, best_y = 0, best_d = -1.0f;
    for (int t = 0; t < c->max_start_tries; t++) {
        uint32_t w[4];
        noise_words(c->seed, e->env_index, e->episode, 0, (uint32_t)t, 4, w);
        float cx = (float)(0.05 * W + (0.95 * W - 0.05 * W) * (double)u24(w[0]));
        float cy = (float)(0.05 * H + (0.95 * H - 0.05 * H) * (double)u24(w[1]));
        float dmin = -1.0f;
        for (int i = 0; i < e->n; i++) {
            float dx = cx - e->pos_x[i], dy = cy - e->pos_y[i];
            float d = sqrtf(dx * dx + dy * dy);
            if (dmin < 0.0f || d < dmin) dmin = d;
        }
        if (e->n == 0) { best_x = cx; best_y = cy; break; }
        if (dmin > best_d) { best_d = dmin; best_x = cx; best_y = cy; }
        if ((double)dmin >= c->min_start_dist) { best_x = cx; best_y = cy; break; }
    }
    e->start_x = best_x; e->start_y = best_y;
}

void orc_reset_keyed(OrcEnv* e, float* obs_out) {
    const OrcConfig* c = &e->cfg;
    float rt[4 * ORC_MAX_SENSORS];
    uint32_t ep = e->episode + 1u;
    orc_noise_reset_tape(c->seed, e->env_index, ep, e->n, rt);
    const int fresh_sf = inherited_sf(e, (double)rt[3 * e->n + 0]);   /* old sensor 0, old start: before both are replaced */
    if ((c->flags & ORC_FLAG_RANDOM_LAYOUT) && c->num_grid_choices > 0) {   /* dqn.py:334 */
        uint32_t w[4];
        noise_words(c->seed, e->env_index, ep, 0, 0, 2, w);
        int g = (int)(((uint64_t)w[3] * (uint32_t)c->num_grid_choices) >> 32);
        e->grid_w = c->grid_choices_w[g]; e->grid_h = c->grid_choices_h[g];
    }
    if (c->flags & ORC_FLAG_RANDOM_LAYOUT)
        orc_noise_positions(c->seed, e->env_index, ep, e->n, e->grid_w, e->grid_h, e->pos_x, e->pos_y);
    reset_common(e, rt, fresh_sf);
    if (c->flags & ORC_FLAG_FAR_START) {
        sample_far_start(e);
        e->uav_x = e->start_x; e->uav_y = e->start_y;               /* dqn.py:364-365 */
    }
    build_observation(e, rt + e->n, rt + 2 * e->n, obs_out);
    e->prev_dist_nearest = dist_nearest_with_data(e);
}

int orc_step_keyed(OrcEnv* e, int action, float* obs_out, double* reward_out, int* truncated_out) {
    float tp[7 * ORC_MAX_SENSORS];
    orc_noise_step_tape(e->cfg.seed, e->env_index, e->episode, (uint32_t)(e->current_step + 1), e->n, tp);
    return orc_step_tape(e, action, tp, obs_out, reward_out, truncated_out);
}

/* ------------------------------------------------------------------------------------------ */
/* Heuristic policies (agents/dqn/dqn_evaluation_results/greedy_agents.py)                      */
/* ------------------------------------------------------------------------------------------ */

/* greedy_agents.py:42-67 GreedyAgent._move_toward (float32 position arithmetic) */
static int policy_move_toward(const OrcEnv* e, float tx, float ty) {
    float dx = tx - e->uav_x, dy = ty - e->uav_y;
    if (fabsf(dx) <= 0.5f && fabsf(dy) <= 0.5f) return 4;
    float nx = e->uav_x, ny = e->uav_y;
    if (fabsf(dx) > fabsf(dy)) nx = e->uav_x + (dx > 0 ? 1.0f : -1.0f);
    else ny = e->uav_y + (dy > 0 ? 1.0f : -1.0f);
    if (nx < 0 || nx >= (float)e->grid_w || ny < 0 || ny >= (float)e->grid_h) return 4;
    float mdx = nx - e->uav_x, mdy = ny - e->uav_y;
    if (mdx > 0) return 3;
    if (mdx < 0) return 2;
    if (mdy > 0) return 0;
    if (mdy < 0) return 1;
    return 4;
}
static float policy_dist(const OrcEnv* e, int i) {          /* np.linalg.norm on float32 2-vectors */
    float dx = e->pos_x[i] - e->uav_x, dy = e->pos_y[i] - e->uav_y;
    return sqrtf(dx * dx + dy * dy);
}

/* Action chosen by a heuristic policy for the CURRENT state; zP[i] is the shadowing sample of the
 * is_in_range() call the policy makes on sensor i (iot_sensors.py:214-219).
 *   ORC_POLICY_NEAREST            greedy_agents.py:73-100  NearestSensorGreedy
 *   ORC_POLICY_MAX_THROUGHPUT_V2  greedy_agents.py:105-216 MaxThroughputGreedyV2 */
int orc_policy_action(const OrcEnv* e, int policy, const float* zP) {
    const OrcConfig* c = &e->cfg;
    int n = e->n;
    if (policy == ORC_POLICY_NEAREST) {
        for (int i = 0; i < n; i++)                                   /* :84-86 */
            if (e->buffer[i] > 0 && sensor_rssi(e, i, (double)zP[i]) >= c->rssi_threshold) return 4;
        int best = -1; float bd = 0.0f;                                /* :94-100 min() keeps the first minimum */
        for (int i = 0; i < n; i++) {
            if (!(e->buffer[i] > 0)) continue;
            float d = policy_dist(e, i);
            if (best < 0 || d < bd) { best = i; bd = d; }
        }
        if (best < 0) return 4;
        return policy_move_toward(e, e->pos_x[best], e->pos_y[best]);
    }
    /* MaxThroughputGreedyV2.select_action :131-160 */
    double battery_pct = e->battery / 274.0;                           /* :133 (hard-coded capacity) */
    int steps_left = c->max_steps - e->current_step;                   /* :134 */
    double r = (double)steps_left / (double)c->max_steps;
    double steps_ratio = r < 1.0 ? r : 1.0;                            /* :207-208 */
    int sf_threshold = (battery_pct > 0.5 && steps_ratio > 0.5) ? 9 : ((battery_pct > 0.2 && steps_ratio > 0.2) ? 10 : 12);
    for (int i = 0; i < n; i++)                                        /* :139-144: any immediate candidate -> COLLECT */
        if (e->buffer[i] > 0 && sensor_rssi(e, i, (double)zP[i]) >= c->rssi_threshold && e->sf[i] <= sf_threshold) return 4;
    double sf_w;                                                       /* :164-169 */
    if (battery_pct < 0.1 || steps_left < 50) sf_w = 1.0;
    else if (battery_pct < 0.3 || steps_left < 150) sf_w = 2.0;
    else sf_w = 5.0;
    double best_score = -INFINITY; int best = -1;
    for (int i = 0; i < n; i++) {                                      /* :174-190 */
        if (e->buffer[i] <= 0) continue;
        float distance = policy_dist(e, i);
        int pr = 13 - e->sf[i]; if (pr < 0) pr = 0;
        double sf_score = pr * 5.0 * sf_w;
        double buffer_score = (e->buffer[i] / c->max_buffer_size) * 10.0;
        double duty_score = (c->duty_cycle / 100.0) * 2.0;
        float distance_penalty = ((distance / (float)e->grid_w) * 5.0f) * 1.0f;        /* float32 chain */
        double score = ((sf_score + buffer_score) + duty_score) - (double)distance_penalty;
        if (score > best_score) { best_score = score; best = i; }
    }
    if (best < 0) return 4;
    return policy_move_toward(e, e->pos_x[best], e->pos_y[best]);
}

int orc_step_policy_tape(OrcEnv* e, int policy, const float* tp7, float* obs_out, double* reward_out, int* truncated_out,
                         int* action_out) {
    int a = orc_policy_action(e, policy, tp7 + 6 * e->n);
    if (action_out) *action_out = a;
    return orc_step_tape(e, a, tp7, obs_out, reward_out, truncated_out);
}

int orc_step_policy_keyed(OrcEnv* e, int policy, float* obs_out, double* reward_out, int* truncated_out, int* action_out) {
    float tp[7 * ORC_MAX_SENSORS];
    orc_noise_step_tape(e->cfg.seed, e->env_index, e->episode, (uint32_t)(e->current_step + 1), e->n, tp);
    return orc_step_policy_tape(e, policy, tp, obs_out, reward_out, truncated_out, action_out);
}

/* cpu_baseline leg: E envs, random policy, auto-reset (what SB3's DummyVecEnv does around the
 * reference: step, and on truncation reset immediately). */
long orc_run_random_policy(const OrcConfig* c, int num_envs, uint32_t env_index_base, int steps,
                           double* reward_checksum) {
    OrcEnv* envs = (OrcEnv*)malloc(sizeof(OrcEnv) * (size_t)num_envs);
    float obs[3 + 5 * ORC_MAX_SENSORS];
    float px[ORC_MAX_SENSORS], py[ORC_MAX_SENSORS];
    double sum = 0.0;
    long count = 0;
    for (int k = 0; k < num_envs; k++) {
        uint32_t idx = env_index_base + (uint32_t)k;
        orc_noise_positions(c->seed, idx, 0xFFFFFFFFu, c->num_sensors, c->grid_w, c->grid_h, px, py);
        orc_init(&envs[k], c, idx, px, py);
        orc_reset_keyed(&envs[k], obs);
    }
    for (int s = 0; s < steps; s++) {
        for (int k = 0; k < num_envs; k++) {
            OrcEnv* e = &envs[k];
            int a = orc_noise_action(c->seed, e->env_index, e->episode, (uint32_t)(e->current_step + 1));
            double r; int tr;
            orc_step_keyed(e, a, obs, &r, &tr);
            sum += r; count++;
            if (tr) orc_reset_keyed(e, obs);
        }
    }
    free(envs);
    if (reward_checksum) *reward_checksum = sum;
    return count;
}

/* Trace runner for the parity tests: `steps` vector steps of `num_envs` keyed environments
 * (global indices base..base+num_envs-1), actions given ([steps][E]) or drawn by the random policy,
 * with SB3-style auto-reset when `auto_reset` (the step's own observation goes to term_out, the
 * reset observation to obs_out).  Any output pointer may be NULL.  final_envs: OrcEnv[num_envs]. */
long orc_trace_keyed(const OrcConfig* c, int num_envs, uint32_t base, int steps, const int32_t* actions, int policy,
                     int auto_reset, float* obs_out, double* rew_out, uint8_t* done_out, float* term_out,
                     int32_t* actions_out, float* reset_obs_out, OrcEnv* final_envs) {
    OrcEnv* envs = (OrcEnv*)malloc(sizeof(OrcEnv) * (size_t)num_envs);
    int D = orc_obs_dim(c, c->num_sensors);
    float obs[3 + 5 * ORC_MAX_SENSORS + 5 * 64];
    float px[ORC_MAX_SENSORS], py[ORC_MAX_SENSORS];
    long count = 0;
    for (int k = 0; k < num_envs; k++) {
        uint32_t idx = base + (uint32_t)k;
        orc_noise_positions(c->seed, idx, 0xFFFFFFFFu, c->num_sensors, c->grid_w, c->grid_h, px, py);
        orc_init(&envs[k], c, idx, px, py);
        orc_reset_keyed(&envs[k], obs);
        if (reset_obs_out) memcpy(reset_obs_out + (size_t)k * D, obs, sizeof(float) * (size_t)D);
    }
    for (int s = 0; s < steps; s++) {
        for (int k = 0; k < num_envs; k++) {
            OrcEnv* e = &envs[k];
            size_t row = (size_t)s * (size_t)num_envs + (size_t)k;
            int a;
            double r = 0.0; int tr = 0;
            if (policy >= 2) orc_step_policy_keyed(e, policy, obs, &r, &tr, &a);
            else {
                a = actions ? actions[row]
                            : orc_noise_action(c->seed, e->env_index, e->episode, (uint32_t)(e->current_step + 1));
                orc_step_keyed(e, a, obs, &r, &tr);
            }
            count++;
            if (actions_out) actions_out[row] = a;
            if (rew_out) rew_out[row] = r;
            if (done_out) done_out[row] = (uint8_t)tr;
            if (tr && auto_reset) {
                if (term_out) memcpy(term_out + row * D, obs, sizeof(float) * (size_t)D);
                orc_reset_keyed(e, obs);
            }
            if (obs_out) memcpy(obs_out + row * D, obs, sizeof(float) * (size_t)D);
        }
    }
    if (final_envs) memcpy(final_envs, envs, sizeof(OrcEnv) * (size_t)num_envs);
    free(envs);
    return count;
}
