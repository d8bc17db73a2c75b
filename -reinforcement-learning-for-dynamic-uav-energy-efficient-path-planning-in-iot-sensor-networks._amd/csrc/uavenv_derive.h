// uavenv_derive.h -- host-only: the default configuration and the constants derived from a configuration, each with the
// reference's own expression.  Shared by the C ABI (uavenv_capi.hip) and by the generator of the compile-time copy of the
// DEFAULT constants (gen_default_consts.cpp -> uavenv_default_consts.inc), so that both come from one place.
#pragma once
#include <cmath>
#include <cstring>

#include "uavenv_internal.h"

// The floating-point constants that exist both in the constants block (struct Consts) and as compile-time literals
// (uavenv_default_consts.inc, struct DefaultConsts in uavenv_kernels.hip): ONE list for the generator and for the test
// that decides whether a handle may run the literal specialisation.
#define UAVENV_LITERAL_CONSTS(X)                                                                                          \
    X(rate) X(bmax) X(thr) X(inv_bmax) X(inv_maxb) X(inv_rate) X(sigma) X(lambda) X(one_minus_lambda) X(tx_power)         \
    X(d_break) X(c_fs) X(fspl_off) X(c_ht) X(c_hr) X(maxb) X(alive_level) X(e_move) X(e_coll)                             \
    X(p_step) X(r_move) X(p_boundary) X(p_battery) X(p_loss)                                                              \
    X(coll_dur) X(p_cycle) X(noise_floor) X(cap_thr) X(e_hover) X(used_hover)                                             \
    X(r_byte) X(r_new) X(r_done) X(r_urg) X(p_revisit) X(p_collision) X(p_hover) X(p_starvation)                          \
    X(p_unvisited) X(p_starved) X(cr_thr) X(fill_lo) X(fill_span) X(min_start_dist) X(prox_eta) X(jain_weight)

namespace uavenv {

inline void fill_default_config(UavEnvConfig* c) {
    std::memset(c, 0, sizeof(*c));
    c->struct_size = (uint32_t)sizeof(*c);
    c->grid_w = 500; c->grid_h = 500;                 // dqn.py EVAL_GRID / BASELINE configs
    c->num_sensors = 20;                              // uav_env.py:270
    c->max_steps = 2100;                              // dqn.py:1069
    c->include_sensor_positions = 0;                  // uav_env.py:286
    c->pad_sensors = 0;
    c->flags = 0;
    c->max_start_tries = 200;                         // dqn.py NAV_CONFIG
    c->use_ema_adr = 1;                               // iot_sensors.py:54
    c->num_grid_choices = 0;
    c->seed = 0;
    c->data_generation_rate = 22.0 / 10;              // uav_env.py:271
    c->max_buffer_size = 1000.0;                      // :272
    c->rssi_threshold = -85.0;                        // :275
    c->duty_cycle = 10.0;                             // :276
    c->start_x = 0.0; c->start_y = 0.0;               // :322-323
    c->max_battery = 274.0;                           // :278
    c->collection_duration = 1.0;                     // :279
    c->tx_power_dbm = 14.0;                           // iot_sensors.py:45
    c->noise_floor_dbm = -105.0;                      // :49
    c->uav_altitude = 100.0;                          // :50
    c->sensor_height = 0.5;                           // :170
    c->wavelength = 0.345;                            // :174
    c->freq_mhz = 868.0; c->fspl_offset_db = 28.0;    // :179
    c->adr_lambda = 0.1;                              // :53
    c->shadowing_std_db = 4.0;                        // :55
    c->capture_threshold_db = 6.0;                    // uav_env.py:567
    c->sf_thresholds[0] = -60.0; c->sf_thresholds[1] = -70.0;   // iot_sensors.py:32-37
    c->sf_thresholds[2] = -78.0; c->sf_thresholds[3] = -85.0;
    c->fill_lo = 0.20; c->fill_hi = 0.60;             // uav_env.py:410
    c->power_move = 500.0; c->power_hover = 700.0;    // uav.py:93-94
    c->alive_fraction = 0.02;                         // uav.py:224
    c->reward_per_byte = 100.0; c->reward_new_sensor = 5000.0; c->reward_completion = 100.0;   // reward_function.py:9-11
    c->reward_urgency_reduction = 20.0;               // uav_env.py:283
    c->reward_movement = 10.0;                        // :285
    c->penalty_revisit = -2.0; c->penalty_boundary = -50.0; c->penalty_collision = -10.0;      // reward_function.py:15-17
    c->penalty_battery = -0.5;                        // uav_env.py:284
    c->penalty_hover = -5.0; c->penalty_step = -0.5;  // reward_function.py:19-20
    c->penalty_data_loss = -1.0;                      // uav_env.py:282
    c->penalty_starvation = -1000.0; c->penalty_unvisited = -5000.0; c->penalty_starved = -1000.0;
    c->starvation_cr_threshold = 0.20;                // reward_function.py:22-25
    c->min_start_dist = 50.0; c->prox_eta = 2.0;      // dqn.py NAV_CONFIG
    c->jain_weight = 0.5;                             // dqn.py:442
}

inline int obs_dim_of(const UavEnvConfig* c) {
    int fps = c->include_sensor_positions ? 5 : 3;
    int slots = c->pad_sensors > c->num_sensors ? c->pad_sensors : c->num_sensors;
    return 3 + fps * slots;
}
// Derived constants: each with the reference's own expression.
inline void derive_consts(const UavEnvConfig& c, Consts& k) {
    std::memset(&k, 0, sizeof(k));
    k.seed = c.seed;
    k.rate = c.data_generation_rate; k.bmax = c.max_buffer_size; k.thr = c.rssi_threshold;
    k.inv_bmax = 1.0 / c.max_buffer_size; k.inv_maxb = 1.0 / c.max_battery;
    k.inv_rate = c.data_generation_rate > 0 ? 1.0 / c.data_generation_rate : 0.0;
    k.p_cycle = c.duty_cycle / 100.0;                                    // iot_sensors.py:105-107
    k.maxb = c.max_battery; k.coll_dur = c.collection_duration;
    k.sigma = c.shadowing_std_db; k.lambda = c.adr_lambda; k.one_minus_lambda = 1 - c.adr_lambda;
    k.tx_power = c.tx_power_dbm; k.noise_floor = c.noise_floor_dbm; k.cap_thr = c.capture_threshold_db;
    k.d_break = (4 * M_PI * c.sensor_height * c.uav_altitude) / c.wavelength;   // iot_sensors.py:174
    k.c_fs = 20 * std::log10(c.freq_mhz); k.fspl_off = c.fspl_offset_db;        // :179
    k.c_ht = 20 * std::log10(c.sensor_height); k.c_hr = 20 * std::log10(c.uav_altitude);   // :183
    for (int i = 0; i < 4; i++) k.sf_thr[i] = c.sf_thresholds[i];
    k.fill_lo = c.fill_lo; k.fill_span = c.fill_hi - c.fill_lo;
    double time_step = 1.0;
    k.e_move = (c.power_move * time_step) / 3600;                        // uav.py:176
    k.e_coll = ((c.power_move * 0.5) * time_step) / 3600;                // uav.py:125,180
    k.e_hover = (c.power_hover * c.collection_duration) / 3600;          // uav.py:204
    k.used_hover = (c.power_hover / (60 * 60)) * c.collection_duration;  // uav.py:260-263, uav_env.py:530
    k.alive_level = c.alive_fraction * c.max_battery;                    // uav.py:224
    k.r_byte = c.reward_per_byte; k.r_new = c.reward_new_sensor; k.r_done = c.reward_completion;
    k.r_urg = c.reward_urgency_reduction; k.r_move = c.reward_movement; k.p_revisit = c.penalty_revisit;
    k.p_boundary = c.penalty_boundary; k.p_collision = c.penalty_collision; k.p_battery = c.penalty_battery;
    k.p_hover = c.penalty_hover; k.p_step = c.penalty_step; k.p_loss = c.penalty_data_loss;
    k.p_starvation = c.penalty_starvation; k.p_unvisited = c.penalty_unvisited; k.p_starved = c.penalty_starved;
    k.cr_thr = c.starvation_cr_threshold;
    k.min_start_dist = c.min_start_dist; k.prox_eta = c.prox_eta; k.jain_weight = c.jain_weight;
    k.alt2 = (float)(c.uav_altitude * c.uav_altitude);                   // iot_sensors.py:164
    k.max_steps = c.max_steps; k.fps = c.include_sensor_positions ? 5 : 3; k.obs_dim = obs_dim_of(&c);
    k.obs_slots = (k.obs_dim - 3) / k.fps;
    k.max_tries = c.max_start_tries; k.use_ema = c.use_ema_adr; k.n_grid_choices = c.num_grid_choices;
    k.flags = c.flags;
    for (int i = 0; i < 8; i++) { k.gw[i] = c.grid_choices_w[i]; k.gh[i] = c.grid_choices_h[i]; }
    k.n_max = c.num_sensors;
    k.inv_small[0] = 0.0;
    for (int i = 1; i <= 64; i++) k.inv_small[i] = 1.0 / (double)i;
}

// true iff every constant that the literal specialisation bakes in is bit-identical to the default configuration's
inline bool consts_are_default(const Consts& k) {
    UavEnvConfig dc;
    fill_default_config(&dc);
    Consts d;
    derive_consts(dc, d);
#define X(f) if (std::memcmp(&k.f, &d.f, sizeof(k.f)) != 0) return false;
    UAVENV_LITERAL_CONSTS(X)
#undef X
    return std::memcmp(&k.alt2, &d.alt2, sizeof(k.alt2)) == 0 && std::memcmp(k.sf_thr, d.sf_thr, sizeof(k.sf_thr)) == 0;
}

}  // namespace uavenv
