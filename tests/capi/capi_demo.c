/* Plain-C host program against the C ABI (include/uavenv.h): no Python, no torch, no C++.
 * Built and run by tests/test_gpu_capi.py on the GPU box; prints a checksum that the test compares
 * with the same rollout driven through the Python binding. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "uavenv.h"

int main(int argc, char** argv) {
    int E = argc > 1 ? atoi(argv[1]) : 8, steps = argc > 2 ? atoi(argv[2]) : 25;
    UavEnvConfig cfg;
    if (uavenv_default_config(&cfg) != UAVENV_OK) return 2;
    cfg.num_sensors = 20; cfg.grid_w = 200; cfg.grid_h = 200; cfg.max_steps = 10; cfg.duty_cycle = 60.0; cfg.seed = 99;
    cfg.flags |= UAVENV_FLAG_AUTO_RESET;
    UavEnv* env = NULL;
    int rc = uavenv_create(&cfg, E, 0, 0, &env);
    if (rc != UAVENV_OK) { fprintf(stderr, "create failed: %s\n", uavenv_last_error(NULL)); return 3; }
    int D = uavenv_env_obs_dim(env);
    float* obs = (float*)calloc((size_t)E * D, sizeof(float));
    double* rew = (double*)calloc(E, sizeof(double));
    uint8_t* done = (uint8_t*)calloc(E, 1);
    int32_t* act = (int32_t*)calloc(E, sizeof(int32_t));
    if (uavenv_reset_host(env, NULL, obs) != UAVENV_OK) return 4;
    double sum = 0.0; long dones = 0;
    for (int s = 0; s < steps; s++) {
        for (int k = 0; k < E; k++) act[k] = (s * 7 + k * 3) % 5;
        rc = uavenv_step_host(env, act, obs, rew, done, NULL);
        if (rc != UAVENV_OK) { fprintf(stderr, "step failed: %s\n", uavenv_last_error(env)); return 5; }
        for (int k = 0; k < E; k++) { sum += rew[k]; dones += done[k]; }
    }
    double osum = 0.0;
    for (int i = 0; i < E * D; i++) osum += obs[i];
    act[0] = 9;                                            /* invalid action: reported, not fatal */
    rc = uavenv_step_host(env, act, obs, rew, done, NULL);
    /* constants of the live handle: read back, a legal change, an illegal one (buffer sizes are fixed at create) */
    UavEnvConfig live;
    int rc_get = uavenv_get_config(env, &live);
    int same = live.num_sensors == 20 && live.duty_cycle == 60.0 && live.seed == 99;
    live.shadowing_std_db = 0.0;
    int rc_set = uavenv_set_config(env, &live);
    live.num_sensors = 21;
    int rc_bad = uavenv_set_config(env, &live);
    printf("obs_dim=%d reward_sum=%.9f obs_sum=%.6f dones=%ld invalid_rc=%d config=%d/%d/%d/%d\n", D, sum, osum, dones, rc, rc_get, same, rc_set, rc_bad);
    uavenv_destroy(env);
    free(obs); free(rew); free(done); free(act);
    return 0;
}
