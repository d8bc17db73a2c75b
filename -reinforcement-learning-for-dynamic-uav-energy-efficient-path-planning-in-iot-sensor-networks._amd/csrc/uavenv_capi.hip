// uavenv_capi.hip -- the extern "C" boundary declared in include/uavenv.h.
//
// Owns the HBM state of one shard of environments (SoA sensor arrays + one 128-byte record per
// environment), derives the kernel constants from UavEnvConfig with the reference's own
// expressions, validates arguments on the host and launches the kernels of uavenv_kernels.hip on
// the caller's stream.  Nothing here throws or aborts: every entry point returns a status code.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <cstdlib>
#include <string>
#include <vector>

#include "uavenv_internal.h"
#include "uavenv_derive.h"

using namespace uavenv;

struct UavEnv {
    UavEnvConfig cfg;
    Consts consts;
    Consts* dev_consts = nullptr;   // device copy read by the kernels through the constant address space
    Ptrs ptrs;
    int32_t num_envs = 0, padded_envs = 0, G = 64, device = 0;
    uint32_t env_index_base = 0;
    void* block = nullptr;          // one hipMalloc holding every state array
    size_t block_bytes = 0;
    // staging for the *_host convenience entry points: ONE device block [obs | reward | done | pad | terminal obs] and a
    // pinned host mirror of it, so that a step moves its results with a single device-to-host transfer
    int32_t* h_actions_dev = nullptr; float* h_obs_dev = nullptr; double* h_rew_dev = nullptr;
    uint8_t* h_done_dev = nullptr; float* h_term_dev = nullptr; uint8_t* h_mask_dev = nullptr;
    char* h_block_dev = nullptr; char* h_block_pin = nullptr; int32_t* h_actions_pin = nullptr;
    size_t h_off_rew = 0, h_off_done = 0, h_off_term = 0, h_block_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float* term_pool = nullptr; uint32_t* term_counter = nullptr; int32_t* term_index = nullptr; int32_t term_rows = 0;
    float* aux_out = nullptr; int32_t aux_capacity = 0;
    void* term_block = nullptr;     // terminal snapshot buffers (uavenv_enable_terminal_snapshot)
    uint32_t* hints = nullptr;      // [2][padded_envs] scheduling hints of the random-policy step (StepArgs::balance)
    int hint_parity = 0;
    bool balance = true;            // UAVENV_NO_BALANCE=1 in the environment keeps the home mapping (A/B timing)
    bool default_consts = false;    // constants bit-identical to the default configuration's: the literal specialisation may run
                                    // (UAVENV_NO_LITERALS=1 in the environment at create time keeps the generic kernels, for A/B runs)
    bool allow_literals = true;
    bool write_through = false;     // sc1 stores for observations / sensor state (StepArgs::write_through): batches of >= 4096 wavefronts;
                                    // UAVENV_WRITE_THROUGH=0 / 1 in the environment at create time forces it off / on (A/B runs)
    std::string err;
};

static thread_local std::string g_create_error;

static int fail(UavEnv* e, int code, const std::string& msg) {
    if (e) e->err = msg; else g_create_error = msg;
    return code;
}
#define HIP_TRY(e, call)                                                                         \
    do {                                                                                         \
        hipError_t _s = (call);                                                                  \
        if (_s != hipSuccess)                                                                    \
            return fail((e), UAVENV_E_HIP, std::string(#call) + ": " + hipGetErrorString(_s));   \
    } while (0)

extern "C" int uavenv_abi_version(void) { return UAVENV_ABI_VERSION; }

extern "C" int uavenv_default_config(UavEnvConfig* c) {
    if (!c) return UAVENV_E_INVALID;
    fill_default_config(c);
    return UAVENV_OK;
}

extern "C" int uavenv_obs_dim(const UavEnvConfig* c) { return c ? obs_dim_of(c) : UAVENV_E_INVALID; }

static size_t field_elem_bytes(int field) {
    switch (field) {
        case UAVENV_F_POS_X: case UAVENV_F_POS_Y: return 4;
        case UAVENV_F_BUFFER: case UAVENV_F_GEN: case UAVENV_F_TX: case UAVENV_F_LOST: case UAVENV_F_AVG_RSSI: return 8;
        case UAVENV_F_FLAGS: return 4;
        default: return 0;
    }
}
static void* field_ptr(UavEnv* e, int field) {
    char* sb = e->ptrs.sensor_base;
    const uint64_t S = e->ptrs.lanes;
    switch (field) {
        case UAVENV_F_POS_X: return sb + kOffPosX * S;     case UAVENV_F_POS_Y: return sb + kOffPosY * S;
        case UAVENV_F_BUFFER: return sb + kOffBuffer * S;  case UAVENV_F_GEN: return sb + kOffGen * S;
        case UAVENV_F_TX: return sb + kOffTx * S;          case UAVENV_F_LOST: return sb + kOffLost * S;
        case UAVENV_F_AVG_RSSI: return sb + kOffAvg * S;   case UAVENV_F_FLAGS: return sb + kOffFlags * S;
        case UAVENV_F_RECORD: return e->ptrs.rec;          case UAVENV_F_EPISODE_STATS: return e->ptrs.stats;
        case UAVENV_F_TERM_RECORD: return e->ptrs.term_rec; case UAVENV_F_TERM_SENSORS: return e->ptrs.term_sensors;
        default: return nullptr;
    }
}
extern "C" size_t uavenv_state_bytes(const UavEnv* e, int32_t field) {
    if (!e) return 0;
    if (field == UAVENV_F_RECORD) return (size_t)e->num_envs * sizeof(UavEnvRecord);
    if (field == UAVENV_F_EPISODE_STATS) return (size_t)e->num_envs * sizeof(UavEnvEpisodeStats);
    if (field == UAVENV_F_TERM_RECORD) return (size_t)e->num_envs * sizeof(UavEnvRecord);
    if (field == UAVENV_F_TERM_SENSORS) return (size_t)e->num_envs * 3u * (size_t)e->G * sizeof(double);
    return (size_t)e->num_envs * (size_t)e->G * field_elem_bytes(field);
}

extern "C" int uavenv_create(const UavEnvConfig* cfg, int32_t num_envs, uint32_t env_index_base, int32_t device,
                             UavEnv** out) {
    if (!cfg || !out) return fail(nullptr, UAVENV_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(UavEnvConfig)) return fail(nullptr, UAVENV_E_INVALID, "UavEnvConfig.struct_size mismatch (ABI)");
    if (num_envs <= 0) return fail(nullptr, UAVENV_E_INVALID, "num_envs must be positive");
    if (cfg->num_sensors < 1 || cfg->num_sensors > 64) return fail(nullptr, UAVENV_E_INVALID, "num_sensors must be in 1..64");
    if (cfg->grid_w < 1 || cfg->grid_h < 1) return fail(nullptr, UAVENV_E_INVALID, "grid must be positive");
    if (!(cfg->max_buffer_size > 0) || !(cfg->max_battery > 0)) return fail(nullptr, UAVENV_E_INVALID, "max_buffer_size and max_battery must be positive");
    if (cfg->num_grid_choices < 0 || cfg->num_grid_choices > 8) return fail(nullptr, UAVENV_E_INVALID, "num_grid_choices must be in 0..8");
    static_assert(sizeof(UavEnvRecord) == 128, "UavEnvRecord must be 128 bytes");
    UavEnv* e = new (std::nothrow) UavEnv();
    if (!e) return fail(nullptr, UAVENV_E_ALLOC, "out of host memory");
    e->cfg = *cfg;
    e->num_envs = num_envs; e->env_index_base = env_index_base; e->device = device;
    e->G = cfg->num_sensors <= 16 ? 16 : (cfg->num_sensors <= 32 ? 32 : 64);
    {   // tuning knob: a wider lane group than the sensor count needs (UAVENV_LANE_GROUP = 32 or 64)
        const char* lg = getenv("UAVENV_LANE_GROUP");
        const int want = lg ? atoi(lg) : 0;
        if ((want == 32 || want == 64) && want > e->G) e->G = want;
    }
    const int per_block = kBlockThreads / e->G;
    e->padded_envs = ((num_envs + per_block - 1) / per_block) * per_block;
    derive_consts(e->cfg, e->consts);
    { const char* nl = getenv("UAVENV_NO_LITERALS"); e->allow_literals = !(nl && nl[0] == '1'); }
#ifdef UAV_FORCE_NOLIT      // A/B timing builds (tools/exp.sh)
    e->allow_literals = false;
#endif
    e->default_consts = e->allow_literals && consts_are_default(e->consts);
    e->write_through = step_uses_big_workgroups(e->G, e->padded_envs);
    { const char* wt = getenv("UAVENV_WRITE_THROUGH"); if (wt && (wt[0] == '0' || wt[0] == '1')) e->write_through = wt[0] == '1'; }
    auto bail = [&](int code, const std::string& m) { uavenv_destroy(e); return fail(nullptr, code, m); };
    hipError_t st = hipSetDevice(device);
    if (st != hipSuccess) return bail(UAVENV_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(st));
    const size_t P = (size_t)e->padded_envs, S = P * (size_t)e->G;
    if (S * 8 >= ((size_t)1 << 32)) { delete e; return fail(nullptr, UAVENV_E_INVALID, "too many environments for one handle (state rows must stay below 4 GiB): shard them"); }
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    size_t o_sens = off; off += al(S * kSensorBytesPerLane);      // pos_x | pos_y | buffer | gen | tx | lost | avg | flags
    size_t o_r = off; off += al(P * sizeof(UavEnvRecord));
    size_t o_s = off; off += al(P * sizeof(UavEnvEpisodeStats));
    size_t o_st = off; off += 256;
    size_t o_c = off; off += al(sizeof(Consts));
    size_t o_h = off; off += al(2 * P * sizeof(uint32_t));
    e->block_bytes = off;
    st = hipMalloc(&e->block, off);
    if (st != hipSuccess) return bail(UAVENV_E_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(st));
    char* base = (char*)e->block;
    e->ptrs.sensor_base = base + o_sens; e->ptrs.lanes = S;
    e->ptrs.rec = (UavEnvRecord*)(base + o_r); e->ptrs.stats = (UavEnvEpisodeStats*)(base + o_s);
    e->ptrs.status = (uint32_t*)(base + o_st);
    e->dev_consts = (Consts*)(base + o_c);
    e->hints = (uint32_t*)(base + o_h);
    { const char* nb = getenv("UAVENV_NO_BALANCE"); e->balance = !(nb && nb[0] == '1'); }
    e->ptrs.step_tape = nullptr; e->ptrs.reset_tape = nullptr; e->ptrs.stamps = nullptr;
    e->ptrs.term_rec = nullptr; e->ptrs.term_sensors = nullptr;
    st = hipMemset(e->block, 0, off);
    if (st != hipSuccess) return bail(UAVENV_E_HIP, std::string("hipMemset: ") + hipGetErrorString(st));
    if (st == hipSuccess) st = hipMemcpy(e->dev_consts, &e->consts, sizeof(Consts), hipMemcpyHostToDevice);
    if (st != hipSuccess) return bail(UAVENV_E_HIP, std::string("consts upload: ") + hipGetErrorString(st));
    st = launch_init(e->G, e->padded_envs, e->consts, e->dev_consts, e->ptrs, env_index_base, cfg->grid_w, cfg->grid_h, cfg->num_sensors,
                     (float)cfg->start_x, (float)cfg->start_y, nullptr);
    if (st == hipSuccess) st = hipDeviceSynchronize();
    if (st != hipSuccess) return bail(UAVENV_E_HIP, std::string("init kernel: ") + hipGetErrorString(st));
    if (hipEventCreate(&e->ev0) != hipSuccess || hipEventCreate(&e->ev1) != hipSuccess)
        return bail(UAVENV_E_HIP, "hipEventCreate failed");
    *out = e;
    return UAVENV_OK;
}

extern "C" int uavenv_destroy(UavEnv* e) {
    if (!e) return UAVENV_OK;
    if (e->block) (void)hipFree(e->block);
    if (e->term_block) (void)hipFree(e->term_block);
    void* tmp[] = {e->h_actions_dev, e->h_block_dev, e->h_mask_dev};
    for (void* t : tmp) if (t) (void)hipFree(t);
    if (e->h_block_pin) (void)hipHostFree(e->h_block_pin);
    if (e->h_actions_pin) (void)hipHostFree(e->h_actions_pin);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
    return UAVENV_OK;
}

extern "C" const char* uavenv_last_error(const UavEnv* e) { return e ? e->err.c_str() : g_create_error.c_str(); }
extern "C" int uavenv_num_envs(const UavEnv* e) { return e ? e->num_envs : UAVENV_E_INVALID; }
extern "C" int uavenv_lane_stride(const UavEnv* e) { return e ? e->G : UAVENV_E_INVALID; }
extern "C" int uavenv_env_obs_dim(const UavEnv* e) { return e ? e->consts.obs_dim : UAVENV_E_INVALID; }

extern "C" int uavenv_set_env_params(UavEnv* e, const int32_t* gw, const int32_t* gh, const int32_t* ns) {
    if (!e) return UAVENV_E_INVALID;
    std::vector<UavEnvRecord> recs((size_t)e->num_envs);
    HIP_TRY(e, hipMemcpy(recs.data(), e->ptrs.rec, recs.size() * sizeof(UavEnvRecord), hipMemcpyDeviceToHost));
    for (int i = 0; i < e->num_envs; i++) {
        if (gw) { if (gw[i] < 1) return fail(e, UAVENV_E_INVALID, "grid_w must be positive"); recs[i].grid_w = gw[i]; recs[i].inv_grid_w = 1.0 / gw[i]; }
        if (gh) { if (gh[i] < 1) return fail(e, UAVENV_E_INVALID, "grid_h must be positive"); recs[i].grid_h = gh[i]; recs[i].inv_grid_h = 1.0 / gh[i]; }
        if (ns) {
            if (ns[i] < 1 || ns[i] > e->cfg.num_sensors) return fail(e, UAVENV_E_INVALID, "per-env num_sensors must be in 1..cfg.num_sensors");
            recs[i].num_sensors = ns[i];
        }
    }
    HIP_TRY(e, hipMemcpy(e->ptrs.rec, recs.data(), recs.size() * sizeof(UavEnvRecord), hipMemcpyHostToDevice));
    return UAVENV_OK;
}

extern "C" int uavenv_set_positions(UavEnv* e, const float* px, const float* py) {
    if (!e || !px || !py) return UAVENV_E_INVALID;
    size_t bytes = (size_t)e->num_envs * e->G * 4;
    HIP_TRY(e, hipMemcpy(field_ptr(e, UAVENV_F_POS_X), px, bytes, hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(field_ptr(e, UAVENV_F_POS_Y), py, bytes, hipMemcpyHostToDevice));
    return UAVENV_OK;
}

// The kernels of earlier launches may still be reading the constants: order the update behind them.
static int upload_consts(UavEnv* e) {
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipMemcpy(e->dev_consts, &e->consts, sizeof(Consts), hipMemcpyHostToDevice));
    return UAVENV_OK;
}

extern "C" int uavenv_set_seed(UavEnv* e, uint64_t seed) {
    if (!e) return UAVENV_E_INVALID;
    e->cfg.seed = seed; e->consts.seed = seed;
    int rc = upload_consts(e);                              // synchronises the device
    if (rc) return rc;
    // the action words the last random-policy launch left behind were drawn with the old seed
    HIP_TRY(e, hipMemset(e->hints, 0, 2 * (size_t)e->padded_envs * sizeof(uint32_t)));
    return UAVENV_OK;
}

extern "C" int uavenv_get_config(const UavEnv* e, UavEnvConfig* out) {
    if (!e || !out) return UAVENV_E_INVALID;
    *out = e->cfg;
    return UAVENV_OK;
}

// Re-derive the constants of a LIVE handle (sim_to_real_sweep.py:109-117 sets shadowing_std_db / path-loss parameters on
// the sensors of an existing environment).  What sizes the buffers and the observation rows must stay as created.
extern "C" int uavenv_set_config(UavEnv* e, const UavEnvConfig* cfg) {
    if (!e || !cfg) return UAVENV_E_INVALID;
    if (cfg->struct_size != sizeof(UavEnvConfig)) return fail(e, UAVENV_E_INVALID, "UavEnvConfig.struct_size mismatch (ABI)");
    if (cfg->num_sensors != e->cfg.num_sensors || cfg->pad_sensors != e->cfg.pad_sensors ||
        (cfg->include_sensor_positions != 0) != (e->cfg.include_sensor_positions != 0))
        return fail(e, UAVENV_E_INVALID, "num_sensors, pad_sensors and include_sensor_positions are fixed at uavenv_create");
    if (!(cfg->max_buffer_size > 0) || !(cfg->max_battery > 0)) return fail(e, UAVENV_E_INVALID, "max_buffer_size and max_battery must be positive");
    if (cfg->num_grid_choices < 0 || cfg->num_grid_choices > 8) return fail(e, UAVENV_E_INVALID, "num_grid_choices must be in 0..8");
    const bool reseeded = cfg->seed != e->cfg.seed;
    e->cfg = *cfg;
    derive_consts(e->cfg, e->consts);
    e->default_consts = e->allow_literals && consts_are_default(e->consts);
    int rc = upload_consts(e);                              // synchronises the device
    if (rc) return rc;
    if (reseeded) HIP_TRY(e, hipMemset(e->hints, 0, 2 * (size_t)e->padded_envs * sizeof(uint32_t)));
    return UAVENV_OK;
}

extern "C" int uavenv_set_grid_choices(UavEnv* e, int32_t count, const int32_t* w, const int32_t* h) {
    if (!e || count < 0 || count > 8 || (count > 0 && (!w || !h))) return UAVENV_E_INVALID;
    e->cfg.num_grid_choices = count; e->consts.n_grid_choices = count;
    for (int i = 0; i < count; i++) {
        if (w[i] < 1 || h[i] < 1) return fail(e, UAVENV_E_INVALID, "grid choice must be positive");
        e->cfg.grid_choices_w[i] = e->consts.gw[i] = w[i];
        e->cfg.grid_choices_h[i] = e->consts.gh[i] = h[i];
    }
    return upload_consts(e);
}

extern "C" int uavenv_set_noise_tape(UavEnv* e, const float* step_tape_dev, const float* reset_tape_dev) {
    if (!e) return UAVENV_E_INVALID;
    e->ptrs.step_tape = step_tape_dev; e->ptrs.reset_tape = reset_tape_dev;
    return UAVENV_OK;
}

extern "C" int uavenv_dump_noise(UavEnv* e, float* step_tape_out, float* reset_tape_out, void* stream) {
    if (!e) return UAVENV_E_INVALID;
    HIP_TRY(e, launch_dump_noise(e->G, e->padded_envs, e->consts, e->dev_consts, e->ptrs, step_tape_out, reset_tape_out, e->num_envs,
                                 (hipStream_t)stream));
    return UAVENV_OK;
}

extern "C" int uavenv_reset(UavEnv* e, const uint8_t* mask_dev, float* obs_out_dev, void* stream) {
    if (!e) return UAVENV_E_INVALID;
    ResetArgs a{mask_dev, obs_out_dev, e->num_envs};
    HIP_TRY(e, launch_reset(e->G, e->padded_envs, e->consts, e->dev_consts, e->ptrs, a, (hipStream_t)stream));
    return UAVENV_OK;
}

static int step_common(UavEnv* e, int32_t policy, const int32_t* actions, int32_t* actions_out, float* obs, double* rew,
                       float* rew32, uint8_t* done, float* term, void* stream, float* aux_override = nullptr) {
    if (!e) return UAVENV_E_INVALID;
    if (policy < UAVENV_POLICY_ACTIONS || policy > UAVENV_POLICY_MAX_THROUGHPUT_V2) return fail(e, UAVENV_E_INVALID, "unknown policy");
    if (policy == UAVENV_POLICY_ACTIONS && !actions) return fail(e, UAVENV_E_INVALID, "actions_dev is NULL");
    StepArgs a{};
    a.actions = actions; a.num_envs = e->num_envs; a.policy = policy;
    a.out.actions_out = actions_out; a.out.obs = obs; a.out.reward = rew; a.out.reward32 = rew32; a.out.done = done; a.out.term_obs = term;
    a.out.term_pool = e->term_pool; a.out.term_counter = e->term_counter; a.out.term_index = e->term_index; a.out.term_rows = e->term_rows;
    a.out.aux = aux_override ? aux_override : e->aux_out; a.out.write_through = e->write_through ? 1 : 0;
    a.hint_in = e->hints + (size_t)e->hint_parity * (size_t)e->padded_envs;          // always a readable buffer
    a.balance = e->balance ? 1 : 0;
    if (policy == UAVENV_POLICY_RANDOM) {                         // this launch leaves the next launch's action words
        a.out.hint_out = e->hints + (size_t)(e->hint_parity ^ 1) * (size_t)e->padded_envs;
        e->hint_parity ^= 1;
    } else if (policy != UAVENV_POLICY_ACTIONS) a.balance = 0;   // no cheap way to know the actions up front
    HIP_TRY(e, launch_step(e->G, e->padded_envs, e->consts, e->dev_consts, e->ptrs, a, e->default_consts, (hipStream_t)stream));
    return UAVENV_OK;
}

extern "C" int uavenv_step(UavEnv* e, const int32_t* actions_dev, float* obs, double* rew, float* rew32, uint8_t* done,
                           float* term, void* stream) {
    return step_common(e, UAVENV_POLICY_ACTIONS, actions_dev, nullptr, obs, rew, rew32, done, term, stream);
}

extern "C" int uavenv_step_random(UavEnv* e, int32_t* actions_out, float* obs, double* rew, float* rew32, uint8_t* done,
                                  float* term, void* stream) {
    return step_common(e, UAVENV_POLICY_RANDOM, nullptr, actions_out, obs, rew, rew32, done, term, stream);
}

// n consecutive single-step launches issued from ONE call: the host side of a replay-ring chunk.  Per launch the C loop costs
// ~2.7 us of host time (less than the 9 us kernel at 4096 environments, so the GPU never starves) and the first kernel starts
// after one packet instead of after a whole graph has been submitted.
extern "C" int uavenv_step_random_n(UavEnv* e, int32_t num_steps, float* obs_out_dev, int64_t obs_stride, float* aux_out_dev,
                                    int64_t aux_stride, float* reward32_out_dev, uint8_t* done_out_dev, void* stream) {
    if (!e) return UAVENV_E_INVALID;
    if (num_steps <= 0 || !obs_out_dev) return fail(e, UAVENV_E_INVALID, "uavenv_step_random_n: num_steps must be positive and obs_out_dev set");
    if (((uintptr_t)aux_out_dev & 15u) != 0 || (aux_stride & 3) != 0) return fail(e, UAVENV_E_INVALID, "aux output must be 16-byte aligned");
    for (int32_t k = 0; k < num_steps; k++) {
        int rc = step_common(e, UAVENV_POLICY_RANDOM, nullptr, nullptr, obs_out_dev + (size_t)k * (size_t)obs_stride, nullptr, reward32_out_dev,
                             done_out_dev, nullptr, stream, aux_out_dev ? aux_out_dev + (size_t)k * (size_t)aux_stride : nullptr);
        if (rc) return rc;
    }
    return UAVENV_OK;
}

extern "C" int uavenv_step_policy(UavEnv* e, int32_t policy, int32_t* actions_out, float* obs, double* rew, float* rew32,
                                  uint8_t* done, float* term, void* stream) {
    if (policy == UAVENV_POLICY_ACTIONS) return e ? fail(e, UAVENV_E_INVALID, "use uavenv_step for given actions") : UAVENV_E_INVALID;
    return step_common(e, policy, nullptr, actions_out, obs, rew, rew32, done, term, stream);
}

extern "C" int uavenv_rollout(UavEnv* e, int32_t num_steps, int32_t policy, const int32_t* actions_dev, int32_t* actions_out,
                              float* obs, double* rew, float* rew32, uint8_t* done, float* term, void* stream) {
    if (!e) return UAVENV_E_INVALID;
    if (num_steps <= 0) return fail(e, UAVENV_E_INVALID, "num_steps must be positive");
    if (policy < UAVENV_POLICY_ACTIONS || policy > UAVENV_POLICY_MAX_THROUGHPUT_V2) return fail(e, UAVENV_E_INVALID, "unknown policy");
    if (policy == UAVENV_POLICY_ACTIONS && !actions_dev) return fail(e, UAVENV_E_INVALID, "actions_dev is NULL");
    if (e->aux_out != nullptr && num_steps > e->aux_capacity)
        return fail(e, UAVENV_E_INVALID, "rollout of more steps than the attached aux output holds ([K][E][4] blocks): detach it or attach a larger one");
    StepArgs a{};
    a.actions = actions_dev; a.num_envs = e->num_envs; a.policy = policy; a.hint_in = nullptr;
    a.out.actions_out = actions_out; a.out.obs = obs; a.out.reward = rew; a.out.reward32 = rew32; a.out.done = done; a.out.term_obs = term;
    a.out.term_pool = e->term_pool; a.out.term_counter = e->term_counter; a.out.term_index = nullptr; a.out.term_rows = e->term_rows;
    a.out.aux = e->aux_out; a.out.write_through = e->write_through ? 1 : 0;       // aux [K][E][4] carries the tickets
    HIP_TRY(e, launch_rollout(e->G, e->padded_envs, e->consts, e->dev_consts, e->ptrs, a, num_steps, e->default_consts, (hipStream_t)stream));
    return UAVENV_OK;
}

extern "C" int uavenv_set_terminal_pool(UavEnv* e, float* pool_dev, int32_t rows, uint32_t* counter_dev, int32_t* index_out_dev) {
    if (!e) return UAVENV_E_INVALID;
    if (pool_dev != nullptr && (rows <= 0 || counter_dev == nullptr))
        return fail(e, UAVENV_E_INVALID, "terminal pool needs rows > 0 and a counter");
    e->term_pool = pool_dev; e->term_rows = rows; e->term_counter = counter_dev; e->term_index = index_out_dev;
    return UAVENV_OK;
}

extern "C" int uavenv_enable_terminal_snapshot(UavEnv* e, int32_t enable) {
    if (!e) return UAVENV_E_INVALID;
    HIP_TRY(e, hipDeviceSynchronize());                // launches in flight may still write the old buffers
    if (!enable) {
        if (e->term_block) (void)hipFree(e->term_block);
        e->term_block = nullptr; e->ptrs.term_rec = nullptr; e->ptrs.term_sensors = nullptr;
        return UAVENV_OK;
    }
    if (e->term_block) return UAVENV_OK;
    const size_t P = (size_t)e->padded_envs;
    const size_t rec_bytes = (P * sizeof(UavEnvRecord) + 255) & ~(size_t)255, sens_bytes = P * 3u * (size_t)e->G * sizeof(double);
    if (hipMalloc(&e->term_block, rec_bytes + sens_bytes) != hipSuccess) { e->term_block = nullptr; return fail(e, UAVENV_E_ALLOC, "terminal snapshot: out of device memory"); }
    HIP_TRY(e, hipMemset(e->term_block, 0, rec_bytes + sens_bytes));
    e->ptrs.term_rec = (UavEnvRecord*)e->term_block;
    e->ptrs.term_sensors = (double*)((char*)e->term_block + rec_bytes);
    return UAVENV_OK;
}

extern "C" int uavenv_set_aux_output(UavEnv* e, float* aux_out_dev, int32_t capacity_steps) {
    if (!e) return UAVENV_E_INVALID;
    if (aux_out_dev != nullptr && capacity_steps < 1) return fail(e, UAVENV_E_INVALID, "aux output needs capacity_steps >= 1");
    if (((uintptr_t)aux_out_dev & 15u) != 0) return fail(e, UAVENV_E_INVALID, "aux output must be 16-byte aligned");
    e->aux_out = aux_out_dev; e->aux_capacity = aux_out_dev ? capacity_steps : 0;
    return UAVENV_OK;
}

extern "C" int uavenv_frame_stack(float* stacked_dev, const float* obs_dev, const uint8_t* done_dev,
                                  const float* terminal_obs_dev, float* terminal_stacked_dev, int32_t num_envs,
                                  int32_t num_frames, int32_t obs_dim, void* stream) {
    if (!stacked_dev || !obs_dev || num_envs <= 0) return UAVENV_E_INVALID;
    hipError_t st = launch_frame_stack(stacked_dev, obs_dev, done_dev, terminal_obs_dev, terminal_stacked_dev, num_envs,
                                       num_frames, obs_dim, (hipStream_t)stream);
    if (st != hipSuccess) return fail(nullptr, st == hipErrorInvalidValue ? UAVENV_E_INVALID : UAVENV_E_HIP,
                                      std::string("frame_stack: ") + hipGetErrorString(st));
    return UAVENV_OK;
}

extern "C" int uavenv_get_state(UavEnv* e, int32_t field, void* dst, size_t bytes, int32_t dst_on_device, void* stream) {
    if (!e || !dst) return UAVENV_E_INVALID;
    void* src = field_ptr(e, field);
    if (!src) return fail(e, UAVENV_E_INVALID, field == UAVENV_F_TERM_RECORD || field == UAVENV_F_TERM_SENSORS
                                                   ? "terminal snapshot is not enabled (uavenv_enable_terminal_snapshot)" : "unknown state field");
    if (bytes != uavenv_state_bytes(e, field)) return fail(e, UAVENV_E_INVALID, "state field size mismatch");
    HIP_TRY(e, hipMemcpyAsync(dst, src, bytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (!dst_on_device) HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
    return UAVENV_OK;
}

extern "C" int uavenv_set_state(UavEnv* e, int32_t field, const void* src, size_t bytes, int32_t src_on_device, void* stream) {
    if (!e || !src) return UAVENV_E_INVALID;
    void* dst = field_ptr(e, field);
    if (!dst) return fail(e, UAVENV_E_INVALID, "unknown state field");
    if (bytes != uavenv_state_bytes(e, field)) return fail(e, UAVENV_E_INVALID, "state field size mismatch");
    // records may change under the action words of the last random-policy launch: drop them (they are re-drawn)
    HIP_TRY(e, hipMemsetAsync(e->hints, 0, 2 * (size_t)e->padded_envs * sizeof(uint32_t), (hipStream_t)stream));
    HIP_TRY(e, hipMemcpyAsync(dst, src, bytes, src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, (hipStream_t)stream));
    if (!src_on_device) HIP_TRY(e, hipStreamSynchronize((hipStream_t)stream));
    return UAVENV_OK;
}

static int ensure_host_staging(UavEnv* e) {
    if (e->h_block_dev) return UAVENV_OK;
    const size_t E = (size_t)e->num_envs, D = (size_t)e->consts.obs_dim;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    e->h_off_rew = al(E * D * 4); e->h_off_done = e->h_off_rew + al(E * 8); e->h_off_term = e->h_off_done + al(E + 16);   // + the status word
    e->h_block_bytes = e->h_off_term + al(E * D * 4);
    HIP_TRY(e, hipMalloc((void**)&e->h_actions_dev, E * 4));
    HIP_TRY(e, hipMalloc((void**)&e->h_mask_dev, E));
    HIP_TRY(e, hipMalloc((void**)&e->h_block_dev, e->h_block_bytes));
    HIP_TRY(e, hipHostMalloc((void**)&e->h_block_pin, e->h_block_bytes, hipHostMallocDefault));
    HIP_TRY(e, hipHostMalloc((void**)&e->h_actions_pin, E * 4, hipHostMallocDefault));
    e->h_obs_dev = (float*)e->h_block_dev; e->h_rew_dev = (double*)(e->h_block_dev + e->h_off_rew);
    e->h_done_dev = (uint8_t*)(e->h_block_dev + e->h_off_done); e->h_term_dev = (float*)(e->h_block_dev + e->h_off_term);
    return UAVENV_OK;
}

static int check_status(UavEnv* e) {
    uint32_t st = 0;
    HIP_TRY(e, hipMemcpy(&st, e->ptrs.status, 4, hipMemcpyDeviceToHost));
    if (st & 1u) {
        HIP_TRY(e, hipMemset(e->ptrs.status, 0, 4));
        return fail(e, UAVENV_E_ACTION, "Invalid action: outside 0..4 (uav_env.py:468)");
    }
    return UAVENV_OK;
}

extern "C" int uavenv_reset_host(UavEnv* e, const uint8_t* mask, float* obs_out) {
    if (!e || !obs_out) return UAVENV_E_INVALID;
    int rc = ensure_host_staging(e); if (rc) return rc;
    size_t E = (size_t)e->num_envs, D = (size_t)e->consts.obs_dim;
    if (mask) HIP_TRY(e, hipMemcpy(e->h_mask_dev, mask, E, hipMemcpyHostToDevice));
    if (mask) HIP_TRY(e, hipMemcpy(e->h_obs_dev, obs_out, E * D * 4, hipMemcpyHostToDevice));   // keep unmasked rows
    rc = uavenv_reset(e, mask ? e->h_mask_dev : nullptr, e->h_obs_dev, nullptr); if (rc) return rc;
    HIP_TRY(e, hipMemcpy(obs_out, e->h_obs_dev, E * D * 4, hipMemcpyDeviceToHost));
    return UAVENV_OK;
}

extern "C" int uavenv_step_host(UavEnv* e, const int32_t* actions, float* obs_out, double* reward_out, uint8_t* done_out,
                                float* terminal_obs_out) {
    if (!e || !actions || !obs_out) return UAVENV_E_INVALID;
    int rc = ensure_host_staging(e); if (rc) return rc;
    const size_t E = (size_t)e->num_envs, D = (size_t)e->consts.obs_dim;
    for (size_t i = 0; i < E; i++)
        if (actions[i] < 0 || actions[i] > 4) return fail(e, UAVENV_E_ACTION, "Invalid action: outside 0..4 (uav_env.py:468)");
    // actions through pinned memory, the step, then ONE transfer of [obs | reward | done | status] (and the terminal rows
    // when asked for) into the pinned mirror: 4096 x 50 = 2.56 MB in ~55 us (tools/pcie_probe.py), then plain memcpys
    std::memcpy(e->h_actions_pin, actions, E * 4);
    HIP_TRY(e, hipMemcpyAsync(e->h_actions_dev, e->h_actions_pin, E * 4, hipMemcpyHostToDevice, nullptr));
    rc = uavenv_step(e, e->h_actions_dev, e->h_obs_dev, e->h_rew_dev, nullptr, e->h_done_dev,
                     terminal_obs_out ? e->h_term_dev : nullptr, nullptr);
    if (rc) return rc;
    // the status word rides in the padding behind the done flags (host-validated actions make it zero unless another
    // entry point flagged something since the last check)
    HIP_TRY(e, hipMemcpyAsync(e->h_block_dev + e->h_off_term - 16, e->ptrs.status, 4, hipMemcpyDeviceToDevice, nullptr));
    const size_t bytes = terminal_obs_out ? e->h_block_bytes : e->h_off_term;
    HIP_TRY(e, hipMemcpyAsync(e->h_block_pin, e->h_block_dev, bytes, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(e, hipStreamSynchronize(nullptr));
    std::memcpy(obs_out, e->h_block_pin, E * D * 4);
    if (reward_out) std::memcpy(reward_out, e->h_block_pin + e->h_off_rew, E * 8);
    if (done_out) std::memcpy(done_out, e->h_block_pin + e->h_off_done, E);
    if (terminal_obs_out) std::memcpy(terminal_obs_out, e->h_block_pin + e->h_off_term, E * D * 4);
    uint32_t st;
    std::memcpy(&st, e->h_block_pin + e->h_off_term - 16, 4);
    return (st & 1u) ? check_status(e) : UAVENV_OK;
}

extern "C" int uavenv_time_steps(UavEnv* e, int32_t steps, float* obs, double* rew, uint8_t* done, void* stream,
                                 float* avg_ms) {
    if (!e || steps <= 0 || !avg_ms) return UAVENV_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(e, hipEventRecord(e->ev0, s));
    for (int i = 0; i < steps; i++) {
        int rc = step_common(e, UAVENV_POLICY_RANDOM, nullptr, nullptr, obs, rew, nullptr, done, nullptr, stream);
        if (rc) return rc;
    }
    HIP_TRY(e, hipEventRecord(e->ev1, s));
    HIP_TRY(e, hipEventSynchronize(e->ev1));
    float ms = 0.f;
    HIP_TRY(e, hipEventElapsedTime(&ms, e->ev0, e->ev1));
    *avg_ms = ms / (float)steps;
    return UAVENV_OK;
}

#ifdef UAVENV_STAMPS
// diagnostic build only: in-kernel timestamps (8 x uint64 per wavefront of the step kernel)
extern "C" int uavenv_debug_set_stamps(UavEnv* e, unsigned long long* stamps_dev) {
    if (!e) return UAVENV_E_INVALID;
    e->ptrs.stamps = stamps_dev;
    return UAVENV_OK;
}
#endif
