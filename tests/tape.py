"""Deterministic, integer-only NOISE TAPE used by the parity tests and the golden-vector generator.

SURVEY.md section 7 (hard part 1): "identical seeds" is redefined as identical *noise*.  The
reference draws, per sensor and per step, up to five standard normals and one uniform:

    slot 0  zA  collect phase ADR sample      (iot_sensors.py:234 via uav_env.py:539)
    slot 1  zB  collect phase link check      (iot_sensors.py:205 via uav_env.py:543)
    slot 2  u   duty-cycle lottery uniform    (uav_env.py:549, stdlib random.random())
    slot 3  zC  collect_data range check      (iot_sensors.py:131 via uav_env.py:581)
    slot 4  zD  observation ADR sample        (iot_sensors.py:234 via uav_env.py:654)
    slot 5  zE  observation in-range check    (iot_sensors.py:215 via uav_env.py:658)
    slot 6  zP  in-range check made by a heuristic policy BEFORE the step (greedy_agents.py:85, :139)

plus, per reset: the buffer pre-fill uniform (uav_env.py:410), the (zD, zE) pair consumed by
the observation built inside reset() (uav_env.py:427) and zS (element 0 only): the ADR sample the DISCARDED
reset observation of DomainRandEnv.reset draws for the old sensor 0, whose SF the fresh sensors inherit
(dqn.py:340-351).

Every value is produced by 64-bit integer hashing (splitmix64) and is exactly representable in
float32, so the tape is bit-identical on every machine and never has to be stored in a fixture:
a fixture stores only `tape_seed`.  The normals are a centred sum of eight 16-bit uniforms scaled by
5*2**-18 (std 1.0206, support +-5.0): close enough to N(0,1) to exercise the realistic regime of
thresholds crossings, and exact.
"""
import numpy as np

SLOT_ZA, SLOT_ZB, SLOT_U, SLOT_ZC, SLOT_ZD, SLOT_ZE, SLOT_ZP = range(7)
NUM_STEP_SLOTS = 7
RSLOT_FILL, RSLOT_ZD, RSLOT_ZE, RSLOT_ZS = range(4)
NUM_RESET_SLOTS = 4

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _hash(seed, stream, a, b, c):
    """64-bit hash of (seed, stream, a, b, c); a/b/c broadcastable uint64 arrays."""
    with np.errstate(over="ignore"):
        h = _splitmix64(np.uint64(seed) ^ np.uint64(0xD1B54A32D192ED03) * np.uint64(stream + 1))
        h = _splitmix64(h ^ a.astype(np.uint64))
        h = _splitmix64(h ^ (b.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)))
        h = _splitmix64(h ^ (c.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)))
        return h


def _normal_from(h0, h1):
    s = np.zeros(h0.shape, dtype=np.int64)
    for h in (h0, h1):
        for k in range(4):
            s += ((h >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.int64)
    return ((s - 262140) * 5).astype(np.float32) * np.float32(2.0 ** -18)


def _uniform_from(h):
    return ((h >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)  # [0,1), 24 bit


def step_tape(tape_seed, env, step, n):
    """float32[7, n] tape for (env, step).  `step` counts vector steps since construction
    (it is NOT reset by episodes), so auto-resets never replay noise."""
    lane = np.arange(n, dtype=np.uint64)
    out = np.empty((NUM_STEP_SLOTS, n), dtype=np.float32)
    e = np.full(n, env, dtype=np.uint64)
    s = np.full(n, step, dtype=np.uint64)
    for slot in range(NUM_STEP_SLOTS):
        h0 = _hash(tape_seed, 2 * slot, e, s, lane)
        if slot == SLOT_U:
            out[slot] = _uniform_from(h0)
        else:
            out[slot] = _normal_from(h0, _hash(tape_seed, 2 * slot + 1, e, s, lane))
    return out


def reset_tape(tape_seed, env, episode, n):
    """float32[4, n] tape for the reset that opens `episode` of `env`: (u_fill, zD, zE, zS)."""
    lane = np.arange(n, dtype=np.uint64)
    e = np.full(n, env, dtype=np.uint64)
    ep = np.full(n, episode, dtype=np.uint64)
    out = np.empty((NUM_RESET_SLOTS, n), dtype=np.float32)
    out[RSLOT_FILL] = _uniform_from(_hash(tape_seed, 100, e, ep, lane))
    out[RSLOT_ZD] = _normal_from(_hash(tape_seed, 101, e, ep, lane), _hash(tape_seed, 102, e, ep, lane))
    out[RSLOT_ZE] = _normal_from(_hash(tape_seed, 103, e, ep, lane), _hash(tape_seed, 104, e, ep, lane))
    out[RSLOT_ZS] = _normal_from(_hash(tape_seed, 105, e, ep, lane), _hash(tape_seed, 106, e, ep, lane))
    return out


def positions(tape_seed, env, n, width, height):
    """float32 sensor positions in [0,W) x [0,H), exactly representable (24-bit uniform * W)."""
    lane = np.arange(n, dtype=np.uint64)
    e = np.full(n, env, dtype=np.uint64)
    z = np.zeros(n, dtype=np.uint64)
    x = _uniform_from(_hash(tape_seed, 200, e, z, lane)) * np.float32(width)
    y = _uniform_from(_hash(tape_seed, 201, e, z, lane)) * np.float32(height)
    return x.astype(np.float32), y.astype(np.float32)


def actions(tape_seed, env, num_steps, p_collect=0.2):
    """int8[num_steps] action sequence: collect (4) with probability p_collect, else a move with a
    drift so the UAV actually crosses the grid (pure uniform moves random-walk around the start)."""
    s = np.arange(num_steps, dtype=np.uint64)
    e = np.full(num_steps, env, dtype=np.uint64)
    z = np.zeros(num_steps, dtype=np.uint64)
    u0 = _uniform_from(_hash(tape_seed, 300, e, s, z))
    u1 = _uniform_from(_hash(tape_seed, 301, e, s, z))
    # moves: UP 0.35, RIGHT 0.35, DOWN 0.15, LEFT 0.15  (reference action ids uav_env.py:497)
    mv = np.where(u1 < 0.35, 0, np.where(u1 < 0.70, 3, np.where(u1 < 0.85, 1, 2)))
    return np.where(u0 < p_collect, 4, mv).astype(np.int8)
