"""Drive the REAL reference environment under an injected, slot-addressed noise tape.

Only usable in the build container, where the reference is mounted at /root/reference; nothing in
`tests/` imports this module at test time (the GPU box has no reference).  It is used by
`make_golden.py` to produce the committed fixtures and by `check_oracle_vs_reference.py`.

Nothing is copied from the reference: its modules are imported from where they lie, behind the
test-only `gymnasium` stand-in in `tests/oracle_stub/`.

Injection points (all process-local monkey patches, restored on exit):
  * `numpy.random.normal`      -> tape normal; the slot is derived from the *call stack*:
        calculate_rssi <- update_spreading_factor <- _execute_collect_action   => zA
        calculate_rssi <- get_success_probability(use_advanced_model=True)     => zB
        calculate_rssi <- get_success_probability(False) <- collect_data       => zC
        calculate_rssi <- update_spreading_factor <- _get_observation          => zD
        calculate_rssi <- is_in_range <- _get_observation                      => zE
        calculate_rssi <- is_in_range <- (a greedy agent's select_action)      => zP
  * `uav_env.random`           -> object whose .random() returns the lottery uniform of the sensor
                                  currently iterated at uav_env.py:535-551
  * `env.np_random`            -> object whose .uniform(lo, hi) returns lo + (hi-lo)*u_fill[k] for
                                  the k-th sensor of the reset loop at uav_env.py:409-410
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
REF_SRC = "/root/reference/src"

sys.path.insert(0, TESTS)
import tape as T  # noqa: E402


def import_reference():
    stub = os.path.join(TESTS, "oracle_stub")
    for p in (os.path.join(REF_SRC, "environment"), REF_SRC, stub):
        if p not in sys.path:
            sys.path.insert(0, p)
    import matplotlib
    matplotlib.use("Agg")
    import uav_env  # noqa: E402  (the reference module itself)
    import iot_sensors  # noqa: E402
    return uav_env, iot_sensors


class _TapeState:
    def __init__(self):
        self.step = None      # float32[6, n] for the current step
        self.reset = None     # float32[3, n] for the current reset
        self.in_reset = False
        self.fill_idx = 0
        self.used = None      # bool[6, n]: which step slots were consumed


class TapedReference:
    """One reference `UAVEnvironment` instance stepped under the tape of tests/tape.py."""

    def __init__(self, n, grid, tape_seed, env_index=0, sigma=None, **env_kwargs):
        self.uav_env, self.iot = import_reference()
        self.n, self.grid, self.tape_seed, self.env_index = n, grid, tape_seed, env_index
        px, py = T.positions(tape_seed, env_index, n, grid[0], grid[1])
        self.pos_x, self.pos_y = px, py
        kw = dict(max_steps=2100, rssi_threshold=-85.0, sensor_duty_cycle=10.0, max_battery=274.0)
        kw.update(env_kwargs)
        self.env = self.uav_env.UAVEnvironment(
            grid_size=grid, sensor_positions=[(float(x), float(y)) for x, y in zip(px, py)], **kw)
        if sigma is not None:     # sim_to_real_sweep.py:113-117 sets it post-construction
            for s in self.env.sensors:
                s.shadowing_std_db = float(sigma)
        self.ts = _TapeState()
        self.vstep = 0            # vector steps since construction (tape address)
        self.episode = 0
        self._orig_normal = None

    # ---- patches -------------------------------------------------------------------------
    def _fake_normal(self, loc=0.0, scale=1.0, size=None):
        f = sys._getframe(1)
        assert f.f_code.co_name == "calculate_rssi", f.f_code.co_name
        sensor = f.f_locals["self"].sensor_id
        c1 = f.f_back
        name1 = c1.f_code.co_name
        if name1 == "update_spreading_factor":
            name2 = c1.f_back.f_code.co_name
            if name2 == "_execute_collect_action":
                slot = T.SLOT_ZA
            elif name2 == "_get_observation":
                slot = T.SLOT_ZD
            else:
                raise AssertionError(name2)
        elif name1 == "get_success_probability":
            slot = T.SLOT_ZB if c1.f_locals["use_advanced_model"] else T.SLOT_ZC
        elif name1 == "is_in_range":
            slot = T.SLOT_ZE if c1.f_back.f_code.co_name == "_get_observation" else T.SLOT_ZP
        else:
            raise AssertionError(name1)
        ts = self.ts
        if ts.in_reset:
            z = ts.reset[{T.SLOT_ZD: T.RSLOT_ZD, T.SLOT_ZE: T.RSLOT_ZE}[slot], sensor]
        else:
            assert not ts.used[slot, sensor], (slot, sensor)
            ts.used[slot, sensor] = True
            z = ts.step[slot, sensor]
        return loc + scale * float(z)

    class _Lottery:
        def __init__(self, outer):
            self.o = outer

        def random(self):
            f = sys._getframe(1)
            assert f.f_code.co_name == "_execute_collect_action"
            sensor = f.f_locals["sensor"].sensor_id
            ts = self.o.ts
            assert not ts.used[T.SLOT_U, sensor]
            ts.used[T.SLOT_U, sensor] = True
            return float(ts.step[T.SLOT_U, sensor])

    class _Fill:
        def __init__(self, outer):
            self.o = outer

        def uniform(self, low, high):
            ts = self.o.ts
            u = float(ts.reset[T.RSLOT_FILL, ts.fill_idx])
            ts.fill_idx += 1
            return low + (high - low) * u     # numpy Generator.uniform: low + (high-low)*U

    def __enter__(self):
        self._orig_normal = np.random.normal
        np.random.normal = self._fake_normal
        self._orig_random_mod = self.uav_env.random
        self.uav_env.random = self._Lottery(self)
        self.env.np_random = self._Fill(self)
        return self

    def __exit__(self, *exc):
        np.random.normal = self._orig_normal
        self.uav_env.random = self._orig_random_mod
        return False

    # ---- API -----------------------------------------------------------------------------
    def reset(self):
        ts = self.ts
        ts.reset = T.reset_tape(self.tape_seed, self.env_index, self.episode, self.n)
        ts.in_reset, ts.fill_idx = True, 0
        obs, info = self.env.reset(seed=None)
        ts.in_reset = False
        self.episode += 1
        return obs, info

    def step(self, action):
        ts = self.ts
        ts.step = T.step_tape(self.tape_seed, self.env_index, self.vstep, self.n)
        ts.used = np.zeros((T.NUM_STEP_SLOTS, self.n), dtype=bool)
        self.vstep += 1
        return self.env.step(int(action))

    def make_agent(self, kind):
        """The reference's own heuristic agents (agents/dqn/dqn_evaluation_results/greedy_agents.py)."""
        d = os.path.join(REF_SRC, "agents", "dqn", "dqn_evaluation_results")
        if d not in sys.path:
            sys.path.insert(0, d)
        import greedy_agents as GA
        return {"nearest": GA.NearestSensorGreedy, "max_throughput_v2": GA.MaxThroughputGreedyV2}[kind](self.env)

    def policy_step(self, agent, obs):
        """select_action (consumes zP samples of THIS step's tape) followed by step(action)."""
        ts = self.ts
        ts.step = T.step_tape(self.tape_seed, self.env_index, self.vstep, self.n)
        ts.used = np.zeros((T.NUM_STEP_SLOTS, self.n), dtype=bool)
        action = int(agent.select_action(obs))
        self.vstep += 1
        return action, self.env.step(action)

    def state(self):
        """Full per-sensor / per-env state as plain arrays (float64 where the reference holds Python floats)."""
        e = self.env
        S = e.sensors
        f64 = lambda g: np.array([g(s) for s in S], dtype=np.float64)
        return dict(
            buffer=f64(lambda s: s.data_buffer), gen=f64(lambda s: s.total_data_generated),
            tx=f64(lambda s: s.total_data_transmitted), lost=f64(lambda s: s.total_data_lost),
            avg_rssi=f64(lambda s: np.nan if s.avg_rssi is None else s.avg_rssi),
            sf=np.array([s.spreading_factor for s in S], dtype=np.int32),
            visited=np.array([s.sensor_id in e.sensors_visited for s in S], dtype=np.uint8),
            data_collected=np.array([bool(s.data_collected) for s in S], dtype=np.uint8),
            uav_x=np.float32(e.uav.position[0]), uav_y=np.float32(e.uav.position[1]),
            battery=np.float64(e.uav.battery), step=np.int32(e.current_step),
            total_reward=np.float64(e.total_reward), total_collected=np.float64(e.total_data_collected),
            capture_triggers=np.int32(e.capture_effect_triggers), boundary_hits=np.int32(e.boundary_hits),
            edge_steps=np.int32(e.edge_steps), last_bytes=np.float64(e.last_step_bytes_collected),
        )
