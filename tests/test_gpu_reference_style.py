"""GPU: the reference's own unit tests for this path, re-expressed against the drop-in
(`UAVEnvironment` over the HIP kernel).  Each class mirrors a class of
/root/reference/tests/test_uav.py or tests/test_iot_sensors.py that still passes against the current
reference code: same scenario, same expectation, driven through env.step()
instead of the UAV / IoTSensor objects (action ids: 0 UP, 1 DOWN, 2 LEFT, 3 RIGHT, 4 COLLECT).
Power defaults are the reference's current ones (500 W move, 700 W hover, uav.py:93-94); the tests
that pass an explicit power_move=600 do the same here through the config."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

UP, DOWN, LEFT, RIGHT, COLLECT = 0, 1, 2, 3, 4


def make(start=(5.0, 5.0), grid=(10, 10), **kw):
    import uavenv_amd as U
    env = U.UAVEnvironment(grid_size=grid, num_sensors=3, uav_start_position=start, **kw)
    env.reset(seed=1)
    return env


class TestUAVMovement:           # test_uav.py:73-127
    def test_move_up(self):
        env = make(); env.step(UP)
        assert env.uav.position[0] == 5.0 and env.uav.position[1] == 6.0

    def test_move_down(self):
        env = make(); env.step(DOWN)
        assert env.uav.position[0] == 5.0 and env.uav.position[1] == 4.0

    def test_move_left(self):
        env = make(); env.step(LEFT)
        assert env.uav.position[0] == 4.0 and env.uav.position[1] == 5.0

    def test_move_right(self):
        env = make(); env.step(RIGHT)
        assert env.uav.position[0] == 6.0 and env.uav.position[1] == 5.0

    def test_sequential_movements(self):
        env = make()
        for a in (UP, RIGHT, RIGHT, DOWN):
            env.step(a)
        assert env.uav.position[0] == 7.0 and env.uav.position[1] == 5.0

    def test_invalid_direction_raises_error(self):
        env = make()
        with pytest.raises(ValueError, match="Invalid action"):
            env.step(17)

    def test_position_dtype(self):   # test_uav.py:58-63
        env = make((5, 5))
        assert isinstance(env.uav.position, np.ndarray) and env.uav.position.dtype == np.float32


class TestBoundaryDetection:     # test_uav.py:137-185
    @pytest.mark.parametrize("start,action,axis,value", [
        ((5.0, 9.0), UP, 1, 9.0), ((5.0, 0.0), DOWN, 1, 0.0), ((0.0, 5.0), LEFT, 0, 0.0), ((9.0, 5.0), RIGHT, 0, 9.0)])
    def test_boundary_collision(self, start, action, axis, value):
        env = make(start)
        _, reward, _, _, info = env.step(action)
        assert env.uav.position[axis] == value and np.array_equal(env.uav.position, np.array(start, np.float32))
        assert info["boundary_hits"] == 1 and reward < -50.0          # penalty_boundary -50 (reward_function.py:16)

    def test_corner_boundary(self):
        env = make((0.0, 0.0))
        env.step(LEFT); env.step(DOWN)
        assert np.array_equal(env.uav.position, np.array([0.0, 0.0])) and env.boundary_hits == 2


class TestBatteryConsumption:    # test_uav.py:188-272
    def test_battery_drains_on_move(self):
        env = make(); b0 = env.uav.battery; env.step(UP)
        assert env.uav.battery < b0

    def test_move_energy_calculation(self):
        env = make(power_move=600.0); b0 = env.uav.battery; env.step(UP)
        assert abs((b0 - env.uav.battery) - (600.0 * 1.0) / 3600) < 1e-12

    def test_hover_energy_calculation(self):
        env = make(power_hover=400.0, collection_duration=5.0); b0 = env.uav.battery; env.step(COLLECT)
        assert abs((b0 - env.uav.battery) - (400.0 * 5.0) / 3600) < 1e-12

    def test_collision_partial_energy(self):
        env = make((9.0, 5.0), power_move=600.0); b0 = env.uav.battery; env.step(RIGHT)
        assert abs((b0 - env.uav.battery) - (600.0 * 0.5 * 1.0) / 3600) < 1e-12

    def test_collision_uses_less_energy_than_move(self):
        e1, e2 = make((5.0, 5.0)), make((9.0, 5.0))
        e1.step(UP); e2.step(RIGHT)
        em, ec = 274.0 - e1.uav.battery, 274.0 - e2.uav.battery
        assert ec < em and abs(ec - em * 0.5) < 1e-12

    def test_multiple_moves_accumulate_energy(self):
        env = make(grid=(100, 100))
        for _ in range(5):
            env.step(UP)
        assert abs((274.0 - env.uav.battery) - 5 * 500.0 / 3600) < 1e-9

    def test_is_alive_threshold(self):       # uav.py:224: alive iff battery > 2 % of capacity
        env = make(max_battery=1.0)
        steps = 0
        trunc = False
        while not trunc:
            _, _, term, trunc, _ = env.step(COLLECT); steps += 1
            assert term is False
        assert not env.uav.is_alive() and env.uav.battery <= 0.02
        assert steps == int(np.ceil((1.0 - 0.02) / (700.0 / 3600)))


class TestEpisodeReset:          # test_uav.py:352-409
    def test_reset_restores_position_battery_counters(self):
        env = make((3.0, 7.0))
        for a in (UP, RIGHT, COLLECT, LEFT):
            env.step(a)
        env.reset()
        assert np.array_equal(env.uav.position, np.array([3.0, 7.0], np.float32)) and env.uav.battery == 274.0
        assert env.current_step == 0 and env.total_reward == 0.0 and env.sensors_visited == set()
        assert np.array_equal(env.uav.start_position, np.array([3.0, 7.0], np.float32))


class TestDataGeneration:        # test_iot_sensors.py:77-127 (the five tests that still pass)
    def test_buffer_accumulates_and_counts_generated(self):
        env = make(data_generation_rate=10.0)
        b0 = [s.data_buffer for s in env.sensors]; g0 = [s.total_data_generated for s in env.sensors]
        env.step(UP)
        for s, b, g in zip(env.sensors, b0, g0):
            assert abs(s.data_buffer - (b + 10.0)) < 1e-9 and abs(s.total_data_generated - (g + 10.0)) < 1e-9

    def test_custom_time_step(self):         # collection_duration is the step duration of a collect step (uav_env.py:450)
        env = make(start=(9.0, 9.0), grid=(2000, 2000), data_generation_rate=10.0, collection_duration=3.0,
                   sensor_positions=[(1500.0, 1500.0)] * 3)
        b0 = env.sensors[0].data_buffer
        env.step(COLLECT)                   # far out of range: nothing drained
        assert abs(env.sensors[0].data_buffer - (b0 + 30.0)) < 1e-9

    def test_buffer_overflow_clamps_and_counts_loss(self):
        env = make(data_generation_rate=500.0, max_buffer_size=1000.0, sensor_positions=[(9.0, 9.0)] * 3, start=(0.0, 0.0),
                   grid=(3000, 3000))
        for _ in range(3):
            env.step(UP)
        for s in env.sensors:
            assert s.data_buffer == 1000.0 and s.total_data_lost > 0
            assert abs(s.total_data_generated - (s.data_buffer + s.total_data_lost + s.total_data_transmitted)) < 1e-9

    def test_data_rate_lookup(self):         # test_iot_sensors.py:67-74, :500-519
        from uavenv_amd.gym_env import SF_DATA_RATES
        assert SF_DATA_RATES[7] == 5470 / 8 and SF_DATA_RATES[12] == 250 / 8 and SF_DATA_RATES[7] > SF_DATA_RATES[12]
