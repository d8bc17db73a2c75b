"""GPU: the C ABI used from a plain-C program (no Python/torch in the process), HIP-graph capture of the step,
and handle lifetime."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "-reinforcement-learning-for-dynamic-uav-energy-efficient-path-planning-in-iot-sensor-networks._amd")


def test_plain_c_program_through_the_abi(tmp_path):
    import torch
    import uavenv_amd as U
    exe = str(tmp_path / "capi_demo")
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-I", os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "capi", "capi_demo.c"), "-L", PKG, "-luavenv_hip",
                           "-Wl,-rpath," + PKG])
    out = subprocess.check_output([exe, "8", "25"], text=True)
    m = re.search(r"obs_dim=(\d+) reward_sum=([-\d.]+) obs_sum=([-\d.]+) dones=(\d+) invalid_rc=(-?\d+) config=(-?\d+)/(\d)/(-?\d+)/(-?\d+)", out)
    assert m, out
    assert [int(m.group(i)) for i in (6, 7, 8, 9)] == [0, 1, 0, -1]      # uavenv_get_config / set_config: ok, values read back, ok, UAVENV_E_INVALID
    E, steps = 8, 25
    env = U.BatchedUAVEnv(E, num_sensors=20, grid_size=(200, 200), max_steps=10, duty_cycle=60.0, seed=99)
    env.reset()
    total, dones = 0.0, 0
    for s in range(steps):
        a = torch.tensor([(s * 7 + k * 3) % 5 for k in range(E)], dtype=torch.int32, device=env.device)
        o, r, d = env.step(a)
        total += float(r.sum().item()); dones += int(d.sum().item())
    assert int(m.group(1)) == env.obs_dim and int(m.group(4)) == dones and dones >= 2 * E - 8
    assert abs(float(m.group(2)) - total) <= 1e-6 * max(1.0, abs(total))
    assert abs(float(m.group(3)) - float(o.double().sum().item())) <= 1e-3
    assert int(m.group(5)) == -3                                  # UAVENV_E_ACTION
    env.close()


def test_step_is_hip_graph_capturable():
    """No allocation / synchronisation inside the launch path: the step can be captured into a HIP graph and
    replayed (cdna_hip_programming.md guideline 9); replays advance the environment exactly like eager steps."""
    import torch
    import uavenv_amd as U
    kw = dict(num_sensors=50, seed=17, max_steps=40)
    a, b = U.BatchedUAVEnv(256, **kw), U.BatchedUAVEnv(256, **kw)
    a.reset(); b.reset()
    for _ in range(3):
        a.step_random(); b.step_random()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            a.step_random()
    for _ in range(4):                 # the capture itself did not execute: the eager twin does its 4 steps
        b.step_random()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward)
    for rep in range(5):
        g.replay()
        for _ in range(4):
            b.step_random()
    torch.cuda.synchronize()
    assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done)
    sa, sb = a.state_dict(), b.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    # an ODD number of random-policy steps per graph: the replays read action words that are two launches old (the
    # launch parity is baked into the graph); their (episode, step) tags no longer match, so the kernel must draw the
    # actions itself -- same results
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        for _ in range(3):
            a.step_random()
    for rep in range(4):
        g3.replay()
        for _ in range(3):
            b.step_random()
        torch.cuda.synchronize()
        assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.actions_taken, b.actions_taken)
    a.close(); b.close()


def test_many_handles_create_destroy():
    import torch
    import uavenv_amd as U
    free0 = torch.cuda.mem_get_info()[0]
    envs = [U.BatchedUAVEnv(512, num_sensors=n, seed=i) for i, n in enumerate((10, 20, 50, 64, 1))]
    outs = []
    for e in envs:
        e.reset(); e.step_random(); outs.append(e.reward.sum().item())
    assert all(np.isfinite(outs))
    for e in envs:
        e.close()
    for _ in range(30):
        e = U.BatchedUAVEnv(2048, num_sensors=50); e.reset(); e.step_random(); e.close()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)      # no leak beyond allocator slack


_BALANCE_CHILD = r'''
import sys, hashlib
sys.path.insert(0, %r)
import torch, uavenv_amd as U
h = hashlib.sha256()
for E, n in ((37, 50), (256, 50), (100, 20), (64, 10), (4096 + 37, 50), (8192 + 5, 20), (16384 + 3, 10)):   # the last three: 16-wave workgroups
    env = U.BatchedUAVEnv(E, num_sensors=n, seed=5, max_steps=30)
    env.reset()
    g = torch.Generator(device="cpu"); g.manual_seed(E)
    for s in range(70 if E < 1000 else 24):
        if s %% 3 == 2:
            env.step_random()
        else:
            a = torch.randint(0, 5, (E,), generator=g, dtype=torch.int32).to(env.device)
            env.step(a)
        h.update(env.obs.cpu().numpy().tobytes()); h.update(env.reward.cpu().numpy().tobytes()); h.update(env.done.cpu().numpy().tobytes())
    ro = env.rollout(6)                                  # the fused kernel too
    for k in ("obs", "reward", "done", "actions"):
        h.update(ro[k].cpu().numpy().tobytes())
    sd = env.state_dict()
    for k in sorted(sd):
        h.update(sd[k].cpu().numpy().tobytes())
    env.close()
print("DIGEST", h.hexdigest())
'''


def test_simd_load_balancing_is_a_pure_scheduling_choice():
    """The step kernel hands the environments of a workgroup to its wavefronts collect-actions-first (StepArgs::balance).
    Which wavefront steps which environment must not change a single bit: same digest with the home mapping."""
    import sys
    outs = []
    for flag in ("0", "1"):
        out = subprocess.check_output([sys.executable, "-c", _BALANCE_CHILD % ROOT], text=True,
                                      env=dict(os.environ, UAVENV_NO_BALANCE=flag))
        outs.append([l for l in out.splitlines() if l.startswith("DIGEST")][0])
    assert outs[0] == outs[1]


def test_ring_chunk_from_one_c_call_equals_the_eager_loop():
    """uavenv_step_random_n / TransitionRing.run_chunk_random: the L launches of a chunk issued by one call into the library
    leave the same ring contents, terminal rows and environment state as L step_random() + commit() calls."""
    import torch
    import uavenv_amd as U
    from uavenv_amd.replay import TransitionRing
    kw = dict(num_sensors=50, seed=9, max_steps=13)
    E, L = 200, 8
    envs, rings = [], []
    for _ in range(2):
        e = U.BatchedUAVEnv(E, **kw)
        r = TransitionRing(3 * L, E, e.obs_dim, e.device, chunk_len=L)
        r.attach(e); e.reset()
        envs.append(e); rings.append(r)
    (ea, eb), (ra, rb) = envs, rings
    all_e = torch.arange(E, device=ea.device)
    n_term = 0
    for rev in range(5):
        for _ in range(L):
            ea.step_random(obs_out=ra.local_obs_slot()); ra.commit()
        rb.run_chunk_random()
        torch.cuda.synchronize()
        assert ra.head == rb.head and ra.size == rb.size
        c0 = ((rb.head - L) % rb.capacity)
        for slot in range(c0, c0 + L):
            aa, ab = ra.aux_at(slot, 0, all_e), rb.aux_at(slot, 0, all_e)
            assert torch.equal(ra.obs_at(slot, 0, all_e), rb.obs_at(slot, 0, all_e)) and torch.equal(aa[:, :3], ab[:, :3])
            ta, tb = ra.tickets_at(slot, 0, all_e).long(), rb.tickets_at(slot, 0, all_e).long()
            assert torch.equal(ta >= 0, tb >= 0)
            m = ta >= 0
            if bool(m.any()):
                assert torch.equal(ra.terminal_at(slot, 0, ta[m]), rb.terminal_at(slot, 0, tb[m])); n_term += int(m.sum())
        assert torch.equal(ea.reward32, eb.reward32) and torch.equal(ea.done, eb.done)
    assert n_term >= 2 * E
    sa, sb = ea.state_dict(), eb.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    ea.close(); eb.close()


def test_literal_constants_and_write_through_stores_change_no_bit():
    """Round-2 code-generation choices: the default configuration's constants as instruction literals (kDefC, chosen when the
    handle's constants block is bit-identical to the defaults) and `sc1` write-through stores (batches of >= 4096 wavefronts).
    Same digest -- observations, rewards, done flags, fused rollouts, final state -- with either switched off."""
    import sys
    digests = {}
    for lit in ("0", "1"):
        for wt in ("0", "1"):
            out = subprocess.check_output([sys.executable, "-c", _BALANCE_CHILD % ROOT], text=True,
                                          env=dict(os.environ, UAVENV_NO_LITERALS=lit, UAVENV_WRITE_THROUGH=wt))
            digests[(lit, wt)] = [l for l in out.splitlines() if l.startswith("DIGEST")][0]
    assert len(set(digests.values())) == 1, digests


def test_action_words_never_outlive_the_seed_or_the_state():
    """The random-policy step leaves the next launch the action it will draw (a word per environment, checked against
    the record's episode/step before use).  Re-seeding, restoring state and fused rollouts must not let a stale word
    decide an action: compare with an environment that has never seen those words."""
    import torch
    import uavenv_amd as U
    kw = dict(num_sensors=50, max_steps=25)
    a = U.BatchedUAVEnv(96, seed=1, **kw)
    a.reset()
    for _ in range(7):
        a.step_random()
    a.seed(7)                                            # words drawn with seed 1 are now wrong
    b = U.BatchedUAVEnv(96, seed=7, **kw)
    b.reset()
    b.load_state_dict(a.state_dict())                    # same records / sensors, no words
    for _ in range(30):                                  # crosses auto-resets (max_steps 25)
        a.step_random(); b.step_random()
        assert torch.equal(a.actions_taken, b.actions_taken) and torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward)
    # state restored to an EARLIER point: the words of the later point carry other (episode, step) tags or other actions
    snap = a.state_dict()
    for _ in range(3):
        a.step_random()
    a.load_state_dict(snap); b.load_state_dict(snap)
    a.rollout(4); b.rollout(4)                           # the fused kernel neither reads nor writes words
    for _ in range(5):
        a.step_random(); b.step_random()
        assert torch.equal(a.actions_taken, b.actions_taken) and torch.equal(a.obs, b.obs)
    a.close(); b.close()


def test_ring_chunk_graphs_equal_the_eager_loop():
    """TransitionRing.capture_chunks: one HIP graph per ring chunk (the bench's launch path) produces the same ring
    contents, terminal rows and environment state as stepping from Python."""
    import torch
    import uavenv_amd as U
    from uavenv_amd.replay import TransitionRing
    kw = dict(num_sensors=20, seed=3, max_steps=11)
    envs, rings = [], []
    for chunk in (1, 4):
        e = U.BatchedUAVEnv(300, **kw)
        r = TransitionRing(8, 300, e.obs_dim, e.device, chunk_len=chunk)
        r.attach(e); e.reset()
        envs.append(e); rings.append(r)
    (ea, eb), (ra, rb) = envs, rings
    for _ in range(8):                                   # both at a chunk boundary
        ea.step_random(obs_out=ra.local_obs_slot()); ra.commit()
        eb.step_random(obs_out=rb.local_obs_slot()); rb.commit()
    graphs = rb.capture_chunks(lambda slot: eb.step_random(obs_out=slot))
    all_e = torch.arange(300, device=ea.device)
    n_term = 0
    for rev in range(3):
        for _ in range(8):
            ea.step_random(obs_out=ra.local_obs_slot()); ra.commit()
        rb.replay_chunk(graphs); rb.replay_chunk(graphs)
        torch.cuda.synchronize()
        # a ring recycles the chunk at its head as a whole: 8 - 1 sampleable slots with chunks of 1, 8 - 4 with chunks of 4
        assert ra.head == rb.head == 0 and ra.size == 7 and rb.size == 4
        # terminal rows are handed out by an atomic counter (order of arrival): compare what the rows hold
        for slot in range(8):
            aa, ab = ra.aux_at(slot, 0, all_e), rb.aux_at(slot, 0, all_e)
            assert torch.equal(ra.obs_at(slot, 0, all_e), rb.obs_at(slot, 0, all_e)) and torch.equal(aa[:, :3], ab[:, :3])
            ta, tb = ra.tickets_at(slot, 0, all_e).long(), rb.tickets_at(slot, 0, all_e).long()
            assert torch.equal(ta >= 0, tb >= 0) and torch.equal(ta >= 0, aa[:, 2] > 0.5)
            m = ta >= 0
            if bool(m.any()):
                assert torch.equal(ra.terminal_at(slot, 0, ta[m]), rb.terminal_at(slot, 0, tb[m]))
                n_term += int(m.sum())
    assert n_term > 300
    sa, sb = ea.state_dict(), eb.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    batch = rb.sample(256)
    assert batch["obs"].shape == (256, eb.obs_dim) and bool(batch["valid"].all())
    batch = rb.sample_stacked(64, 4)
    assert batch["obs"].shape == (64, 4 * eb.obs_dim)
    ea.close(); eb.close()
