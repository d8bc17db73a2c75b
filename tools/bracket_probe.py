"""GPU box: what the timing bracket of bench.py costs around a SHORT region (20 steps = one graph replay)."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import uavenv_amd as U
from uavenv_amd.replay import TransitionRing
E, L = 4096, 20
env = U.BatchedUAVEnv(E, num_sensors=50, seed=0)
ring = TransitionRing(2 * L, E, env.obs_dim, env.device, chunk_len=L); ring.attach(env)
env.reset()
graphs = ring.capture_chunks(lambda slot: env.step_random(obs_out=slot))
for _ in range(20): ring.replay_chunk(graphs)
torch.cuda.synchronize()
kern = env.time_steps(1000) * 1e3
def region(body):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record(); body(); t1 = time.perf_counter(); e1.record()
    while not e1.query(): pass
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    return (t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t0) * 1e6, e0.elapsed_time(e1) * 1e3
def med(f, n=31):
    rs = sorted((region(f) for _ in range(n)), key=lambda r: r[0]); return rs[n // 2]
hip = C.CDLL("libamdhip64.so")
def raw_replay():
    c = ring.head // ring.L
    g = graphs[c]
    hip.hipGraphLaunch(C.c_void_p(g.raw_cuda_graph_exec()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    ring._advance(ring.L, zero_counts=False); ring._point_env()
print(f"kernel-only {kern:.2f} us/launch -> 20 steps = {20 * kern:.1f} us")
for name, f in (("empty region", lambda: None), ("one 20-step graph (ring.replay_chunk)", lambda: ring.replay_chunk(graphs)),
                ("one 20-step graph (hipGraphLaunch via ctypes)", raw_replay),
                ("two 20-step graphs", lambda: (ring.replay_chunk(graphs), ring.replay_chunk(graphs)))):
    try:
        w, enq, det, ev = med(f)
        print(f"{name:48s} wall {w:7.1f} us   enqueue returns at {enq:6.1f}   completion seen at {det:7.1f}   device events {ev:7.1f} us")
    except Exception as ex:
        print(name, "failed:", repr(ex)[:200])
def plain(f, how):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); f()
    if how == "device": torch.cuda.synchronize()
    elif how == "stream": torch.cuda.current_stream().synchronize()
    return (time.perf_counter() - t0) * 1e6
for how in ("device", "stream"):
    for name, f in (("empty", lambda: None), ("one 20-step graph", lambda: ring.replay_chunk(graphs))):
        rs = sorted(plain(f, how) for _ in range(31))
        print(f"no events, {how}-synchronize: {name:20s} wall median {rs[15]:7.1f} us  min {rs[0]:7.1f}")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for g in graphs:
    rc = hip.hipGraphUpload(C.c_void_p(g.raw_cuda_graph_exec()), st)
torch.cuda.synchronize()
print("hipGraphUpload rc", rc)
for name, f in (("empty", lambda: None), ("one 20-step graph after hipGraphUpload", lambda: ring.replay_chunk(graphs)), ("raw hipGraphLaunch after upload", raw_replay)):
    rs = sorted(plain(f, "device") for _ in range(31))
    print(f"no events, device-synchronize: {name:40s} wall median {rs[15]:7.1f} us  min {rs[0]:7.1f}")
# eager launches for comparison (no graph)
def eager20():
    for _ in range(20):
        env.step_random(obs_out=ring.local_obs_slot()); ring.commit()
rs = sorted(plain(eager20, "device") for _ in range(15))
print(f"no events, 20 eager step_random + commit            wall median {rs[7]:7.1f} us  min {rs[0]:7.1f}")
