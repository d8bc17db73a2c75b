#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched UAV-IoT environment step() on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE pass of the hot path over the whole batch: one launch of the HIP step kernel
(uavenv_step_random: uniform-random policy drawn in-kernel) advancing all
4096 environments x 50 sensors of this GPU by one env-step each -- ageing, move/collect with
Capture-Effect resolution, reward, truncation, auto-reset, float32 observation -- with every
output (obs [E,153], reward, done) written to HBM every step.  State and outputs are resident in
HBM; nothing crosses PCIe inside the timed region.

Multi-GPU (--gpus N > 1, one process per GPU): weak scaling, every rank owns 4096 environments
(global indices rank*4096 ..), writes its observations straight into its slice of a shared replay
ring and publishes the transition block to all ranks with an RCCL all-gather per step, issued on a
side stream so it overlaps the next step (BASELINE config 4).  `value` = env-steps of ALL ranks /
max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
REFERENCE_ENV_STEPS_PER_S_N50 = 764.0     # BASELINE.md section 2 (reference Python step(), N = 50, one core, survey container)


def algorithmic_bytes_per_env_step(n, padded_to=None):
    """B(N) = 68*N + 104 (+ 12*(pad-N) if padded)."""
    b = 68 * n + 104
    if padded_to and padded_to > n:
        b += 12 * (padded_to - n)
    return b


def cpu_baseline(num_sensors, grid, seconds_target=12.0):
    """The CPU oracle (a C restatement of the reference, kind "port") timed on ONE host core over a
    bounded sample of the same workload: the first 256 of the 4096 environments, random policy,
    auto-reset, for as many vector steps as fit the time target."""
    from oracle import oracle as O
    cfg = O.default_config(num_sensors=num_sensors, grid_size=grid, seed=0)
    envs = 256
    t0 = time.perf_counter()
    n, _ = O.run_random_policy(cfg, envs, 40)            # calibrate
    rate = n / (time.perf_counter() - t0)
    steps = max(50, int(seconds_target * rate / envs))
    t0 = time.perf_counter()
    n, _ = O.run_random_policy(cfg, envs, steps)
    dt = time.perf_counter() - t0
    # the same port on P host threads over disjoint environment shards (ctypes releases the GIL), P = the box's
    # CPU share for one GPU (16) or fewer; reported beside the single-core figure, never instead of it
    from concurrent.futures import ThreadPoolExecutor
    P = max(1, min(16, os.cpu_count() or 1))
    psteps = max(50, steps // 3)
    t1 = time.perf_counter()
    with ThreadPoolExecutor(P) as ex:
        res = list(ex.map(lambda k: O.run_random_policy(cfg, envs, psteps, env_index_base=k * envs)[0], range(P)))
    dtp = time.perf_counter() - t1
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port", "cpu_model": model,
            "all_cores": {"value": sum(res) / dtp, "cores": P, "sample": f"{P} threads x {envs} envs x {psteps} vector steps"},
            "sample": f"first {envs} of the 4096 envs x {num_sensors} sensors, {steps} vector steps "
                      f"({n} env-steps, {dt:.1f} s), oracle/uavenv_oracle.c, 1 thread",
            "reference_python_1core_survey": {"n20": 1692, "n50": 764, "unit": "env-steps/s",
                                              "source": "BASELINE.md section 2 (real reference, build container)"}}


def latest_traffic(workload_key):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/*traffic*.json), or None."""
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            d = json.load(open(p))
            if d.get("workload") == workload_key:
                best = dict(d, file=os.path.relpath(p, ROOT))
        except Exception:
            pass
    return best


def build_commit():
    """The commit tools/gpu.sh stamped into the tree (the GPU box has no .git), or None."""
    p = os.path.join(ROOT, ".build_commit")
    return open(p).read().strip() if os.path.exists(p) else None


def latest_sq_counters(envs):
    """Per-wave SQ counters of the step kernel at `envs` environments from the committed summary
    (profiles/*sq_counters*.txt, tools/pmc_sq.sh: separate rocprofv3 --pmc passes), or None."""
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*sq_counters*.txt"))):
        cur = {}
        try:
            for line in open(p):
                f = line.split()
                if len(f) >= 6 and f[0] == "step" and f[1].startswith("envs=") and f[3].startswith("SQ_") and f[4] == "per":
                    if int(f[2]) == envs:
                        cur[f[3]] = float(f[6])
                elif len(f) >= 6 and f[0] == "step" and f[1] == "envs=" + str(envs) and f[2].startswith("SQ_"):
                    cur[f[2]] = float(f[5])
        except Exception:
            cur = {}
        if cur:
            best = dict(cur, file=os.path.relpath(p, ROOT))
    return best


class stdout_to_stderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the contract is ONE JSON line there.
    Everything written to file descriptor 1 inside this block goes to stderr instead."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def issue_split(E, n):
    sq = latest_sq_counters(E) if n == 50 else None
    if not sq or "SQ_WAVE_CYCLES" not in sq:
        return None
    wc = sq["SQ_WAVE_CYCLES"]
    return {"valu_insts_per_wave": sq.get("SQ_INSTS_VALU"), "scalar_insts_per_wave": sq.get("SQ_INSTS_SALU"),
            "wave_cycles": 4 * wc, "frac_issuing": sq.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            "frac_parked_in_waitcnt": sq.get("SQ_WAIT_ANY", 0) / wc, "frac_issue_stalled": sq.get("SQ_WAIT_INST_ANY", 0) / wc,
            "frac_valu_active": sq.get("SQ_ACTIVE_INST_VALU", 0) / wc, "source": sq["file"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--sensors", type=int, default=50)
    ap.add_argument("--grid", type=int, default=500)
    ap.add_argument("--exchange", choices=["auto", "none", "allgather"], default="auto")
    ap.add_argument("--ring", type=int, default=128, help="replay-ring slots used by the bench (two chunks = two HIP graphs / collectives).  "
                    "Measured at 4096 x 50 (2.6 MB per slot): 8.73 / 8.50 / 8.39 / 8.38 us per step for rings of 32 / 50 / 80 / 128 slots, "
                    "8.95 for 250 and 500 -- a ring beyond the 256 MB Infinity Cache makes every output store miss it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from Python instead of chunk by chunk")
    ap.add_argument("--launch", choices=["batch", "graph"], default="graph",
                    help="how a ring chunk of L steps is issued: 'batch' = L single-step launches from one call into the C "
                         "library (uavenv_step_random_n), 'graph' = one HIP-graph replay of the captured L launches")
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of K steps each (median reported); 0 = as many as ~1500 steps need, at most 15")
    ap.add_argument("--fused", type=int, default=16, help="steps per launch of the additional fused-rollout measurement (0 = skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    # Rehearsal switch (never the driver's configuration): UAVENV_BENCH_REHEARSE=gloo runs the N > 1 path with every rank on
    # cuda:0 and the gloo backend -- a one-GPU box can then exercise the sharded launch, the ring exchange and the max-over-
    # ranks timing end to end (RCCL itself needs one GPU per rank).
    rehearse = os.environ.get("UAVENV_BENCH_REHEARSE") == "gloo"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    selftest = os.environ.get("UAVENV_BENCH_SELFTEST_DIST") == "1"     # dev: run the N>1 code path on one rank
    distributed = world > 1 or selftest
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        with stdout_to_stderr():
            if rehearse:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)                  # brings the communicator (and its banner) up now
            torch.cuda.synchronize(dev)

    import uavenv_amd as U
    from uavenv_amd.replay import TransitionRing

    E, n, K, W = args.envs, args.sensors, args.steps, args.warmup
    env = U.BatchedUAVEnv(E, device=local_rank, env_index_base=rank * E, auto_reset=True, seed=0,
                          num_sensors=n, grid_size=(args.grid, args.grid))
    exchange = args.exchange
    if exchange == "auto":
        exchange = "allgather" if distributed else "none"
    if not distributed:
        exchange = "none"
    use_graph = not args.no_graph
    # slots per chunk = steps per HIP graph = steps per collective.  The chunk length is chosen so that the K timed steps
    # are whole graph replays: K itself when it fits half the ring (a short run is ONE replay), else the largest divisor
    # of K in 16..ring/2, else ring/2 with the remainder launched step by step.
    half = max(1, args.ring // 2)
    L = 1
    if use_graph:
        if K <= half:
            L = K
        else:
            divs = [d for d in range(16, half + 1) if K % d == 0]
            L = max(divs) if divs else half
    ring_slots = 2 * L if use_graph else args.ring

    def make_ring(shared):
        r_ = TransitionRing(ring_slots, E, env.obs_dim, dev, world_size=world if shared else 1, rank=rank if shared else 0,
                            chunk_len=L, always_exchange=shared and selftest)
        r_.attach(env)    # the kernel writes obs AND (action, reward, done, terminal row) straight into the ring slot
        return r_

    ring = make_ring(exchange == "allgather")
    env.reset()
    exchange_error = None
    if exchange == "allgather":
        # One probe exchange before anything is timed: if the collective cannot run on this node, say so in the
        # JSON line and measure the shards without it rather than dying without a result.
        try:
            for _ in range(L):
                env.step_random(obs_out=ring.local_obs_slot())
                ring.commit()                     # the L-th commit completes a chunk and issues its all-gather
            ring.drain()
            torch.cuda.synchronize(dev)
        except Exception as ex:          # pragma: no cover - depends on the node
            exchange_error = repr(ex)[:300]
        flag = torch.tensor([1 if exchange_error else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            exchange = "none"
            ring = make_ring(False)

    def one_step():
        env.step_random(obs_out=ring.local_obs_slot())
        ring.commit()     # N > 1: one in-place RCCL all-gather of this rank's transition block, on a side stream

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # A Python -> ctypes -> hipLaunchKernel round trip costs ~12 us, more than the step kernel, and a collective call
    # several times that: the loop is captured, one chunk of the ring (`--ring`/2 steps) per HIP graph, and replayed;
    # ranks that share the ring all-gather a chunk (one in-place RCCL collective of L transition blocks per rank, side
    # stream, overlapping the next chunk's steps) when it is complete.  Steps that do not fill a chunk launch eagerly.
    graphs = None
    chunked = use_graph
    if use_graph:
        while ring.head % L:
            one_step()
        if args.launch == "graph":
            graphs = ring.capture_chunks(lambda slot: env.step_random(obs_out=slot))

    def run_chunk():
        if graphs is not None:
            ring.replay_chunk(graphs)
        else:
            ring.run_chunk_random()

    def run_steps(n, align=False):
        q, r = divmod(n, L) if chunked else (0, n)
        for _ in range(q):
            run_chunk()
        for _ in range(r):
            one_step()
        while align and chunked and ring.head % L:      # untimed: back to a chunk boundary
            one_step()

    run_steps(W, align=True)
    ring.drain()

    def timed_region(with_events=False, steps=None):
        """EXACTLY K steps (or `steps`, for the diagnostic event region) between two barrier + synchronize brackets; returns (wall s, enqueue s, device-event ms or None).
        The regions that produce `value` record NO events: an event pair costs ~10 us of completion latency on an idle
        GPU (tools/bracket_probe.py: 22 us around an empty region with events, 3 us without), which is noise of the order
        of a step in a short region.  The device-event figure comes from one extra region that is not counted."""
        barrier()
        ev0 = ev1 = None
        if with_events:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if with_events:
            ev0.record()                   # torch's current stream IS the stream the kernel is launched on
        run_steps(K if steps is None else steps)
        t_enq = time.perf_counter() - t0
        if with_events:
            ev1.record()
        ring.drain()
        barrier()
        dt_ = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        return dt_, t_enq, (ev0.elapsed_time(ev1) if with_events else None)

    # A region of a few dozen steps lasts a fraction of a millisecond, where one scheduling hiccup of the host moves the
    # figure by 10 %: short regions are REPEATED (each one exactly K steps in its own brackets, re-aligned to a chunk
    # boundary in between, untimed) and the MEDIAN region is reported, with the spread beside it.
    repeats = args.repeats if args.repeats > 0 else max(1, min(15, -(-1500 // K)) | 1)
    regions = []
    for _ in range(repeats):
        regions.append(timed_region())
        run_steps(0, align=True)
    regions.sort(key=lambda x: x[0])
    dt, t_enq, _ = regions[len(regions) // 2]
    ev_ms = timed_region(with_events=True)[2]          # diagnostic only
    run_steps(0, align=True)

    # The step kernel's average launch duration for `roofline`: HIP events on the launch stream around the bench's OWN launch
    # path (the same graph replays into the ring), over a region long enough that the bracket's ~20 us do not show (a region
    # of its own, not one that produces `value`: see timed_region).  rocprofv3's average for the kernel in the same command
    # (profiles/*_kernel_stats.md) is the figure it has to agree with.  `back_to_back_ms` is the same kernel launched
    # back-to-back from a C loop with one fixed observation buffer (no graph, no ring): what tools/exp.sh compares.
    ev_steps = max(K, 1000)
    if chunked:
        ev_steps = -(-ev_steps // L) * L
    kern_ms = timed_region(with_events=True, steps=ev_steps)[2] / ev_steps
    run_steps(0, align=True)
    back_to_back_ms = env.time_steps(min(K, 1000))
    torch.cuda.synchronize(dev)
    launch_source = f"HIP events on the launch stream around {ev_steps} steps of this run's launch path"
    if exchange == "allgather":
        # with the exchange the launch stream also waits for the chunk collectives (a chunk is not overwritten before its
        # previous gather is done): events around the loop then time the exchange, not the kernel
        kern_ms = back_to_back_ms
        launch_source = "HIP events around back-to-back launches into one buffer (the launch path of this run waits for its collectives)"

    # Additional measurement for N > 1 (never `value`): the same K steps WITHOUT the exchange, every rank filling a ring of
    # its own.  The exchange replicates every observation on every GPU (2.6 MB per rank and step) and is bound by the
    # xGMI links, not by the step; this number shows how the sharded path itself scales.
    shard_only = None
    if exchange == "allgather":
        ring.drain()
        ring = make_ring(False)
        graphs = ring.capture_chunks(lambda slot: env.step_random(obs_out=slot)) if (use_graph and args.launch == "graph") else None
        run_steps(W, align=True)
        barrier()
        t1 = time.perf_counter()
        run_steps(K)
        barrier()
        dt1 = time.perf_counter() - t1
        t = torch.tensor([dt1], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        shard_only = {"env_steps_per_s": E * world * K / float(t.item()), "ms_per_step": float(t.item()) / K * 1e3,
                      "note": "same steps, no all-gather (per-rank replay rings)"}

    # Additional measurement (never `value`): the fused rollout entry point, `--fused` steps per launch, writing
    # every step's obs/reward/done block into consecutive ring slots.  Same results bit for bit.
    fused = None
    if args.fused > 0 and exchange == "none":
        F = args.fused
        launches = max(20, min(K, 1600) // F)
        ring.drain()
        env.set_aux_output(None)           # the ring's aux slot holds ONE [E][4] block, a rollout writes F of them
        env.set_terminal_pool(None, None, None)
        slab = torch.empty(F, E, env.obs_dim, dtype=torch.float32, device=dev)
        for _ in range(3):
            env.rollout(F, obs_out=slab)
        torch.cuda.synchronize(dev)
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record()
        for _ in range(launches):
            env.rollout(F, obs_out=slab)
        f1.record()
        torch.cuda.synchronize(dev)
        fms = f0.elapsed_time(f1) / launches
        fused = {"steps_per_launch": F, "launches": launches, "ms_per_launch": fms,
                 "env_steps_per_s": E * F / (fms * 1e-3),
                 "achieved_GBps": algorithmic_bytes_per_env_step(n) * E * F / (fms * 1e-3) / 1e9,
                 "kernel": "uav_rollout_kernel<64, true, true>"}

    total_env_steps = E * K * world
    value = total_env_steps / dt
    B = algorithmic_bytes_per_env_step(n)
    per_launch_bytes = B * E
    achieved = per_launch_bytes / (kern_ms * 1e-3) / 1e9
    workload_key = f"{E}x{n}@{args.grid}"
    tr = latest_traffic(workload_key)
    out = {
        "metric": "env-steps/sec at 4096 envs\u00d750 sensors, 1/2/4/8 MI355X; HBM GB/s vs peak",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
        # BASELINE.md publishes no simulator throughput of the reference's own (section 1: "None"); the only number it holds for
        # THIS metric's unit and workload shape is the reference's Python step() measured in the survey container (section 2:
        # 764 env-steps/s at N = 50, one core).  vs_baseline is the ratio to THAT figure, per GPU-count as measured.
        "vs_baseline": value / REFERENCE_ENV_STEPS_PER_S_N50 if n == 50 else None,
        "vs_baseline_basis": "BASELINE.md section 2: reference uav_env.step(), N=50, 1 CPU core, 764 env-steps/s (measured in the survey container; the reference publishes no throughput)",
        "timed_regions": {"repeats": repeats, "reported": "median", "steps_each": K,
                          "ms_per_step_min": regions[0][0] / K * 1e3, "ms_per_step_max": regions[-1][0] / K * 1e3},
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{E} envs/GPU x {n} sensors, {args.grid}x{args.grid} grid, BASE_ENV_CONFIG, "
                               f"uniform-random policy (in-kernel Philox), auto-reset, obs+reward+done written every step",
                   "envs_per_gpu": E, "sensors": n, "grid": args.grid, "obs_dim": env.obs_dim,
                   "exchange": (f"one in-place rccl all_gather per {L} steps of every rank's {L} transition blocks into a shared replay ring"
                                if exchange == "allgather" else "none (observations written in place into the replay ring)"),
                   "parallelism": f"env-shard x{world}" + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearse else ""),
                   "launch": (f"one step per kernel launch; launches replayed as HIP graphs of {L} steps (one ring chunk)" if graphs is not None
                              else f"one step per kernel launch; the {L} launches of a ring chunk issued by one call into the C library "
                                   f"(uavenv_step_random_n)" if chunked else "one step per kernel launch, launched from Python"),
                   "host_enqueue_us_per_step": t_enq / K * 1e6,
                   "arithmetic": "float64 state and rewards, float32 distance chain and observations (the reference's own mix)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": (tr or {}).get("hbm_bytes_per_launch"),
                     # the PMC passes are separate profiler runs (tools/profile_bench.sh): say which code they measured
                     "traffic_source": None if tr is None else {"file": tr.get("file"), "measured_on_commit": tr.get("commit"),
                                                                "this_run_commit": build_commit()},
                     "achieved_from_traffic": None if not (tr or {}).get("hbm_bytes_per_launch") else
                     tr["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9,
                     # measured HBM bytes over algorithmic bytes: > 1 = traffic the byte model does not count (float64 instead of
                     # float32 state rows; 64 lanes fetched for 50 sensors), not re-reads -- DESIGN.md section 4
                     "traffic_over_algorithmic": None if not (tr or {}).get("hbm_bytes_per_launch") else
                     tr["hbm_bytes_per_launch"] / per_launch_bytes,
                     "kernel": "uav_step_kernel<64, true, 16, true> (lane group 64, lean, 16-wave workgroups, default-config literals)",
                     "algorithmic_bytes_per_launch": per_launch_bytes,
                     "algorithmic_bytes_per_env_step": B, "avg_launch_ms": kern_ms,
                     "avg_launch_source": launch_source,
                     "timed_region_event_ms_per_step": ev_ms / K, "back_to_back_ms": back_to_back_ms,
                     "note": "path is VALU/latency bound (Philox + float64 physics per sensor), not HBM bound: "
                             "see DESIGN.md"},
        # what the launch is actually bound by, for context (not a roofline the contract asks for): the measured split of a wave's
        # life into issuing / parked in s_waitcnt / issue stalls (SQ counters, quad-cycles; separate profiler passes)
        "issue_split": issue_split(E, n),
    }
    if fused:
        out["fused_rollout"] = fused
    if shard_only:
        out["shard_only"] = shard_only
    if distributed and exchange_error:
        out["config"]["exchange_error"] = exchange_error
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n, (args.grid, args.grid))
        out["cpu_baseline"]["cores_available"] = os.cpu_count()
        out["cpu_baseline"]["host"] = {"cpu_count": os.cpu_count()}
    elif rank == 0:
        out["cpu_baseline"] = None
    env.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
