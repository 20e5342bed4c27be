#!/usr/bin/env python3
"""Benchmark of the PSO latent-discovery hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--particles P]

One step = one swarm iteration over P particles per GPU (default 512, the
"DCGAN-MNIST 28x28, 512 particles, 1xMI355X, fp32" configuration): generator forward,
per-image rescale, ResNet50 assessor forward, fitness, personal/global best, move, and
the mean-pairwise-distance diagnostic and history appends the reference's loop also does
every iteration (src/pso/util_pso.py:174-189).  Weights are synthetic (random-init),
latents start N(0,1), r1/r2 come from the device RNG.  Swarm state, weights and
workspaces are resident in HBM before the timed region; nothing crosses PCIe inside it.

With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank holds P
particles of one N*P-particle swarm (weak scaling) and the ranks exchange one
{value, index, position} record per iteration over RCCL.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# MI355X dense matrix-instruction peaks by operand type: fp32-input MFMA (= the fp32 vector peak), and
# the 16-bit forms (about 2.5 PFLOP/s dense; the headline figures twice that include 2:1 sparsity)
PEAK_MFMA_TFLOPS = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0}
# PMC summaries (tools/refresh_profiles.sh) of the configurations that have one: (dtype, z_dim, side, particles) -> file
TRAFFIC_FILES = {("f32", 10, 28, 512): "r3_hbm_traffic.json", ("bf16", 100, 28, 512): "r3_bf16_hbm_traffic.json"}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(z_dim, n_class, class_index, seconds_budget=20.0):
    """The oracle's restatement of the reference loop, timed on this box's host cores on bounded
    samples of the benchmark's workload (64 particles, as many iterations as fit each mode's share
    of the budget, at least 1; pairwise-distance diagnostic on, no plotting):

    * faithful: one particle per forward pass, the way the reference's loop calls its networks
      (src/pso/util_pso.py:117-125) -- `value` is this mode on all the threads used;
    * batched: one torch-CPU batch per iteration (the same arithmetic, not how the reference runs);
    each on `cores` threads (this job's CPU share: 16 per GPU on the GPU box, or every CPU of a
    smaller host) and on 1 thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pso_oracle as oracle
    synth = importlib.import_module("gan-discovery-pso_amd.synth")
    host_cpus = os.cpu_count() or 1
    threads = min(16, host_cpus)
    gsd = synth.to_torch(synth.generator_state(z_dim, 1, 64, seed=11))
    csd = synth.to_torch(synth.resnet_state("ResNet50", 1, n_class, seed=12))
    fit = oracle.Fitness(gsd, csd, class_index)
    n = 64

    def timed(batch, nthreads, budget, max_iters):
        torch.set_num_threads(nthreads)
        rs = np.random.RandomState(42)
        pos, vel = oracle.init_swarm(n, z_dim, rs)
        fit(pos[:2])  # first-touch of the torch CPU kernels is not what is being measured
        t0 = time.perf_counter()
        oracle.run_swarm(fit, pos, vel, 1, 0.73, 1.496, 1.496, early_stopping=False, rng=rs, batch=batch, record=False)
        one = time.perf_counter() - t0
        iters = int(max(1, min(max_iters, (budget - one) // max(one, 1e-3))))
        t0 = time.perf_counter()
        oracle.run_swarm(fit, pos, vel, iters, 0.73, 1.496, 1.496, early_stopping=False, rng=rs, batch=batch, record=False)
        dt = time.perf_counter() - t0
        return round(n * iters / dt, 2), iters

    share = seconds_budget / 4
    f_all, it_fa = timed(1, threads, 2 * share, 8)
    b_all, it_ba = timed(n, threads, share / 2, 8)
    f_one, it_f1 = timed(1, 1, share, 2)
    b_one, it_b1 = timed(n, 1, share / 2, 2)
    torch.set_num_threads(threads)
    return {"value": f_all, "unit": "particle-evals/s", "cores": threads, "kind": "port",
            "sample": f"{n} particles x {it_fa} iterations, one particle per forward pass (reference-style loop), torch CPU fp32",
            "host_cpus": host_cpus, "cpu_model": _cpu_model(),
            "modes": {"faithful": {f"{threads}_threads": f_all, "1_thread": f_one, "iterations": [it_fa, it_f1]},
                      "batched": {f"{threads}_threads": b_all, "1_thread": b_one, "iterations": [it_ba, it_b1]}}}


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: run the same command line under
    torch.distributed.run (one child process per GPU, rendezvous on 127.0.0.1 at a free port) and
    print the one JSON line rank 0 writes.  Returns the exit code of the launch."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def reference_run_shape(eng, ev, spec, z_dim, dev):
    """The shape the reference really runs (configs/dcgan_mnist.yaml:146-156, src/training/pso_discovery.py:174-218),
    beside the steady-state headline: (a) one swarm of the benchmark's size for T = 50 iterations started on a
    device that has idled (the clock ramp is inside the run); (b) its 8 classes x 32 particles x 50 iterations,
    every iteration of the 8 swarms as one batch, from an idle device too."""
    import time as _time
    p, t = 512, 50
    rs = np.random.RandomState(43)
    r = rs.standard_normal((p, 2, z_dim))
    sw = eng.HipSwarm(ev, p, z_dim, t, history=True, diagnostics=True)
    sw.load(r[:, 0], (r[:, 1] - 0.5) / 10, 0.73)
    torch.cuda.synchronize(dev)
    _time.sleep(1.5)  # long enough for the device to fall back to its idle clock
    t0 = _time.perf_counter()
    sw.iterate(1, t, spec, 1.496, 1.496, False, False, 1e-5, True, seed=42)
    torch.cuda.synchronize(dev)
    cold = _time.perf_counter() - t0
    sizes, iters = [32] * 8, [t] * 8
    batch = eng.HipSwarmBatch(ev, sizes, z_dim, iters, history=True, diagnostics=True)
    specs = [eng.FitnessSpec(class_index=c) for c in range(8)]
    params = [dict(w_cogn=1.496, w_soci=1.496, schedule_inertia=False, early_stopping=False, tolerance=1e-5, diagnostics=True, seed=7 + c)
              for c in range(8)]

    def load():
        for c, s in enumerate(batch.swarms):
            q = np.random.RandomState(100 + c).standard_normal((32, 2, z_dim))
            s.load(q[:, 0], (q[:, 1] - 0.5) / 10, 0.73)

    load()
    batch.iterate(1, 2, specs, params)  # plans and buffers of the 256-image batch
    torch.cuda.synchronize(dev)
    load()
    _time.sleep(1.5)
    t0 = _time.perf_counter()
    batch.iterate(1, t, specs, params)
    torch.cuda.synchronize(dev)
    multi = _time.perf_counter() - t0
    return {"t50_from_idle_ms_per_iteration": round(1e3 * cold / t, 4), "t50_from_idle_particle_evals_per_s": round(p * t / cold, 1),
            "multi_swarm_8x32_t50_ms_per_iteration": round(1e3 * multi / t, 4),
            "multi_swarm_8x32_t50_particle_evals_per_s": round(256 * t / multi, 1),
            "how": "wall clock around 50 iterations enqueued at once on a device that has idled for 1.5 s (host synchronised either side); "
                   "not the headline: `value` is the steady state"}


N_PHASE_PASSES, N_PHASE_TIMED = 48, 18  # generator / assessor passes ahead of the warm-up steps, and how many of them are timed


def rank_spread(dist, dt, device=None):
    """(max, min) of a per-rank duration over the ranks."""
    hi = torch.tensor([dt], dtype=torch.float64, device=device)
    lo = hi.clone()
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return float(hi.item()), float(lo.item())


def result_line(args, world, *, value, dt, dt_min, k, w, n_global, p, z_dim, side, flop_per_eval, t_enqueued, roofline, phases, ranks_seen,
                extra=None):
    """The one JSON object rank 0 prints (the dry run fills it with what it has, so that both carry the same keys)."""
    ms = None if dt is None else round(1e3 * dt / k, 4)
    out = {
        "metric": "particle-evaluations/sec (swarm x gen-fwd) DCGAN-MNIST", "value": None if value is None else round(value, 1),
        "unit": "particle-evals/s", "n_gpus": world, "steps": k, "warmup": w, "ms_per_step": ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"DCGAN-MNIST {side}x{side}, {p} particles per GPU, ResNet50 assessor (n_class=8), z_dim={z_dim}, "
                               f"{n_global} particles total", "particles_per_gpu": p, "z_dim": z_dim,
                   "flop_per_particle_eval": flop_per_eval, "parallelism": f"particle shards x{world}"},
        "end_to_end_tflops": None if value is None or flop_per_eval is None else round(value * flop_per_eval / 1e12, 3),
        "host_enqueue_ms_per_step": None if t_enqueued is None else round(1e3 * t_enqueued / k, 4),
        "order": f"{N_PHASE_PASSES} generator / assessor passes (per-phase timing; they also take the device from idle to its working "
                 f"clock: the driver's --warmup steps come behind them), {w} warm-up steps, {k} timed steps, "
                 + (f"{k} more steps of the sharded loop with HIP events either side of the all-gather (collective_ms), " if world > 1 or (phases or {}).get("collective_ms") is not None else "")
                 + f"{k} more steps with HIP events around every convolution launch (roofline)",
        "roofline": roofline,
        "phases": phases,
        # per-rank spread of the timed region (value uses the slowest rank)
        "ranks": {"n_ranks_seen": ranks_seen, "ms_per_step_max": ms, "ms_per_step_min": None if dt_min is None else round(1e3 * dt_min / k, 4)},
    }
    out.update(extra or {})
    return out


def dry_run(args, world, rank, result_fd):
    """GDPSO_BENCH_DRYRUN=1: everything around the measurement without a GPU (rendezvous of the
    ranks over gloo, one exchange of candidate records and positions through engine.TorchComm, the
    max- and min-over-ranks reductions, rank 0's result line with every key of a real one), so that the
    launch path of `--gpus N` can be tested where there is no GPU.  The line says "dry_run": true, its
    times are made up (rank r "takes" r + 1 ms a step) and it carries no value."""
    import torch.distributed as dist
    eng = importlib.import_module("gan-discovery-pso_amd.engine")
    lib = importlib.import_module("gan-discovery-pso_amd._lib")
    k = args.steps
    dt, dt_min, coll_ms, seen = 1e-3 * k, 1e-3 * k, None, 1
    if world > 1:
        dist.init_process_group(backend="gloo")
        comm = eng.TorchComm()
        d, p = args.z_dim, 4
        cand = torch.zeros(world, lib.cand_doubles(d), dtype=torch.float64)
        cand[rank, 0], cand[rank, 1] = 1.0 + rank, rank * p
        pos = torch.full((p, d), float(rank), dtype=torch.float64)
        handle = comm.start_gather_positions(pos, world * p)
        cand[rank, d + 2] = 10.0 + rank
        t0 = time.perf_counter()
        comm.all_gather(cand, cand[rank])
        coll_ms = 1e3 * (time.perf_counter() - t0)
        allp = handle.wait()
        assert cand[:, 1].tolist() == [float(r * p) for r in range(world)]
        assert cand[:, d + 2].tolist() == [10.0 + r for r in range(world)]
        assert allp[:, 0].tolist() == [float(r) for r in range(world) for _ in range(p)]
        dt, dt_min = rank_spread(dist, 1e-3 * k * (rank + 1))
        assert (dt, dt_min) == (1e-3 * k * world, 1e-3 * k)
        seen = dist.get_world_size()
        dist.barrier()
    if rank == 0:
        phases = {"generator_ms": None, "assessor_ms": None, "pso_update_and_diagnostic_ms": None,
                  "collective_ms": None if coll_ms is None else round(coll_ms, 4), "how": "dry run: one host-timed gloo all-gather"}
        out = result_line(args, world, value=None, dt=dt, dt_min=dt_min, k=k, w=args.warmup, n_global=args.particles * world,
                          p=args.particles, z_dim=args.z_dim, side=args.side, flop_per_eval=None, t_enqueued=None, roofline=None,
                          phases=phases, ranks_seen=seen, extra={"dry_run": True})
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--particles", type=int, default=512, help="particles per GPU")
    ap.add_argument("--z-dim", type=int, default=10)
    ap.add_argument("--side", type=int, default=28, help="image side (28: DCGAN-MNIST; 64: the CLARO-sized generator)")
    ap.add_argument("--dtype", choices=sorted(PEAK_MFMA_TFLOPS), default="f32",
                    help="storage / matrix-instruction input type of the networks (f32: the reference's precision and the headline "
                         "configuration; bf16: the VQ-VAE-decoder configuration with --z-dim 100; f16: the 64x64 one with --side 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-shape", action="store_true",
                    help="skip the two extra measurements of the reference's own run shape (T = 50 from an idle device; 8 swarms x 32 particles)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started directly (python bench.py --gpus N): this process has not touched the GPU and never
        # will; it starts one fresh rank per GPU and relays rank 0's result line.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    # Libraries write to stdout too (RCCL prints a version banner there when a process group comes
    # up): everything but the result line goes to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("GDPSO_BENCH_DRYRUN") == "1":
        return dry_run(args, world, rank, result_fd)
    # GDPSO_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the multi-rank path
    backend = os.environ.get("GDPSO_BENCH_BACKEND", "nccl")
    if backend != "nccl" and world > 1:
        # ranks that share one GPU: a chained launch needs all its workgroups resident at once, which two
        # processes launching side by side cannot both have; one launch per layer there
        os.environ.setdefault("GDPSO_CONV_CHAIN", "0")
    local_rank %= max(1, torch.cuda.device_count()) if backend != "nccl" else 1 << 30
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    # GDPSO_BENCH_FORCE_COMM=1: a single rank still goes through the process group and the
    # sharded loop (rehearsal of the RCCL calls on a one-GPU box)
    force_comm = world == 1 and os.environ.get("GDPSO_BENCH_FORCE_COMM") == "1"
    if force_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_comm:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    eng = importlib.import_module("gan-discovery-pso_amd.engine")
    synth = importlib.import_module("gan-discovery-pso_amd.synth")
    lib = importlib.import_module("gan-discovery-pso_amd._lib")
    if world > 1 or force_comm:
        comm = eng.TorchComm()

    z_dim, n_class, class_index = args.z_dim, 8, 6
    gen = eng.PackedGenerator(synth.generator_state(z_dim, 1, 64, seed=11, first_kernel=args.side // 4), dev, dtype=args.dtype)
    ass = eng.PackedAssessor(synth.resnet_state("ResNet50", 1, n_class, seed=12), dev, img_side=gen.side, dtype=args.dtype)
    ev = eng.Evaluator(gen, ass)
    spec = eng.FitnessSpec(class_index=class_index)
    p, k, w = args.particles, args.steps, args.warmup
    n_global = p * world
    lo = rank * p
    # several ranks: equal shards, so the positions travel in the exchange table behind the records
    sw = eng.HipSwarm(ev, p, z_dim, 2 * (w + k), first_index=lo, n_ranks=world, rank=rank, n_global=n_global,
                      history=True, diagnostics=True,
                      table_positions=comm is not None and os.environ.get("GDPSO_BENCH_TABLE_POSITIONS", "1") != "0")
    rs = np.random.RandomState(42)
    r = rs.standard_normal((n_global, 2, z_dim))
    sw.load(r[lo:lo + p, 0], (r[lo:lo + p, 1] - 0.5) / 10, 0.73)

    if os.environ.get("GDPSO_BENCH_DEBUG"):
        import faulthandler
        faulthandler.dump_traceback_later(25, exit=True)

    def dbg(msg):
        if os.environ.get("GDPSO_BENCH_DEBUG"):
            print(f"[rank {rank}] {msg}", file=sys.stderr, flush=True)

    def run(first, count):
        dbg(f"run {first}..{first + count - 1}")
        if comm is None and not os.environ.get("GDPSO_BENCH_STEPWISE"):
            sw.iterate(first, count, spec, 1.496, 1.496, False, False, 1e-5, True, seed=42)
        elif comm is None:  # the per-phase calls of the sharded loop without the collectives (host-overhead check)
            for i in range(first, first + count):
                sw.evaluate(i, spec)
                sw.local_best(i)
                sw.global_best(i, False, False, 1e-5)
                sw.move(i, 1.496, 1.496, seed=42)
                sw.mean_distance(i, sw.pos)
        else:
            deep = os.environ.get("GDPSO_BENCH_DEBUG") == "1"  # any other value: progress lines and the hang dump only

            def hook(i, name):
                if deep:
                    torch.cuda.synchronize(dev)
                    dbg(f"{i} {name} ok")

            eng.sharded_iterations(sw, comm, first, count, spec, 1.496, 1.496, False, False, 1e-5, True, seed=42,
                                   after_phase=hook)

    def fence():
        torch.cuda.synchronize(dev)
        if comm is not None:
            comm.dist.barrier()
            torch.cuda.synchronize(dev)

    # First pass, every rank: where an iteration's time goes.  Generator and assessor are run on the swarm's
    # positions as calls of their own, back to back without a host synchronisation, with HIP events between
    # them on the launch stream; the PSO update (personal / global best, move, history) and the distance
    # diagnostic are what an iteration of the timed loop takes beyond the two.  It runs ahead of the warm-up
    # steps because it also brings the device from idle to its working clock, which takes ~25 iterations
    # (tools/iter_ramp.py: 0.96 ms per iteration at first, 0.83 from the 26th on, again after every idle
    # second); the numbers come from its last passes.
    marks = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(N_PHASE_PASSES)]
    for it in range(N_PHASE_PASSES):
        marks[it][0].record()
        img01, _, _ = ev.images(sw.pos)
        marks[it][1].record()
        ev.assess(img01, spec)
        marks[it][2].record()
    torch.cuda.synchronize(dev)
    gen_ms = float(np.mean([m[0].elapsed_time(m[1]) for m in marks[-N_PHASE_TIMED:]]))
    ass_ms = float(np.mean([m[1].elapsed_time(m[2]) for m in marks[-N_PHASE_TIMED:]]))

    run(1, w)
    dbg("warm-up enqueued")
    fence()
    dbg("warm-up done")
    t0 = time.perf_counter()
    run(w + 1, k)
    t_enqueued = time.perf_counter() - t0  # host time to enqueue the K steps (the device runs behind it)
    fence()
    dt = dt_min = time.perf_counter() - t0
    ranks_seen = 1
    if comm is not None:
        dt, dt_min = rank_spread(comm.dist, dt, dev if backend == "nccl" else None)
        ranks_seen = comm.dist.get_world_size()
    dbg("timed region done")
    ctrl = sw.control()
    assert int(ctrl[lib.CTRL_ITERS_DONE]) == w + k, "the loop did not run every iteration"

    # With a process group, every rank: K more steps of the sharded loop with HIP events on the launch stream
    # either side of the all-gather (the one collective of an iteration; it runs in stream order there).
    coll_ms = None
    if comm is not None:
        ev_pairs = [[torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)] for _ in range(k)]

        def mark(i, name):
            j = i - (w + k + 1)
            if 0 <= j < k and name == "local_best":
                ev_pairs[j][0].record()
            elif 0 <= j < k and name == "exchange":
                ev_pairs[j][1].record()

        eng.sharded_iterations(sw, comm, w + k + 1, k, spec, 1.496, 1.496, False, False, 1e-5, True, seed=42, after_phase=mark)
        fence()
        coll_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_pairs]))

    # second pass of K steps with HIP events around every convolution launch (same stream)
    roofline = None
    if rank == 0:
        L = lib.lib()
        lib.check(L.gdpso_conv_timing_begin(64 * k))
        if comm is None:
            run(w + k + 1, k)
            torch.cuda.synchronize(dev)
        else:  # price the kernel on this rank's own shard without the other ranks
            for i in range(w + k + 1, w + 2 * k + 1):
                sw.evaluate(i, spec)
            torch.cuda.synchronize(dev)
        ms, launches, flops = C.c_double(), C.c_int(), C.c_double()
        lib.check(L.gdpso_conv_timing_end(C.byref(ms), C.byref(launches), C.byref(flops)))
        achieved = flops.value / (ms.value * 1e-3) / 1e12
        alg_bytes = L.gdpso_conv_timing_bytes() / launches.value
        executed = L.gdpso_conv_timing_executed_flops()
        # HBM bytes per launch from the committed PMC passes of this same command (counters cannot be
        # read from inside the process); null when the summary is absent or was taken at another size
        traffic, traffic_file = None, TRAFFIC_FILES.get((args.dtype, z_dim, args.side, p))
        try:
            if traffic_file:
                with open(os.path.join(ROOT, "profiles", traffic_file)) as f:
                    traffic = json.load(f)["kernels"]["conv_gemm (all instantiations)"]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        peak = PEAK_MFMA_TFLOPS[args.dtype]
        tb_per_s = alg_bytes * launches.value / (ms.value * 1e-3) / 1e12
        # fp32-input MFMA is slow enough for the matrix pipe to be the roof.  With 16-bit inputs the pipe is 16x
        # faster and these layers are short: neither roof is near, what bounds the launches is the fixed cost of a
        # tile and of a step of the chained launch (prologue, hand-off, operand fetch, epilogue)
        bound = "mfma" if args.dtype == "f32" else "latency"
        roofline = {"bound": bound,
                    "bound_note": None if bound == "mfma" else
                    f"neither roof is near (matrix pipe {achieved / peak:.3f} of its peak, HBM {tb_per_s / 8.0:.3f} of 8 TB/s on the algorithmic bytes): "
                    "per-tile and per-step fixed costs (prologue, hand-off between dependent layers, operand fetch, epilogue); "
                    "`frac` prices the launches against the 16-bit MFMA peak all the same",
                    "kernel": f"every convolution launch of an iteration ({args.dtype}-input MFMA): conv_chain_kernel (the late ResNet stages in one "
                              "persistent launch of the implicit-GEMM tile code), stage_resident_kernel (stem + first two stages, two images' maps "
                              "resident in a workgroup's LDS; 28x28 images), gen_resident_kernel (the generator's three layers the same way, "
                              "first and last layer included in its FLOPs and its time) and conv_gemm_kernel / _pair / _fused (the same tile "
                              "code, one launch a layer: whatever the three above do not cover at this size)",
                    "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic,
                    "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE" + (", profiles/" + traffic_file if traffic_file else "; no PMC summary of this configuration") + ")",
                    "algorithmic_bytes_per_launch": round(alg_bytes),
                    "launches_per_step": launches.value // k, "avg_launch_us": round(1e3 * ms.value / launches.value, 3),
                    "conv_ms_per_step": round(ms.value / k, 4),
                    "algorithmic_gflop_per_step": round(flops.value / k / 1e9, 3),
                    # what the MFMAs really multiplied: taps of border pixels that fall in the padding are skipped
                    "executed_gflop_per_step": round(executed / k / 1e9, 3),
                    "executed_frac": round(executed / (ms.value * 1e-3) / 1e12 / peak, 4),
                    # the same launches priced against HBM: algorithmic bytes over kernel time
                    "algorithmic_tb_per_s": round(tb_per_s, 3)}
    # rank 0's own generator / assessor passes, the all-gather between HIP events (when there is one), and what
    # an iteration of the timed loop takes beyond them
    phases = {"generator_ms": round(gen_ms, 4), "assessor_ms": round(ass_ms, 4),
              "pso_update_and_diagnostic_ms": round(max(0.0, 1e3 * dt / k - gen_ms - ass_ms - (coll_ms or 0.0)), 4),
              "collective_ms": None if coll_ms is None else round(coll_ms, 4),
              "how": f"HIP events on the launch stream around the last {N_PHASE_TIMED} of {N_PHASE_PASSES} generator / assessor passes enqueued "
                     "back to back ahead of the warm-up steps (rank 0's)"
                     + ("; collective_ms: HIP events on the launch stream either side of the in-place all-gather of the exchange table "
                        f"(candidate records + positions, {sw.exchange_table[0].numel() * 8} B per rank), mean over {k} steps of a pass of its own, "
                        "which includes waiting for the slowest rank's records" if coll_ms is not None else "")
                     + "; the rest = ms_per_step of the timed loop minus the others"}
    extra = {}
    if rank == 0 and world == 1 and comm is None and args.dtype == "f32" and not args.no_reference_shape:
        extra["reference_run_shape"] = reference_run_shape(eng, ev, spec, z_dim, dev)
    if comm is not None:
        comm.dist.barrier()

    if rank == 0:
        value = n_global * k / dt
        flop_per_eval = gen.flops_per_image + ass.flops_per_image
        out = result_line(args, world, value=value, dt=dt, dt_min=dt_min, k=k, w=w, n_global=n_global, p=p, z_dim=z_dim, side=gen.side,
                          flop_per_eval=flop_per_eval, t_enqueued=t_enqueued, roofline=roofline, phases=phases, ranks_seen=ranks_seen,
                          extra=extra)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(z_dim, n_class, class_index)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.dist.destroy_process_group()


if __name__ == "__main__":
    main()
