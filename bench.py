#!/usr/bin/env python3
"""
bench.py — whole-job QP solves/s of the MI355X batched solve path (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic robot instances:
one DCM-MPC QP (N = 50, BASELINE configs[1]) and one Jacobian QP-IK (iCub 23 DoF,
BASELINE configs[2]) per instance, i.e. 2 QP solves per robot-tick, `--batch`
instances per GPU (default 4096, the batch both configs are quoted on), every
instance cold-started.  Inputs are resident in HBM before the timed region starts.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Instances are independent, so ranks shard the batch with no data-path collective
(`scaling: weak`, fixed per-GPU batch); `--exchange` adds the RCCL scatter of inputs from
rank 0 and gather of solutions to every step and reports that rate too.

The JSON line carries
  roofline      HBM roofline of the dominant kernel (the IK kernel): algorithmic bytes per
                launch (5240 B/IK-QP x batch) / its average launch duration,
                measured with HIP events on the launch stream over the timed steps;
  cpu_baseline  oracle/wc_oracle.c (OSQP-algorithm restatement for the MPC, dense dual
                active set for the IK) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "QP solves/sec (whole node), iCub IK-QP + DCM-MPC batch at 1/2/4/8 MI355X"
IK_BYTES_PER_QP = 5240      # 632 doubles in + 23 doubles out
MPC_BYTES_PER_QP = 1056     # 130 doubles in + 2 doubles out
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="robot instances per GPU")
    ap.add_argument("--ik-vmax", type=float, default=0.5, help="joint velocity limit of the synthetic robots [rad/s]")
    ap.add_argument("--ik-form", choices=["qpoases", "osqp"], default="qpoases")
    ap.add_argument("--exchange", action="store_true", help="RCCL scatter inputs / gather solutions every step")
    ap.add_argument("--workload", choices=["qp", "tick", "kin"], default="qp",
                    help="qp: configs[1]+[2] cold-start batches (default); tick: the device-resident receding-horizon "
                         "MPC->glue->IK tick of configs[3]/[4], one hipGraph replay per step")
    ap.add_argument("--no-graph", action="store_true", help="tick workload: plain launches instead of hipGraph replay")
    ap.add_argument("--streams", type=int, choices=[0, 1, 2], default=0,
                    help="qp workload: 2 = the (independent) MPC and IK batches go to two HIP streams and may overlap; "
                         "0 = auto: 2 up to 32768 robots per GPU (+18 %% at 4096, +18 %% at 8192, +9 %% at 16384, +2 %% at 32768: the MPC "
                         "kernel fits beside the IK kernel), 1 above (at 65536 the overlap slows the IK kernel more than it saves)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work of the baseline sample")
    args = ap.parse_args()

    import torch
    import walking_controllers_amd as wca

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the solve path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(1, ndev)          # == local_rank on a full node; lets a 1-GPU box rehearse N > 1
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    backend = os.environ.get("WCQP_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; "gloo" only for rehearsals
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    assert world == max(1, args.gpus) or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    B = args.batch
    first = rank * B
    if args.workload == "tick":
        return bench_tick(args, wca, torch, dist, dev, world, rank, B, first)
    if args.workload == "kin":
        return bench_kin(args, wca, torch, dist, dev, world, rank, B, first)
    # ---- synthetic inputs (each rank generates its own shard: identical to the rows a rank-0
    # scatter would hand it, walking-controllers_amd/synth.py is counter-based) -------------
    mb = wca.synth.synth_mpc_batch(B, seed=1234, first=first)
    ib = wca.synth.synth_ik_batch(B, seed=4321, first=first)

    def up(a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(dev) if dtype is None else t.to(dev, dtype)

    d = {k: up(mb[k]) for k in ("x0", "ref", "u_prev", "hull_A", "hull_b")}
    d["hull_nc"] = up(mb["hull_nc"])
    d.update({k: up(ib[k]) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")})
    u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    mstat = torch.zeros(B, dtype=torch.int32, device=dev)
    mact = torch.zeros(B, dtype=torch.int32, device=dev)
    mmar = torch.zeros(B, dtype=torch.float64, device=dev)
    dq = torch.zeros(B, 23, dtype=torch.float64, device=dev)
    istat = torch.zeros(B, dtype=torch.int32, device=dev)
    ilo = torch.zeros(B, dtype=torch.int32, device=dev)
    iup = torch.zeros(B, dtype=torch.int32, device=dev)
    iit = torch.zeros(B, dtype=torch.int32, device=dev)

    mpc = wca.MpcSolver(horizon=50)
    ik = wca.IkSolver(form=wca.IK_FORM_QPOASES if args.ik_form == "qpoases" else wca.IK_FORM_OSQP, v_max=args.ik_vmax)
    stream = torch.cuda.current_stream(dev)
    sp = stream.cuda_stream
    n_streams = args.streams if args.streams else (2 if B <= 32768 else 1)
    two_streams = n_streams == 2 and not (args.exchange and world > 1)
    stream_mpc = torch.cuda.Stream(dev) if two_streams else stream
    sp_mpc = stream_mpc.cuda_stream
    N1 = mb["ref"].shape[1]

    def launch_mpc():
        mpc.solve_device(B, d["x0"].data_ptr(), d["ref"].data_ptr(), N1, d["u_prev"].data_ptr(),
                         d["hull_A"].data_ptr(), d["hull_b"].data_ptr(), d["hull_nc"].data_ptr(),
                         u0.data_ptr(), mstat.data_ptr(), mact.data_ptr(), mmar.data_ptr(), sp_mpc)

    def launch_ik():
        ik.solve_device(B, d["J_left"].data_ptr(), d["J_right"].data_ptr(), d["J_neck"].data_ptr(),
                        d["J_com"].data_ptr(), d["q"].data_ptr(), d["state"].data_ptr(),
                        dq.data_ptr(), istat.data_ptr(), ilo.data_ptr(), iup.data_ptr(), 0, iit.data_ptr(), sp)

    # optional RCCL exchange (rank 0 owns the whole batch)
    exch = None
    if args.exchange and world > 1:
        in_keys = ("x0", "ref", "u_prev", "hull_A", "hull_b", "J_left", "J_right", "J_neck", "J_com", "q", "state")
        if rank == 0:
            full = {k: [torch.empty_like(d[k]) for _ in range(world)] for k in in_keys}
            for k in in_keys:
                for r in range(world):
                    full[k][r].copy_(d[k])           # shape-true stand-ins: only the traffic matters here
            gat_u0 = [torch.empty_like(u0) for _ in range(world)]
            gat_dq = [torch.empty_like(dq) for _ in range(world)]
        else:
            full, gat_u0, gat_dq = None, None, None

        def exch_in():
            for k in in_keys:
                dist.scatter(d[k], full[k] if rank == 0 else None, src=0)

        def exch_out():
            dist.gather(u0, gat_u0 if rank == 0 else None, dst=0)
            dist.gather(dq, gat_dq if rank == 0 else None, dst=0)
        exch = (exch_in, exch_out)

    bracket_ik = [False]      # two streams with fewer than 2 sampled steps: fall back to bracketing the IK launch

    def step(ev=None):
        if exch:
            exch[0]()
        launch_mpc()
        if ev is not None:
            ev[0].record(stream)
        launch_ik()
        if ev is not None and (not two_streams or bracket_ik[0]):
            ev[1].record(stream)
        if exch:
            exch[1]()

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events around the IK launch of every `stride`-th timed step (a pair of event records costs about
    # as much as a launch, so bracketing every step would slow the thing being measured).  With two streams
    # the IK stream carries nothing but IK launches: one event every `stride` steps, and the kernel's
    # average duration is the time between consecutive events / stride (inter-launch gap included).
    stride = max(1, int(os.environ.get("WCQP_BENCH_EVENT_STRIDE", str(max(1, min(8, args.steps // 8))))))
    events = {k: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for k in range(0, args.steps, stride)}
    bracket_ik[0] = len(events) < 2
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events.get(k))
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(dist, torch, dev, elapsed)
    if two_streams and len(events) >= 2:
        keys = sorted(events)
        ik_ms = float(np.mean([events[a][0].elapsed_time(events[b][0]) / (b - a) for a, b in zip(keys[:-1], keys[1:])]))
    else:
        ik_ms = float(np.mean([a.elapsed_time(b) for a, b in events.values()]))      # IK kernel, HIP events on its stream
    # MPC kernel duration: a short separately timed run (it is not the dominant kernel)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record(stream_mpc)
    for _ in range(50):
        launch_mpc()
    e1.record(stream_mpc)
    torch.cuda.synchronize(dev)
    mpc_ms = e0.elapsed_time(e1) / 50.0

    # sanity: the timed work really solved the problems
    n_ok_ik = int((istat == 0).sum().item())
    n_ok_mpc = int((mstat == 0).sum().item())
    ik_iters = float(iit.double().mean().item())
    frac_active = float(((ilo | iup) != 0).double().mean().item())

    total_qp = 2 * B * world * args.steps
    value = total_qp / elapsed
    out = {
        "metric": METRIC, "value": value, "unit": "QP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[1]+[2] per GPU: DCM-MPC QP (N=50, n=202, cold start) B=%d + QP-IK "
                         "(iCub 23 DoF, 15 eq rows, %s form, v_max=%.2f rad/s) B=%d; 2 QP solves per robot-tick"
                         % (B, args.ik_form, args.ik_vmax, B)),
            "batch_per_gpu": B, "global_batch": B * world, "horizon": 50, "dof": 23,
            "parallelism": "batch sharded over %d GPU(s), no data-path collective%s%s" % (world, " + RCCL scatter/gather" if exch else "", "; MPC and IK batches on two HIP streams" if two_streams else ""),
        },
        "roofline": {
            "bound": "hbm", "kernel": "ik3_kernel",
            "achieved": IK_BYTES_PER_QP * B / (ik_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": IK_BYTES_PER_QP * B / (ik_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "avg_launch_ms": ik_ms, "algorithmic_bytes_per_launch": IK_BYTES_PER_QP * B,
        },
        "kernels": {
            "ik_ms": ik_ms, "ik_qps_per_gpu": B / (ik_ms * 1e-3),
            # secondary bound: ~30 kflop per IK-QP (elimination 13.5 k, Gram product 8.1 k, sweep 5.5 k,
            # tables / x / active set ~3 k) against the 78.6 TFLOP/s fp64 vector peak
            "ik_fp64_tflops": 30e3 * B / (ik_ms * 1e-3) / 1e12, "ik_fp64_frac": 30e3 * B / (ik_ms * 1e-3) / 78.6e12,
            "mpc_ms": mpc_ms, "mpc_qps_per_gpu": B / (mpc_ms * 1e-3),
            "mpc_hbm_frac": MPC_BYTES_PER_QP * B / (mpc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        },
        "solved": {"ik": n_ok_ik, "mpc": n_ok_mpc, "of": B, "ik_mean_active_set_changes": ik_iters,
                   "ik_frac_with_active_bounds": frac_active},
    }
    # HBM bytes per launch from the PMC passes of tools/pmc/collect.sh (committed summary)
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file)).get("per_batch", {}).get(str(B))
            if tr:
                out["roofline"]["traffic"] = tr.get("ik_hbm_bytes_per_launch")
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(mb, ib, args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def max_over_ranks(dist, torch, dev, value):
    if dist is None:
        return value
    on = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def bench_kin(args, wca, torch, dist, dev, world, rank, B, first):
    """Auxiliary line: the kinematics kernel alone - Jacobians and poses of B robots per step.
    Not the BASELINE metric: `metric` says so; the roofline is the kernel's own (HBM: it is bound by the 4.4 KB it
    writes per robot)."""
    kb = wca.synth.synth_kin_batch(B, first=first)
    kin = wca.KinModel(wca.synth.icub_like_model())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    base, q = t(kb["base"]), t(kb["q"])
    JL = torch.zeros(B, 6, 29, dtype=torch.float64, device=dev); JR = torch.zeros_like(JL)
    JN = torch.zeros(B, 3, 29, dtype=torch.float64, device=dev); JC = torch.zeros_like(JN)
    state = torch.zeros(B, 87, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        kin.jacobians_device(B, base.data_ptr(), q.data_ptr(), JL.data_ptr(), JR.data_ptr(), JN.data_ptr(), JC.data_ptr(),
                             state.data_ptr(), stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(args.steps):
        step()
    e1.record(stream)
    barrier()
    elapsed = max_over_ranks(dist, torch, dev, time.perf_counter() - t0)
    k_ms = e0.elapsed_time(e1) / args.steps
    bytes_per = 12 * 8 + 23 * 8 + (6 + 6 + 3 + 3) * 29 * 8 + 36 * 8           # base + q in; Jacobians + actual poses out
    out = {
        "metric": "robots/sec through the kinematics kernel (auxiliary; NOT the BASELINE metric)",
        "value": B * world * args.steps / elapsed, "unit": "robots/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "forward kinematics + MIXED free-floating Jacobians (2 feet, neck, CoM) of an iCub-shaped 23-DoF tree, B=%d" % B,
                   "batch_per_gpu": B, "global_batch": B * world, "dof": 23},
        "roofline": {"bound": "hbm", "kernel": "kin_jacobians_kernel", "achieved": bytes_per * B / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": bytes_per * B / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "avg_launch_ms": k_ms, "algorithmic_bytes_per_launch": bytes_per * B},
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def bench_tick(args, wca, torch, dist, dev, world, rank, B, first):
    """BASELINE configs[3]/[4]: every step is one robot-tick of the whole batch — MPC on the
    receding window of the per-instance DCM trajectory, ZMP-CoM glue, IK, joint integration —
    with all solver and plant state resident in HBM and the six launches replayed from a hipGraph."""
    T = args.steps + args.warmup
    # `--streams 2` (auto from 8192 robots per GPU; measured 39.6 -> 36.2 us per tick there, no gain at 4096): the batch is cut into two independent halves, each its
    # own pipeline on its own HIP stream, so that one half's (HBM-bound) MPC kernel overlaps the other half's
    # IK kernel.  Synthetic robots are counter-based, so the two halves are exactly the rows of the full batch.
    n_streams = args.streams if args.streams else (2 if B >= 8192 and B % 2 == 0 else 1)
    parts = [(first, B)] if n_streams == 1 else [(first, B // 2), (first + B // 2, B - B // 2)]
    ik_form = wca.IK_FORM_QPOASES if args.ik_form == "qpoases" else wca.IK_FORM_OSQP
    pipes = []
    for f0, cnt in parts:
        pp = wca.TickPipeline(cnt, T, wca.MpcSolver(horizon=50), wca.IkSolver(form=ik_form, v_max=args.ik_vmax), first=f0)
        pp.upload(wca.synth.synth_tick_batch(cnt, T, first=f0))
        pipes.append(pp)
    stream = torch.cuda.current_stream(dev)
    streams = [stream] + [torch.cuda.Stream(dev) for _ in pipes[1:]]
    graph = not args.no_graph

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run(n):
        for pp, st in zip(pipes, streams):
            pp.run(n, use_graph=graph, stream=st.cuda_stream)

    run(args.warmup)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    run(args.steps)
    for st in streams[1:]:
        stream.wait_stream(st)
    e1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(dist, torch, dev, elapsed)
    states = [pp.download() for pp in pipes]
    out_state = {"tick": min(x["tick"] for x in states), "mpc_fail": np.concatenate([x["mpc_fail"] for x in states]),
                 "ik_fail": np.concatenate([x["ik_fail"] for x in states])}
    dev_ms = e0.elapsed_time(e1) / args.steps
    value = 2 * B * world * args.steps / elapsed
    bytes_per_tick = 6296 + 2 * 8 * (2 * 10 + 23 * 3)      # algorithmic I/O + resident controller/plant state read+written
    out = {
        "metric": METRIC, "value": value, "unit": "QP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[3]/[4] per GPU: receding-horizon robot-tick (DCM-MPC N=50 on the advancing "
                         "reference window -> ZMP-CoM glue -> QP-IK 23 DoF %s form v_max=%.2f -> joint integration), "
                         "B=%d robots, %s, contact pair changes every 70-110 ticks; 2 QP solves per robot-tick"
                         % (args.ik_form, args.ik_vmax, B, "hipGraph replay" if graph else "plain launches")),
            "batch_per_gpu": B, "global_batch": B * world, "horizon": 50, "dof": 23, "ticks": args.steps,
            "parallelism": "batch sharded over %d GPU(s), no data-path collective%s" % (world, "; two half-batches on two HIP streams" if len(pipes) > 1 else ""),
        },
        "roofline": {"bound": "hbm", "kernel": "whole tick (2 launches: mpc_condensed, ik3_kernel<TICK> with glue and post fused)", "achieved": bytes_per_tick * B / (dev_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_per_tick * B / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "avg_launch_ms": dev_ms, "algorithmic_bytes_per_launch": bytes_per_tick * B},
        "solved": {"ticks_executed": out_state["tick"], "mpc_fail": int(out_state["mpc_fail"].sum()),
                   "ik_fail": int(out_state["ik_fail"].sum()), "of": B * T},
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(mb, ib, args):
    """oracle/wc_oracle.c on the host cores: the reference's CPU algorithms (OSQP for the MPC,
    an active-set method for the qpOASES-form IK) restated in C — `kind: port`."""
    from oracle import c_oracle as co
    from oracle import qp_spec as qs
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:                                                  # honour a cgroup CPU quota if there is one
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            avail = min(avail, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    cores = max(1, min(co.num_threads(), avail))
    mp = qs.MPCParams()
    ipar = qs.IKParams(v_max=args.ik_vmax * np.ones(23))

    def take(b, n):
        return {k: v[:n] for k, v in b.items()}
    # calibrate single-threaded on a small slice, then size the sample to ~cpu_seconds of CPU work
    t = time.perf_counter(); co.mpc_batch_osqp(mp, take(mb, 32), nthreads=1); t_m = (time.perf_counter() - t) / 32
    t = time.perf_counter(); co.ik_batch(ipar, take(ib, 32), args.ik_form, nthreads=1); t_i = (time.perf_counter() - t) / 32
    n = len(mb["x0"])                                     # the whole per-GPU batch
    reps = max(1, int(round(args.cpu_seconds / max((t_m + t_i) * n, 1e-9))))
    co.mpc_batch_osqp(mp, take(mb, n), nthreads=cores)    # warm the thread pool
    t = time.perf_counter()
    for _ in range(reps):
        co.mpc_batch_osqp(mp, take(mb, n), nthreads=cores)
    wall_m = time.perf_counter() - t
    co.ik_batch(ipar, take(ib, n), args.ik_form, nthreads=cores)
    t = time.perf_counter()
    for _ in range(reps):
        co.ik_batch(ipar, take(ib, n), args.ik_form, nthreads=cores)
    wall_i = time.perf_counter() - t
    return {"value": 2 * n * reps / (wall_m + wall_i), "unit": "QP/s", "cores": cores, "kind": "port",
            "sample": "%d x the first %d instances of the same workload (1 MPC via OSQP-restatement + 1 IK via %s per instance), "
                      "OpenMP static split" % (reps, n, "dense active set" if args.ik_form == "qpoases" else "OSQP-restatement"),
            "mpc_qps": n * reps / wall_m, "ik_qps": n * reps / wall_i,
            "single_thread_qps": {"mpc": 1.0 / t_m, "ik": 1.0 / t_i}}


if __name__ == "__main__":
    main()
