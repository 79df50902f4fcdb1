#!/usr/bin/env python3
"""
bench.py — whole-job QP solves/s of the MI355X batched solve path (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic robot instances:
one DCM-MPC QP (N = 50, BASELINE configs[1]) and one Jacobian QP-IK (iCub 23 DoF,
BASELINE configs[2]) per instance, i.e. 2 QP solves per robot-tick, `--batch`
instances per GPU (default 4096, the batch both configs are quoted on), every
instance cold-started.  Inputs are resident in HBM before the timed region starts.

    python bench.py                       # 1 GPU
    python bench.py --gpus N              # starts N ranks itself (one per GPU; refuses when fewer GPUs are visible)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Consecutive steps are independent batches (cold start: nothing is carried from one step to the next), so a step does not
have to be a launch: the K timed steps are handed to the library once (`wcqp_qp_plan_create`) and enqueued as ONE launch
(`wcqp_qp_plan_enqueue`) in which a wavefront owns four robots and walks through the steps on its own - the MPC of a step on
the IK's lanes, under its Jacobian loads - with `--plan-ways` wavefronts per robot group (way w takes steps w, w + ways, ...;
each way writes its own output buffers).  `--plan-ways 0` is the round-2 form: one launch per step (`qp_pair_kernel`), step i
on pipeline i % 3 (own stream, own outputs), all handed over in one `wcqp_qp_enqueue_steps` call.
Every step reads cold inputs (input sets > 1 GiB in total, visited round-robin) and the outputs of every way's last batch
are compared with the committed golden vectors after the timed region (`solved.golden_*`).
Timed region: barrier + torch.cuda.synchronize() -> K steps -> every stream's completion event (hipEventSynchronize),
MAX over ranks; the device-wide synchronize follows the clock (`ms_per_step_incl_device_sync` keeps it inside: on this
ROCm stack that call costs the host 20-75 us with the device already idle, a fifth of a 20-step region).

Instances are independent, so ranks shard the batch with no data-path collective
(`scaling: weak`, fixed per-GPU batch); `--exchange` adds the RCCL scatter of inputs from
rank 0 and gather of solutions to every step and reports that rate too.

The JSON line carries
  roofline      HBM roofline of the dominant kernel - the timed launch itself, `qp_plan_kernel`: algorithmic bytes per launch
                (6296 B per robot-tick = 5240 B/IK-QP + 1056 B/MPC-QP, SURVEY.md §8d, x batch x steps) / the duration of a launch
                over `steps` cold records, measured with HIP events on the launch stream in a pass of its own after the timed
                region (rocprofv3's average duration of that kernel for the same command is `avg_launch_ms`); `traffic` = HBM
                bytes per step from the PMC passes (profiles/traffic.json, quoted only while the kernel sources are the ones
                it was measured on); `kernels` keeps the stand-alone IK and MPC kernels (`--plan-ways 0`: `qp_pair_kernel`);
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
    ap.add_argument("--streams", type=int, choices=[0, 1, 2, 3, 4], default=0,
                    help="qp workload: 0/1 = the MPC and the IK of a step are ONE launch (wcqp_qp_enqueue_steps, both on one stream: "
                         "default); 2 = two launches on two streams.  tick workload: robot groups, each with a pipeline and a stream of its own "
                         "(0 = 2, or 3 with per-tick kinematics, from 8192 robots per GPU)")
    ap.add_argument("--pipelines", type=int, default=0,
                    help="qp workload: consecutive steps are independent batches (cold start, nothing carried over), so step i goes to "
                         "pipeline i %% P, each with its own streams and output buffers, and the load phase of one batch overlaps the "
                         "arithmetic of the previous one; 0 = 3 (1 with --exchange)")
    ap.add_argument("--horizon", type=int, default=50, help="qp workload: MPC horizon N (BASELINE: 50; the shipped controllerHorizon 2 s is N = 200: auxiliary line)")
    ap.add_argument("--input-sets", type=int, default=0, help="qp workload: distinct input sets visited round-robin (0 = enough for > 320 MB, at least 2)")
    ap.add_argument("--ik-jac", choices=["mixed", "auto", "general"], default="mixed",
                    help="wcqp_ik_params.jacobian_structure: mixed = the caller states what the reference always passes (iDynTree MIXED "
                         "free-floating Jacobians; checked per instance, one launch), auto = + the general kernel over non-conforming "
                         "instances (one more, nearly empty, launch), general = the general kernel only")
    ap.add_argument("--tick-tables", action="store_true", help="tick workload: constant uploaded Jacobians and precomputed hull tables "
                    "(round-1 form) instead of per-tick kinematics")
    ap.add_argument("--tick-kin-handoff", choices=["fused", "dense", "compact"], default="fused",
                    help="tick workload with kinematics: fused = the solve kernel evaluates the kinematics itself (one launch per tick / many ticks "
                         "per launch); dense / compact = a kinematics launch per tick handing over four dense Jacobians / per-joint records (A/B; same results)")
    ap.add_argument("--ticks-per-launch", type=int, default=0, help="tick workload without per-tick kinematics: ticks the fused kernel runs per launch (0 = the library's default, 1 = one launch per tick)")
    ap.add_argument("--tick-cold-ik", action="store_true", help="tick workload: no IK hot start (every tick walks the active set from the unconstrained optimum)")
    ap.add_argument("--plan-ways", type=int, default=-1,
                    help="qp workload: the timed steps as ONE launch (wcqp_qp_plan_*): consecutive steps are independent batches, so a wavefront "
                         "owns four robots and walks through the steps on its own, the MPC of a step in the shadow of its IK's Jacobian loads; "
                         "W wavefronts share a robot group (way w takes steps w, w + W, ...; each way has its own output buffers, like a "
                         "pipeline).  0 = one launch per step on --pipelines streams (round-2 form); -1 (default) = as many ways as put >= 16384 "
                         "workgroups into the launch (8 x the card's resident wavefronts: the hardware's workgroup dispatch then evens out the "
                         "launch's ends), at least 4, at most one per step: 16 at 4096 robots, 4 from 16384 on")
    ap.add_argument("--plan-queue", type=int, choices=[0, 1], default=0,
                    help="plan mode: 1 = the one launch hands out (step, robot group) units from a work queue instead of fixed ways - as many "
                         "wavefronts as are resident at once, each taking the next unit when it is done with one; every timed step then has "
                         "output buffers of its own (any two steps may be in flight together) and EVERY timed step is checked against the goldens")
    ap.add_argument("--step-graph", action="store_true", help="qp workload: replay ONE hipGraph that holds the timed steps as P parallel chains (one per "
                    "pipeline) instead of enqueueing them launch by launch.  Measured and left off: 8.9-9.2 us per step against 8.5-9.1 in "
                    "the driver's 20-step form, 8.9 against 7.0 at 200 steps (profiles/r03_step_graph_ab.txt)")
    ap.add_argument("--resident-pass", action="store_true", help="plan mode: also time the dominant kernel re-reading ONE input set (frac_resident_inputs).  Off by "
                    "default: every qp_plan_kernel launch of a run then walks through the same number of cold records, so that rocprofv3's "
                    "average duration and the PMC bytes per launch of that kernel are the numbers of the roofline object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work of the baseline sample")
    args = ap.parse_args()

    # `--gpus N` without a launcher: this process becomes the launcher - N children, one rank per GPU, started BEFORE
    # anything here touches the GPU (the parent never does); rank 0's JSON line is relayed, any child's failure is ours
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)

    import torch
    import walking_controllers_amd as wca

    launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node == --gpus, "
                         "or without a launcher: bench.py then starts the ranks itself)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the solve path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("WCQP_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; "gloo" only for rehearsals on fewer GPUs
    if ndev < world and backend != "gloo":
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s): one rank per GPU (WCQP_DIST_BACKEND=gloo lets a smaller box "
                         "REHEARSE the launch with the ranks sharing its GPUs - not a measurement)" % (world, ndev))
    dev_index = local_rank % max(1, ndev)          # == local_rank on a full node
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if launched:                                   # under a launcher, also with ONE rank (RCCL with world size 1 is a valid group)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    B = args.batch
    first = rank * B
    if args.workload == "tick":
        return bench_tick(args, wca, torch, dist, dev, world, rank, B, first)
    if args.workload == "kin":
        return bench_kin(args, wca, torch, dist, dev, world, rank, B, first)
    # ---- synthetic inputs (each rank generates its own shard: identical to the rows a rank-0
    # scatter would hand it, walking-controllers_amd/synth.py is counter-based) -------------
    NH = args.horizon
    mpc_bytes = 8 * (2 + 2 * (NH + 1) + 2 + 24) + 16          # x0, reference window, u_prev, padded hull in; u0 out (1056 B at N = 50)
    mb = wca.synth.synth_mpc_batch(B, seed=1234, first=first, horizon=NH)
    ib = wca.synth.synth_ik_batch(B, seed=4321, first=first)

    def up(a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(dev) if dtype is None else t.to(dev, dtype)

    base = {k: up(mb[k]) for k in ("x0", "ref", "u_prev", "hull_A", "hull_b")}
    base["hull_nc"] = up(mb["hull_nc"])
    base.update({k: up(ib[k]) for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")})
    # K input sets in distinct HBM allocations, visited round-robin, so that every step reads COLD inputs: one set is
    # 6.1 KB x B (25 MB at 4096 robots) and would otherwise sit in the 256 MiB Infinity Cache from the previous step.
    # Set k is the same batch rotated by k B / K instances: same work per launch, different bytes at every address.
    set_bytes = B * (mpc_bytes - 16 + IK_BYTES_PER_QP - 184)
    # (> 1 GiB in total: with 320 MB - 14 sets at 4096 robots - the PMC passes of round 3 showed the one-launch plan getting 31 % of
    # its input bytes from the 256 MiB Infinity Cache, HBM traffic 0.69 x the algorithmic bytes; launch by launch it was 1.04 x)
    # AND far apart in the plan's own order: a wavefront of a 4-way plan comes back to input set k after K / gcd(K, ways) of its
    # records, and in that time the card's resident wavefronts have read that many x 2048 x 25 KB.  With 3 sets of 400 MB (65536
    # robots: 1.2 GB "cold" by the first rule) that is 154 MB - the sets were coming out of the Infinity Cache, and the kernel showed
    # 0.65 of the roofline where it is 0.55 with inputs of its own for every step (profiles/r03_plan_queue_distinct_inputs.txt).
    # Plan mode therefore gives EVERY step input arrays of its own (below); the launch-per-step form keeps > 1 GiB of sets, at least 13.
    import math
    if args.plan_ways < 0:
        args.plan_ways = int(max(1, min(args.steps, max(4, -(-16384 // ((B + 3) // 4))))))
    if args.input_sets > 0:
        K = args.input_sets
    elif args.plan_queue and args.workload == "qp":
        K = args.steps + args.warmup
        if K * set_bytes > 96e9:
            raise SystemExit("bench.py: --plan-queue 1 gives every step input arrays of its own; %d steps x %.0f MB do not fit" % (K, set_bytes / 1e6))
    elif args.plan_ways > 0 and args.workload == "qp":
        # plan mode: every step input arrays of its own while they fit 64 GB (a 4-way plan behind 13 coprime sets is cold, but e.g. 50 ways
        # behind 47 sets are not: way w reads set k as its first record and way w - 3 the same set as its second, 63 us = 260 MB of
        # the card's reads apart - 1.41e9 QP/s where it is 1.32e9; profiles/r03_plan_ways_20steps.txt); beyond that as many as fit
        # (and never fewer than the > 1 GiB of the launch-per-step rule: the sets behind the timed ones are what the set-up pass reads
        # last, so the timed region does not find the tail of the set-up in the Infinity Cache)
        K = int(min(max(args.steps + args.warmup, 13, -(-(1 << 30) // set_bytes)), max(13, 64e9 // set_bytes)))
        if K < args.steps + args.warmup:
            while math.gcd(K, max(1, args.plan_ways)) != 1:
                K += 1
    else:
        K = int(max(13, min(64, -(-(1 << 30) // set_bytes))))
    sets = [base] + [{k: torch.roll(v, shifts=j * max(1, B // K), dims=0).contiguous() for k, v in base.items()} for j in range(1, K)]
    use_plan = (args.plan_ways > 0 and not (args.exchange and dist is not None) and args.streams in (0, 1) and args.ik_jac == "mixed"
                and not args.step_graph)
    plan_ways = 0 if args.plan_queue else args.plan_ways                        # wcqp_qp_plan_create: 0 = work queue
    if use_plan:
        # one output buffer set per way; with the work queue one per timed step
        P = args.pipelines if args.pipelines > 0 else (args.steps if args.plan_queue else args.plan_ways)
    else:
        P = args.pipelines if args.pipelines > 0 else (1 if (args.exchange and dist is not None) else 3)

    def outputs():
        return dict(u0=torch.zeros(B, 2, dtype=torch.float64, device=dev), mstat=torch.zeros(B, dtype=torch.int32, device=dev),
                    mact=torch.zeros(B, dtype=torch.int32, device=dev), mmar=torch.zeros(B, dtype=torch.float64, device=dev),
                    dq=torch.zeros(B, 23, dtype=torch.float64, device=dev), istat=torch.zeros(B, dtype=torch.int32, device=dev),
                    ilo=torch.zeros(B, dtype=torch.int32, device=dev), iup=torch.zeros(B, dtype=torch.int32, device=dev),
                    iit=torch.zeros(B, dtype=torch.int32, device=dev))
    outs = [outputs() for _ in range(P)]
    u0, dq = outs[0]["u0"], outs[0]["dq"]

    mpc = wca.MpcSolver(horizon=NH)
    ik_form = wca.IK_FORM_QPOASES if args.ik_form == "qpoases" else wca.IK_FORM_OSQP
    jac = {"mixed": wca.IK_JAC_MIXED, "auto": wca.IK_JAC_AUTO, "general": wca.IK_JAC_GENERAL}[args.ik_jac]
    ik = wca.IkSolver(form=ik_form, v_max=args.ik_vmax, jacobian_structure=jac)
    # pipeline 0 gets a stream of its own as well (a hipGraph cannot be captured on the default stream); the exchange path
    # stays on the current stream, where torch.distributed enqueues its collectives
    torch.cuda.synchronize(dev)                    # inputs and zero-filled outputs were enqueued on the default stream
    stream = torch.cuda.current_stream(dev) if (args.exchange and dist is not None) else torch.cuda.Stream(dev)
    sp = stream.cuda_stream
    # one stream per pipeline: wcqp_qp_enqueue_steps then makes the two calls of a step as ONE launch (IK and MPC workgroups
    # side by side); --streams 2 keeps them as two launches on two streams (the exchange path always does)
    n_streams = args.streams if args.streams else (2 if (args.exchange and dist is not None and B <= 32768) else 1)
    two_streams = n_streams == 2
    stream_mpc = torch.cuda.Stream(dev) if two_streams else stream
    sp_mpc = stream_mpc.cuda_stream
    # pipeline 0 = (stream, stream_mpc); further pipelines get streams of their own
    pipes = [(stream, stream_mpc)] + [((stream, None) if use_plan else (torch.cuda.Stream(dev), torch.cuda.Stream(dev) if two_streams else None)) for _ in range(P - 1)]
    pipes = [(a, b if b is not None else a) for a, b in pipes]
    all_streams = []
    for pr in pipes:
        for x in pr:
            if not any(x is y for y in all_streams):
                all_streams.append(x)
    if not any(stream is y for y in all_streams):
        all_streams.append(stream)
    N1 = mb["ref"].shape[1]

    # raw device addresses, looked up once: a step is two kernels of 5 and 15 us, and a dozen Tensor.data_ptr() calls per
    # launch cost the host about as much as the launch itself
    for d in sets:
        d["_mpc"] = tuple(d[k].data_ptr() for k in ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc"))
        d["_ik"] = tuple(d[k].data_ptr() for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state"))
    optr = [{k: v.data_ptr() for k, v in o.items()} for o in outs]
    sptr = [(a.cuda_stream, b.cuda_stream) for a, b in pipes]

    def launch_mpc(d, on=None, k=0):
        o, m = optr[k], d["_mpc"]
        mpc.solve_device(B, m[0], m[1], N1, m[2], m[3], m[4], m[5], o["u0"], o["mstat"], o["mact"], o["mmar"], sptr[k][1] if on is None else on)

    def launch_ik(d, solver=None, k=0):
        o, m = optr[k], d["_ik"]
        (solver or ik).solve_device(B, m[0], m[1], m[2], m[3], m[4], m[5], o["dq"], o["istat"], o["ilo"], o["iup"], 0, o["iit"], sptr[k][0])

    # optional RCCL exchange (rank 0 owns the whole batch)
    exch = None
    if args.exchange and dist is not None:
        in_keys = ("x0", "ref", "u_prev", "hull_A", "hull_b", "J_left", "J_right", "J_neck", "J_com", "q", "state")
        d0 = sets[0]
        if rank == 0:
            full = {k: [torch.empty_like(d0[k]) for _ in range(world)] for k in in_keys}
            for k in in_keys:
                for r in range(world):
                    full[k][r].copy_(d0[k])           # shape-true stand-ins: only the traffic matters here
            gat_u0 = [torch.empty_like(u0) for _ in range(world)]
            gat_dq = [torch.empty_like(dq) for _ in range(world)]
        else:
            full, gat_u0, gat_dq = None, None, None

        def exch_in():
            for k in in_keys:
                dist.scatter(d0[k], full[k] if rank == 0 else None, src=0)

        def exch_out():
            dist.gather(u0, gat_u0 if rank == 0 else None, dst=0)
            dist.gather(dq, gat_dq if rank == 0 else None, dst=0)
        exch = (exch_in, exch_out)

    def step(i):
        d = sets[0] if exch else sets[i % K]
        if exch:
            exch[0]()
            stream_mpc.wait_stream(stream)          # the MPC stream starts behind the scatter ...
        launch_mpc(d, k=i % P)
        launch_ik(d, k=i % P)
        if exch:
            stream.wait_stream(stream_mpc)          # ... and the gather behind both solves: the two batches still overlap
            exch[1]()

    def barrier():
        """Every stream the steps use has finished (an event at its tail, polled, then hipEventSynchronize'd), all ranks
        have, and the device is idle.  Returns the host time at which this rank's work was COMPLETE - the end of a timed
        region - and the time after the device-wide synchronize that follows: on this ROCm stack the first
        hipDeviceSynchronize after a burst of launches costs the host 55-75 us with the device already idle
        (tools/sync_cost.py; it is 4 us on a quiet process), which is 20 % of a 20-step region and no part of the steps."""
        evs = barrier.events                            # created once: the timed region pays for the records only
        for e, st_ in zip(evs, all_streams):
            e.record(st_)
        while not all(e.query() for e in evs):          # polled: a blocking wait wakes up ~50 us late
            pass
        for e in evs:
            e.synchronize()
        t_done = time.perf_counter()
        torch.cuda.synchronize(dev)
        t_sync = time.perf_counter()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        return t_done, t_sync

    barrier.events = [torch.cuda.Event() for _ in all_streams]
    # the inputs and the zero-filled outputs were enqueued on the default stream: the MPC stream starts behind them
    for st_ in all_streams:
        if st_ is not stream:
            st_.wait_stream(stream)
    # set-up, not warm-up: every input set is read once, so that none of them is touched for the first time (page-table
    # walks of a fresh allocation) inside a short timed region; the W warm-up steps follow
    if not exch:
        for i in range(K):
            step(i)
        barrier()
    def fill_record(r, i):
        """Argument record of step i: input set i % K, output buffers i % P."""
        d, k = sets[i % K], i % P
        o, m, q_ = optr[k], d["_mpc"], d["_ik"]
        r.x0, r.ref, r.ref_len, r.u_prev, r.hull_A, r.hull_b, r.hull_nc = m[0], m[1], N1, m[2], m[3], m[4], m[5]
        r.u0, r.mpc_status, r.mpc_active, r.mpc_margin, r.mpc_stream = o["u0"], o["mstat"], o["mact"], o["mmar"], sptr[k][1] or None
        r.J_left, r.J_right, r.J_neck, r.J_com, r.q, r.state = q_
        r.dq, r.ik_status, r.active_lower, r.active_upper, r.foot_err, r.iters = o["dq"], o["istat"], o["ilo"], o["iup"], None, o["iit"]
        r.ik_stream = sptr[k][0] or None

    # the W warm-up steps, in the form the timed steps take: in plan mode as a plan of their own (one launch of qp_plan_kernel -
    # otherwise the timed launch would be the first launch of that kernel in the process)
    if use_plan and not exch and args.warmup > 0 and (not args.plan_queue or args.warmup <= P):
        wrecs = (wca.capi.QpStep * args.warmup)()
        for i in range(args.warmup):
            fill_record(wrecs[i], i)
        wplan = wca.capi.QpPlan(mpc, ik, B, wrecs, ways=plan_ways)
        wplan.enqueue(sp)
        barrier()
        wplan.close()
    else:
        for i in range(args.warmup):
            step(i)
    barrier()
    # the argument records of the timed steps (set-up): step i = the MPC and the IK call of step(i), handed to the
    # library in ONE host call (wcqp_qp_enqueue_steps) - through ctypes a launch costs the host ~4.4 us, two kernels of 5
    # and 15 us per step leave the card waiting for the host otherwise
    recs = None
    if not exch:
        recs = (wca.capi.QpStep * args.steps)()
        for n in range(args.steps):
            fill_record(recs[n], args.warmup + n)
    # --step-graph: the K timed steps as ONE hipGraph - P parallel chains of one-launch steps (pipeline p's steps in order on
    # its stream, the chains forked from and joined into the capture stream), captured from the very wcqp_qp_enqueue_steps
    # call the launch-by-launch form makes; set-up: capture, instantiate, one replay (the first launch of a graph uploads
    # it).  A/B'd in round 3 and left OFF: the graph's dependent kernel nodes are dispatched no faster than the streams do it.
    plan = wca.capi.QpPlan(mpc, ik, B, recs, ways=plan_ways) if (use_plan and recs is not None) else None
    step_graph = None
    if recs is not None and args.step_graph:
        try:
            side = [st_ for st_ in all_streams if st_ is not stream]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                for st_ in side:
                    st_.wait_stream(stream)
                wca.capi.qp_enqueue_steps(mpc, ik, B, recs)
                for st_ in side:
                    stream.wait_stream(st_)
            g.replay()
            barrier()
            step_graph = g
        except Exception as e:                      # no graph: the launch-by-launch form still measures the same steps
            print("bench.py: step graph not available (%r): enqueueing launch by launch" % (e,), file=sys.stderr)
            torch.cuda.synchronize(dev)
    # ---- the timed region: K steps, nothing but the launches (no events, no host reads) ----------------------------
    t0 = time.perf_counter()
    if plan is not None:
        plan.enqueue(sp)
    elif step_graph is not None:
        step_graph.replay()
    elif recs is not None:
        wca.capi.qp_enqueue_steps(mpc, ik, B, recs)
    else:
        for i in range(args.steps):
            step(args.warmup + i)
    t_enq = time.perf_counter() - t0
    t_done, t_sync = barrier()
    elapsed = t_done - t0
    elapsed_sync = max_over_ranks(dist, torch, dev, t_sync - t0)
    elapsed = max_over_ranks(dist, torch, dev, elapsed)
    for st_ in all_streams:
        if st_ is not stream:
            stream.wait_stream(st_)

    # sanity: the timed work really solved the problems (every pipeline's last batch)
    n_ok_ik = min(int((o["istat"] == 0).sum().item()) for o in outs)
    n_ok_mpc = min(int((o["mstat"] == 0).sum().item()) for o in outs)
    ik_iters = float(outs[0]["iit"].double().mean().item())
    frac_active = float(((outs[0]["ilo"] | outs[0]["iup"]) != 0).double().mean().item())

    # what was timed is what the committed golden vectors hold (tests/golden/: the exact fp64 optimum of the same seeded
    # instances): every pipeline's LAST batch of the timed region against them.  Input set k is the batch rolled by k B / K
    # rows, so output row r of a step that read set k belongs to instance (r - k B / K) mod B.
    golden = {"golden_max_abs_err": None, "golden_active_set_mismatches": None}
    if first == 0 and NH == 50 and args.ik_form == "qpoases" and abs(args.ik_vmax - 0.5) < 1e-12 and not exch:
        try:
            gm = np.load(os.path.join(ROOT, "tests", "golden", "mpc_cfg2_b4096.npz"), allow_pickle=False)
            gi = np.load(os.path.join(ROOT, "tests", "golden", "ik_qpoases_v050_b4096.npz"), allow_pickle=False)
            err, mism, rows_checked = 0.0, 0, 0
            last = args.warmup + args.steps - 1
            for p_ in range(P):
                i_last = last - ((last - p_) % P)
                if i_last < args.warmup:
                    continue
                inst = (np.arange(B) - (i_last % K) * max(1, B // K)) % B
                o = {k: v.cpu().numpy() for k, v in outs[p_].items()}
                m = inst < int(gm["count"])
                err = max(err, float(np.abs(o["u0"][m] - gm["u0"][inst[m]]).max()))
                sure = (gm["mu_min_active"][inst[m]] > 1e-7) & (gm["slack_min_inactive"][inst[m]] > 1e-7)
                mism += int((o["mact"][m].astype(np.uint32)[sure] != gm["active"][inst[m]][sure]).sum())
                mism += int((o["mstat"][m] != 0).sum())
                n = inst < int(gi["count"])
                err = max(err, float(np.abs(o["dq"][n] - gi["dq"][inst[n]]).max()))
                sure = (gi["mu_min_active"][inst[n]] > 1e-7) & (gi["slack_min_inactive"][inst[n]] > 1e-7) & (gi["status"][inst[n]] == 0)
                mism += int(((o["ilo"][n].astype(np.uint32) != gi["active_lower"][inst[n]]) | (o["iup"][n].astype(np.uint32) != gi["active_upper"][inst[n]]))[sure].sum())
                mism += int((o["istat"][n] != gi["status"][inst[n]]).sum())
                rows_checked += int(m.sum()) + int(n.sum())
            golden = {"golden_max_abs_err": err, "golden_active_set_mismatches": mism, "golden_rows_checked": rows_checked,
                      "golden": "every pipeline's last timed batch vs tests/golden/mpc_cfg2_b4096.npz (u0, active rows, status) and "
                                "ik_qpoases_v050_b4096.npz (dq, active bounds, status); active sets where the strict-complementarity margin exceeds 1e-7"}
        except Exception as e:                      # a bench line without the check is still a bench line; say why
            golden["golden_error"] = repr(e)

    # ---- kernel durations, measured AFTER the timed region in short passes of their own: HIP events on the launch
    # stream around n back-to-back launches of one kernel (a pair of event records costs about as much as a launch, so
    # nothing is bracketed singly); `cold` rotates over the K input sets like the timed region, `resident` re-reads one
    def kernel_ms(launch, on, cold, n=None):
        n = n or max(24, 2 * K)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(K):
            launch(sets[i % K if cold else 0])
        torch.cuda.synchronize(dev)
        e0.record(on)
        for i in range(n):
            launch(sets[i % K if cold else 0])
        e1.record(on)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n
    # the IK kernel alone: a MIXED-structure handle launches exactly the one kernel (AUTO adds the nearly empty
    # fall-back launch behind it, timed separately below)
    ik_one = ik if args.ik_jac != "auto" else wca.IkSolver(form=ik_form, v_max=args.ik_vmax, jacobian_structure=wca.IK_JAC_MIXED)
    ik_ms = kernel_ms(lambda d: launch_ik(d, ik_one), pipes[0][0], True)
    ik_ms_res = kernel_ms(lambda d: launch_ik(d, ik_one), pipes[0][0], False)
    mpc_ms = kernel_ms(lambda d: launch_mpc(d, sp), stream, True)
    mpc_ms_res = kernel_ms(lambda d: launch_mpc(d, sp), stream, False)
    # BASELINE config 2 on its own (batched DCM-MPC): the MPC-only plan - one launch walks through the batches, 16-17 wavefronts per
    # robot group - over MPC input sets of its own (> 1 GiB of them: the K sets above hold far less MPC data than the Infinity Cache)
    mpc_plan_ms = None
    if not exch:
        try:
            m_bytes = B * (mpc_bytes - 16)
            KM = int(max(13, min(512, -(-(1 << 30) // m_bytes))))
            wm = 16
            while math.gcd(KM, wm) != 1:
                wm += 1
            mkeys = ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc")
            msets = [{k: base[k] for k in mkeys}] + [{k: torch.roll(base[k], shifts=j * max(1, B // KM), dims=0).contiguous() for k in mkeys} for j in range(1, KM)]
            mouts = [dict(u0=torch.zeros(B, 2, dtype=torch.float64, device=dev), mstat=torch.zeros(B, dtype=torch.int32, device=dev),
                          mact=torch.zeros(B, dtype=torch.int32, device=dev), mmar=torch.zeros(B, dtype=torch.float64, device=dev)) for _ in range(wm)]
            nm = max(args.steps, 200)
            mrecs = (wca.capi.QpStep * nm)()
            for t_ in range(nm):
                d_, o_, r = msets[t_ % KM], mouts[t_ % wm], mrecs[t_]
                r.x0, r.ref, r.ref_len, r.u_prev, r.hull_A, r.hull_b, r.hull_nc = (d_["x0"].data_ptr(), d_["ref"].data_ptr(), N1, d_["u_prev"].data_ptr(),
                                                                                     d_["hull_A"].data_ptr(), d_["hull_b"].data_ptr(), d_["hull_nc"].data_ptr())
                r.u0, r.mpc_status, r.mpc_active, r.mpc_margin = o_["u0"].data_ptr(), o_["mstat"].data_ptr(), o_["mact"].data_ptr(), o_["mmar"].data_ptr()
            mpl = wca.capi.QpPlan(mpc, None, B, mrecs, ways=wm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            mpl.enqueue(sp); mpl.enqueue(sp)
            torch.cuda.synchronize(dev)
            e0.record(stream)
            for _ in range(5):
                mpl.enqueue(sp)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            mpc_plan_ms = e0.elapsed_time(e1) / 5 / nm
            mpc_plan_ok = min(int(((o_["mstat"] == 0) | (o_["mstat"] == 3)).sum().item()) for o_ in mouts)
            mpl.close()
            del msets, mouts
        except Exception as e:
            print("bench.py: MPC-only plan pass skipped (%r)" % (e,), file=sys.stderr)
    # BASELINE config 3 on its own (batched QP-IK): the IK-only plan - qp_plan_kernel without its MPC share - over the K cold input sets
    ik_plan_ms = None
    if not exch and args.ik_jac == "mixed":
        try:
            wi = 5
            while math.gcd(K, wi) != 1:
                wi += 1
            iouts = [outputs() for _ in range(wi)]
            ni = max(args.steps, 100)
            irecs = (wca.capi.QpStep * ni)()
            for t_ in range(ni):
                q_, o_, r = sets[t_ % K]["_ik"], iouts[t_ % wi], irecs[t_]
                r.J_left, r.J_right, r.J_neck, r.J_com, r.q, r.state = q_
                r.dq, r.ik_status, r.active_lower, r.active_upper, r.foot_err, r.iters = (o_["dq"].data_ptr(), o_["istat"].data_ptr(), o_["ilo"].data_ptr(),
                                                                                           o_["iup"].data_ptr(), None, o_["iit"].data_ptr())
            ipl = wca.capi.QpPlan(None, ik, B, irecs, ways=wi)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ipl.enqueue(sp); ipl.enqueue(sp)
            torch.cuda.synchronize(dev)
            e0.record(stream)
            for _ in range(5):
                ipl.enqueue(sp)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            ik_plan_ms = e0.elapsed_time(e1) / 5 / ni
            ik_plan_ok = min(int((o_["istat"] == 0).sum().item()) for o_ in iouts)
            ipl.close()
            del iouts
        except Exception as e:
            print("bench.py: IK-only plan pass skipped (%r)" % (e,), file=sys.stderr)
    ik_auto = wca.IkSolver(form=ik_form, v_max=args.ik_vmax, jacobian_structure=wca.IK_JAC_AUTO)
    ik_auto_ms = kernel_ms(lambda d: launch_ik(d, ik_auto), pipes[0][0], True)

    # the launch the timed region is made of when a step's two calls share a stream: IK and MPC workgroups in one grid
    # the launch the timed region is made of in plan mode: qp_plan_kernel over n records (cold: rotating over the K input sets;
    # resident: one set), timed alone with events; per-step time = launch / n
    plan_ms = plan_ms_res = None
    if plan is not None:
        def plan_pass(cold, n):
            r2 = (wca.capi.QpStep * n)()
            for t_ in range(n):
                d, k = sets[t_ % K if cold else 0], t_ % P
                o, m, q_ = optr[k], d["_mpc"], d["_ik"]
                r = r2[t_]
                r.x0, r.ref, r.ref_len, r.u_prev, r.hull_A, r.hull_b, r.hull_nc = m[0], m[1], N1, m[2], m[3], m[4], m[5]
                r.u0, r.mpc_status, r.mpc_active, r.mpc_margin = o["u0"], o["mstat"], o["mact"], o["mmar"]
                r.J_left, r.J_right, r.J_neck, r.J_com, r.q, r.state = q_
                r.dq, r.ik_status, r.active_lower, r.active_upper, r.foot_err, r.iters = o["dq"], o["istat"], o["ilo"], o["iup"], None, o["iit"]
            pl = wca.capi.QpPlan(mpc, ik, B, r2, ways=plan_ways)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(2, min(10, 2000 // max(1, n)))          # several launches in one bracket: the host work since the timed region let the clocks drop
            pl.enqueue(sp); pl.enqueue(sp)
            torch.cuda.synchronize(dev)
            e0.record(stream)
            for _ in range(reps):
                pl.enqueue(sp)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / reps
            pl.close()
            return ms / n
        n_plan = args.steps                       # the timed launch's own length: rocprofv3's average over the run's launches of this kernel is then THIS number
        plan_ms = plan_pass(True, n_plan)
        plan_ms_res = plan_pass(False, n_plan) if args.resident_pass else None
    pair_mode = plan is None and recs is not None and not two_streams and args.ik_jac != "general"
    pair_ms = pair_ms_res = None
    if pair_mode:
        one = {}
        for d in sets:
            r1 = (wca.capi.QpStep * 1)()
            r, o, m, q_ = r1[0], optr[0], d["_mpc"], d["_ik"]
            r.x0, r.ref, r.ref_len, r.u_prev, r.hull_A, r.hull_b, r.hull_nc = m[0], m[1], N1, m[2], m[3], m[4], m[5]
            r.u0, r.mpc_status, r.mpc_active, r.mpc_margin, r.mpc_stream = o["u0"], o["mstat"], o["mact"], o["mmar"], sptr[0][0] or None
            r.J_left, r.J_right, r.J_neck, r.J_com, r.q, r.state = q_
            r.dq, r.ik_status, r.active_lower, r.active_upper, r.foot_err, r.iters = o["dq"], o["istat"], o["ilo"], o["iup"], None, o["iit"]
            r.ik_stream = sptr[0][0] or None
            one[id(d)] = r1
        pair_ms = kernel_ms(lambda d: wca.capi.qp_enqueue_steps(mpc, ik_one, B, one[id(d)]), pipes[0][0], True)
        pair_ms_res = kernel_ms(lambda d: wca.capi.qp_enqueue_steps(mpc, ik_one, B, one[id(d)]), pipes[0][0], False)

    total_qp = 2 * B * world * args.steps
    value = total_qp / elapsed
    ik_kernel = {"mixed": "ik4_kernel", "auto": "ik4_kernel", "general": "ik3_kernel"}[args.ik_jac]
    ik_gbs = IK_BYTES_PER_QP * B / (ik_ms * 1e-3) / 1e9
    out = {
        "metric": METRIC, "value": value, "unit": "QP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[1]+[2] per GPU: DCM-MPC QP (N=%d, n=%d, cold start) B=%d + QP-IK "
                         "(iCub 23 DoF, 15 eq rows, %s form, v_max=%.2f rad/s) B=%d; 2 QP solves per robot-tick"
                         % (NH, 4 * NH + 2, B, args.ik_form, args.ik_vmax, B)),
            "batch_per_gpu": B, "global_batch": B * world, "horizon": NH, "dof": 23,
            "input_sets": K, "input_bytes_per_set": set_bytes, "ik_jacobian_structure": args.ik_jac, "pipelines": P,
            "host_enqueue_us_per_step": 1e6 * t_enq / args.steps,
            "timed_steps_enqueued_as": (("ONE launch (wcqp_qp_plan_enqueue): qp_plan_kernel, as many wavefronts as are resident at once take (step, robot group) units of the %d steps from a work queue" % args.steps) if args.plan_queue else
                                        ("ONE launch (wcqp_qp_plan_enqueue): qp_plan_kernel walks through the %d steps, %d wavefronts per robot group" % (args.steps, P))) if plan is not None
                                       else (("one hipGraph launch: %d parallel chains of one-launch steps" % P) if step_graph is not None else "one wcqp_qp_enqueue_steps call, launch by launch"),
            "timed_region": "barrier + torch.cuda.synchronize() -> K steps -> completion events of every stream used (hipEventSynchronize), "
                            "MAX over ranks; the device-wide synchronize that follows adds %.0f us of host time with the device idle "
                            "(ms_per_step_incl_device_sync)" % (1e6 * (elapsed_sync - elapsed)),
            "ms_per_step_incl_device_sync": 1e3 * elapsed_sync / args.steps, "value_incl_device_sync": total_qp / elapsed_sync,
            "parallelism": "batch sharded over %d GPU(s), no data-path collective%s%s" % (world, " + RCCL scatter/gather" if exch else "", "; MPC and IK batches on two HIP streams" if two_streams else ("; MPC and IK of a step in one launch" if not exch else "")) + ((("; work queue over (step, robot group) units" if args.plan_queue else "; %d wavefronts per robot group (step i -> way i %% %d)" % (P, P)) if plan is not None else "; %d independent batches in flight (step i -> pipeline i %% %d)" % (P, P)) if P > 1 else ""),
        },
        # the dominant kernel of the timed region: the one-launch step (IK + MPC workgroups; algorithmic bytes = both QPs'
        # per robot-tick) - or the IK kernel where the two calls are separate launches
        "roofline": ({
            # the dominant kernel IS the timed region: one launch that walks through the steps.  achieved = algorithmic bytes of
            # a step (6296 B per robot-tick) x steps per launch / launch duration (events, a pass of its own over
            # `steps` records like the timed launch: rocprofv3's average duration of qp_plan_kernel for this command is avg_launch_ms)
            "bound": "hbm", "kernel": "qp_plan_kernel",
            "achieved": (IK_BYTES_PER_QP + mpc_bytes) * B / (plan_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (IK_BYTES_PER_QP + mpc_bytes) * B / (plan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "avg_launch_ms": plan_ms * n_plan, "steps_per_launch": n_plan, "avg_ms_per_step": plan_ms,
            "algorithmic_bytes_per_launch": (IK_BYTES_PER_QP + mpc_bytes) * B * n_plan,
            "inputs": ("cold: %d input sets of %.1f MB (%.2f GB in total), " % (K, set_bytes / 1e6, K * set_bytes / 1e9)) +
                      ("every step of the run reads input arrays of its own" if K >= args.steps + args.warmup else
                       "visited round-robin; a wavefront returns to a set after %d of its records" % (K // math.gcd(K, max(1, plan_ways if use_plan else 1)))),
            "frac_resident_inputs": ((IK_BYTES_PER_QP + mpc_bytes) * B / (plan_ms_res * 1e-3) / 1e9 / HBM_PEAK_GBS) if plan_ms_res else None, "avg_ms_per_step_resident_inputs": plan_ms_res,
            "timed_region": {"achieved": (IK_BYTES_PER_QP + mpc_bytes) * B * args.steps / elapsed / 1e9,
                             "frac": (IK_BYTES_PER_QP + mpc_bytes) * B * args.steps / elapsed / 1e9 / HBM_PEAK_GBS},
        } if plan is not None else {
            "bound": "hbm", "kernel": "qp_pair_kernel",
            "achieved": (IK_BYTES_PER_QP + mpc_bytes) * B / (pair_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (IK_BYTES_PER_QP + mpc_bytes) * B / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": None, "avg_launch_ms": pair_ms, "algorithmic_bytes_per_launch": (IK_BYTES_PER_QP + mpc_bytes) * B,
            "inputs": ("cold: %d input sets of %.1f MB (%.2f GB in total), " % (K, set_bytes / 1e6, K * set_bytes / 1e9)) +
                      ("every step of the run reads input arrays of its own" if K >= args.steps + args.warmup else
                       "visited round-robin; a wavefront returns to a set after %d of its records" % (K // math.gcd(K, max(1, plan_ways if use_plan else 1)))),
            "frac_resident_inputs": (IK_BYTES_PER_QP + mpc_bytes) * B / (pair_ms_res * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms_resident_inputs": pair_ms_res,
            # `frac` prices ONE launch running alone (back-to-back launches of the kernel on one stream: what `rocprofv3 --stats`
            # reports for a --pipelines 1 run); with P batches in flight the launches overlap, each takes longer, and the
            # card as a whole moves this many algorithmic bytes per second through the timed region:
            "timed_region": {"batches_in_flight": P, "achieved": (IK_BYTES_PER_QP + mpc_bytes) * B * args.steps / elapsed / 1e9,
                             "frac": (IK_BYTES_PER_QP + mpc_bytes) * B * args.steps / elapsed / 1e9 / HBM_PEAK_GBS},
        } if pair_mode else {
            "bound": "hbm", "kernel": ik_kernel,
            "achieved": ik_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ik_gbs / HBM_PEAK_GBS,
            "traffic": None, "avg_launch_ms": ik_ms, "algorithmic_bytes_per_launch": IK_BYTES_PER_QP * B,
            "inputs": ("cold: %d input sets of %.1f MB (%.2f GB in total), " % (K, set_bytes / 1e6, K * set_bytes / 1e9)) +
                      ("every step of the run reads input arrays of its own" if K >= args.steps + args.warmup else
                       "visited round-robin; a wavefront returns to a set after %d of its records" % (K // math.gcd(K, max(1, plan_ways if use_plan else 1)))),
            "frac_resident_inputs": IK_BYTES_PER_QP * B / (ik_ms_res * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms_resident_inputs": ik_ms_res,
        }),
        "kernels": {
            "ik_kernel": ik_kernel, "ik_ms": ik_ms, "ik_ms_resident_inputs": ik_ms_res, "ik_qps_per_gpu": B / (ik_ms * 1e-3),
            "ik_hbm_frac": ik_gbs / HBM_PEAK_GBS, "ik_hbm_frac_resident_inputs": IK_BYTES_PER_QP * B / (ik_ms_res * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # secondary bounds.  fp64 vector: ~8.5 kflop per IK-QP with the base-eliminated kernel (row
            # operations 1.6 k, sweep 2.5 k, x / bounds ~1.4 k, rhs / rot errors ~0.5 k + the Gram tile below) against
            # 78.6 TFLOP/s; fp64 MFMA: 6 v_mfma_f64_16x16x4 per IK-QP = 12.3 kflop issued (the Gram product C C', 24 x 13 x 13
            # useful) against the 78.6 TFLOP/s fp64 matrix peak
            "ik_fp64_valu_frac": 6.0e3 * B / (ik_ms * 1e-3) / 78.6e12,
            "ik_mfma_f64_tflops": 12288.0 * B / (ik_ms * 1e-3) / 1e12, "ik_mfma_f64_frac": 12288.0 * B / (ik_ms * 1e-3) / 78.6e12,
            "ik_auto_ms": ik_auto_ms, "ik_auto_fallback_launch_ms": ik_auto_ms - ik_ms,
            "mpc_ms": mpc_ms, "mpc_ms_resident_inputs": mpc_ms_res, "mpc_qps_per_gpu": B / (mpc_ms * 1e-3),
            "mpc_hbm_frac": mpc_bytes * B / (mpc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "mpc_bytes_per_qp": mpc_bytes,
            # BASELINE config 2 alone, as ONE launch over many batches (wcqp_qp_plan_* with MPC-only records, mpc_plan_kernel)
            "mpc_plan_ms_per_batch": mpc_plan_ms, "mpc_plan_qps_per_gpu": (B / (mpc_plan_ms * 1e-3)) if mpc_plan_ms else None,
            "mpc_plan_hbm_frac": (mpc_bytes * B / (mpc_plan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if mpc_plan_ms else None,
            "mpc_plan_solved": (mpc_plan_ok if mpc_plan_ms else None),
            # BASELINE config 3 alone, as ONE launch over many batches (wcqp_qp_plan_* with IK-only records: qp_plan_kernel without its MPC share)
            "ik_plan_ms_per_batch": ik_plan_ms, "ik_plan_qps_per_gpu": (B / (ik_plan_ms * 1e-3)) if ik_plan_ms else None,
            "ik_plan_hbm_frac": (IK_BYTES_PER_QP * B / (ik_plan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ik_plan_ms else None,
            "ik_plan_solved": (ik_plan_ok if ik_plan_ms else None),
        },
        "solved": dict({"ik": n_ok_ik, "mpc": n_ok_mpc, "of": B, "ik_mean_active_set_changes": ik_iters,
                        "ik_frac_with_active_bounds": frac_active}, **golden),
    }
    # HBM bytes per launch from the PMC passes of tools/pmc/collect.sh (committed summary)
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            tj = json.load(open(traffic_file))
            tr = tj.get("per_batch", {}).get(str(B))
            # only while the kernels are the ones the PMC passes ran on (tools/pmc/make_traffic.py stamps the sources' hash)
            if tj.get("csrc_sha256") != wca.capi.source_hash():
                out["roofline"]["traffic_note"] = "profiles/traffic.json was measured on other kernel sources (re-run tools/pmc/collect.sh + make_traffic.py): not quoted"
            elif tr and NH == 50:                                                   # measured on the BASELINE horizon
                out["roofline"]["traffic"] = tr.get("plan_hbm_bytes_per_step") if plan is not None else tr.get("pair_hbm_bytes_per_launch" if pair_mode else "ik_hbm_bytes_per_launch")
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(mb, ib, args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def spawn_ranks(n):
    """python bench.py --gpus N: start N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* per child, rendezvous on
    127.0.0.1), relay rank 0's JSON line, exit non-zero when any rank does.  The parent makes no GPU call at all."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %s" % rcs)


def max_over_ranks(dist, torch, dev, value):
    if dist is None:
        return value
    on = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, torch, dev, value):
    if dist is None:
        return value
    on = dev if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=on)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
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
    """BASELINE configs[3]/[4]: every step is one robot-tick of the whole batch - per-tick kinematics at the integrated
    joint state (Jacobians, actual poses, support polygon on a contact change), then MPC on the receding window of the
    per-instance DCM trajectory, ZMP-CoM glue, IK and joint integration in ONE fused launch - with all solver and
    plant state resident in HBM and the launches replayed from a hipGraph.  `--tick-tables`: the round-1 form
    (constant uploaded Jacobians, precomputed hull tables; one launch per tick)."""
    T = args.steps + args.warmup
    kin_mode = not args.tick_tables
    # the walking robots carry per-joint velocity limits (synth.WALK_VMAX: legs 1.5, torso and arms 0.3 rad/s - DESIGN.md 8.2)
    vmax = args.ik_vmax if args.tick_tables else wca.synth.WALK_VMAX
    # Robot groups, each a pipeline on a HIP stream of its own (synthetic robots are counter-based, so the groups are exactly
    # the rows of the full batch).  When a tick is ONE launch of the fused kernel (constant Jacobians, or kinematics fused
    # in) the library runs all the ticks of a run() call in one launch and one group is best (three concurrent launches of a
    # third of the robots each: 5.2e8 against 7.7e8 QP/s with kinematics at 8192 robots).  With a kinematics launch per tick
    # (--tick-kin-handoff dense / compact) or one tick per launch, groups overlap one group's kernel with another's.
    multi_tick = args.ticks_per_launch != 1 and (not kin_mode or args.tick_kin_handoff == "fused")
    n_streams = args.streams if args.streams else (1 if multi_tick else ((3 if (kin_mode and B <= 16384) else 2) if B >= 8192 else 1))
    cuts = [B * k // n_streams for k in range(n_streams + 1)]
    parts = [(first + cuts[k], cuts[k + 1] - cuts[k]) for k in range(n_streams)]
    ik_form = wca.IK_FORM_QPOASES if args.ik_form == "qpoases" else wca.IK_FORM_OSQP
    pipes = []
    kin = wca.KinModel(wca.synth.icub_like_model()) if kin_mode else None
    for f0, cnt in parts:
        if kin_mode:
            kb = wca.synth.synth_walk_kin_batch(cnt, first=f0)
            poses = kin.jacobians_host(kb["base"], kb["q"], state=np.zeros((cnt, 87)))["state"]
            data = wca.synth.synth_walk_batch(cnt, T, poses, kb, first=f0)
            iks = wca.IkSolver(form=ik_form, v_max=vmax, joint_reg_rad=np.deg2rad(wca.synth.WALK_POSTURE_DEG))
        else:
            data = wca.synth.synth_tick_batch(cnt, T, first=f0)
            iks = wca.IkSolver(form=ik_form, v_max=vmax)
        pp = wca.TickPipeline(cnt, T, wca.MpcSolver(horizon=50), iks, first=f0, kin=kin, ik_hot_start=not args.tick_cold_ik,
                              kin_handoff={"fused": 0, "dense": 1, "compact": 2}[args.tick_kin_handoff], ticks_per_launch=args.ticks_per_launch)
        pp.upload(data)
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
                 "ik_fail": np.concatenate([x["ik_fail"] for x in states]),
                 "hot_try": int(sum(x["hot_try"].sum() for x in states)), "hot_hit": int(sum(x["hot_hit"].sum() for x in states))}
    dev_ms = e0.elapsed_time(e1) / args.steps
    # QP solves = 2 per robot-tick, minus the robot-ticks of STOPPED robots (a robot whose IK failed keeps dq = 0 and its
    # active-set walk is skipped from then on: ik_fail counts the failing tick and every tick after it; the failing
    # tick itself did walk, so it stays counted).  Ranks > 0 report through the same all-reduce as the time.
    stopped_ticks = int(np.maximum(out_state["ik_fail"] - 1, 0).sum())
    stopped_ticks_all = int(round(sum_over_ranks(dist, torch, dev, float(stopped_ticks))))
    value = (2 * B * world * args.steps - stopped_ticks_all) / elapsed
    # Roofline of the tick, two denominators:
    #  `frac`      SURVEY.md 8d's algorithmic bytes of a robot-tick, 6296 B (one cold-start MPC + one IK with four dense Jacobians
    #              in, u0 + dq out) - what any implementation of the two QPs has to be given and has to return;
    #  `frac_own`  the bytes THIS pipeline moves through HBM per robot-tick: the IK's pose block and joint state, the MPC's
    #              window (one new stage per tick, the rest re-read from L2), resident controller / plant state read + written,
    #              logs; with constant Jacobians the 4464 B of Jacobians every tick; with a kinematics launch per tick its
    #              hand-off written and read (dense 4464 + 4464, compact 1440 + 1440); with fused kinematics no Jacobian bytes at all.
    state_rw = 2 * 8 * (16 + 10 + 23 * 2) + 87 * 8 + 23 * 8 + 16 + 8       # mst + hand + q_des/dq_prev r/w, pose block, dq out, one reference stage, words
    jac_bytes = 4464 if not kin_mode else {"fused": 0, "compact": 2 * 1440 + 288, "dense": 2 * 4464 + 288}[args.tick_kin_handoff]
    own_bytes = state_rw + jac_bytes
    bytes_per_tick = 6296
    if kin_mode:
        launches = {"fused": "1 launch per run() call: ik4_kernel<TICK, fused kinematics> walks through the ticks (kinematics, MPC of the next tick, glue, IK, post step)",
                    "compact": "2 launches per tick: kin_jacobians_kernel<TICK, compact>, ik4_kernel<TICK>",
                    "dense": "2 launches per tick: kin_jacobians_kernel<TICK>, ik4_kernel<TICK>"}[args.tick_kin_handoff]
    else:
        launches = "1 launch per run() call: ik4_kernel<TICK> walks through the ticks (MPC of the next tick, glue, IK, post step)"
    if args.ticks_per_launch == 1:
        launches = launches.replace("1 launch per run() call", "1 launch per tick (hipGraph of 8)").replace("walks through the ticks", "")
    out = {
        "metric": METRIC, "value": value, "unit": "QP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[3]/[4] per GPU: receding-horizon robot-tick (%sDCM-MPC N=50 on the advancing "
                         "reference window -> ZMP-CoM glue -> QP-IK 23 DoF %s form v_max=%s -> joint integration), "
                         "B=%d robots, %s, contact pair changes every 70-110 ticks; 2 QP solves per robot-tick"
                         % ("forward kinematics + Jacobians + support polygon at the integrated joint state -> " if kin_mode else "constant Jacobians, ",
                            args.ik_form, ("%.2f" % vmax) if args.tick_tables else "legs 1.5 / upper body 0.3", B, "hipGraph replay" if graph else "plain launches")),
            "batch_per_gpu": B, "global_batch": B * world, "horizon": 50, "dof": 23, "ticks": args.steps, "per_tick_kinematics": kin_mode,
            "parallelism": "batch sharded over %d GPU(s), no data-path collective%s" % (world, ("; %d robot groups on %d HIP streams" % (len(pipes), len(pipes))) if len(pipes) > 1 else ""),
        },
        "roofline": {"bound": "hbm", "kernel": "whole tick (%s)" % launches, "achieved": bytes_per_tick * B / (dev_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_per_tick * B / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "avg_launch_ms": dev_ms, "algorithmic_bytes_per_launch": bytes_per_tick * B,
                     "algorithmic_bytes_basis": "SURVEY.md 8d: 6296 B per robot-tick (1056 MPC + 5240 IK), per TICK (a launch runs many)",
                     "own_hbm_bytes_per_robot_tick": own_bytes, "frac_own": own_bytes * B / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "inputs": ("constant Jacobians, %.0f MB for the batch, re-read every tick: %s" % (4464 * B / 1e6, "they fit the 256 MiB Infinity Cache - this line is priced "
                                "against the HBM peak but not fed from HBM" if 4464 * B < 200e6 else "more than the 256 MiB Infinity Cache holds: fed from HBM")) if not kin_mode
                               else ("device-resident robot state only (joint state, MPC chain records, reference window): no Jacobian bytes cross HBM" if args.tick_kin_handoff == "fused"
                                     else "device-resident robot state + the kinematics launch's Jacobian hand-off, written and read back every tick (%s)" % args.tick_kin_handoff)},
        "solved": {"ticks_executed": out_state["tick"], "mpc_fail": int(out_state["mpc_fail"].sum()),
                   "ik_fail": int(out_state["ik_fail"].sum()), "robots_with_ik_fail": int((out_state["ik_fail"] > 0).sum()), "of": B * T,
                   "stopped_robot_ticks_not_counted": stopped_ticks_all,
                   # IK hot start: robot-ticks on which the previous tick's active bounds were tried first / accepted
                   "ik_hot_start_tried": out_state["hot_try"], "ik_hot_start_accepted": out_state["hot_hit"],
                   "ik_hot_start_hit_rate": (out_state["hot_hit"] / out_state["hot_try"]) if out_state["hot_try"] else None},
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def wca_synth_mpc(count, horizon):
    import walking_controllers_amd as wca
    return wca.synth.synth_mpc_batch(count, seed=1234, horizon=horizon)


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
    # the MPC as the reference runs it on >= 97 % of its ticks (no contact change): ONE persistent OSQP workspace per
    # robot, bounds + gradient updated, warm-started solve (WM/src/MPCSolver.cpp:157-173, 249-258).  The number above
    # (`mpc_qps`) pays set-up, ordering, symbolic + numeric LDL' and scaling for every QP: it is the COLD number (what
    # the reference pays on a contact change, ...PredictiveController.cpp:415-420).
    warm_ticks = 40
    nw = min(n, 2048)
    wb = wca_synth_mpc(nw, 50 + warm_ticks)
    # wall time, like the cold numbers: the same call with 8 and with `warm_ticks` warm ticks - the difference is 32 warm
    # ticks of every robot without the cold solve and the set-up in front of them
    t = time.perf_counter(); co.mpc_batch_osqp_warm(mp, wb, 8, nthreads=cores); wall_8 = time.perf_counter() - t
    t = time.perf_counter()
    _, warm_iters, warm_thread_s, warm_fail = co.mpc_batch_osqp_warm(mp, wb, warm_ticks, nthreads=cores)
    wall_w = time.perf_counter() - t
    mpc_warm_qps = nw * (warm_ticks - 8) / (wall_w - wall_8) if wall_w > wall_8 else None
    return {"value": 2 * n * reps / (wall_m + wall_i), "unit": "QP/s", "cores": cores, "kind": "port",
            "mpc_warm_qps": mpc_warm_qps, "mpc_warm_mean_iters": warm_iters, "mpc_warm_nonconverged": warm_fail,
            "mpc_warm_sample": "%d robots x %d warm ticks each after one cold solve: persistent workspace, bounds + gradient update, "
                               "warm-started OSQP-restatement solve; WALL time of the call with %d warm ticks minus the one with 8 "
                               "(set-up and cold solve cancel)" % (nw, warm_ticks, warm_ticks),
            "sample": "%d x the first %d instances of the same workload (1 MPC via OSQP-restatement + 1 IK via %s per instance), "
                      "OpenMP static split" % (reps, n, "dense active set" if args.ik_form == "qpoases" else "OSQP-restatement"),
            "mpc_qps": n * reps / wall_m, "mpc_qps_is": "cold start: a new OSQP workspace per QP", "ik_qps": n * reps / wall_i,
            "single_thread_qps": {"mpc": 1.0 / t_m, "ik": 1.0 / t_i}}


if __name__ == "__main__":
    main()
