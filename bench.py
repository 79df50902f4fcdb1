#!/usr/bin/env python3
"""
bench.py — whole-job QP solves/s of the MI355X batched solve path (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic robot instances: one DCM-MPC QP (N = 50, BASELINE configs[1])
and one Jacobian QP-IK (iCub 23 DoF, BASELINE configs[2]) per instance, i.e. 2 QP solves per robot-tick, `--batch` instances per
GPU (default 4096, the batch both configs are quoted on), every instance cold-started, inputs resident in HBM.

    python bench.py                       # 1 GPU
    python bench.py --gpus N              # starts N ranks itself (one per GPU; refuses when fewer GPUs are visible)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

The K timed steps are independent batches (cold start: nothing is carried from one to the next), so they are handed to the
library once (`wcqp_qp_plan_create`) and enqueued as ONE launch (`wcqp_qp_plan_enqueue`) in which a wavefront owns four robots and
walks through the steps on its own, the MPC of a step on the IK's lanes under its Jacobian loads, `--plan-ways` wavefronts per robot
group.  `--plan-ways 0` is the launch-per-step form (`qp_pair_kernel`, step i on pipeline i % 3).  EVERY step of the run reads input
arrays of its own (cold: never out of the 256 MiB Infinity Cache).

Timed region: barrier + torch.cuda.synchronize() -> K steps -> completion event of every stream used (hipEventSynchronize), MAX
over ranks.  It is repeated `--repeats` (5) times, each time over K steps with inputs no earlier step has read; `value` is the
MEDIAN repeat (`value_min` / `value_max` beside it) and `ms_per_step` x `steps` is that repeat's region.  After the last repeat the
outputs of every way's last batch are compared with the committed golden vectors (`solved.golden_*`: all 4096 rows of both QPs).

The same line carries
  roofline      HBM roofline of the dominant kernel - the timed launch itself, `qp_plan_kernel`: algorithmic bytes per launch
                (6296 B per robot-tick = 5240 B/IK-QP + 1056 B/MPC-QP, SURVEY.md 8d, x batch x steps) / the duration of a launch over
                `steps` cold records (HIP events on the launch stream, a pass of its own after the timed region: rocprofv3's average
                duration of that kernel for the same command is `avg_launch_ms`); `traffic` = HBM bytes per step from the PMC
                passes (profiles/traffic.json, quoted only while the kernel sources are the ones it was measured on);
  tick          BASELINE configs[3]/[4] per GPU - the closed-loop receding-horizon tick, `--tick-batch` (8192) robots x `--tick-ticks`
                (1000) ticks in ONE wcqp_tick_run call, with per-tick kinematics fused into the solve kernel AND with constant
                Jacobians: QP/s, us per tick, roofline against SURVEY's 6296 B and against the bytes the pipeline really moves
                (bound: instruction issue), failure / hot-start counts, and a sample of robots replayed through oracle/tick_spec.py
                over the first ticks (what was timed is what was checked);
  exchange      N > 1 only: the same steps with the north_star's RCCL exchange around every step - ONE scatter of per-rank input
                slabs from rank 0, the solve reading them where they landed, ONE gather of output slabs (include/wcqp.h: shard
                slabs) - as a short second pass; `value` itself stays the no-exchange number;
  cpu_baseline  on rank 0: oracle/wc_oracle.c on this box's host cores - the reference's CPU algorithms restated (`kind: port`: OSQP
                for the MPC, a dense active set for the IK) and, as `same_algorithm_qps`, the device kernels' OWN direct methods in
                plain C, so that the record separates "a better algorithm" from "an MI355X".
"""
import argparse
import json
import math
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
MKEYS = ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc")
IKEYS = ("J_left", "J_right", "J_neck", "J_com", "q", "state")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="robot instances per GPU")
    ap.add_argument("--repeats", type=int, default=5, help="qp workload: timed regions, each over --steps cold steps of its own; value = the median")
    ap.add_argument("--ik-vmax", type=float, default=0.5, help="joint velocity limit of the synthetic robots [rad/s]")
    ap.add_argument("--ik-form", choices=["qpoases", "osqp"], default="qpoases")
    ap.add_argument("--exchange", action="store_true", help="time the steps WITH the RCCL exchange (one scatter of input slabs, one gather of output slabs per step) as the value")
    ap.add_argument("--workload", choices=["qp", "tick", "kin"], default="qp",
                    help="qp: configs[1]+[2] cold-start batches, with the tick configs as a `tick` object (default); tick: the closed-loop tick of "
                         "configs[3]/[4] alone, one tick per step; kin: the kinematics kernel alone (auxiliary)")
    ap.add_argument("--no-graph", action="store_true", help="tick workload: plain launches instead of hipGraph replay (one tick per launch only)")
    ap.add_argument("--streams", type=int, choices=[0, 1, 2, 3, 4], default=0, help="tick workload: robot groups, each with a pipeline and a stream of its own (0 = 1 when a launch walks through the ticks)")
    ap.add_argument("--pipelines", type=int, default=0, help="qp workload, --plan-ways 0: step i goes to pipeline i %% P (own stream, own outputs); 0 = 3")
    ap.add_argument("--horizon", type=int, default=50, help="MPC horizon N (BASELINE: 50; the shipped controllerHorizon 2 s is N = 200: auxiliary line)")
    ap.add_argument("--input-sets", type=int, default=0, help="qp workload: distinct input sets (0 = one per step of the whole run while they fit 64 GB)")
    ap.add_argument("--ik-jac", choices=["mixed", "auto", "general"], default="mixed", help="wcqp_ik_params.jacobian_structure (plans need mixed)")
    ap.add_argument("--tick-tables", action="store_true", help="tick workload: constant uploaded Jacobians instead of per-tick kinematics")
    ap.add_argument("--tick-kin-handoff", choices=["fused", "dense", "compact"], default="fused")
    ap.add_argument("--ticks-per-launch", type=int, default=0, help="tick workload: ticks the fused kernel runs per launch (0 = all of a run call, 1 = one launch per tick)")
    ap.add_argument("--tick-cold-ik", action="store_true", help="tick workload: no IK hot start")
    ap.add_argument("--plan-ways", type=int, default=-1,
                    help="qp workload: W wavefronts share a robot group (way w takes steps w, w + W, ...; own outputs per way).  0 = one launch per step; "
                         "-1 (default) = as many ways as put >= 16384 workgroups into the launch (8 x the card's resident wavefronts), at least 4, at most one per step")
    ap.add_argument("--plan-queue", type=int, choices=[0, 1], default=0, help="plan mode: 1 = (step, robot group) units from a device-side work queue instead of fixed ways")
    ap.add_argument("--resident-pass", action="store_true", help="plan mode: also time the dominant kernel re-reading ONE input set (frac_resident_inputs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU work of the baseline sample")
    ap.add_argument("--no-tick", action="store_true", help="qp workload: leave the `tick` object out")
    ap.add_argument("--tick-batch", type=int, default=8192, help="`tick` object: robots per GPU (BASELINE configs[3]/[4]: 65536 over 8 GPUs)")
    ap.add_argument("--tick-ticks", type=int, default=1000, help="`tick` object: ticks of the timed wcqp_tick_run call")
    ap.add_argument("--exchange-steps", type=int, default=10, help="N > 1: steps of the second pass that measures the exchange")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------- launching the ranks
def spawn_ranks(n, argv=None, script=None, poll_s=0.05):
    """python bench.py --gpus N: start N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* per child, rendezvous on
    127.0.0.1), relay rank 0's standard output, exit non-zero when any rank does.  The parent makes no GPU call at all.
    Supervision: ALL children are polled; the first one that ends non-zero takes the others with it (a rank that dies before the
    rendezvous would otherwise leave the rest waiting for its timeout) and the parent exits non-zero at once; SIGTERM / SIGINT
    to the parent are forwarded, so that nothing is left holding a GPU."""
    import signal
    import socket
    import subprocess
    import tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = script or os.path.abspath(__file__)
    argv = sys.argv[1:] if argv is None else argv
    out0 = tempfile.TemporaryFile()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=out0 if r == 0 else subprocess.DEVNULL))

    def stop_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass
        t_end = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()

    def on_signal(signum, frame):
        stop_all(signal.SIGTERM)
        raise SystemExit(128 + signum)
    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                rc = p.poll()
                if rc is not None and rc != 0:
                    failed = (r, rc)
                    break
            time.sleep(poll_s)
        if failed is None:
            failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.returncode), None)
        if failed is not None:
            stop_all()
        rcs = [p.wait() for p in procs]
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit("bench.py: rank exit codes %s%s" % (rcs, (" (rank %d ended first with %d; the others were stopped)" % failed) if failed else ""))


class Ctx:
    """torch, the package, the process group and this rank's device."""
    pass


def init(args):
    import datetime
    import torch
    import walking_controllers_amd as wca
    c = Ctx()
    c.torch, c.wca, c.args = torch, wca, args
    launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    c.rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if c.world != max(1, args.gpus):
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node == --gpus, "
                         "or without a launcher: bench.py then starts the ranks itself)" % (args.gpus, c.world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the solve path has no CPU fallback")
    ndev = torch.cuda.device_count()
    c.backend = os.environ.get("WCQP_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; "gloo" only for rehearsals on fewer GPUs
    if ndev < c.world and c.backend != "gloo":
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s): one rank per GPU (WCQP_DIST_BACKEND=gloo lets a smaller box "
                         "REHEARSE the launch with the ranks sharing its GPUs - not a measurement)" % (c.world, ndev))
    dev_index = local_rank % max(1, ndev)          # == local_rank on a full node
    torch.cuda.set_device(dev_index)
    c.dev = torch.device("cuda", dev_index)
    c.dist = c.host_group = None
    if launched:                                   # under a launcher, also with ONE rank (RCCL with world size 1 is a valid group)
        import torch.distributed as dist
        to = datetime.timedelta(seconds=int(os.environ.get("WCQP_DIST_TIMEOUT_S", "120")))     # a rank that never arrives fails the job in minutes, not in half an hour
        if c.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=c.dev, timeout=to)
        else:
            dist.init_process_group(backend=c.backend, timeout=to)
        c.dist = dist
        if c.world > 1:
            # a HOST-side group for the one long wait of the run: rank 0's CPU-baseline leg (tens of seconds) while the others are done.
            # (one node: gloo over the loopback interface - the container's hostname may not resolve; without the group the baseline leg is
            # skipped at N > 1 rather than risking the device-side group's timeout)
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            try:
                c.host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=900))
            except Exception as e:
                print("bench.py: no host-side (gloo) group (%r): cpu_baseline is left out of this N > 1 line" % (e,), file=sys.stderr)
                c.host_group = None
    return c


def reduce_over_ranks(c, value, op="max"):
    if c.dist is None:
        return value
    on = c.dev if c.dist.get_backend() == "nccl" else c.torch.device("cpu")
    t = c.torch.tensor([value], dtype=c.torch.float64, device=on)
    c.dist.all_reduce(t, op=c.dist.ReduceOp.MAX if op == "max" else c.dist.ReduceOp.SUM)
    return float(t.item())


def main():
    args = parse_args()
    # `--gpus N` without a launcher: this process becomes the launcher - N children, one rank per GPU, started BEFORE
    # anything here touches the GPU (the parent never does)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    c = init(args)
    if args.workload == "tick":
        out = bench_tick(c)
    elif args.workload == "kin":
        out = bench_kin(c)
    else:
        out = bench_qp(c)
    if c.host_group is not None:
        c.dist.barrier(group=c.host_group)      # rank 0 may still be in its CPU-baseline leg: the others wait here, on the host
    if c.rank == 0:
        print(json.dumps(out), flush=True)
    if c.dist is not None:
        c.dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------- configs[1] + [2]: cold-start batches
def bench_qp(c):
    torch, wca, args, dev, dist, world, rank = c.torch, c.wca, c.args, c.dev, c.dist, c.world, c.rank
    B, NH, S, W = args.batch, args.horizon, args.steps, args.warmup
    first = rank * B
    mpc_bytes = 8 * (2 + 2 * (NH + 1) + 2 + 24) + 16          # x0, reference window, u_prev, padded hull in; u0 out (1056 B at N = 50)
    step_bytes = IK_BYTES_PER_QP + mpc_bytes
    # synthetic inputs: each rank generates its own shard - identical to the rows a rank-0 scatter would hand it (synth.py is counter-based)
    mb = wca.synth.synth_mpc_batch(B, seed=1234, first=first, horizon=NH)
    ib = wca.synth.synth_ik_batch(B, seed=4321, first=first)
    N1 = mb["ref"].shape[1]

    def up(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    base = {k: up(mb[k]) for k in MKEYS}
    base.update({k: up(ib[k]) for k in IKEYS})
    set_bytes = B * (mpc_bytes - 16 + IK_BYTES_PER_QP - 184)
    exch_value = bool(args.exchange)                          # --exchange: the timed steps themselves carry the exchange
    use_plan = args.plan_ways != 0 and args.ik_jac == "mixed" and not exch_value
    if args.plan_ways < 0:
        # enough workgroups for the hardware's dispatcher to even out the launch's ends (8 x the 2048 resident wavefronts), at
        # least 4, at most one per step: 16 at 4096 robots, 4 from 16384 on (profiles/r03_plan_ways_20steps.txt; 20 ways at 4096
        # robots were tried in round 4: 1.16e9 against 1.20e9 QP/s in the 20-step form)
        args.plan_ways = int(max(1, min(S, max(4, -(-16384 // ((B + 3) // 4))))))
    # ---- input sets.  Every step of the run reads input arrays of its OWN (set k = the batch rotated by k B / K rows: same work,
    # other bytes at every address): a set that is read twice within ~256 MB of the card's reads comes out of the Infinity Cache
    # and the kernel then shows 0.62-0.66 of the roofline where it is 0.55.  R repeats x S steps + the warm-up
    # steps need R S + W sets; when they do not fit 64 GB the repeats are cut, never the rule.
    R = max(1, args.repeats)
    k_cap = int(max(13, 64e9 // set_bytes))
    if args.input_sets > 0:
        K = args.input_sets
    else:
        while R > 1 and W + R * S > k_cap:
            R -= 1
        K = int(min(max(W + R * S, 13, -(-(1 << 30) // set_bytes)), k_cap))
    cold = K >= W + R * S
    if not cold:
        while math.gcd(K, max(1, args.plan_ways)) != 1:       # a way then returns to a set only after K of its records
            K += 1
    sets = [base] + [{k: torch.roll(v, shifts=j * max(1, B // K), dims=0).contiguous() for k, v in base.items()} for j in range(1, K)]
    plan_ways = 0 if args.plan_queue else args.plan_ways      # wcqp_qp_plan_create: 0 = work queue
    if use_plan:
        P = S if args.plan_queue else args.plan_ways          # one output buffer set per way; with the work queue one per step
    else:
        P = 1 if exch_value else (args.pipelines if args.pipelines > 0 else 3)

    def outputs():
        return dict(u0=torch.zeros(B, 2, dtype=torch.float64, device=dev), mstat=torch.zeros(B, dtype=torch.int32, device=dev),
                    mact=torch.zeros(B, dtype=torch.int32, device=dev), mmar=torch.zeros(B, dtype=torch.float64, device=dev),
                    dq=torch.zeros(B, 23, dtype=torch.float64, device=dev), istat=torch.zeros(B, dtype=torch.int32, device=dev),
                    ilo=torch.zeros(B, dtype=torch.int32, device=dev), iup=torch.zeros(B, dtype=torch.int32, device=dev),
                    iit=torch.zeros(B, dtype=torch.int32, device=dev))
    outs = [outputs() for _ in range(P)]
    mpc = wca.MpcSolver(horizon=NH)
    ik_form = wca.IK_FORM_QPOASES if args.ik_form == "qpoases" else wca.IK_FORM_OSQP
    jac = {"mixed": wca.IK_JAC_MIXED, "auto": wca.IK_JAC_AUTO, "general": wca.IK_JAC_GENERAL}[args.ik_jac]
    ik = wca.IkSolver(form=ik_form, v_max=args.ik_vmax, jacobian_structure=jac)
    torch.cuda.synchronize(dev)                    # inputs and zero-filled outputs were enqueued on the default stream
    cur = torch.cuda.current_stream(dev)           # torch.distributed enqueues its collectives here
    streams = [cur] if exch_value else [torch.cuda.Stream(dev) for _ in range(1 if use_plan else P)]
    stream = streams[0]
    sp = stream.cuda_stream
    for d in sets:                                 # raw device addresses, looked up once
        d["_mpc"] = tuple(d[k].data_ptr() for k in MKEYS)
        d["_ik"] = tuple(d[k].data_ptr() for k in IKEYS)
    optr = [{k: v.data_ptr() for k, v in o.items()} for o in outs]

    def fill_record(r, i, out_k=None, on=None):
        """Argument record of step i: input set i % K, output buffers i % P (its stream in the launch-per-step form)."""
        d, k = sets[i % K], (i % P if out_k is None else out_k)
        o, m, q_ = optr[k], d["_mpc"], d["_ik"]
        st_ = on if on is not None else (None if use_plan else streams[k % len(streams)].cuda_stream)
        r.x0, r.ref, r.ref_len, r.u_prev, r.hull_A, r.hull_b, r.hull_nc = m[0], m[1], N1, m[2], m[3], m[4], m[5]
        r.u0, r.mpc_status, r.mpc_active, r.mpc_margin, r.mpc_stream = o["u0"], o["mstat"], o["mact"], o["mmar"], st_
        r.J_left, r.J_right, r.J_neck, r.J_com, r.q, r.state = q_
        r.dq, r.ik_status, r.active_lower, r.active_upper, r.foot_err, r.iters = o["dq"], o["istat"], o["ilo"], o["iup"], None, o["iit"]
        r.ik_stream = st_

    def records(lo, n, **kw):
        rr = (wca.capi.QpStep * n)()
        for t_ in range(n):
            fill_record(rr[t_], lo + t_, **kw)
        return rr

    events = [torch.cuda.Event() for _ in streams]

    def barrier():
        """Every stream the steps use has finished (an event at its tail, polled, then hipEventSynchronize'd), all ranks have, and
        the device is idle.  Returns the host time at which this rank's work was COMPLETE - the end of a timed region - and the
        time after the device-wide synchronize that follows: on this ROCm stack the first hipDeviceSynchronize after a burst of
        launches costs the host 20-75 us with the device already idle (tools/sync_cost.py), no part of the steps."""
        for e, st_ in zip(events, streams):
            e.record(st_)
        while not all(e.query() for e in events):       # polled: a blocking wait wakes up ~50 us late
            pass
        for e in events:
            e.synchronize()
        t_done = time.perf_counter()
        torch.cuda.synchronize(dev)
        t_sync = time.perf_counter()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        return t_done, t_sync
    for st_ in streams:
        if st_ is not cur:
            st_.wait_stream(cur)

    # ---- the exchange: per-rank slabs, ONE scatter + ONE gather per step; rank 0's shard is the golden batch
    slabs = None
    if dist is not None or exch_value:
        slabs = wca.sharding.ShardSlabs(dist, B, N1, device=dev)
        if rank == 0:
            for r_ in range(world):                           # shape-true stand-ins for the other ranks' blocks: only the traffic matters
                slabs.fill_peer(r_, {**{k: mb[k] for k in MKEYS}, **{k: ib[k] for k in IKEYS}})
        slab_step = (wca.capi.QpStep * 1)(slabs.step())
        torch.cuda.synchronize(dev)

    def exchange_step():
        slabs.scatter()
        wca.capi.qp_enqueue_steps(mpc, ik, B, slab_step)      # one launch: IK and MPC workgroups side by side, reading the slab in place (default stream)
        slabs.gather()

    # ---- set-up (not warm-up): every input set is read once, so that none is touched for the first time inside a timed region
    wca.capi.qp_enqueue_steps(mpc, ik, B, records(0, K, out_k=0, on=sp))
    barrier()
    # ---- the W warm-up steps, in the form the timed steps take
    if exch_value:
        for _ in range(W):
            exchange_step()
    elif use_plan and W > 0 and (not args.plan_queue or W <= P):
        wplan = wca.capi.QpPlan(mpc, ik, B, records(0, W), ways=plan_ways)
        wplan.enqueue(sp)
        barrier()
        wplan.close()
    elif W > 0:
        wca.capi.qp_enqueue_steps(mpc, ik, B, records(0, W))
    barrier()
    # ---- R timed regions: repeat r = steps W + r S ... W + (r + 1) S - 1, nothing in the region but the enqueue of its S steps
    work = []
    for r_ in range(R):
        lo = W + r_ * S
        work.append(None if exch_value else (wca.capi.QpPlan(mpc, ik, B, records(lo, S), ways=plan_ways) if use_plan else records(lo, S)))
    regions, regions_sync, t_enq = [], [], 0.0
    for r_ in range(R):
        t0 = time.perf_counter()
        if exch_value:
            for _ in range(S):
                exchange_step()
        elif use_plan:
            work[r_].enqueue(sp)
        else:
            wca.capi.qp_enqueue_steps(mpc, ik, B, work[r_])
        t_enq = time.perf_counter() - t0
        t_done, t_sync = barrier()
        regions.append(reduce_over_ranks(c, t_done - t0))
        regions_sync.append(reduce_over_ranks(c, t_sync - t0))
    order = sorted(range(R), key=lambda i_: regions[i_])
    med = order[(R - 1) // 2]                       # (the lower median: a region that WAS measured, not a mean of two)
    elapsed, elapsed_sync = regions[med], regions_sync[med]
    total_qp = 2 * B * world * S
    for st_ in streams:
        if st_ is not cur:
            cur.wait_stream(st_)

    # ---- sanity + goldens: the timed work really solved the problems, and solved them right (every way's LAST batch of the last repeat)
    if exch_value:
        ov = slabs.out_views()
        outs_chk = [dict(u0=ov["u0"], mstat=ov["mpc_status"], mact=ov["mpc_active"], dq=ov["dq"], istat=ov["ik_status"], ilo=ov["active_lower"], iup=ov["active_upper"], iit=ov["iters"])]
    else:
        outs_chk = outs
    n_ok_ik = min(int((o["istat"] == 0).sum().item()) for o in outs_chk)
    n_ok_mpc = min(int((o["mstat"] == 0).sum().item()) for o in outs_chk)
    ik_iters = float(outs_chk[0]["iit"].double().mean().item())
    frac_active = float(((outs_chk[0]["ilo"] | outs_chk[0]["iup"]) != 0).double().mean().item())
    golden = {"golden_max_abs_err": None, "golden_active_set_mismatches": None}
    if first == 0 and NH == 50 and args.ik_form == "qpoases" and abs(args.ik_vmax - 0.5) < 1e-12:
        try:
            gm = np.load(os.path.join(ROOT, "tests", "golden", "mpc_cfg2_b4096.npz"), allow_pickle=False)
            gi = np.load(os.path.join(ROOT, "tests", "golden", "ik_qpoases_v050_b4096.npz"), allow_pickle=False)
            err, mism, rows_checked = 0.0, 0, 0
            last = W + R * S - 1
            for p_ in range(len(outs_chk)):
                i_last = last - ((last - p_) % P)
                if i_last < W and not exch_value:
                    continue
                # input set k is the batch rolled by k B / K rows: output row r of a step that read set k belongs to instance (r - k B / K) mod B
                inst = np.arange(B) if exch_value else (np.arange(B) - (i_last % K) * max(1, B // K)) % B
                o = {k: v.cpu().numpy() for k, v in outs_chk[p_].items()}
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
                      "golden": "every way's last timed batch vs tests/golden/mpc_cfg2_b4096.npz (u0, active rows, status) and ik_qpoases_v050_b4096.npz "
                                "(dq, active bounds, status): every row of both QPs; active sets where the strict-complementarity margin exceeds 1e-7"}
        except Exception as e:                      # a bench line without the check is still a bench line; say why
            golden["golden_error"] = repr(e)
    for w_ in work:
        if w_ is not None and use_plan:
            w_.close()

    # ---- kernel durations, measured AFTER the timed regions in short passes of their own: HIP events on the launch stream around
    # n back-to-back launches of one kernel; `cold` rotates over the K input sets, `resident` re-reads one
    def kernel_ms(launch, cold_, n=None):
        n = n or max(24, min(2 * K, 96))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(min(K, 32)):
            launch(sets[i % K if cold_ else 0])
        torch.cuda.synchronize(dev)
        e0.record(stream)
        for i in range(n):
            launch(sets[(i + 32) % K if cold_ else 0])
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n
    o0 = optr[0]
    ik_one = ik if args.ik_jac != "auto" else wca.IkSolver(form=ik_form, v_max=args.ik_vmax, jacobian_structure=wca.IK_JAC_MIXED)

    def launch_ik(d, solver=None):
        m = d["_ik"]
        (solver or ik_one).solve_device(B, m[0], m[1], m[2], m[3], m[4], m[5], o0["dq"], o0["istat"], o0["ilo"], o0["iup"], 0, o0["iit"], sp)

    def launch_mpc(d):
        m = d["_mpc"]
        mpc.solve_device(B, m[0], m[1], N1, m[2], m[3], m[4], m[5], o0["u0"], o0["mstat"], o0["mact"], o0["mmar"], sp)
    ik_ms, ik_ms_res = kernel_ms(launch_ik, True), kernel_ms(launch_ik, False)
    mpc_ms, mpc_ms_res = kernel_ms(launch_mpc, True), kernel_ms(launch_mpc, False)
    ik_auto = wca.IkSolver(form=ik_form, v_max=args.ik_vmax, jacobian_structure=wca.IK_JAC_AUTO)
    ik_auto_ms = kernel_ms(lambda d: launch_ik(d, ik_auto), True)
    pair_recs = {}

    def launch_pair(d):
        if id(d) not in pair_recs:
            r1 = (wca.capi.QpStep * 1)()
            r, m, q_ = r1[0], d["_mpc"], d["_ik"]
            r.x0, r.ref, r.ref_len, r.u_prev, r.hull_A, r.hull_b, r.hull_nc = m[0], m[1], N1, m[2], m[3], m[4], m[5]
            r.u0, r.mpc_status, r.mpc_active, r.mpc_margin, r.mpc_stream = o0["u0"], o0["mstat"], o0["mact"], o0["mmar"], sp
            r.J_left, r.J_right, r.J_neck, r.J_com, r.q, r.state = q_
            r.dq, r.ik_status, r.active_lower, r.active_upper, r.foot_err, r.iters, r.ik_stream = o0["dq"], o0["istat"], o0["ilo"], o0["iup"], None, o0["iit"], sp
            pair_recs[id(d)] = r1
        wca.capi.qp_enqueue_steps(mpc, ik_one, B, pair_recs[id(d)])
    pair_ms = pair_ms_res = None
    if args.ik_jac != "general":
        pair_ms, pair_ms_res = kernel_ms(launch_pair, True), kernel_ms(launch_pair, False)

    def plan_pass(recs_, mpc_, ik_, ways_, reps=None):
        """ms per record of a plan over the given records: several launches inside ONE event bracket."""
        pl = wca.capi.QpPlan(mpc_, ik_, B, recs_, ways=ways_)
        n = len(recs_)
        reps = reps or max(2, min(10, 2000 // max(1, n)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pl.enqueue(sp); pl.enqueue(sp)
        torch.cuda.synchronize(dev)
        e0.record(stream)
        for _ in range(reps):
            pl.enqueue(sp)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / reps / n
        pl.close()
        return ms
    plan_ms = plan_ms_res = None
    if use_plan:
        # the timed launch's own length and own input sets (the last repeat's): rocprofv3's average over the run's launches of this kernel is then THIS number
        plan_ms = plan_pass(records(W + (R - 1) * S, S), mpc, ik, plan_ways)
        if args.resident_pass:
            rr = (wca.capi.QpStep * S)()
            for t_ in range(S):
                fill_record(rr[t_], 0, out_k=t_ % P)
            plan_ms_res = plan_pass(rr, mpc, ik, plan_ways)
    # BASELINE configs 2 and 3 on their own: MPC-only / IK-only plans (one launch walks through the batches)
    mpc_plan_ms = ik_plan_ms = mpc_plan_ok = ik_plan_ok = None
    if args.ik_jac == "mixed" and not exch_value:
        try:
            m_bytes = B * (mpc_bytes - 16)
            KM = int(max(13, min(512, -(-(1 << 30) // m_bytes))))
            wm = 16
            while math.gcd(KM, wm) != 1:
                wm += 1
            msets = [{k: base[k] for k in MKEYS}] + [{k: torch.roll(base[k], shifts=j * max(1, B // KM), dims=0).contiguous() for k in MKEYS} for j in range(1, KM)]
            mouts = [dict(u0=torch.zeros(B, 2, dtype=torch.float64, device=dev), mstat=torch.zeros(B, dtype=torch.int32, device=dev),
                          mact=torch.zeros(B, dtype=torch.int32, device=dev), mmar=torch.zeros(B, dtype=torch.float64, device=dev)) for _ in range(wm)]
            nm = max(S, 200)
            mrecs = (wca.capi.QpStep * nm)()
            for t_ in range(nm):
                d_, o_, r = msets[t_ % KM], mouts[t_ % wm], mrecs[t_]
                r.x0, r.ref, r.ref_len, r.u_prev, r.hull_A, r.hull_b, r.hull_nc = (d_["x0"].data_ptr(), d_["ref"].data_ptr(), N1, d_["u_prev"].data_ptr(),
                                                                                     d_["hull_A"].data_ptr(), d_["hull_b"].data_ptr(), d_["hull_nc"].data_ptr())
                r.u0, r.mpc_status, r.mpc_active, r.mpc_margin = o_["u0"].data_ptr(), o_["mstat"].data_ptr(), o_["mact"].data_ptr(), o_["mmar"].data_ptr()
            mpc_plan_ms = plan_pass(mrecs, mpc, None, wm, reps=5)
            mpc_plan_ok = min(int(((o_["mstat"] == 0) | (o_["mstat"] == 3)).sum().item()) for o_ in mouts)
            del msets, mouts
            wi = 5
            while math.gcd(K, wi) != 1:
                wi += 1
            iouts = [outputs() for _ in range(wi)]
            ni = max(S, 100)
            irecs = (wca.capi.QpStep * ni)()
            for t_ in range(ni):
                q_, o_, r = sets[t_ % K]["_ik"], iouts[t_ % wi], irecs[t_]
                r.J_left, r.J_right, r.J_neck, r.J_com, r.q, r.state = q_
                r.dq, r.ik_status, r.active_lower, r.active_upper, r.foot_err, r.iters = (o_["dq"].data_ptr(), o_["istat"].data_ptr(), o_["ilo"].data_ptr(),
                                                                                           o_["iup"].data_ptr(), None, o_["iit"].data_ptr())
            ik_plan_ms = plan_pass(irecs, None, ik, wi, reps=5)
            ik_plan_ok = min(int((o_["istat"] == 0).sum().item()) for o_ in iouts)
            del iouts
        except Exception as e:
            print("bench.py: MPC-only / IK-only plan passes skipped (%r)" % (e,), file=sys.stderr)

    value = total_qp / elapsed
    ik_kernel = {"mixed": "ik4_kernel", "auto": "ik4_kernel", "general": "ik3_kernel"}[args.ik_jac]
    ik_gbs = IK_BYTES_PER_QP * B / (ik_ms * 1e-3) / 1e9
    inputs_note = (("cold: %d input sets of %.1f MB (%.2f GB in total), " % (K, set_bytes / 1e6, K * set_bytes / 1e9)) +
                   ("every step of the run (warm-up and all %d repeats) reads input arrays of its own" % R if cold else
                    "visited round-robin; a wavefront returns to a set after %d of its records" % (K // math.gcd(K, max(1, plan_ways if use_plan else 1)))))
    gbs = lambda ms: step_bytes * B / (ms * 1e-3) / 1e9
    if use_plan:
        roofline = {
            # the dominant kernel IS the timed region: one launch that walks through the steps
            "bound": "hbm", "kernel": "qp_plan_kernel", "achieved": gbs(plan_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs(plan_ms) / HBM_PEAK_GBS,
            "traffic": None, "avg_launch_ms": plan_ms * S, "steps_per_launch": S, "avg_ms_per_step": plan_ms,
            "algorithmic_bytes_per_launch": step_bytes * B * S, "inputs": inputs_note,
            "frac_resident_inputs": (gbs(plan_ms_res) / HBM_PEAK_GBS) if plan_ms_res else None, "avg_ms_per_step_resident_inputs": plan_ms_res,
            "timed_region": {"achieved": step_bytes * B * S / elapsed / 1e9, "frac": step_bytes * B * S / elapsed / 1e9 / HBM_PEAK_GBS},
            "single_launch_form": {"kernel": "qp_pair_kernel", "avg_launch_ms": pair_ms, "frac": (gbs(pair_ms) / HBM_PEAK_GBS) if pair_ms else None,
                                   "note": "what a caller with ONE batch per call gets (wcqp_qp_enqueue_steps, one launch of both QPs): the launch's ramp-up and its slowest wave are not amortised"},
        }
    elif pair_ms is not None:
        roofline = {
            "bound": "hbm", "kernel": "qp_pair_kernel", "achieved": gbs(pair_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs(pair_ms) / HBM_PEAK_GBS,
            "traffic": None, "avg_launch_ms": pair_ms, "algorithmic_bytes_per_launch": step_bytes * B, "inputs": inputs_note,
            "frac_resident_inputs": gbs(pair_ms_res) / HBM_PEAK_GBS, "avg_launch_ms_resident_inputs": pair_ms_res,
            "timed_region": {"batches_in_flight": P, "achieved": step_bytes * B * S / elapsed / 1e9, "frac": step_bytes * B * S / elapsed / 1e9 / HBM_PEAK_GBS},
        }
    else:
        roofline = {"bound": "hbm", "kernel": ik_kernel, "achieved": ik_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ik_gbs / HBM_PEAK_GBS, "traffic": None,
                    "avg_launch_ms": ik_ms, "algorithmic_bytes_per_launch": IK_BYTES_PER_QP * B, "inputs": inputs_note}
    if exch_value:
        timed_as = "per step: ONE scatter of per-rank input slabs, one wcqp_qp_enqueue_steps launch reading the slab in place, ONE gather of output slabs"
    elif use_plan:
        timed_as = ("ONE launch (wcqp_qp_plan_enqueue): qp_plan_kernel, as many wavefronts as are resident at once take (step, robot group) units of the %d steps from a work queue" % S) \
            if args.plan_queue else ("ONE launch (wcqp_qp_plan_enqueue): qp_plan_kernel walks through the %d steps, %d wavefronts per robot group" % (S, P))
    else:
        timed_as = "one wcqp_qp_enqueue_steps call, launch by launch (qp_pair_kernel), %d batches in flight on %d streams" % (P, len(streams))
    out = {
        "metric": METRIC, "value": value, "unit": "QP/s", "n_gpus": world, "steps": S, "warmup": W,
        "ms_per_step": 1e3 * elapsed / S, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "value_min": total_qp / max(regions), "value_max": total_qp / min(regions), "repeats": R,
        "config": {
            "workload": ("BASELINE configs[1]+[2] per GPU: DCM-MPC QP (N=%d, n=%d, cold start) B=%d + QP-IK (iCub 23 DoF, 15 eq rows, %s form, v_max=%.2f rad/s) B=%d; "
                         "2 QP solves per robot-tick" % (NH, 4 * NH + 2, B, args.ik_form, args.ik_vmax, B)),
            "batch_per_gpu": B, "global_batch": B * world, "horizon": NH, "dof": 23,
            "input_sets": K, "input_bytes_per_set": set_bytes, "ik_jacobian_structure": args.ik_jac, "pipelines": P,
            "host_enqueue_us_per_step": 1e6 * t_enq / S, "timed_steps_enqueued_as": timed_as,
            "value_is": "the median of %d timed regions of %d steps each, every region over input arrays no earlier step has read; regions [us]: %s"
                        % (R, S, ", ".join("%.1f" % (1e6 * x) for x in regions)),
            "timed_region": "barrier + torch.cuda.synchronize() -> K steps -> completion events of every stream used (hipEventSynchronize), MAX over ranks; the "
                            "device-wide synchronize that follows adds %.0f us of host time with the device idle (ms_per_step_incl_device_sync)" % (1e6 * (elapsed_sync - elapsed)),
            "ms_per_step_incl_device_sync": 1e3 * elapsed_sync / S, "value_incl_device_sync": total_qp / elapsed_sync,
            "parallelism": "batch sharded over %d GPU(s), %s" % (world, "RCCL scatter / gather of per-rank slabs around every step" if exch_value else "no data-path collective"),
        },
        "roofline": roofline,
        "kernels": {
            "ik_kernel": ik_kernel, "ik_ms": ik_ms, "ik_ms_resident_inputs": ik_ms_res, "ik_qps_per_gpu": B / (ik_ms * 1e-3),
            "ik_hbm_frac": ik_gbs / HBM_PEAK_GBS, "ik_hbm_frac_resident_inputs": IK_BYTES_PER_QP * B / (ik_ms_res * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # secondary bounds: fp64 vector ~6 kflop per IK-QP against 78.6 TFLOP/s; fp64 MFMA 6 v_mfma_f64_16x16x4 per IK-QP = 12.3 kflop issued
            "ik_fp64_valu_frac": 6.0e3 * B / (ik_ms * 1e-3) / 78.6e12,
            "ik_mfma_f64_tflops": 12288.0 * B / (ik_ms * 1e-3) / 1e12, "ik_mfma_f64_frac": 12288.0 * B / (ik_ms * 1e-3) / 78.6e12,
            "ik_auto_ms": ik_auto_ms, "ik_auto_fallback_launch_ms": ik_auto_ms - ik_ms,
            "mpc_ms": mpc_ms, "mpc_ms_resident_inputs": mpc_ms_res, "mpc_qps_per_gpu": B / (mpc_ms * 1e-3),
            "mpc_hbm_frac": mpc_bytes * B / (mpc_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "mpc_bytes_per_qp": mpc_bytes,
            "pair_ms": pair_ms, "pair_hbm_frac": (gbs(pair_ms) / HBM_PEAK_GBS) if pair_ms else None,
            "mpc_plan_ms_per_batch": mpc_plan_ms, "mpc_plan_qps_per_gpu": (B / (mpc_plan_ms * 1e-3)) if mpc_plan_ms else None,
            "mpc_plan_hbm_frac": (mpc_bytes * B / (mpc_plan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if mpc_plan_ms else None, "mpc_plan_solved": mpc_plan_ok,
            "ik_plan_ms_per_batch": ik_plan_ms, "ik_plan_qps_per_gpu": (B / (ik_plan_ms * 1e-3)) if ik_plan_ms else None,
            "ik_plan_hbm_frac": (IK_BYTES_PER_QP * B / (ik_plan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ik_plan_ms else None, "ik_plan_solved": ik_plan_ok,
        },
        "solved": dict({"ik": n_ok_ik, "mpc": n_ok_mpc, "of": B, "ik_mean_active_set_changes": ik_iters, "ik_frac_with_active_bounds": frac_active}, **golden),
    }
    # HBM bytes per launch from the PMC passes (committed summary), only while the kernels are the ones they ran on
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            tj = json.load(open(traffic_file))
            tr = tj.get("per_batch", {}).get(str(B))
            if tj.get("csrc_sha256") != wca.capi.source_hash():
                out["roofline"]["traffic_note"] = "profiles/traffic.json was measured on other kernel sources (re-run the PMC passes + tools/pmc/make_traffic.py): not quoted"
            elif tr and NH == 50:
                out["roofline"]["traffic"] = tr.get("plan_hbm_bytes_per_step") if use_plan else tr.get("pair_hbm_bytes_per_launch" if pair_ms is not None else "ik_hbm_bytes_per_launch")
        except Exception:
            pass
    # ---- N > 1: what the exchange costs, as a short second pass (value stays the no-exchange number)
    if dist is not None and not exch_value:
        try:
            ne = max(2, args.exchange_steps)
            for _ in range(2):
                exchange_step()
            torch.cuda.synchronize(dev)
            dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(ne):
                exchange_step()
            torch.cuda.synchronize(dev)
            t_ex = reduce_over_ranks(c, time.perf_counter() - t0)
            dist.barrier()
            ov = slabs.gathered() if rank == 0 else None
            per_peer = slabs.bytes_per_step
            out["exchange"] = {
                "value": 2 * B * world * ne / t_ex, "unit": "QP/s", "steps": ne, "ms_per_step": 1e3 * t_ex / ne,
                "bytes_per_step_per_peer": per_peer, "bytes_per_step": per_peer * world,
                "link_GBps": per_peer / (t_ex / ne) / 1e9,
                "collectives_per_step": 2, "backend": dist.get_backend(),
                "form": "rank 0 -> ONE scatter of per-rank input slabs [x0|ref|u_prev|hull_A|hull_b|hull_nc|J_left|J_right|J_neck|J_com|q|state] -> one launch of both QPs "
                        "reading the slab in place -> ONE gather of output slabs (include/wcqp.h: shard slabs)",
                "link_GBps_is": "bytes one peer receives + returns per step / the step's wall time: a LOWER bound of the link rate (the solve is inside the step)",
                "gathered_ok": (bool((ov["ik_status"] == 0).all()) and bool((ov["mpc_status"] == 0).all())) if ov is not None else None,
            }
        except Exception as e:
            out["exchange"] = {"error": repr(e)}
    del sets, base
    torch.cuda.empty_cache()
    if not args.no_tick:
        out["tick"] = tick_object(c)
    if rank == 0 and not args.no_cpu_baseline and (world == 1 or c.host_group is not None):
        out["cpu_baseline"] = cpu_baseline(mb, ib, args)       # (the other ranks wait at main()'s host-side barrier)
    return out


# ------------------------------------------------------------------------------------------------- configs[3] / [4]: the closed-loop tick
def tick_measure(c, B, T, W, kin_mode, first=0, handoff="fused", ticks_per_launch=0, n_groups=1, hot_start=True, ik_form="qpoases",
                 vmax_tables=0.5, graph=True, check_ticks=16, external=False):
    """One closed-loop run: B robots per GPU, W warm-up ticks (the first `check_ticks` of them logged and replayed through
    oracle/tick_spec.py for a sample of robots), then T timed ticks in ONE wcqp_tick_run call per robot group.
    external: wcqp_tick_params.plant = EXTERNAL - every tick behind a wcqp_tick_set_feedback_device call and a run call of its own (the
    measured state fed here is the robots' initial one, held: this run prices the mode, the parity of it is tests/test_tick_pipeline.py's).
    Returns (results, elapsed seconds MAX over ranks)."""
    torch, wca, dev, dist = c.torch, c.wca, c.dev, c.dist
    S_ = wca.synth
    L = 0 if external else min(check_ticks, W)
    vmax = S_.WALK_VMAX if kin_mode else vmax_tables * np.ones(23)
    form = wca.IK_FORM_QPOASES if ik_form == "qpoases" else wca.IK_FORM_OSQP
    kin = wca.KinModel(S_.icub_like_model()) if kin_mode else None
    cuts = [B * k // n_groups for k in range(n_groups + 1)]
    pipes, datas = [], []
    for g in range(n_groups):
        f0, cnt = first + cuts[g], cuts[g + 1] - cuts[g]
        if kin_mode:
            kb = S_.synth_walk_kin_batch(cnt, first=f0)
            poses = kin.jacobians_host(kb["base"], kb["q"], state=np.zeros((cnt, 87)))["state"]
            data = S_.synth_walk_batch(cnt, T + W, poses, kb, first=f0)
            iks = wca.IkSolver(form=form, v_max=vmax, joint_reg_rad=np.deg2rad(S_.WALK_POSTURE_DEG))
        else:
            data = S_.synth_tick_batch(cnt, T + W, first=f0)
            iks = wca.IkSolver(form=form, v_max=vmax)
        pp = wca.TickPipeline(cnt, T + W, wca.MpcSolver(horizon=50), iks, first=f0, kin=kin, ik_hot_start=hot_start, log_ticks=L,
                              kin_handoff={"fused": 0, "dense": 1, "compact": 2}[handoff], ticks_per_launch=ticks_per_launch, external_feedback=external)
        pp.upload(data)
        if external:
            pp._fb = [torch.from_numpy(np.ascontiguousarray(data[k])).to(dev) for k in ("dcm0", "com0", "u_init")]
        pipes.append(pp); datas.append((f0, cnt, data))
    stream = torch.cuda.current_stream(dev)
    streams = [stream] + [torch.cuda.Stream(dev) for _ in pipes[1:]]

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run(n):
        for pp, st in zip(pipes, streams):
            if external:
                for _ in range(n):
                    pp.set_feedback_device(pp._fb[0].data_ptr(), pp._fb[1].data_ptr(), pp._fb[2].data_ptr(), 0, st.cuda_stream)
                    pp.run(1, use_graph=False, stream=st.cuda_stream)
            else:
                pp.run(n, use_graph=graph, stream=st.cuda_stream)
    if W > 0:
        run(W)
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    run(T)
    for st in streams[1:]:
        stream.wait_stream(st)
    e1.record(stream)
    barrier()
    elapsed = reduce_over_ranks(c, time.perf_counter() - t0)
    states = [pp.download() for pp in pipes]
    dev_ms = e0.elapsed_time(e1) / T
    ik_fail = np.concatenate([x["ik_fail"] for x in states]); mpc_fail = np.concatenate([x["mpc_fail"] for x in states])
    hot_try, hot_hit = int(sum(x["hot_try"].sum() for x in states)), int(sum(x["hot_hit"].sum() for x in states))
    # QP solves = 2 per robot-tick, minus the robot-ticks of STOPPED robots (a robot whose IK failed keeps dq = 0 and its active-set walk is skipped from then on)
    stopped = int(np.maximum(ik_fail - 1, 0).sum())
    # ---- what was timed is what is checked: the first L ticks of a sample of robots through the CPU restatement
    check = {"ticks": L, "robots": [], "max_abs_err_u0": None, "max_abs_err_dq": None, "ok": None}
    if L > 0 and c.rank == 0:
        try:
            from oracle import qp_spec as qs, tick_spec as ts
            f0, cnt, data = datas[0]
            st0 = states[0]
            sample = sorted(set([0, 1, cnt // 3, cnt // 2, cnt - 2, cnt - 1]))
            eu = ed = 0.0
            ipar = qs.IKParams(v_max=np.asarray(vmax, float).copy(), **({"joint_reg_deg": S_.WALK_POSTURE_DEG.copy()} if kin_mode else {}))
            for i in sample:
                one = {k: (v[i:i + 1] if isinstance(v, np.ndarray) else v) for k, v in data.items()}
                one["first"] = int(f0 + i)
                ref = ts.run_ticks(ts.TickParams(), one, L, ipar, ik_form=ik_form, **({"kin_model": S_.icub_like_model(), "foot_rect": S_.FOOT_RECT} if kin_mode else {}))
                eu = max(eu, float(np.abs(st0["u0_log"][:L, i] - ref["u0_log"][:, 0]).max()))
                ed = max(ed, float(np.abs(st0["dq_log"][:L, i] - ref["dq_log"][:, 0]).max()))
            check.update(robots=[int(f0 + i) for i in sample], max_abs_err_u0=eu, max_abs_err_dq=ed, ok=bool(eu <= 1e-9 and ed <= 1e-8),
                         against="oracle/tick_spec.py (exact QPs%s), tolerances 1e-9 (u0) / 1e-8 (dq)" % (" + kin_spec + hull_spec" if kin_mode else ""))
        except Exception as e:
            check["error"] = repr(e)
    for pp in pipes:
        pp.close()
    state_rw = 2 * 8 * (16 + 10 + 23 * 2) + 87 * 8 + 23 * 8 + 16 + 8       # mst + hand + q_des/dq_prev r/w, pose block, dq out, one reference stage, words
    jac_bytes = 4464 if not kin_mode else {"fused": 0, "compact": 2 * 1440 + 288, "dense": 2 * 4464 + 288}[handoff]
    res = {"dev_ms": dev_ms, "ik_fail": ik_fail, "mpc_fail": mpc_fail, "hot_try": hot_try, "hot_hit": hot_hit, "stopped": stopped,
           "tick": min(x["tick"] for x in states), "own_bytes_model": state_rw + jac_bytes, "check": check, "n_groups": len(pipes)}
    return res, elapsed


def tick_object(c):
    """The `tick` object of the default line: BASELINE configs[3]/[4] per GPU, with per-tick kinematics (fused) and with constant Jacobians."""
    args, world = c.args, c.world
    B, T, W = args.tick_batch, args.tick_ticks, 24
    pmc = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj.get("csrc_sha256") == c.wca.capi.source_hash():
            pmc = tj.get("tick", {})
    except Exception:
        pass
    obj = {"config": "BASELINE configs[3]/[4] per GPU: closed-loop receding-horizon robot-tick (DCM-MPC N=50 on the advancing reference window -> ZMP-CoM law -> "
                     "QP-IK 23 DoF qpoases form -> joint integration -> LIPM plant), %d robots x %d ticks in ONE wcqp_tick_run call (one launch walks through the ticks), "
                     "2 QP solves per robot-tick; %d warm-up ticks" % (B, T, W),
           "batch_per_gpu": B, "ticks": T}
    for name, kin_mode in (("fused_kinematics", True), ("constant_jacobians", False)):
        try:
            t_all = time.perf_counter()
            r, elapsed = tick_measure(c, B, T, W, kin_mode, first=c.rank * B)
            stopped_all = int(round(reduce_over_ranks(c, float(r["stopped"]), "sum")))
            value = (2 * B * world * T - stopped_all) / elapsed
            own_pmc = (pmc.get(name) or {}).get("hbm_bytes_per_robot_tick")
            own = own_pmc if own_pmc else r["own_bytes_model"]
            us = 1e3 * r["dev_ms"]
            obj[name] = {
                "value": value, "unit": "QP/s", "us_per_tick": 1e6 * elapsed / T, "kernel_us_per_tick": us,
                "per_tick_kinematics": kin_mode,
                "launch": "ik4_kernel<TICK%s>: 1 launch for the %d ticks (+ tick_mpc_prime_kernel)" % (", fused kinematics" if kin_mode else "", T),
                "roofline": {"bound": "latency", "kernel_bound_note": "one wave's chain of dependent LDS round trips and arithmetic with ONE other wave per SIMD to hide behind (256 VGPRs; VALU and LDS pipe each a little over half busy), not bytes (DESIGN.md 8.2): the HBM fractions below are reported, not the limit",
                             "achieved": 6296 * B / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 6296 * B / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes_basis": "SURVEY.md 8d: 6296 B per robot-tick (1056 MPC + 5240 IK), per TICK",
                             "own_hbm_bytes_per_robot_tick": own, "own_bytes_from": "PMC (profiles/traffic.json)" if own_pmc else "model of the pipeline's reads and writes (no current PMC pass)",
                             "frac_own": own * B / (us * 1e-6) / 1e9 / HBM_PEAK_GBS},
                "solved": {"ticks_executed": r["tick"], "robots_with_ik_fail": int((r["ik_fail"] > 0).sum()), "mpc_fail": int(r["mpc_fail"].sum()),
                           "stopped_robot_ticks_not_counted": stopped_all, "ik_hot_start_tried": r["hot_try"], "ik_hot_start_accepted": r["hot_hit"],
                           "ik_hot_start_hit_rate": (r["hot_hit"] / r["hot_try"]) if r["hot_try"] else None},
                "oracle_check": r["check"], "wall_s_incl_setup": time.perf_counter() - t_all,
            }
        except Exception as e:
            obj[name] = {"error": repr(e)}
    # what the EXTERNAL-feedback mode costs: a feedback copy + the MPC of the tick + its IK as three launches per tick, one run call per tick
    try:
        Te = min(T, 200)
        r, elapsed = tick_measure(c, B, Te, 8, True, first=c.rank * B, external=True)
        obj["external_feedback"] = {
            "value": (2 * B * world * Te - int(round(reduce_over_ranks(c, float(r["stopped"]), "sum")))) / elapsed, "unit": "QP/s", "ticks": Te,
            "us_per_tick": 1e6 * elapsed / Te, "kernel_us_per_tick": 1e3 * r["dev_ms"], "per_tick_kinematics": True,
            "launch": "per tick: tick_feedback_kernel + tick_mpc_prime_kernel<EXT> + ik4_kernel<TICK, fused kinematics, EXT> (wcqp_tick_set_feedback_device + wcqp_tick_run(1))",
            "what": "wcqp_tick_params.plant = EXTERNAL: every tick reads the caller's measured DCM / CoM / ZMP from device arrays, in order (no MPC ahead of its feedback); "
                    "the measured state fed here is the initial one, held", "robots_with_ik_fail": int((r["ik_fail"] > 0).sum())}
    except Exception as e:
        obj["external_feedback"] = {"error": repr(e)}
    return obj


def bench_tick(c):
    """--workload tick: BASELINE configs[3]/[4] alone - every step is one robot-tick of the whole batch; the JSON line's value is the tick's."""
    args, world, B = c.args, c.world, c.args.batch
    kin_mode = not args.tick_tables
    multi_tick = args.ticks_per_launch != 1 and (not kin_mode or args.tick_kin_handoff == "fused")
    # robot groups on streams of their own only pay when a tick is several launches (kinematics launch + solve) or one launch per tick
    n_groups = args.streams if args.streams else (1 if multi_tick else ((3 if (kin_mode and B <= 16384) else 2) if B >= 8192 else 1))
    r, elapsed = tick_measure(c, B, args.steps, args.warmup, kin_mode, first=c.rank * B, handoff=args.tick_kin_handoff, ticks_per_launch=args.ticks_per_launch,
                              n_groups=n_groups, hot_start=not args.tick_cold_ik, ik_form=args.ik_form, vmax_tables=args.ik_vmax, graph=not args.no_graph)
    stopped_all = int(round(reduce_over_ranks(c, float(r["stopped"]), "sum")))
    value = (2 * B * world * args.steps - stopped_all) / elapsed
    dev_ms = r["dev_ms"]
    if kin_mode:
        launches = {"fused": "1 launch per run() call: ik4_kernel<TICK, fused kinematics> walks through the ticks (kinematics, MPC of the next tick, glue, IK, post step)",
                    "compact": "2 launches per tick: kin_jacobians_kernel<TICK, compact>, ik4_kernel<TICK>",
                    "dense": "2 launches per tick: kin_jacobians_kernel<TICK>, ik4_kernel<TICK>"}[args.tick_kin_handoff]
    else:
        launches = "1 launch per run() call: ik4_kernel<TICK> walks through the ticks (MPC of the next tick, glue, IK, post step)"
    if args.ticks_per_launch == 1:
        launches = launches.replace("1 launch per run() call", "1 launch per tick (hipGraph of 8)").replace("walks through the ticks", "")
    own = r["own_bytes_model"]
    return {
        "metric": METRIC, "value": value, "unit": "QP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": ("BASELINE configs[3]/[4] per GPU: receding-horizon robot-tick (%sDCM-MPC N=50 on the advancing reference window -> ZMP-CoM glue -> QP-IK 23 DoF %s form "
                         "v_max=%s -> joint integration), B=%d robots, contact pair changes every 70-110 ticks; 2 QP solves per robot-tick"
                         % ("forward kinematics + Jacobians + support polygon at the integrated joint state -> " if kin_mode else "constant Jacobians, ",
                            args.ik_form, ("%.2f" % args.ik_vmax) if args.tick_tables else "legs 1.5 / upper body 0.3", B)),
            "batch_per_gpu": B, "global_batch": B * world, "horizon": 50, "dof": 23, "ticks": args.steps, "per_tick_kinematics": kin_mode,
            "parallelism": "batch sharded over %d GPU(s), no data-path collective%s" % (world, ("; %d robot groups on %d HIP streams" % (r["n_groups"], r["n_groups"])) if r["n_groups"] > 1 else ""),
        },
        "roofline": {"bound": "hbm", "kernel": "whole tick (%s)" % launches, "achieved": 6296 * B / (dev_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 6296 * B / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": None, "avg_launch_ms": dev_ms, "algorithmic_bytes_per_launch": 6296 * B,
                     "algorithmic_bytes_basis": "SURVEY.md 8d: 6296 B per robot-tick (1056 MPC + 5240 IK), per TICK (a launch runs many)",
                     "own_hbm_bytes_per_robot_tick": own, "frac_own": own * B / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "binding_resource": "instruction issue (two waves per SIMD at 256 VGPRs), not bytes: DESIGN.md 8"},
        "solved": {"ticks_executed": r["tick"], "mpc_fail": int(r["mpc_fail"].sum()), "ik_fail": int(r["ik_fail"].sum()),
                   "robots_with_ik_fail": int((r["ik_fail"] > 0).sum()), "of": B * (args.steps + args.warmup), "stopped_robot_ticks_not_counted": stopped_all,
                   "ik_hot_start_tried": r["hot_try"], "ik_hot_start_accepted": r["hot_hit"], "ik_hot_start_hit_rate": (r["hot_hit"] / r["hot_try"]) if r["hot_try"] else None,
                   "oracle_check": r["check"]},
    }


def bench_kin(c):
    """Auxiliary line: the kinematics kernel alone - Jacobians and poses of B robots per step.  Not the BASELINE metric."""
    torch, wca, args, dev, dist, world, B = c.torch, c.wca, c.args, c.dev, c.dist, c.world, c.args.batch
    kb = wca.synth.synth_kin_batch(B, first=c.rank * B)
    kin = wca.KinModel(wca.synth.icub_like_model())
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    base, q = t(kb["base"]), t(kb["q"])
    JL = torch.zeros(B, 6, 29, dtype=torch.float64, device=dev); JR = torch.zeros_like(JL)
    JN = torch.zeros(B, 3, 29, dtype=torch.float64, device=dev); JC = torch.zeros_like(JN)
    state = torch.zeros(B, 87, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        kin.jacobians_device(B, base.data_ptr(), q.data_ptr(), JL.data_ptr(), JR.data_ptr(), JN.data_ptr(), JC.data_ptr(), state.data_ptr(), stream.cuda_stream)

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
    elapsed = reduce_over_ranks(c, time.perf_counter() - t0)
    k_ms = e0.elapsed_time(e1) / args.steps
    bytes_per = 12 * 8 + 23 * 8 + (6 + 6 + 3 + 3) * 29 * 8 + 36 * 8           # base + q in; Jacobians + actual poses out
    return {
        "metric": "robots/sec through the kinematics kernel (auxiliary; NOT the BASELINE metric)",
        "value": B * world * args.steps / elapsed, "unit": "robots/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "forward kinematics + MIXED free-floating Jacobians (2 feet, neck, CoM) of an iCub-shaped 23-DoF tree, B=%d" % B,
                   "batch_per_gpu": B, "global_batch": B * world, "dof": 23},
        "roofline": {"bound": "hbm", "kernel": "kin_jacobians_kernel", "achieved": bytes_per * B / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": bytes_per * B / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "avg_launch_ms": k_ms, "algorithmic_bytes_per_launch": bytes_per * B},
    }


# ------------------------------------------------------------------------------------------------- the CPU beside it
def cpu_baseline(mb, ib, args):
    """oracle/wc_oracle.c on the host cores.  `value`: the reference's CPU algorithms (OSQP for the MPC, an active-set method for the
    qpOASES-form IK) restated in C - `kind: port`.  `same_algorithm_qps`: the device kernels' own direct methods (condensed MPC +
    2-D projection; base-eliminated range-space IK) in plain C on the same cores, so that the ratio to the GPU number is the
    hardware's share and the ratio between the two CPU numbers the algorithm's."""
    import walking_controllers_amd as wca
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
    mp = qs.MPCParams(horizon=args.horizon)
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
    # the MPC as the reference runs it on >= 97 % of its ticks (no contact change): ONE persistent OSQP workspace per robot, bounds +
    # gradient updated, warm-started solve (WM/src/MPCSolver.cpp:157-173, 249-258); `mpc_qps` above is the COLD number
    warm_ticks = 40
    nw = min(n, 2048)
    wb = wca.synth.synth_mpc_batch(nw, seed=1234, horizon=args.horizon + warm_ticks)
    t = time.perf_counter(); co.mpc_batch_osqp_warm(mp, wb, 8, nthreads=cores); wall_8 = time.perf_counter() - t
    t = time.perf_counter()
    _, warm_iters, warm_thread_s, warm_fail = co.mpc_batch_osqp_warm(mp, wb, warm_ticks, nthreads=cores)
    wall_w = time.perf_counter() - t
    mpc_warm_qps = nw * (warm_ticks - 8) / (wall_w - wall_8) if wall_w > wall_8 else None
    # the device's own algorithms on the host cores (~2 s of CPU work each)
    same = {}
    try:
        gains = co.mpc_condensed_gains(mp)
        co.mpc_batch_condensed(mp, mb, gains, nthreads=cores); co.ik_batch_range_space(ipar, ib, args.ik_form, nthreads=cores)
        t = time.perf_counter(); co.mpc_batch_condensed(mp, take(mb, 256), gains, nthreads=1); s_m = (time.perf_counter() - t) / 256
        t = time.perf_counter(); co.ik_batch_range_space(ipar, take(ib, 256), args.ik_form, nthreads=1); s_i = (time.perf_counter() - t) / 256
        r_m = max(3, int(2.0 * cores / max(s_m * n, 1e-9))); r_i = max(3, int(2.0 * cores / max(s_i * n, 1e-9)))
        t = time.perf_counter()
        for _ in range(r_m):
            co.mpc_batch_condensed(mp, mb, gains, nthreads=cores)
        w_m = time.perf_counter() - t
        t = time.perf_counter()
        for _ in range(r_i):
            co.ik_batch_range_space(ipar, ib, args.ik_form, nthreads=cores)
        w_i = time.perf_counter() - t
        same = {"same_algorithm_qps": 2.0 / (w_m / (n * r_m) + w_i / (n * r_i)),
                "same_algorithm": {"mpc_qps": n * r_m / w_m, "ik_qps": n * r_i / w_i, "single_thread_qps": {"mpc": 1.0 / s_m, "ik": 1.0 / s_i},
                                   "what": "oracle/wc_oracle.c part (4): condensed MPC (gains of the constant equality KKT + 2-D projection onto the support polygon by enumeration) "
                                           "and base-eliminated range-space IK with a dual active set - the device kernels' methods in plain C, OpenMP static split over %d cores; "
                                           "%d / %d passes over the batch" % (cores, r_m, r_i)}}
    except Exception as e:
        same = {"same_algorithm_qps": None, "same_algorithm_error": repr(e)}
    return dict({"value": 2 * n * reps / (wall_m + wall_i), "unit": "QP/s", "cores": cores, "kind": "port",
                 "mpc_warm_qps": mpc_warm_qps, "mpc_warm_mean_iters": warm_iters, "mpc_warm_nonconverged": warm_fail,
                 "mpc_warm_sample": "%d robots x %d warm ticks each after one cold solve: persistent workspace, bounds + gradient update, warm-started OSQP-restatement solve; WALL time "
                                    "of the call with %d warm ticks minus the one with 8 (set-up and cold solve cancel)" % (nw, warm_ticks, warm_ticks),
                 "sample": "%d x the first %d instances of the same workload (1 MPC via OSQP-restatement + 1 IK via %s per instance), OpenMP static split"
                           % (reps, n, "dense active set" if args.ik_form == "qpoases" else "OSQP-restatement"),
                 "mpc_qps": n * reps / wall_m, "mpc_qps_is": "cold start: a new OSQP workspace per QP", "ik_qps": n * reps / wall_i,
                 "single_thread_qps": {"mpc": 1.0 / t_m, "ik": 1.0 / t_i}}, **same)


if __name__ == "__main__":
    main()
