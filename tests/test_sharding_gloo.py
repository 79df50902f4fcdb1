"""
N > 1 path on CPU: two processes, gloo backend.  The per-shard solver here is the C oracle
(tests may use it as the checker's stand-in; on a GPU box the same plumbing wraps the HIP
path — see test_gpu_sharded_equals_single below).  What is verified is the sharding
contract: contiguous block split, no data-path collective, optional scatter/gather, and
SHARD-COUNT INVARIANCE — the gathered result is bitwise the single-process result.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, exchange, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import walking_controllers_amd as wca
    from oracle import c_oracle as co, qp_spec as qs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ip = qs.IKParams(v_max=0.4 * np.ones(23))
    keys = ("J_left", "J_right", "J_neck", "J_com", "q", "state")

    def make(first, count):
        b = wca.synth.synth_ik_batch(count, seed=21, first=first)
        return {k: b[k] for k in keys}

    def solve(inp):
        dq, status, lo, up, _ = co.ik_batch(ip, inp, "qpoases", nthreads=1)
        return {"dq": dq, "status": status, "lo": lo.astype(np.int64), "up": up.astype(np.int64)}

    out = wca.sharding.solve_sharded(dist, 64, make, solve, exchange=exchange)
    if rank == 0:
        q.put({k: v for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", [False, True])
def test_world_size_2_gloo_matches_single_process(exchange):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import walking_controllers_amd as wca
    from oracle import c_oracle as co, qp_spec as qs
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, exchange, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b = wca.synth.synth_ik_batch(64, seed=21)
    dq, status, lo, up, _ = co.ik_batch(qs.IKParams(v_max=0.4 * np.ones(23)), b, "qpoases", nthreads=1)
    assert np.array_equal(got["dq"], dq) and np.array_equal(got["status"], status)       # bitwise
    assert np.array_equal(got["lo"], lo.astype(np.int64)) and np.array_equal(got["up"], up.astype(np.int64))


def _slab_worker(rank, world, port, q):
    """The slab form of the exchange on CPU: ONE scatter of per-rank input slabs, the shard solved from the arrays where they
    landed (the C oracle stands in for the HIP path, which reads the same addresses through the step record), ONE gather."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import walking_controllers_amd as wca
    from oracle import c_oracle as co, qp_spec as qs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ip, mp = qs.IKParams(v_max=0.4 * np.ones(23)), qs.MPCParams()
    gains = co.mpc_condensed_gains(mp)
    calls = {"scatter": 0, "gather": 0}
    real_scatter, real_gather = dist.scatter, dist.gather
    dist.scatter = lambda *a, **k: (calls.__setitem__("scatter", calls["scatter"] + 1), real_scatter(*a, **k))[1]
    dist.gather = lambda *a, **k: (calls.__setitem__("gather", calls["gather"] + 1), real_gather(*a, **k))[1]

    def make(first, count):
        bi = wca.synth.synth_ik_batch(count, seed=21, first=first)
        bm = wca.synth.synth_mpc_batch(count, seed=22, first=first, uprev_sigma=0.04)
        return {**{k: bi[k] for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state")},
                **{k: bm[k] for k in ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc")}}
    holder = {}

    def solve_step(rec, count):
        sl = holder["sl"]
        v = {k: t.numpy() for k, t in sl.in_views().items()}
        # the step record points at exactly these arrays
        assert rec.J_left == sl.in_views()["J_left"].data_ptr() and rec.state == sl.in_views()["state"].data_ptr()
        assert rec.hull_nc == sl.in_views()["hull_nc"].data_ptr() and rec.ref_len == 51 and rec.dq == sl.out_views()["dq"].data_ptr()
        dq, st, lo, up, it = co.ik_batch_range_space(ip, v, "qpoases", nthreads=1)
        u0, act, mst = co.mpc_batch_condensed(mp, v, gains, nthreads=1)
        o = sl.out_views()
        o["dq"].copy_(torch.from_numpy(dq)); o["ik_status"].copy_(torch.from_numpy(st)); o["u0"].copy_(torch.from_numpy(u0))
        o["active_lower"].copy_(torch.from_numpy(lo.view(np.int32))); o["active_upper"].copy_(torch.from_numpy(up.view(np.int32)))
        o["mpc_status"].copy_(torch.from_numpy(mst)); o["mpc_active"].copy_(torch.from_numpy(act.view(np.int32)))
    # (solve_sharded_slabs builds the slabs itself; the solver above needs to see them)
    real_init = wca.sharding.ShardSlabs.__init__

    def init(self, *a, **k):
        real_init(self, *a, **k); holder["sl"] = self
    wca.sharding.ShardSlabs.__init__ = init
    out = wca.sharding.solve_sharded_slabs(dist, 64, 51, make, solve_step)
    assert calls == {"scatter": 1, "gather": 1}, calls                # ONE collective each way per step
    if rank == 0:
        q.put({k: v for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_slab_exchange_matches_single_process():
    """Two ranks, gloo, the slab form (wcqp_slab_layout_for / ShardSlabs): exactly one scatter and one gather, arrays used where
    they landed, gathered result bitwise the single-process one."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    import walking_controllers_amd as wca
    from oracle import c_oracle as co, qp_spec as qs
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slab_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bi = wca.synth.synth_ik_batch(64, seed=21)
    bm = wca.synth.synth_mpc_batch(64, seed=22, uprev_sigma=0.04)
    dq, st, lo, up, _ = co.ik_batch_range_space(qs.IKParams(v_max=0.4 * np.ones(23)), bi, "qpoases", nthreads=1)
    u0, act, mst = co.mpc_batch_condensed(qs.MPCParams(), bm, nthreads=1)
    assert np.array_equal(got["dq"], dq) and np.array_equal(got["ik_status"], st) and np.array_equal(got["u0"], u0)       # bitwise
    assert np.array_equal(got["active_lower"], lo) and np.array_equal(got["active_upper"], up)
    assert np.array_equal(got["mpc_status"], mst) and np.array_equal(got["mpc_active"], act)


def test_slab_layout_is_aligned_and_disjoint():
    sys.path.insert(0, ROOT)
    import walking_controllers_amd as wca
    for B, n in ((1, 51), (4096, 51), (8191, 201), (65536, 51)):
        L = wca.capi.SlabLayout.make(B, n)
        for shapes, total in ((L.in_shapes(), L.in_bytes), (L.out_shapes(), L.out_bytes)):
            end = 0
            for k, (off, dt, shp) in shapes.items():
                assert off % 256 == 0 and off >= end, (k, off, end)
                end = off + int(np.prod(shp)) * np.dtype(dt).itemsize
            assert end <= total and total % 256 == 0
        # SURVEY 8d's algorithmic bytes of a robot-tick are 6296; the slabs add only the status words and padding
        # (mpc_margin 8 B, hull_nc + six status / mask / iteration words 4 B each)
        per = 6296 + (n - 51) * 16 + 8 + 7 * 4
        assert per <= (L.in_bytes + L.out_bytes) / B < per + (21 * 256) / B + 1


def test_shard_ranges_cover_the_batch():
    sys.path.insert(0, ROOT)
    import walking_controllers_amd as wca
    for B, G in ((65536, 8), (4096, 2), (10, 4), (7, 8)):
        seen = []
        for r in range(G):
            f, c = wca.sharding.shard_range(B, G, r)
            seen += list(range(f, f + c))
        assert seen == list(range(B))


@pytest.mark.gpu
def test_gpu_sharded_equals_single(wca):
    """On the GPU box: solving two half-batches separately (what two ranks would do) gives
    bitwise the rows of the full-batch solve — per-instance results do not depend on which
    shard, wave or lane half an instance lands in."""
    B = 1024
    full_i = wca.synth.synth_ik_batch(B, seed=5)
    full_m = wca.synth.synth_mpc_batch(B, seed=6, uprev_sigma=0.04)
    ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.4)
    mpc = wca.MpcSolver()
    ki = ("J_left", "J_right", "J_neck", "J_com", "q", "state")
    km = ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc")
    one_i = ik.solve_host(*[full_i[k] for k in ki])
    one_m = mpc.solve_host(*[full_m[k] for k in km])
    for world in (2, 8):
        parts_i, parts_m = [], []
        for r in range(world):
            f, c = wca.sharding.shard_range(B, world, r)
            si = wca.synth.synth_ik_batch(c, seed=5, first=f)
            sm = wca.synth.synth_mpc_batch(c, seed=6, first=f, uprev_sigma=0.04)
            parts_i.append(ik.solve_host(*[si[k] for k in ki]))
            parts_m.append(mpc.solve_host(*[sm[k] for k in km]))
        assert np.array_equal(np.concatenate([p["dq"] for p in parts_i]), one_i["dq"])
        assert np.array_equal(np.concatenate([p["active_upper"] for p in parts_i]), one_i["active_upper"])
        assert np.array_equal(np.concatenate([p["u0"] for p in parts_m]), one_m["u0"])


def _gpu_worker(rank, world, port, exchange, q):
    """One rank of the sharded HIP path: its own process, its own HIP context on the one GPU of the box, gloo for the
    scatter / gather (what `nccl` = RCCL does between GPUs of a node)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import walking_controllers_amd as wca
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.4)
    mpc = wca.MpcSolver()
    ki = ("J_left", "J_right", "J_neck", "J_com", "q", "state")
    km = ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc")

    def make(first, count):
        bi = wca.synth.synth_ik_batch(count, seed=21, first=first)
        bm = wca.synth.synth_mpc_batch(count, seed=22, first=first, uprev_sigma=0.04)
        d = {k: bi[k] for k in ki}
        d.update({k: (bm[k].astype(np.int64) if k == "hull_nc" else bm[k]) for k in km})
        return d

    def solve(inp):
        oi = ik.solve_host(*[inp[k] for k in ki])
        om = mpc.solve_host(*[(inp[k].astype(np.int32) if k == "hull_nc" else inp[k]) for k in km])
        return {"dq": oi["dq"], "status": oi["status"].astype(np.int64), "up": oi["active_upper"].astype(np.int64),
                "lo": oi["active_lower"].astype(np.int64), "u0": om["u0"], "mstatus": om["status"].astype(np.int64)}

    if exchange == "slabs":
        # ONE scatter of per-rank slabs, the HIP solvers read / write the slabs in place through a step record, ONE gather
        def make_slab(first, count):
            d = make(first, count)
            d["hull_nc"] = d["hull_nc"].astype(np.int32)
            return d

        def solve_step(rec, count):
            assert wca.capi.qp_enqueue_steps(mpc, ik, count, (wca.capi.QpStep * 1)(rec)) == 1
            torch.cuda.synchronize()
        o = wca.sharding.solve_sharded_slabs(dist, 512, 51, make_slab, solve_step, device="cuda:0")
        out = None if o is None else {"dq": o["dq"], "status": o["ik_status"].astype(np.int64), "up": o["active_upper"].astype(np.int64),
                                      "lo": o["active_lower"].astype(np.int64), "u0": o["u0"], "mstatus": o["mpc_status"].astype(np.int64)}
    else:
        out = wca.sharding.solve_sharded(dist, 512, make, solve, exchange=exchange)
    if rank == 0:
        q.put({k: v for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", [False, True, "slabs"])
def test_two_ranks_on_the_gpu_match_single_process(wca, exchange):
    """the N > 1 path with the HIP solver - two spawned ranks (fresh processes, no exec of a
    GPU-initialised one) share the box's one GPU, rank 0 scatters the inputs and gathers the solutions over gloo
    (`exchange`), or every rank generates its own block; the gathered result is bitwise the single-process solve."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, exchange, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    bi = wca.synth.synth_ik_batch(512, seed=21)
    bm = wca.synth.synth_mpc_batch(512, seed=22, uprev_sigma=0.04)
    oi = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.4).solve_host(bi["J_left"], bi["J_right"], bi["J_neck"], bi["J_com"], bi["q"], bi["state"])
    om = wca.MpcSolver().solve_host(bm["x0"], bm["ref"], bm["u_prev"], bm["hull_A"], bm["hull_b"], bm["hull_nc"])
    assert np.array_equal(got["dq"], oi["dq"]) and np.array_equal(got["status"], oi["status"])           # bitwise
    assert np.array_equal(got["up"], oi["active_upper"]) and np.array_equal(got["lo"], oi["active_lower"])
    assert np.array_equal(got["u0"], om["u0"]) and np.array_equal(got["mstatus"], om["status"])


def _run_bench(extra_args, env_extra, timeout=600):
    import json
    import subprocess
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.gpu
def test_bench_rccl_path_with_one_rank():
    """The RCCL branch of bench.py (backend "nccl" IS RCCL on ROCm) executed on the one-GPU box: a launcher-style environment
    with WORLD_SIZE = 1 makes bench.py initialise the process group on the device, scatter the inputs / gather the solutions
    every step (--exchange), all-reduce the timing and destroy the group - everything the N-GPU run does, with one rank."""
    env = dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WCQP_DIST_BACKEND="nccl")
    r, line = _run_bench(["--gpus", "1", "--steps", "4", "--warmup", "2", "--batch", "512", "--exchange", "--no-cpu-baseline", "--no-tick"], env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 1 and "RCCL scatter / gather of per-rank slabs" in line["config"]["parallelism"]
    assert line["solved"]["ik"] == 512 and line["solved"]["mpc"] == 512
    # the slab path, bitwise: what came back through scatter -> solve in place -> gather IS the golden optimum of the same rows
    assert line["solved"]["golden_rows_checked"] == 1024 and line["solved"]["golden_active_set_mismatches"] == 0 and line["solved"]["golden_max_abs_err"] <= 1e-9
    # ... and without --exchange a launcher-started run reports the exchange as a second pass (value stays the no-exchange number)
    r, line = _run_bench(["--gpus", "1", "--steps", "4", "--warmup", "2", "--batch", "512", "--no-cpu-baseline", "--no-tick"], env)
    assert r.returncode == 0, r.stderr[-2000:]
    ex = line["exchange"]
    assert ex["collectives_per_step"] == 2 and ex["backend"] == "nccl" and ex["gathered_ok"] is True and ex["bytes_per_step_per_peer"] > 512 * 6296
    assert "no data-path collective" in line["config"]["parallelism"] and line["value"] > ex["value"]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (the parent never touches the GPU) and reports
    n_gpus = 2; on this one-GPU box that is only allowed as a gloo REHEARSAL, and refused otherwise."""
    args = ["--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "512", "--no-cpu-baseline", "--no-tick"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600,
                       env=dict(env, WCQP_DIST_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 1024
    assert line["exchange"]["collectives_per_step"] == 2 and line["exchange"]["gathered_ok"] is True      # the slab exchange over gloo (staged through the host)
    sys.path.insert(0, ROOT)
    import walking_controllers_amd as wca
    if wca.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600,
                           env=dict(env, WCQP_DIST_BACKEND="nccl"))
        assert r.returncode != 0 and "one rank per GPU" in (r.stderr + r.stdout)
