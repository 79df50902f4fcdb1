"""
CPU tests of the C ABI library and the host logic around it: the library loads, exports
every symbol include/wcqp.h declares, builds the reference's constant blocks, condenses
them correctly, validates arguments — and REFUSES to solve without a GPU (no CPU fallback).
"""
import ctypes as C
import os
import re

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(wca):
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "wcqp.h")).read()
    declared = set(re.findall(r"\b(wcqp_[a-z_]+)\s*\(", hdr))
    assert declared == set(wca.capi.ABI_SYMBOLS)
    lib = wca.capi.lib()
    for s in declared:
        assert getattr(lib, s) is not None
    assert lib.wcqp_version() == 400
    assert lib.wcqp_strerror(-4).decode().startswith("HIP")


def test_mpc_constants_match_reference_blocks(wca, qs):
    """wcqp_mpc_create restates initializeMatrices; compare with the oracle's blocks exactly."""
    for N in (2, 7, 50):
        m = wca.MpcSolver(horizon=N)
        P, A, G = m.matrices()
        c = qs.mpc_constants(qs.MPCParams(horizon=N))
        assert np.array_equal(P, c.P) and np.array_equal(A, c.A_eq) and np.array_equal(G, c.grad_sub)


def test_mpc_condensing_is_the_kkt_inverse(wca, qs):
    """u0_unc = sum Gr_i r_i + Gx x0 + Gu u_prev must equal the equality-only KKT solve, and
    Sigma0 the u0 block of K^-1 — for isotropic and for coupled (non-diagonal) weights."""
    rng = np.random.default_rng(3)
    for Q, R in ((7500.0 * np.eye(2), 9e6 * np.eye(2)),
                 (np.array([[7.0, 2.0], [2.0, 5.0]]), np.array([[90.0, -10.0], [-10.0, 40.0]]))):
        N = 12
        m = wca.MpcSolver(horizon=N, Q=Q, R=R)
        Gr, Gx, Gu, S0 = m.condensed()
        c = qs.mpc_constants(qs.MPCParams(horizon=N, Q=Q, R=R))
        x0, up, ref = rng.normal(size=2), rng.normal(size=2), rng.normal(size=(N + 1, 2))
        P, q, A, l, u = qs.mpc_assemble(c, x0, ref, up, np.zeros((0, 2)), np.zeros(0))
        z, lam = qs._kkt_solve(P, q, A, u)
        u0 = np.einsum("iab,ib->a", Gr, ref) + Gx @ x0 + Gu @ up
        assert np.abs(u0 - z[c.n_x:c.n_x + 2]).max() < 1e-9 * max(1.0, np.abs(z).max())
        K = np.block([[P, A.T], [A, np.zeros((c.n_x, c.n_x))]])
        assert np.allclose(S0, np.linalg.inv(K)[c.n_x:c.n_x + 2, c.n_x:c.n_x + 2], rtol=1e-8, atol=0)


def test_argument_validation(wca):
    lib = wca.capi.lib()
    h = C.c_void_p()
    assert lib.wcqp_mpc_create(None, C.byref(h)) == -1
    with pytest.raises(wca.WcqpError):
        wca.MpcSolver(horizon=0)
    with pytest.raises(wca.WcqpError):
        wca.MpcSolver(Q=np.array([[1.0, 2.0], [3.0, 1.0]]))      # non-symmetric weight
    with pytest.raises(wca.WcqpError):
        wca.IkSolver(dof=12, joint_reg_weights=np.ones(12), joint_reg_gains=np.ones(12),
                     joint_reg_rad=np.zeros(12), v_max=np.ones(12))   # kernels are built for 23 DoF
    with pytest.raises(wca.WcqpError):
        wca.IkSolver(v_min=np.ones(23), v_max=-np.ones(23))


def test_no_cpu_fallback(wca):
    """On a host without a GPU the solve entry points fail loudly; nothing routes to the oracle."""
    if wca.device_count() > 0:
        pytest.skip("GPU present")
    b = wca.synth.synth_mpc_batch(4)
    with pytest.raises(wca.WcqpError, match="HIP"):
        wca.MpcSolver().solve_host(b["x0"], b["ref"], b["u_prev"], b["hull_A"], b["hull_b"], b["hull_nc"])
    ib = wca.synth.synth_ik_batch(2)
    with pytest.raises(wca.WcqpError, match="HIP"):
        wca.IkSolver().solve_host(ib["J_left"], ib["J_right"], ib["J_neck"], ib["J_com"], ib["q"], ib["state"])
    src = open(os.path.join(os.path.dirname(wca.capi.__file__), "capi.py")).read()
    assert "oracle" not in src.replace("no fallback", "")


def test_synthetic_batches_are_shard_invariant(wca):
    full_m = wca.synth.synth_mpc_batch(64, seed=5)
    full_i = wca.synth.synth_ik_batch(64, seed=6)
    for first, count in ((0, 16), (16, 32), (48, 16)):
        sm = wca.synth.synth_mpc_batch(count, seed=5, first=first)
        si = wca.synth.synth_ik_batch(count, seed=6, first=first)
        for k in ("x0", "ref", "u_prev", "hull_A", "hull_b", "hull_nc"):
            assert np.array_equal(sm[k], full_m[k][first:first + count])
        for k in ("J_left", "J_right", "J_neck", "J_com", "q", "state"):
            assert np.array_equal(si[k], full_i[k][first:first + count])


def test_hull_rows_convention(wca):
    """CCW hull, unit outward normals, a.u <= b, padded to 8 rows with 0.u <= 1e30 (D-4)."""
    pts = wca.synth.foot_corners(np.zeros(2), 0.0)
    A, b, nc = wca.synth.hull_rows(pts)
    assert nc == 4 and np.allclose(np.linalg.norm(A[:4], axis=1), 1.0)
    assert sorted(np.round(b[:4], 12)) == [0.02, 0.025, 0.025, 0.05]
    assert not A[4:].any() and (b[4:] == 1e30).all()
    assert (A[:4] @ np.array([0.01, 0.0]) <= b[:4]).all()
    two = np.vstack([pts, wca.synth.foot_corners(np.array([0.05, -0.16]), 0.2)])
    A2, b2, nc2 = wca.synth.hull_rows(two)
    assert 5 <= nc2 <= 8 and ((A2[:nc2] @ two.T) <= b2[:nc2, None] + 1e-12).all()


def test_qp_step_record_layout_matches_the_header(tmp_path):
    """The ctypes mirror of wcqp_qp_step (an array of these crosses the FFI) has the header's size and field offsets."""
    import subprocess
    import walking_controllers_amd as wca
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "abi_layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "abi_layout.c"), "-o", str(exe)])
    lines = [ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines()]
    out = lines[0]
    S = wca.capi.QpStep
    mine = [C.sizeof(S)] + [getattr(S, f).offset for f in ("x0", "ref_len", "u_prev", "hull_nc", "mpc_stream", "J_left", "ik_stream")]
    assert out[0] == "wcqp_qp_step" and [int(x) for x in out[1:]] == mine
    # wcqp_tick_params grew this round (kin_handoff, ticks_per_launch): the ctypes mirror follows the header field for field
    out = lines[1]
    T = wca.capi.TickParams
    mine = [C.sizeof(T)] + [getattr(T, f).offset for f in ("seed", "mpc", "ik", "ik_cold_start_only", "kin", "foot_rect", "kin_handoff", "ticks_per_launch")]
    assert out[0] == "wcqp_tick_params" and [int(x) for x in out[1:]] == mine


def test_source_hash_follows_the_kernel_sources(wca, tmp_path, monkeypatch):
    """profiles/traffic.json is stamped with capi.source_hash(); bench.py quotes it only while the hash is the current one."""
    h = wca.capi.source_hash()
    assert len(h) == 64 and h == wca.capi.source_hash()
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tj = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    assert set(tj["per_batch"]) == {"4096", "65536"} and len(tj["csrc_sha256"]) == 64
    for B in ("4096", "65536"):
        assert 0.95 < tj["per_batch"][B]["plan_ratio"] < 1.15          # no wasted traffic in the committed measurement


def test_bench_gpus_n_starts_ranks_and_fails_with_them(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts two ranks before touching the GPU; on this GPU-less host both ranks
    refuse (there is no CPU path) and the parent's exit code says so - it never reports a 1-GPU number as the 2-GPU point."""
    import subprocess
    import sys
    import walking_controllers_amd as wca
    if wca.device_count() > 0:
        pytest.skip("GPU present (the GPU suite runs the real thing)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "rank exit codes" in (r.stderr + r.stdout) and "{" not in r.stdout
    # and a launcher environment that disagrees with --gpus is refused before any work
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300,
                       env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_spawn_ranks_fails_fast_when_one_rank_dies(tmp_path):
    """bench.spawn_ranks supervises ALL its children: a rank that dies before the rendezvous (a GPU fault, an OOM kill) takes the others
    with it and the launcher exits non-zero at once - not after rank 0 has sat in init_process_group until its timeout.  Here rank 1
    exits with code 3 immediately while rank 0 would sleep for ten minutes; a SIGTERM to the launcher is forwarded as well."""
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = tmp_path / "rank.py"
    child.write_text("import os, sys, time\n"
                     "open(os.path.join(%r, 'started_' + os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                     "if os.environ['RANK'] == '1' and '--die' in sys.argv: sys.exit(3)\n"
                     "print('{\"rank0\": true}', flush=True)\n"
                     "time.sleep(600)\n" % str(tmp_path))
    launcher = tmp_path / "launch.py"
    launcher.write_text("import sys\nsys.path.insert(0, %r)\nimport bench\nbench.spawn_ranks(2, argv=sys.argv[1:], script=%r)\n" % (root, str(child)))
    t0 = time.time()
    r = subprocess.run([sys.executable, str(launcher), "--die"], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "rank exit codes" in r.stderr and "rank 1 ended first with 3" in r.stderr, r.stderr
    assert time.time() - t0 < 30
    assert "rank0" in r.stdout                                   # what rank 0 had printed is still relayed
    pid0 = int((tmp_path / "started_0").read_text())
    time.sleep(0.2)
    with pytest.raises(ProcessLookupError):
        os.kill(pid0, 0)                                         # rank 0 is gone, not orphaned
    # SIGTERM to the launcher: both ranks are stopped
    for f in ("started_0", "started_1"):
        (tmp_path / f).unlink()
    p = subprocess.Popen([sys.executable, str(launcher)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    for _ in range(200):
        if (tmp_path / "started_0").exists() and (tmp_path / "started_1").exists():
            break
        time.sleep(0.05)
    time.sleep(0.2)
    pids = [int((tmp_path / f).read_text()) for f in ("started_0", "started_1")]
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) != 0
    time.sleep(0.2)
    for pid in pids:
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_kernel_resources_match_the_committed_table(wca):
    """Registers, scratch and LDS of every shipped kernel (read from the built code objects' metadata, no GPU needed) against
    profiles/kernel_resources.json - the table DESIGN.md quotes.  A kernel edit that moves a count fails here until
    `python tools/kernel_resources.py --write` refreshes the table (and the text that cites it): VERDICT r3 found the 252 / 256 VGPRs and
    36 / 40 B of scratch of two kernels stale in DESIGN.md."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kernel_resources.py"), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    import json
    tab = json.load(open(os.path.join(root, "profiles", "kernel_resources.json")))
    # the walking kernels hold two waves per SIMD (<= 256 registers of the 512-entry file) and what the product runs by default has no scratch
    # beyond the fused-kinematics tick's few spilled registers
    for k in ("ik4:qp_plan_kernel", "ik4:ik4_kernel<true, 0, false, false>", "ik4:ik4_kernel<true, 2, false, false>", "ik4:qp_pair_kernel", "ik4:ik4_kernel<false, 0, false, false>"):
        assert tab[k]["vgpr"] + tab[k]["agpr"] <= 256 and tab[k]["waves_per_simd_by_registers"] >= 2, k
    assert tab["ik4:qp_plan_kernel"]["scratch_bytes"] == 0 and tab["ik4:ik4_kernel<true, 2, false, false>"]["scratch_bytes"] <= 64
    assert tab["ik4:ik4_kernel<true, 2, false, false>"]["lds_bytes"] <= 20480             # 8 workgroups per CU: two waves per SIMD


def test_compact_jacobian_layout_of_the_icub_shaped_tree(wca):
    """The tick's compact kinematics -> IK hand-off (csrc/tick_device.h: compact_offset): one record per joint - 4 doubles for
    a joint on no frame path (CoM column + pad), 6 on the neck's, 10 on a foot's - derived from the tree's path masks.  Restated
    here for the iCub-shaped tree: legs and torso branch at the root link, so the masks are disjoint and a robot's block is 180
    doubles (1440 B) against the 558 of four dense Jacobians."""
    m = wca.synth.icub_like_model()
    parent = list(m["parent"])
    def path(j):
        out = 0
        while j >= 0:
            out |= 1 << j; j = parent[j]
        return out
    mL, mR, mN = (path(int(j)) for j in m["frame_joint"])
    assert mL & mR == 0 and mL & mN == 0 and mR & mN == 0
    assert [bin(x).count("1") for x in (mL, mR, mN)] == [6, 6, 3]
    off, end = [], 0
    for c in range(23):
        below = (1 << c) - 1
        off.append(4 * c + 6 * bin((mL | mR) & below).count("1") + 2 * bin(mN & below).count("1"))
        end = off[-1] + (10 if ((mL | mR) >> c) & 1 else (6 if (mN >> c) & 1 else 4))
    assert off == sorted(off) and all(o % 2 == 0 for o in off) and end == 170          # + 9 (+1) doubles of p_frame - p_base = 180
    # depth-first numbering: every subtree is an index range (what the prefix-sum subtree moments of the kernels rely on)
    for j in range(23):
        desc = [k for k in range(23) if (path(k) >> j) & 1]
        assert desc == list(range(j, j + len(desc)))


def test_kernels_hold_no_flat_memory_instructions(wca, tmp_path):
    """Every global-memory access of the shipped kernels is a global_* instruction.  A pointer that a kernel reads out of a
    record in device memory (the tick's TickDev, the plan's step records) is generic unless it is marked (csrc/gptr.h), and an
    access through a generic pointer is a flat_* instruction: it counts on vmcnt AND lgkmcnt and forces full drains, which
    silently undid the 'state block first, Jacobians stay in flight' order of the walking kernels for most of round 3
    (DESIGN.md 4.4).  Checked on the device code inside the built objects (no GPU needed)."""
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    build = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "walking-controllers_amd", "csrc", "build")
    checked = 0
    for name in ("mpc", "ik2", "ik3", "ik4", "tick", "hull", "kin"):
        obj = os.path.join(build, name + ".hip.o")
        assert os.path.exists(obj), obj
        fat, dev = str(tmp_path / (name + ".fatbin")), str(tmp_path / (name + ".co"))
        subprocess.run([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, str(tmp_path / "unused.o")], check=True)
        subprocess.run([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + dev], check=True)
        asm = subprocess.run([llvm + "/llvm-objdump", "-d", dev], check=True, capture_output=True, text=True).stdout
        assert len(re.findall(r"\bglobal_(?:load|store)", asm)) > 0, name
        flat = re.findall(r"\bflat_(?:load|store|atomic)\w*", asm)
        assert not flat, (name, len(flat), sorted(set(flat)))
        checked += 1
    assert checked == 7
