#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime cycles of the tick kernel's body (fused kinematics or constant Jacobians), from the stamps of every
workgroup's last tick.  Needs the stamp build:  tools/build_variant.sh tstamps -DWCQP_TICK_STAMPS  and
WCQP_LIB_PATH=walking-controllers_amd/csrc/build/diag/libwcqp_tstamps.so"""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import walking_controllers_amd as wca

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
kin_mode = (sys.argv[3] if len(sys.argv) > 3 else "kin") == "kin"
S = wca.synth
if kin_mode:
    kin = wca.KinModel(S.icub_like_model())
    kb = S.synth_walk_kin_batch(B)
    poses = kin.jacobians_host(kb["base"], kb["q"], state=np.zeros((B, 87)))["state"]
    d = S.synth_walk_batch(B, T, poses, kb)
    vmax = S.WALK_VMAX.copy() if hasattr(S.WALK_VMAX, "copy") else np.broadcast_to(S.WALK_VMAX, (23,)).copy()
    if len(sys.argv) > 4:
        vmax[:] = float(sys.argv[4])           # e.g. 5.0: no joint-velocity bound ever binds
    ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=vmax, joint_reg_rad=np.deg2rad(S.WALK_POSTURE_DEG))
else:
    kin, d = None, S.synth_tick_batch(B, T)
    ik = wca.IkSolver(form=wca.IK_FORM_QPOASES, v_max=0.5)
p = wca.TickPipeline(B, T, wca.MpcSolver(), ik, kin=kin)
p.upload(d); p.run(T)
nw = (B + 3) // 4
buf = np.zeros(nw * 16, np.uint64)
lib = wca.capi.lib()
lib.wcqp_tick_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
wca.capi.check(lib.wcqp_tick_debug_stamps(p._h, buf.ctypes.data_as(C.c_void_p), nw * 16))
t = buf.reshape(nw, 16).astype(np.float64)
if kin_mode and os.environ.get("WCQP_KSTAMPS"):
    # the kinematics phase in detail (library built with -DWCQP_TICK_KSTAMPS): slots 1..9 are its sub-phases
    order = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12]
    names = ["MPC loads landed, partial sums, stash", "joint rotations (sin / cos)", "pointer jumping (3 rounds) + frame stores", "attached frames in base coordinates",
             "base pose, attached frames in world coordinates", "own joints in world coordinates, axes, first moments", "frame columns", "prefix sums through LDS",
             "total, CoM columns", "base vectors, CoM stash, pose block / constants / hand-off loads issued"]
    seg = np.diff(t[:, order], axis=1)
    print(json.dumps({"B": B, "ticks": T, "kinematics_detail": True, "median_cycles": dict(zip(names, np.median(seg, 0).tolist())),
                      "kinematics_median": float(np.median(t[:, 12] - t[:, 0])), "tick_median": float(np.median(t[:, 14] - t[:, 0]))}))
    sys.exit(0)
# stamp ids in program order (fused kinematics: 12 = end of the kinematics phase; 13 = MPC(t+1) finished; 14 = post step done)
order = ([0, 12, 13, 1, 2, 3, 4, 10, 11, 5, 6, 7, 8, 14] if kin_mode else [0, 13, 1, 2, 3, 4, 10, 11, 5, 6, 7, 8, 14])
names = ((["kinematics (+ MPC loads, partial sums)", "MPC(t+1) arithmetic + glue"] if kin_mode else ["loads issued .. MPC(t+1) arithmetic + glue"]) +
         ["pose block / Jacobians landed", "rhs", "pattern / dB", "row operations", "C^T stores", "mfma", "tile + rows", "sweep", "y, x", "active set", "outputs + post step"])
seg = np.diff(t[:, order], axis=1)
print(json.dumps({"B": B, "ticks": T, "kinematics": kin_mode, "median_cycles": dict(zip(names, np.median(seg, 0).tolist())),
                  "p90_active_set": float(np.percentile(seg[:, -2], 90)), "tick_median": float(np.median(t[:, 14] - t[:, 0])),
                  "end_of_tick_fence_median": float(np.median(t[:, 15] - t[:, 14])), "end_of_tick_fence_p90": float(np.percentile(t[:, 15] - t[:, 14], 90)),
                  "tick_p90": float(np.percentile(t[:, 14] - t[:, 0], 90))}))
