"""profiles/traffic.json from a PMC summary (tools/pmc/summarize.py output): HBM-side bytes per launch of the IK kernel, the
MPC kernel and the one-launch step (qp_pair_kernel), FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950).
    python3 tools/pmc/make_traffic.py gpurun_out/r02/pmc_summary.json > profiles/traffic.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import walking_controllers_amd as wca
d = json.load(open(sys.argv[1]))
IK, MPC = 5240, 1056
out = {"note": "HBM-side bytes from rocprofv3 PMC (separate passes: tools/gpurun.sh profiles, summary in profiles/r04_pmc_summary.json). "
               "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of streamed bytes; calibrated in "
               "round 1 on tools/pmc/pmc_calib.hip: 512 MiB streamed with 8 B/lane and 16 B/lane reads report x0.5000, 29-wide rows x0.517, writes exact). "
               "qp_plan_kernel: per STEP (every launch of the PMC runs holds `records_per_launch` records); tick kernels: per TICK (all launches / all ticks).",
       "kernel_version": "qp_plan_kernel (the timed launch of bench.py), ik4_kernel / mpc_condensed_kernel (warm-up launches), ik4_kernel<TICK> (tick workload)",
       # bench.py quotes these numbers only while the kernel sources are the ones they were measured on
       "csrc_sha256": wca.capi.source_hash(), "per_batch": {}, "tick": {}}
for B, recs in ((4096, 88), (65536, 24)):
    f, w = d.get("bench_%d_FETCH_SIZE" % B, {}), d.get("bench_%d_WRITE_SIZE" % B, {})
    row = {}
    for key, kern, alg, per in (("ik", "ik4_kernel", IK, 1), ("mpc", "mpc_condensed_kernel", MPC, 1), ("plan", "qp_plan_kernel", IK + MPC, recs)):
        if kern in f and kern in w:
            b = (2 * 1024 * f[kern]["FETCH_SIZE"]["mean_per_launch"] + 1024 * w[kern]["WRITE_SIZE"]["mean_per_launch"]) / per
            row[key + ("_hbm_bytes_per_step" if per > 1 else "_hbm_bytes_per_launch")] = b
            row[key + "_algorithmic_bytes"] = alg * B
            row[key + "_ratio"] = b / (alg * B)
    if "plan_hbm_bytes_per_step" in row:
        row["records_per_launch"] = recs
    out["per_batch"][str(B)] = row
for tag, name in (("tickkin_8192", "fused_kinematics"), ("ticktab_8192", "constant_jacobians")):
    f, w = d.get(tag + "_FETCH_SIZE", {}), d.get(tag + "_WRITE_SIZE", {})
    k = "ik4_tick_kernel"
    if k in f and k in w:
        ticks = 224                        # bench.py --workload tick --steps 200 --warmup 24
        b = (2 * 1024 * f[k]["FETCH_SIZE"]["total"] + 1024 * w[k]["WRITE_SIZE"]["total"]) / ticks
        out["tick"][name] = {"batch": 8192, "hbm_bytes_per_tick": b, "hbm_bytes_per_robot_tick": b / 8192, "survey_algorithmic_bytes_per_robot_tick": IK + MPC}
print(json.dumps(out, indent=1))
