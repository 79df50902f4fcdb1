"""profiles/traffic.json from a PMC summary (tools/pmc/summarize.py output): HBM-side bytes per launch of the IK kernel, the
MPC kernel and the one-launch step (qp_pair_kernel), FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950).
    python3 tools/pmc/make_traffic.py gpurun_out/r02/pmc_summary.json > profiles/traffic.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import walking_controllers_amd as wca
d = json.load(open(sys.argv[1]))
IK, MPC = 5240, 1056
out = {"note": "HBM-side bytes per launch from rocprofv3 PMC (separate passes: tools/gpu_r02_profiles.sh, summary in profiles/r02_pmc_summary.json). "
               "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of streamed bytes; calibrated in "
               "round 1 on tools/pmc/pmc_calib.hip: 512 MiB streamed with 8 B/lane and 16 B/lane reads report x0.5000, 29-wide rows x0.517, writes exact).",
       "kernel_version": "ik4_kernel, mpc_condensed_kernel and qp_pair_kernel (both in one launch)",
       # bench.py quotes these numbers only while the kernel sources are the ones they were measured on
       "csrc_sha256": wca.capi.source_hash(), "per_batch": {}}
for B in (4096, 65536):
    f, w = d.get("bench_%d_FETCH_SIZE" % B, {}), d.get("bench_%d_WRITE_SIZE" % B, {})
    row = {}
    for key, kern, alg in (("ik", "ik4_kernel", IK), ("mpc", "mpc_condensed_kernel", MPC), ("pair", "qp_pair_kernel", IK + MPC)):
        if kern in f and kern in w:
            b = 2 * 1024 * f[kern]["FETCH_SIZE"]["mean_per_launch"] + 1024 * w[kern]["WRITE_SIZE"]["mean_per_launch"]
            row[key + "_hbm_bytes_per_launch"] = b
            row[key + "_algorithmic_bytes"] = alg * B
            row[key + "_ratio"] = b / (alg * B)
    out["per_batch"][str(B)] = row
print(json.dumps(out, indent=1))
