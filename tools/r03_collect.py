"""Copies the round-3 evidence worth keeping from gpurun_out/r03/ (tools/history/r03_profiles.sh, tools/history/r03_tick_kernel_stats.sh) into profiles/:
kernel stats CSVs, bench JSON lines, the PMC summary, traffic.json, and a per-wave digest of the detailed PMC passes."""
import glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(R, "gpurun_out", "r03"), os.path.join(R, "profiles")
for f in glob.glob(os.path.join(O, "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(P, "r03_" + os.path.basename(f)))
for f in glob.glob(os.path.join(O, "bench_*.json")):
    lines = [ln for ln in open(f).read().splitlines() if ln.startswith("{")]
    if lines:
        open(os.path.join(P, "r03_" + os.path.basename(f)), "w").write(lines[-1] + "\n")
shutil.copy(os.path.join(O, "pmc_summary.json"), os.path.join(P, "r03_pmc_summary.json"))
shutil.copy(os.path.join(O, "traffic.json"), os.path.join(P, "traffic.json"))
d = json.load(open(os.path.join(O, "pmc_summary.json")))


def digest(prefix, kern, units_per_launch, what, units_total=None):
    c, launches = {}, 0
    for p in ("p1", "p2", "p3"):
        for k, v in d.get("%s_%s" % (prefix, p), {}).get(kern, {}).items():
            c[k] = v["total"]
            launches = v["launches"]
    if not c:
        return None
    units = units_total if units_total else launches * units_per_launch
    waves_x_units = units                       # wave-records / wave-ticks the counters cover
    g = lambda k: c.get(k, 0.0)
    return {"what": what, "wave_units": waves_x_units,
            "per_wave_unit": {"valu_insts": g("SQ_INSTS_VALU") / units, "lds_insts": g("SQ_INSTS_LDS") / units, "salu_insts": g("SQ_INSTS_SALU") / units,
                              "smem_insts": g("SQ_INSTS_SMEM") / units, "vmem_read_insts": g("SQ_INSTS_VMEM_RD") / units,
                              "mfma_f64_insts": g("SQ_INSTS_VALU_MFMA_F64") / units, "mfma_busy_cycles": g("SQ_VALU_MFMA_BUSY_CYCLES") / units,
                              "wave_cycles_x4": 4 * g("SQ_WAVE_CYCLES") / units},
            "shares_of_wave_cycles": {"valu_issue": g("SQ_ACTIVE_INST_VALU") / max(g("SQ_WAVE_CYCLES"), 1), "lds_issue": g("SQ_ACTIVE_INST_LDS") / max(g("SQ_WAVE_CYCLES"), 1),
                                      "any_issue": g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), "waiting_on_anything": g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
                                      "waiting_inst_any": g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), "waiting_inst_lds": g("SQ_WAIT_INST_LDS") / max(g("SQ_WAVE_CYCLES"), 1)},
            "lds_bank_conflict_share_of_lds_active": g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1),
            "raw_totals": c}


out = {"note": "rocprofv3 --pmc passes of tools/history/r03_profiles.sh (one counter group per run, --kernel-trace only). per_wave_unit: counter totals over all "
               "launches of the run / (workgroups x records or ticks they walked through); SQ_WAVE_CYCLES counts in units of 4 cycles.",
       # bench.py --steps 88: every qp_plan_kernel launch of the run (the timed one and the roofline pass's) walks through 88 records, 1024 robot groups
       "qp_plan_kernel_b4096": digest("plan_4096", "qp_plan_kernel", 88 * 1024, "one wave-record = the IK and the MPC of 4 robots of one step"),
       # bench.py --workload tick --steps 200 --warmup 24: 224 ticks in two launches, 2048 workgroups
       "ik4_tick_kernel_fused_kinematics_b8192": digest("tickkin_8192", "ik4_tick_kernel", None, "one wave-tick = kinematics + MPC(t+1) + IK + post step of 4 robots", units_total=224 * 2048)}
json.dump(out, open(os.path.join(P, "r03_pmc_detail.json"), "w"), indent=1)
for k in ("qp_plan_kernel_b4096", "ik4_tick_kernel_fused_kinematics_b8192"):
    if out[k]:
        print(k, json.dumps(out[k]["per_wave_unit"]), json.dumps(out[k]["shares_of_wave_cycles"]), out[k]["lds_bank_conflict_share_of_lds_active"])
