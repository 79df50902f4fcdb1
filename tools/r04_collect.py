"""Copies the round-4 evidence worth keeping from gpurun_out/r04/ (tools/gpurun.sh profiles / diet / pipe / twopass) into profiles/:
kernel stats CSVs, bench JSON lines, the PMC summary, traffic.json, a per-wave digest of the instruction-mix passes, the A/B lines."""
import glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out", "r04"), os.path.join(R, "profiles")
O = os.path.join(G, "profiles")


def last_json(f):
    lines = [ln for ln in open(f).read().splitlines() if ln.startswith("{")]
    return lines[-1] if lines else None


for f in glob.glob(os.path.join(O, "*_kernel_stats.csv")):
    shutil.copy(f, os.path.join(P, "r04_" + os.path.basename(f)))
for f in glob.glob(os.path.join(O, "bench_*.json")):
    ln = last_json(f)
    if ln:
        open(os.path.join(P, "r04_" + os.path.basename(f)), "w").write(ln + "\n")
if os.path.exists(os.path.join(O, "pmc_summary.json")):
    shutil.copy(os.path.join(O, "pmc_summary.json"), os.path.join(P, "r04_pmc_summary.json"))
    if os.path.exists(os.path.join(O, "traffic.json")) and os.path.getsize(os.path.join(O, "traffic.json")) > 0:
        shutil.copy(os.path.join(O, "traffic.json"), os.path.join(P, "traffic.json"))
    d = json.load(open(os.path.join(O, "pmc_summary.json")))

    def digest(prefix, kern, units, what):
        c = {}
        for p in ("p1", "p2"):
            for k, v in d.get("%s_%s" % (prefix, p), {}).get(kern, {}).items():
                c[k] = v["total"]
        if not c:
            return None
        g = lambda k: c.get(k, 0.0)
        return {"what": what, "wave_units": units,
                "per_wave_unit": {"valu_insts": g("SQ_INSTS_VALU") / units, "lds_insts": g("SQ_INSTS_LDS") / units, "salu_insts": g("SQ_INSTS_SALU") / units,
                                  "smem_insts": g("SQ_INSTS_SMEM") / units, "vmem_read_insts": g("SQ_INSTS_VMEM_RD") / units, "mfma_insts": g("SQ_INSTS_MFMA") / units,
                                  "wave_cycles_x4": 4 * g("SQ_WAVE_CYCLES") / units},
                "shares_of_wave_cycles": {"valu_issue": g("SQ_ACTIVE_INST_VALU") / max(g("SQ_WAVE_CYCLES"), 1), "lds_issue": g("SQ_ACTIVE_INST_LDS") / max(g("SQ_WAVE_CYCLES"), 1),
                                          "any_issue": g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1), "waiting_on_anything": g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
                                          "waiting_inst_any": g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1)},
                "lds_bank_conflict_share_of_lds_active": g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1), "raw_totals": c}
    launches = d.get("plan_4096_p1", {}).get("qp_plan_kernel", {}).get("SQ_WAVES", {}).get("launches", 0)
    out = {"note": "rocprofv3 --pmc passes of tools/gpurun.sh profiles (one counter group per run, --kernel-trace only). per_wave_unit: counter totals over all launches of "
                   "the run / (workgroups x records or ticks they walked through); SQ_WAVE_CYCLES counts in units of 4 cycles.  Round 3's digest: r03_pmc_detail.json.",
           "qp_plan_kernel_b4096": digest("plan_4096", "qp_plan_kernel", max(1, launches) * 88 * 1024, "one wave-record = the IK and the MPC of 4 robots of one step (bench.py --steps 88 --warmup 88 --repeats 1)"),
           "ik4_tick_kernel_fused_kinematics_b8192": digest("tickkin_8192", "ik4_tick_kernel", 224 * 2048, "one wave-tick = kinematics + MPC(t+1) + IK + post step of 4 robots (224 ticks, 2048 workgroups)")}
    json.dump(out, open(os.path.join(P, "r04_pmc_detail.json"), "w"), indent=1)
    for k, v in out.items():
        if isinstance(v, dict):
            print(k, json.dumps(v["per_wave_unit"]), v["lds_bank_conflict_share_of_lds_active"])
# the diet / pipe / two-pass experiments: their bench lines, condensed
rows = []
for sub in ("diet", "pipe"):
    for f in sorted(glob.glob(os.path.join(G, sub, "bench*.json"))):
        ln = last_json(f)
        if ln:
            j = json.loads(ln)
            rows.append("%-34s value %.4g QP/s  %.3f us/step  roofline.frac %.3f  (value_min %.4g, value_max %.4g)" % (sub + "/" + os.path.basename(f)[:-5], j["value"], 1e3 * j["ms_per_step"],
                        j["roofline"]["frac"], j.get("value_min", float("nan")), j.get("value_max", float("nan"))))
if rows:
    open(os.path.join(P, "r04_ab_bench_lines.txt"), "w").write("# bench lines of the round-4 A/B experiments (tools/gpurun.sh diet <tag> / pipe), one per run\n" + "\n".join(rows) + "\n")
f = os.path.join(G, "twopass", "two_pass.json")
if os.path.exists(f) and last_json(f):
    open(os.path.join(P, "r04_two_pass_timing.json"), "w").write(last_json(f) + "\n")
