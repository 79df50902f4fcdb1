"""Summarise the PMC passes of tools/pmc/collect.sh into per-kernel, per-launch numbers."""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
def load(d):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            short = ("qp_plan_kernel" if ("qp_plan_kernel" in name or "qp_plan_pipe_kernel" in name) else "tick_mpc_prime_kernel" if "tick_mpc_prime" in name else "ik4_tick_kernel" if "ik4_kernel<true" in name
                     else "qp_pair_kernel" if "qp_pair_kernel" in name else "ik4_kernel" if "ik4_kernel" in name else "kin_jacobians_kernel" if "kin_jacobians" in name else "ik3_kernel" if "ik3_kernel" in name else "ik2_kernel" if "ik2_kernel" in name else "ik_kernel" if "ik_kernel" in name
                     else "mpc_condensed_kernel" if "mpc_condensed" in name else name.split("(")[0][-30:])
            out[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out
res = {}
for d in sorted(os.listdir(root)):
    if not os.path.isdir(os.path.join(root, d)): continue
    data = load(d)
    for k, ctrs in data.items():
        if not any(s in k for s in ("ik_kernel", "ik2_kernel", "ik3_kernel", "ik4_kernel", "ik4_tick_kernel", "qp_plan_kernel", "tick_mpc_prime", "qp_pair_kernel", "kin_jacobians", "mpc_condensed", "read8", "read16", "read_rows29", "write8")): continue
        for c, vals in ctrs.items():
            res.setdefault(d, {}).setdefault(k, {})[c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals), "total": sum(vals)}
print(json.dumps(res, indent=1))
