"""Per-launch means of the IK-only PMC passes (tools/pmc/collect_ik.sh)."""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
NAMES = tuple(sys.argv[2:]) or ("ik4_kernel", "ik3_kernel", "ik2_kernel", "ik_kernel")
res = collections.defaultdict(dict)
for d in sorted(os.listdir(root)):
    if not os.path.isdir(os.path.join(root, d)): continue
    tag = d.rsplit("_p", 1)[0]
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(n in r["Kernel_Name"] for n in NAMES):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        v = v[2:] if len(v) > 4 else v
        res[tag][c] = sum(v) / len(v)
for tag, c in res.items():
    w = c.get("SQ_WAVES", 32768.0)
    c["_per_wave"] = {k: round(v / w, 1) for k, v in c.items() if k.startswith("SQ_") and k != "SQ_WAVES"}
print(json.dumps(res, indent=1))
