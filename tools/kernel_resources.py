#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in the built objects (walking-controllers_amd/csrc/build/*.hip.o), read from the
code objects' metadata notes with llvm-readelf (no GPU needed).

    python tools/kernel_resources.py                 # table
    python tools/kernel_resources.py --json          # machine-readable
    python tools/kernel_resources.py --check         # compare with profiles/kernel_resources.json, exit 1 when anything drifted
    python tools/kernel_resources.py --write         # refresh profiles/kernel_resources.json

DESIGN.md quotes these numbers from profiles/kernel_resources.json; tests/test_abi_host.py runs --check, so a kernel edit that
changes a register count fails the CPU suite until the file (and the text that cites it) is refreshed."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
BUILD = os.path.join(ROOT, "walking-controllers_amd", "csrc", "build")
COMMITTED = os.path.join(ROOT, "profiles", "kernel_resources.json")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return [re.sub(r"\(anonymous namespace\)::", "", x) for x in out[:len(names)]]


def short(name):
    """kernel name with its template arguments, without the parameter list"""
    depth, out = 0, []
    for ch in name:
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return re.sub(r"^void ", "", "".join(out)).strip()


def code_object(obj, tmp):
    fat, dev = os.path.join(tmp, "x.fatbin"), os.path.join(tmp, "x.co")
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj, os.path.join(tmp, "unused.o")], check=True)
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + fat, "--output=" + dev], check=True)
    return dev


def kernels_of(obj):
    sections = subprocess.run([LLVM + "/llvm-readelf", "-S", obj], capture_output=True, text=True, check=True).stdout
    if ".hip_fatbin" not in sections:              # host code only (ik.hip without -DWCQP_DIAG_KERNELS: ABI + dispatch)
        return []
    with tempfile.TemporaryDirectory() as tmp:
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", code_object(obj, tmp)], capture_output=True, text=True, check=True).stdout
    res, cur = [], None
    for line in notes.split("\n"):
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count":                      # first key of a kernel's record
            cur = {"agpr": int(v)}
            res.append(cur)
        elif cur is not None and k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
                                       "vgpr_spill_count", "sgpr_spill_count", "max_flat_workgroup_size"):
            cur[{"vgpr_count": "vgpr", "sgpr_count": "sgpr", "private_segment_fixed_size": "scratch_bytes", "group_segment_fixed_size": "lds_bytes",
                 "vgpr_spill_count": "vgpr_spills", "sgpr_spill_count": "sgpr_spills", "max_flat_workgroup_size": "wg"}[k]] = int(v)
        elif cur is not None and k == "name" and "name" not in cur and not v.endswith(".kd"):
            cur["name"] = v
    res = [r for r in res if "name" in r]
    for r, d in zip(res, demangle([r["name"] for r in res])):
        r["kernel"] = short(d)
        del r["name"]
        regs = -(-(r["vgpr"] + r["agpr"]) // 8) * 8          # unified register file, granule 8 (gfx90a and later)
        r["waves_per_simd_by_registers"] = min(8, 512 // max(regs, 1))
    return res


def collect():
    out = {}
    for f in sorted(os.listdir(BUILD)):
        if f.endswith(".hip.o"):
            for r in kernels_of(os.path.join(BUILD, f)):
                out[f[:-6] + ":" + r.pop("kernel")] = r
    return out


def main():
    cur = collect()
    if "--write" in sys.argv:
        json.dump(cur, open(COMMITTED, "w"), indent=1, sort_keys=True)
        print("wrote", COMMITTED)
        return 0
    if "--check" in sys.argv:
        old = json.load(open(COMMITTED))
        keys = ("vgpr", "agpr", "scratch_bytes", "lds_bytes", "vgpr_spills", "sgpr_spills")
        bad = [(k, {x: (old.get(k, {}).get(x), cur.get(k, {}).get(x)) for x in keys if old.get(k, {}).get(x) != cur.get(k, {}).get(x)})
               for k in sorted(set(old) | set(cur)) if any(old.get(k, {}).get(x) != cur.get(k, {}).get(x) for x in keys)]
        for k, d in bad:
            print("DRIFT %s: %s (committed, built)" % (k, d))
        if bad:
            print("profiles/kernel_resources.json is stale: python tools/kernel_resources.py --write, then refresh the DESIGN.md text that quotes it")
        return 1 if bad else 0
    if "--json" in sys.argv:
        print(json.dumps(cur, indent=1, sort_keys=True))
        return 0
    print("%-64s %5s %5s %5s %8s %7s %6s %s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds", "spills", "waves/SIMD"))
    for k, r in sorted(cur.items()):
        print("%-64s %5d %5d %5d %8d %7d %6d %d" % (k[:64], r["vgpr"], r["agpr"], r["sgpr"], r["scratch_bytes"], r["lds_bytes"],
                                                     r.get("vgpr_spills", 0) + r.get("sgpr_spills", 0), r["waves_per_simd_by_registers"]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
