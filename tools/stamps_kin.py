"""Diagnostic: per-phase cycles of kin_jacobians_kernel from s_memtime stamps (tools/build_variant.sh kstamps -DWCQP_KIN_STAMPS)."""
import os, sys, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import walking_controllers_amd as wca
from walking_controllers_amd import capi
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda", 0)
kb = wca.synth.synth_kin_batch(B)
kin = wca.KinModel(wca.synth.icub_like_model())
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
base, q = t(kb["base"]), t(kb["q"])
JL = torch.zeros(B, 6, 29, dtype=torch.float64, device=dev); JR = torch.zeros_like(JL)
JN = torch.zeros(B, 3, 29, dtype=torch.float64, device=dev); JC = torch.zeros_like(JN)
state = torch.zeros(B, 87, dtype=torch.float64, device=dev)
nw = (B + 1) // 2
dbg = torch.zeros(nw * 8, dtype=torch.int64, device=dev)
capi.lib().wcqp_kin_set_debug(ctypes.c_void_p(dbg.data_ptr()))
for _ in range(3):
    kin.jacobians_device(B, base.data_ptr(), q.data_ptr(), JL.data_ptr(), JR.data_ptr(), JN.data_ptr(), JC.data_ptr(), state.data_ptr(), 0)
torch.cuda.synchronize()
full = dbg.cpu().numpy().reshape(nw, 8).astype(np.float64)[:, :7]
names = ["loads issued + pose landed", "q landed + local frame", "pointer jumping", "frames + base + world", "moments", "columns + stores issued"]
seg = np.diff(full, axis=1)
start = full[:, 0] - full[:, 0].min()
print(json.dumps({"B": B, "median_cycles": dict(zip(names, np.median(seg, 0).tolist())), "p90": dict(zip(names, np.percentile(seg, 90, axis=0).tolist())),
                  "total_median": float(np.median(full[:, -1] - full[:, 0])), "span_all_waves": float(full[:, -1].max() - full[:, 0].min()),
                  "start_quartiles": np.percentile(start, [25, 50, 75, 100]).tolist()}))
