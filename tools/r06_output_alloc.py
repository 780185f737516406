"""Round 6: BASELINE configs[1] (c2) and its Linear / f32 siblings into three caller-style buffers (torch.empty) and three
library-owned ones (ndi_output_alloc), one fresh process per line -- bench.long_rows_leg is the unit.
    for i in 1 2 3 4 5; do python3 tools/r06_output_alloc.py >> gpurun_out/r06_output_alloc.jsonl; done"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = bench.load_package()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
res = bench.long_rows_leg(pkg, torch, dev, None)
out = {}
for k, v in res.items():
    out[k] = {"caller_ms": v["kernel_ms_per_output_buffer"], "caller_frac": v["frac_per_output_buffer"],
              "owned_ms": v["kernel_ms_library_owned_outputs"], "owned_frac": v["frac_library_owned_outputs"],
              "owned_info": v.get("library_owned_info")}
    c, o = v["kernel_ms_per_output_buffer"], v["kernel_ms_library_owned_outputs"]
    out[k]["caller_spread"] = round(max(c) / min(c) - 1, 4)
    out[k]["owned_spread"] = round(max(o) / min(o) - 1, 4) if o else None
print(json.dumps(out), flush=True)
