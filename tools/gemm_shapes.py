#!/usr/bin/env python3
"""Per-shape durations of the encoder kernels from a rocprofv3 kernel trace (tools/prof.sh output dir).
Encoder GEMMs are told apart by grid size and by their predecessor in the launch sequence."""
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
prev = ""
for r in rows:
    n = r["Kernel_Name"]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "flash" in n:
        d["flash attention"].append(us); prev = "flash"
    elif "gemm_bf16_tile" in n:
        g = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        name = {5625: "qkv  N3840 K1280", 7500: "fc1  N5120 K1280", 3750: "n2560/conv1"}.get(g)
        if g == 1875:
            name = {"flash": "out  N1280 K1280", "fc1": "fc2  N1280 K5120"}.get(prev, "conv2 N1280 K3840")
        if name is None: name = f"grid {g}"
        d[name].append(us); prev = name[:3]
    elif "layernorm" in n and "reduce" not in n:
        pass
for k, v in sorted(d.items()):
    v.sort()
    print(f"{k:22s} n={len(v):4d} median={v[len(v)//2]:8.1f} min={v[0]:8.1f} us")
