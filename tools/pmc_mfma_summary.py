#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16, SQ_BUSY_CU_CYCLES and
GRBM_GUI_ACTIVE (tools/pmc.sh <name> <those counters>): per-kernel MFMA utilisation.

  kernel cycles  = GRBM_GUI_ACTIVE / 8        (rocprofv3 reports the sum over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back")
  MFMA util      = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs)      (the MfmaUtil expression of rocprofv3 -L)
  flop per cycle = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 / kernel cycles; dense bf16 peak = 1024 flop/cycle/SIMD x 1024 SIMDs
"""
import collections, csv, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> value
for r in csv.DictReader(open(f)):
    per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (k, _), c in per.items():
    for n, v in c.items():
        agg[k][n].append(v)
print(f"# {f}")
print(f"{'kernel':70s} {'launches':>8s} {'cycles':>10s} {'mfma_util':>9s} {'bf16 flop/cyc':>13s} {'of_peak':>7s}")
rows = []
for k, c in agg.items():
    n = len(c.get("GRBM_GUI_ACTIVE", []))
    if not n:
        continue
    cyc = sum(c["GRBM_GUI_ACTIVE"]) / 8
    busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
    mops = sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0]))
    rows.append((cyc, k, n, cyc / n, busy / (cyc * 1024) if cyc else 0, mops * 512 / cyc if cyc else 0))
for cyc, k, n, cpl, util, fpc in sorted(rows, reverse=True)[:14]:
    print(f"{k[:70]:70s} {n:8d} {cpl:10.0f} {util:9.3f} {fpc:13.0f} {fpc / (1024 * 1024):7.3f}")
