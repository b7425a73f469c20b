#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of the 4-lane bench: how the decode loops' kernels fare while another lane's encoder GEMM is running.
For every decode kernel launch: duration and the gap to the previous kernel of the same queue, split by whether an encoder GEMM
(gemm_bf16_tile256_kernel) or the encoder's flash attention was executing on ANOTHER queue at its start.
usage: python tools/enc_overlap_trace.py <dir with *kernel_trace.csv>"""
import bisect, collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
enc = [(s, e, q) for s, e, q, n in rows if "tile256" in n or "attn_flash" in n]
enc_starts = [s for s, _, _ in enc]
def encoder_active(t, q):
    i = bisect.bisect_right(enc_starts, t)
    for j in range(max(0, i - 8), i):
        s, e, qq = enc[j]
        if s <= t < e and qq != q:
            return True
    return False
last_end = {}
stats = collections.defaultdict(lambda: [0, 0.0, 0.0])      # (short name, during encoder) -> [n, sum duration, sum gap]
for s, e, q, n in rows:
    short = n.split("(")[0].split("<")[0].replace("void ", "").replace("wx::", "")[:40]
    gap = s - last_end.get(q, s)
    last_end[q] = e
    if "tile256" in n or "attn_flash" in n or gap > 200000:      # (gaps > 0.2 ms: the host, not the queue)
        continue
    k = (short, encoder_active(s, q))
    st = stats[k]; st[0] += 1; st[1] += e - s; st[2] += max(gap, 0)
names = sorted({k[0] for k in stats}, key=lambda nm: -sum(stats[(nm, b)][1] for b in (False, True) if (nm, b) in stats))[:12]
print(f"{'kernel':42s} {'n (no enc)':>10s} {'dur us':>8s} {'gap us':>8s} | {'n (enc)':>9s} {'dur us':>8s} {'gap us':>8s}")
for nm in names:
    a, b = stats.get((nm, False), [0, 0, 0]), stats.get((nm, True), [0, 0, 0])
    fa = lambda x: (x[1] / x[0] / 1e3, x[2] / x[0] / 1e3) if x[0] else (0, 0)
    print(f"{nm:42s} {a[0]:10d} {fa(a)[0]:8.2f} {fa(a)[1]:8.2f} | {b[0]:9d} {fa(b)[0]:8.2f} {fa(b)[1]:8.2f}")
tot = {b: sum(v[1] + v[2] for k, v in stats.items() if k[1] == b) / max(1, sum(v[0] for k, v in stats.items() if k[1] == b)) / 1e3 for b in (False, True)}
print(f"mean (duration + gap) per decode launch: {tot[False]:.2f} us without an encoder beside it, {tot[True]:.2f} us with one")
