"""How many launches of a kernel run side by side: reads a rocprofv3 --kernel-trace CSV (Kernel_Name, Start_Timestamp,
End_Timestamp) and prints, per kernel name, the launches, their mean duration and the time-weighted mean number in flight."""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"][:60]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for name, iv in sorted(rows.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    if len(iv) < 4:
        continue
    iv = iv[len(iv) // 4:]                       # (skip the warm-up quarter)
    ev = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    busy = area = 0
    cur, last, peak = 0, ev[0][0], 0
    for t, d in ev:
        if cur > 0:
            busy += t - last
            area += cur * (t - last)
        cur += d
        peak = max(peak, cur)
        last = t
    span = ev[-1][0] - ev[0][0]
    print("%-60s n %5d  mean %9.3f ms  in flight while any runs: mean %5.2f peak %3d  busy %5.1f %% of the span (%.1f ms)" %
          (name, len(iv), sum(e - s for s, e in iv) / len(iv) / 1e6, area / max(busy, 1), peak, 100.0 * busy / max(span, 1), span / 1e6))
