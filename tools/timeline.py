#!/usr/bin/env python
"""Per-step GPU timeline from a rocprofv3 --kernel-trace CSV: kernel sequence of the last bench step with start
offsets, durations and the idle gaps between consecutive kernels.  usage: timeline.py <kernel_trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n.split("::")[-1][:48] if "rocprim" in n or "at::native" in n else n[:60]
# a step starts with preprocess_fwd_kernel
starts = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r["Kernel_Name"]]
i0, i1 = starts[-3], starts[-2]   # a timed step in the middle of the last ones
# include the fills that precede the step's preprocess (zero-init of outputs belongs to the previous backward)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end or e, e)
    busy += e - s
span = (int(rows[i1]["Start_Timestamp"]) - t0) / 1e3
print(f"step span {span:.1f} us, kernel busy {busy / 1e3:.1f} us, idle {span - busy / 1e3:.1f} us")
