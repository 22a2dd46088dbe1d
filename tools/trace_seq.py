"""Diagnostic: the kernel sequence of ONE training step from a rocprofv3 --kernel-trace CSV: start offset, duration, gap in front.
usage: python3 tools/trace_seq.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
names = [e[2] for e in ev]
first_layer = [i for i, n in enumerate(names) if "k_gcn_layer_ps" in n and ", 1," in n]
n_steps = len(first_layer) // 3
lo, hi = first_layer[(n_steps - 2) * 3], first_layer[(n_steps - 1) * 3]
t0, end = ev[lo][0], ev[lo][0]
for s, e, n in ev[lo:hi]:
    short = n.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", "")
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {max(0, s - end) / 1e3:7.1f}  {short[:110]}")
    end = max(end, e)
