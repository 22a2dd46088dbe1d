"""Diagnostic: the kernel sequence of inference steps (hipGraph replays) from a rocprofv3 --kernel-trace CSV: what runs between
two fused last-layer launches, with durations and gaps.  usage: python3 tools/trace_infer_seq.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
cls = [i for i, e in enumerate(ev) if "k_gcn_layer_ps<true" in e[2]]
# take a window in the middle of the run: 3 consecutive steps
mid = len(cls) // 2
lo, hi = cls[mid], cls[mid + 3]
end = ev[lo][1]
t0 = ev[lo][0]
for s, e, n in ev[lo + 1:hi + 1]:
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {max(0, s - end) / 1e3:6.1f}  {n.replace('void ', '')[:90]}")
    end = max(end, e)
steps = [(ev[cls[i + 1]][1] - ev[cls[i]][1]) / 1e3 for i in range(mid - 10, mid + 10)]
print("step-to-step (end of the last-layer kernel), us:", [round(x, 1) for x in steps])
