"""Diagnostic: GPU idle time inside a training step from a rocprofv3 --kernel-trace CSV (kernel start / end timestamps).
usage: python3 tools/trace_gaps.py <kernel_trace.csv> [steps]
Prints the union of busy intervals, the idle time between kernels and the largest gaps with the kernels on either side."""
import csv
import sys


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    # the timed region: the last `steps` occurrences of the optimizer's first kernel bracket the steps; simpler: take the window
    # between the first and the last k_gcn_layer_ps launch of the trace's second half
    names = [e[2] for e in ev]
    first_layer = [i for i, n in enumerate(names) if "k_gcn_layer_ps" in n and ", 1," in n]
    per_step = 3
    n_steps = len(first_layer) // per_step
    use = min(steps, n_steps - 1)
    lo = first_layer[(n_steps - use - 1) * per_step]
    hi = first_layer[(n_steps - 1) * per_step]
    win = ev[lo:hi]
    t0, t1 = win[0][0], ev[hi][0]
    busy, gaps, end = 0, [], win[0][0]
    for i, (s, e, n) in enumerate(win):
        if s > end:
            gaps.append((s - end, win[i - 1][2][:60] if i else "", n[:60]))
            busy += e - s
        else:
            busy += max(0, e - max(s, end))
        end = max(end, e)
    total = t1 - t0
    print(f"{use} steps: {total / use / 1e6:.3f} ms per step, busy {busy / use / 1e6:.3f} ms, idle {(total - busy) / use / 1e6:.3f} ms, "
          f"{len(win) / use:.0f} kernels per step")
    small = sum(e - s for s, e, n in win if e - s < 20000)
    print(f"kernels shorter than 20 us: {sum(1 for s, e, n in win if e - s < 20000) / use:.0f} per step, {small / use / 1e6:.3f} ms")
    gaps.sort(reverse=True)
    for g, a, b in gaps[:25]:
        print(f"  gap {g / 1e3:8.1f} us   after {a}   before {b}")
    hist = {}
    for g, a, b in gaps:
        k = (a.split("(")[0][:40], b.split("(")[0][:40])
        hist[k] = hist.get(k, 0) + g
    print("idle by (kernel before, kernel after), ms per step:")
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:20]:
        print(f"  {v / use / 1e6:7.3f}  {k[0]}  ->  {k[1]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5)
