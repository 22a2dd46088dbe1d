#!/bin/bash
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TD|TCC|SQ|GRBM)_[A-Z0-9_a-z]*" | sort -u > /tmp/counters.txt
echo "n counters: $(wc -l < /tmp/counters.txt)"; grep -E "UTCL|TLB|PENDING|STALL|TCP_TCC|TA_BUSY|TA_.*STALL|LATENCY" /tmp/counters.txt | tr '\n' ' '; echo
for pmc in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum"; do
  n=$(echo $pmc | tr ' ' '_' | cut -c1-60)
  rocprofv3 --pmc $pmc --output-format csv -d /tmp/p4_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel-iters 3 > /tmp/p4_$n.log 2>&1 || tail -3 /tmp/p4_$n.log
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob("/tmp/p4_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_gcn_layer" in row["Kernel_Name"]:
            agg[row["Counter_Name"]][0]+=float(row["Counter_Value"]); agg[row["Counter_Name"]][1]+=1
for k,(v,c) in sorted(agg.items()): print(f"{k:45s} {v/c:.4g}")
PY
