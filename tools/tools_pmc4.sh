#!/bin/bash
export TMPDIR=/tmp
timeout 40 rocprofv3 -L > /tmp/counters_raw.txt 2>&1; grep -oE "\b(TCP|TA|TD|TCC)_[A-Za-z0-9_]*" /tmp/counters_raw.txt | sort -u > /tmp/counters.txt
echo "n counters: $(wc -l < /tmp/counters.txt)"; grep -E "UTCL|TLB|PENDING|TCP_TCC_READ|TA_BUSY|TA_.*STALL|LATENCY" /tmp/counters.txt | tr '\n' ' '; echo
for pmc in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $pmc | tr ' ' '_' | cut -c1-60)
  timeout 90 rocprofv3 --pmc $pmc --output-format csv -d /tmp/p4_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --kernel-iters 3 > /tmp/p4_$n.log 2>&1 || { echo "FAILED: $pmc"; tail -2 /tmp/p4_$n.log; }
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
