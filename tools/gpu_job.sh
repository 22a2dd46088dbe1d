#!/bin/bash
# One parametrised GPU-box job list (replaces the per-experiment run_r*.sh scripts of rounds 1-4).
#   gpurun --timeout 900 -- 'bash tools/gpu_job.sh <tag> <step> [<step> ...]'
# steps: micro:<name>   build tools/<name>.hip and run it            -> gpurun_out/<tag>_<name>.txt
#        tests[:<expr>] pytest -m gpu [-k expr]                      -> gpurun_out/<tag>_tests.log
#        bench[:<args>] python bench.py <args> (commas = spaces)     -> gpurun_out/<tag>_bench.json / .log
#        chain          tools/tools_chain.py (ECHOGLAD_LIB honoured) -> gpurun_out/<tag>_chain.txt
#        py:<script>[:args]  python tools/<script> args              -> gpurun_out/<tag>_<script>.txt
#        trace:train|infer   kernel sequence of one step from a rocprofv3 --kernel-trace
#        pmc:<c1>,<c2>,..[:train]  one rocprofv3 --pmc pass, per-kernel means of the counters
#        ab:<variant>:<variant>[:...][:reps=N] same-box A/B of library builds ("base" = shipped) with tools_chain.py + bench --steps 300
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" == "$step" ] && rest=""
  case $kind in
    micro)
      hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/$rest tools/$rest.hip 2>/dev/null && timeout 300 /tmp/$rest > gpurun_out/${tag}_$rest.txt 2>&1
      tail -40 gpurun_out/${tag}_$rest.txt ;;
    tests)
      if [ -n "$rest" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -k "$rest" > gpurun_out/${tag}_tests.log 2>&1
      else timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; fi
      tail -5 gpurun_out/${tag}_tests.log ;;
    bench)
      timeout 900 python bench.py ${rest//,/ } > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.log
      tail -c 1500 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.log ;;
    chain)
      timeout 300 python tools/tools_chain.py > gpurun_out/${tag}_chain.txt 2>&1; cat gpurun_out/${tag}_chain.txt ;;
    py)
      script=${rest%%:*}; args=${rest#*:}; [ "$args" == "$rest" ] && args=""
      timeout 900 python tools/$script ${args//,/ } > gpurun_out/${tag}_${script%.py}.txt 2>&1; tail -40 gpurun_out/${tag}_${script%.py}.txt ;;
    trace)
      # trace:train|infer   kernel sequence of one step (start, duration, gap in front) from a rocprofv3 --kernel-trace of bench.py
      rm -rf /tmp/trace_$tag
      if [ "$rest" == "train" ]; then
        timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$tag -- python3 bench.py --mode train --batch ${EG_B:-32} --steps 5 --warmup 2 --no-other-configs --no-graph-replay > gpurun_out/${tag}_trace_run.log 2>&1
        python3 tools/trace_seq.py $(find /tmp/trace_$tag -name "*kernel_trace.csv" | head -1) > gpurun_out/${tag}_trace_train.txt 2>&1
        python3 tools/trace_gaps.py $(find /tmp/trace_$tag -name "*kernel_trace.csv" | head -1) >> gpurun_out/${tag}_trace_train.txt 2>&1
        tail -5 gpurun_out/${tag}_trace_train.txt
      else
        timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$tag -- python3 bench.py --steps 60 --warmup 10 --repeats 0 --no-cpu-baseline --no-other-configs > gpurun_out/${tag}_trace_run.log 2>&1
        python3 tools/trace_infer_seq.py $(find /tmp/trace_$tag -name "*kernel_trace.csv" | head -1) > gpurun_out/${tag}_trace_infer.txt 2>&1
        tail -12 gpurun_out/${tag}_trace_infer.txt
      fi ;;
    pmc)
      # pmc:<counter>,<counter>,...[:train]  one rocprofv3 --pmc pass over the default bench command (or the training step), per-kernel means
      IFS=: read -r ctrs mode <<< "$rest"
      cmd="bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --repeats 0"
      [ "$mode" == "train" ] && cmd="bench.py --mode train --batch 32 --steps 5 --warmup 2 --no-other-configs --no-graph-replay"
      rm -rf /tmp/pmc_$tag
      timeout 600 rocprofv3 --pmc ${ctrs//,/ } --output-format csv -d /tmp/pmc_$tag -- python3 $cmd > gpurun_out/${tag}_pmc_run.log 2>&1
      python3 - <<PY > gpurun_out/${tag}_pmc_${ctrs%%,*}.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("/tmp/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "").replace("eg::", "")
        a = agg[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in sorted(agg.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
    if not k.startswith("k_"): continue
    print(k[:70], {c: round(v / n, 1) for c, (v, n) in sorted(cs.items())}, "launches", max(n for _, n in cs.values()))
PY
      head -12 gpurun_out/${tag}_pmc_${ctrs%%,*}.txt ;;
    ab)
      # ab:<variant>:<variant>[:...][:reps=N]  ("base" = the shipped library); every variant in turn, N rounds (default 2)
      reps=2; vars=""
      IFS=: read -ra parts <<< "$rest"
      for p in "${parts[@]}"; do case $p in reps=*) reps=${p#reps=} ;; *) vars="$vars $p" ;; esac; done
      for r in $(seq $reps); do for v in $vars; do
        lib=echoglad_amd/lib/libechoglad_hip.$v.so; [ "$v" == "base" ] && lib=echoglad_amd/lib/libechoglad_hip.so
        echo "== $v round $r" >> gpurun_out/${tag}_ab.txt
        ECHOGLAD_LIB=$lib timeout 300 python tools/tools_chain.py >> gpurun_out/${tag}_ab.txt 2>&1
        ECHOGLAD_LIB=$lib timeout 300 python bench.py --steps 300 --warmup 20 --repeats 0 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step ms', d['ms_per_step'], 'frac', d['roofline']['frac'])" >> gpurun_out/${tag}_ab.txt 2>&1
      done; done
      cat gpurun_out/${tag}_ab.txt ;;
  esac
done
