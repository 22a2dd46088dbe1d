#!/bin/bash
# One parametrised GPU-box job list (replaces the per-experiment run_r*.sh scripts of rounds 1-4).
#   gpurun --timeout 900 -- 'bash tools/gpu_job.sh <tag> <step> [<step> ...]'
# steps: micro:<name>   build tools/<name>.hip and run it            -> gpurun_out/<tag>_<name>.txt
#        tests[:<expr>] pytest -m gpu [-k expr]                      -> gpurun_out/<tag>_tests.log
#        bench[:<args>] python bench.py <args> (commas = spaces)     -> gpurun_out/<tag>_bench.json / .log
#        chain          tools/tools_chain.py (ECHOGLAD_LIB honoured) -> gpurun_out/<tag>_chain.txt
#        py:<script>[:args]  python tools/<script> args              -> gpurun_out/<tag>_<script>.txt
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
