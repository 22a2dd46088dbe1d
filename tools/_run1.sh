timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; cat gpurun_out/bench_default.json | cut -c1-400
timeout 1500 bash tools/profile_round.sh r01 2>&1 | tail -12
