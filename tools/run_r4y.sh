#!/bin/bash
# round 4, after the step-time work: full GPU suite, default bench, then the r04 profiles on the frozen sources
OUT=gpurun_out
python -m pytest tests -x -q -m gpu > $OUT/r4y_pytest.log 2>&1; tail -3 $OUT/r4y_pytest.log
python3 bench.py > $OUT/r4y_bench.json 2> $OUT/r4y_bench.err; python3 -c "
import json; d=json.loads(open('$OUT/r4y_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('other_configs',{}).get('cfg4_train',{}).get('ms_per_step'))"
bash tools/run_r4n.sh > $OUT/r4y_profiles.log 2>&1
tail -30 $OUT/r4y_profiles.log
