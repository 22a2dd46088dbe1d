#!/bin/bash
# Everything profiles/<tag>_* of a round is made of, in one GPU-box job:  gpurun --timeout 2400 -- 'bash tools/round_profiles.sh r06'
#   <tag>_kernel_stats.csv, <tag>_pmc.json            inference step (tools/profile_round.sh)
#   <tag>_train_kernel_stats.csv, <tag>_train_pmc.json training step (tools/tools_train_profile.sh, tools/profile_train_pmc.sh)
#   <tag>_bench_default.json                           the default bench command's full result (--details inline)
TAG=${1:-r06}
export TMPDIR=/tmp
cd /tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
bash tools/profile_round.sh $TAG > gpurun_out/profiles/${TAG}_profile_round.log 2>&1
bash tools/profile_train_pmc.sh $TAG > gpurun_out/profiles/${TAG}_profile_train_pmc.log 2>&1
bash tools/tools_train_profile.sh 32 > gpurun_out/profiles/${TAG}_train_profile.txt 2>&1
cp gpurun_out/profiles/train_kernel_stats.csv gpurun_out/profiles/${TAG}_train_kernel_stats.csv
# the PMC summaries have to be where bench.py looks for them before the bench run below reports `traffic`
cp gpurun_out/profiles/${TAG}_pmc.json gpurun_out/profiles/${TAG}_train_pmc.json profiles/ 2>/dev/null
timeout 1200 python bench.py --details inline > gpurun_out/profiles/${TAG}_bench_default.json 2> gpurun_out/profiles/${TAG}_bench_default.log
tail -c 600 gpurun_out/profiles/${TAG}_bench_default.json; tail -12 gpurun_out/profiles/${TAG}_profile_train_pmc.log
