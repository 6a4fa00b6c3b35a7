#!/bin/bash
# Kernel traces (rocprofv3 --kernel-trace --stats, no counters) of the other shapes on the GPU box:
#   tools/profile_traces.sh  -> gpurun_out/prof_<shape>/trace, gpurun_out/prof_<shape>/bench.json
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for P in map-ont:9400 dense:100 dense:1000; do
  N=${P%%:*}; K=${P##*:}; D=$R/gpurun_out/prof_${N}_$K
  mkdir -p $D
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $R/bench.py --preset $N --reads-per-gpu $K --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $D/bench.json 2> $D/err.log || exit 1
done
