#!/bin/bash
# Dense-shape probe on the GPU box: bench.py --preset dense at a few batch sizes and ring capacities.
#   tools/dense_probe.sh <tag>     -> gpurun_out/dense_<tag>.log
set -u
TAG=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/dense_$TAG.log
mkdir -p $R/gpurun_out
: > $OUT
for READS in 100 1000; do
  for RING in ${RINGS:-128 512}; do
    echo "== dense reads=$READS ring=$RING" >> $OUT
    timeout -k 10 300 python3 $R/bench.py --preset dense --reads-per-gpu $READS --ring $RING --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>> $OUT | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'G_anchors_per_s': round(d['value']/1e9,4), 'ms_per_step': round(d['ms_per_step'],3), 'anchors': d['config']['anchors_per_gpu'], 'units': d['config']['units_per_gpu'], 'kernel_ms': {k: round(v,3) for k,v in d['kernel_ms'].items()}}))" >> $OUT || exit 1
  done
done
cat $OUT
