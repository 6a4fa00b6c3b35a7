#!/bin/bash
# Interleaved A/B of several builds of the library on one GPU box, with a parity check of each build first:
#   tools/exp_ab.sh <rounds> <name1> <name2> ...      (names of minimap2_chaindp_amd/csrc/variants/libchaindp_<name>.so)
# Writes gpurun_out/exp_ab.log.
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; shift
mkdir -p $R/gpurun_out
LOG=$R/gpurun_out/exp_ab.log
: > $LOG
for v in "$@"; do
  L=$R/minimap2_chaindp_amd/csrc/variants/libchaindp_$v.so
  if CHAINDP_LIB=$L timeout -k 10 300 python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_fuzz.py -x -q -m gpu > $R/gpurun_out/parity_$v.log 2>&1; then echo "parity $v ok: $(tail -1 $R/gpurun_out/parity_$v.log)" >> $LOG; else echo "parity $v FAILED: $(tail -3 $R/gpurun_out/parity_$v.log | tr '\n' ' ')" >> $LOG; fi
done
for i in $(seq 1 $N); do
  for v in "$@"; do
    L=$R/minimap2_chaindp_amd/csrc/variants/libchaindp_$v.so
    CHAINDP_LIB=$L timeout -k 10 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']/1e9,3), {k: round(v,4) for k,v in d['kernel_ms'].items()})" >> $LOG
  done
done
cat $LOG
