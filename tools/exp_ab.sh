#!/bin/bash
# Interleaved A/B of several builds of the library on one GPU box, with a parity check of each build first:
#   tools/exp_ab.sh <rounds> <name1> <name2> ...      (names of minimap2_chaindp_amd/csrc/variants/libchaindp_<name>.so; "main" = the
#   library in csrc/).  PARITY=0 skips the parity runs; BENCH_ARGS adds bench.py arguments.  Writes gpurun_out/exp_ab.log.
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; shift
mkdir -p $R/gpurun_out
LOG=$R/gpurun_out/exp_ab.log
: > $LOG
# "main" = the library in csrc/; "quad" = the same library with CHAINDP_QUAD=1 (one-table batches of ordinary units four per wave)
libof() { if [ "$1" = main ] || [ "$1" = quad ]; then echo $R/minimap2_chaindp_amd/csrc/libchaindp_hip.so; else echo $R/minimap2_chaindp_amd/csrc/variants/libchaindp_$1.so; fi; }
envof() { if [ "$1" = quad ]; then echo CHAINDP_QUAD=1; else echo CHAINDP_UNUSED=1; fi; }
if [ "${PARITY:-1}" = 1 ]; then
for v in "$@"; do
  if env $(envof $v) CHAINDP_LIB=$(libof $v) timeout -k 10 400 python -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_fuzz.py $R/tests/test_gpu_fullsize.py -x -q -m gpu > $R/gpurun_out/parity_$v.log 2>&1; then echo "parity $v ok: $(tail -1 $R/gpurun_out/parity_$v.log)" >> $LOG; else echo "parity $v FAILED: $(tail -5 $R/gpurun_out/parity_$v.log | tr '\n' ' ')" >> $LOG; fi
done
fi
for i in $(seq 1 $N); do
  for v in "$@"; do
    env $(envof $v) CHAINDP_LIB=$(libof $v) timeout -k 10 200 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']/1e9,3), {k: round(v,4) for k,v in d['kernel_ms'].items()})" >> $LOG
  done
done
cat $LOG
