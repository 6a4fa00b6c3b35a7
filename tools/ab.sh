#!/bin/bash
# A/B timing of two builds of libchaindp_hip.so on the SAME GPU, interleaved (cdna guide rule 24):
#   tools/ab.sh <libA.so> <libB.so> [rounds] [bench args...]
# Prints chain_dp kernel ms per round for each build.
A=$1; B=$2; N=${3:-3}; shift 3 || true
for i in $(seq 1 $N); do
  for L in $A $B; do
    CHAINDP_LIB=$L python bench.py --steps 10 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.path.basename('$L'), round(d['value']/1e9,3), {k: round(v,3) for k,v in d['kernel_ms'].items()})"
  done
done
