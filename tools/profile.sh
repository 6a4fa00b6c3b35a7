#!/bin/bash
# Profiling recipe for the GPU box (run through gpurun from the repo root):
#   tools/profile.sh <tag> [bench.py args...]
# 1) rocprofv3 --kernel-trace --stats of a short bench.py run  -> gpurun_out/prof_<tag>/trace
# 2) separate --pmc passes (never combined with other trace domains) -> gpurun_out/prof_<tag>/pmc_*
# 3) tools/pmc_summary.py condenses both into gpurun_out/prof_<tag>/summary.txt (copy that to profiles/).
set -u
TAG=${1:-run}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras "$@" > $OUT/bench_under_trace.json 2> $OUT/trace.err || exit 1
i=0
for PMC in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc_$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $OUT/pmc_$i.err || exit 1
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
