# Per-kernel times of seed collection on the large dump under different limits of the LDS sort (GPU box, via gpurun).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, env...
  name=$1; shift
  ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sp_$name -- python3 $R/tools/seed_scale_run.py 3 ${MULT:-1} ) > $R/gpurun_out/sp_$name.log 2>&1
  tail -1 $R/gpurun_out/sp_$name.log
  f=$(find $R/gpurun_out/sp_$name -name '*kernel_stats.csv' | sort | tail -1)
  python3 - "$f" <<'PY'
import csv,sys
for row in csv.DictReader(open(sys.argv[1])):
    nm=row["Name"].split("(")[0]
    if "seed" in nm: print(f'   {nm:34s} calls {row["Calls"]:>3s} avg {float(row["AverageNs"])/1e3:9.1f} us total {float(row["TotalDurationNs"])/1e6:8.2f} ms')
PY
}
run default X=1
run exact CHAINDP_SEED_FORCE_EXACT=1
if [ "${1:-}" = "limits" ]; then
run m8192 CHAINDP_SEED_MAX_N=8192,8192
run m8192_exact CHAINDP_SEED_MAX_N=8192,8192 CHAINDP_SEED_FORCE_EXACT=1
run m4096 CHAINDP_SEED_MAX_N=4096,4096
run m4096_exact CHAINDP_SEED_MAX_N=4096,4096 CHAINDP_SEED_FORCE_EXACT=1
fi
