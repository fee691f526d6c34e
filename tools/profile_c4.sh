#!/bin/bash
# Profiles of the C4 bench on the GPU box (run through gpurun from the repo root):
#   tools/profile_c4.sh TAG [passes]     passes: any of "trace sq fetch write" (default: all four); extra bench.py arguments in $BENCH_ARGS
# Writes rocprofv3 output under gpurun_out/prof_TAG/, and the condensed files profiles/ keeps:
#   gpurun_out/prof_TAG/TAG_kernel_stats.csv, TAG_pmc_summary.txt, TAG_walk_traffic.json
# PMC passes are separate runs (counters alone, no trace domains), one per counter group (MI355X_MICROARCH.md, PMC slots).
set -e
TAG=$1; shift
PASSES=${@:-trace sq fetch write}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
for p in $PASSES; do
  case $p in
    trace) rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $O/trace.log 2>&1
           cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/${TAG}_kernel_stats.csv ;;
    sq)    rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $O/pmc_sq -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $O/pmc_sq.log 2>&1 ;;
    fetch) rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $O/pmc_fetch.log 2>&1 ;;
    write) rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $O/pmc_write.log 2>&1 ;;
  esac
  echo "pass $p done"
done
D=""
for p in pmc_sq pmc_fetch pmc_write; do [ -d $O/$p ] && D="$D $O/$p"; done
[ -n "$D" ] && python3 $R/tools/summarize_pmc.py $O/${TAG}_pmc_summary.txt $O/${TAG}_walk_traffic.json $D
# keep the merge small: drop the raw rocprofv3 trees, keep logs and the condensed files
rm -rf $O/trace $O/pmc_sq $O/pmc_fetch $O/pmc_write
ls -la $O
