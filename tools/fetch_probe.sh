#!/bin/bash
# FETCH_SIZE of the walk kernels for one bench configuration: tools/fetch_probe.sh TAG [bench args...]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fetch_$TAG
mkdir -p $O
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-accuracy "$@" > $O/log.txt 2>&1
python3 $R/tools/summarize_pmc.py $O/summary.txt $O/traffic.json $O/pmc_fetch > /dev/null 2>&1 || true
grep "k_walk_group2" $O/summary.txt | grep "dispatch " | cut -c1-140
rm -rf $O/pmc_fetch
