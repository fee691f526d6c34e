#!/bin/bash
# A/B of two builds of libngravs_hip.so on ONE GPU box (the boxes of the pool differ by +-1 ms in the evaluation kernel):
#   tools/ab_so.sh TAG A.so B.so [rounds]   -> gpurun_out/TAG_ab.txt  (C4 bench, alternating; then the SQ counter pass of each)
# The two files are copied over gadget-2.0.7-ngravs_amd/libngravs_hip.so in turn; the last copy is B.
TAG=$1; A=$(realpath $2); B=$(realpath $3); N=${4:-2}
R=$GRAFT_REPO_ROOT
L=$R/gadget-2.0.7-ngravs_amd/libngravs_hip.so
O=$R/gpurun_out/${TAG}_ab.txt
: > $O
one() {
  cp $2 $L
  timeout -k 10 200 python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > /tmp/ab.json 2> /tmp/ab.err || { echo "$1 bench failed" >> $O; tail -3 /tmp/ab.err >> $O; return 1; }
  python3 - "$1" >> $O <<PY
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], "%.1f M/s" % (d["value"] / 1e6), "step %.2f ms" % d["ms_per_step"], "eval %.2f ms" % d["roofline"]["avg_launch_ms"], "trips/group %.1f" % d["config"]["walk_force_iters_per_group"], "rms %.3e" % d["accuracy"]["rms"])
PY
}
for i in $(seq $N); do one A $A && one B $B || exit 1; done
cd /tmp && export TMPDIR=/tmp
for v in A B; do
  [ $v = A ] && cp $A $L || cp $B $L
  rm -rf /tmp/pmc_$v
  rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_$v -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc_$v.log 2>&1
  python3 $R/tools/summarize_pmc.py /tmp/pmc_$v.txt /tmp/pmc_$v.json /tmp/pmc_$v > /dev/null 2>&1
  echo "== $v" >> $O
  grep "k_eval_ring.* dispatch" /tmp/pmc_$v.txt >> $O
done
cat $O
