#!/bin/bash
# A second set of SQ counters for the evaluation kernel of the C4 bench (one rocprofv3 --pmc pass; counters alone, no trace domains):
#   tools/pmc_extra.sh TAG  ->  gpurun_out/TAG_pmc_extra.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf /tmp/pmcx
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES -d /tmp/pmcx -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-accuracy > /tmp/pmcx.log 2>&1
python3 $R/tools/summarize_pmc.py /tmp/pmcx.txt /tmp/pmcx.json /tmp/pmcx > /dev/null 2>&1
grep "k_eval_ring.* dispatch" /tmp/pmcx.txt > $O/$1_pmc_extra.txt
cat $O/$1_pmc_extra.txt
