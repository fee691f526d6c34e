#!/bin/bash
# per-dispatch kernel trace of tools/probe_one_task.py (one task over RCCL): busy / idle time of the GPU per step
#   tools/trace_one_task.sh [log2n] [pmgrid]  ->  gpurun_out/trace_one_task/dispatches.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_one_task
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/tools/probe_one_task.py ${1:-23} ${2:-256} nokept > $O/log.txt 2>&1
F=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 - "$F" > $O/dispatches.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last step: from the last k_minmax (ngravs_dd_local_extent) to the end
starts=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('k_minmax')]
i0=starts[-1]
t0=int(rows[i0]['Start_Timestamp']); prev_end=t0; busy=0; gaps=[]
for r in rows[i0:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    gap=(s-prev_end)/1e3
    if gap>20: gaps.append((gap,(s-t0)/1e6,r['Kernel_Name'][:50]))
    busy+=(e-max(s,prev_end)) if e>prev_end else 0
    print("%9.3f ms  %8.1f us  gap %7.1f us  %s"%((s-t0)/1e6,(e-s)/1e3,gap,r['Kernel_Name'][:70]))
    prev_end=max(prev_end,e)
tot=(prev_end-t0)/1e6
print("STEP: %.3f ms from first to last kernel, GPU busy %.3f ms, idle %.3f ms in %d gaps > 20 us"%(tot,busy/1e6,tot-busy/1e6,len(gaps)))
for g in sorted(gaps,reverse=True)[:25]: print("  gap %8.1f us before %s at %.3f ms"%(g[0],g[2],g[1]))
PY
rm -rf $O/t
tail -32 $O/dispatches.txt
