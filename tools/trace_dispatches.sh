#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_build
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-accuracy > $O/log.txt 2>&1
F=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 - "$F" > $O/build_dispatches.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    n=r['Kernel_Name']
    if any(k in n for k in ('k_tb_','k_split','k_link','k_moments','k_scan_blocks','k_level','k_keys','k_gather','rocprim','k_minmax','k_finish','k_cic','k_force_mesh','k_green','fft','k_walk')):
        print("%10.3f ms  %8.1f us  grid=%s wg=%s  %s"%((int(r['Start_Timestamp'])-t0)/1e6,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Grid_Size_X',r.get('Grid_Size','?')),r.get('Workgroup_Size_X',r.get('Workgroup_Size','?')),n[:60]))
PY
rm -rf $O/t
tail -130 $O/build_dispatches.txt
