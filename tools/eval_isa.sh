#!/bin/bash
# ISA of the ring-pool evaluation kernel's C4 instantiation k_eval_ring<2,true,true> (hipcc -S, no GPU needed): tools/eval_isa.sh [outdir]
O=${1:-/tmp/isa}; mkdir -p $O
R=$(cd $(dirname $0)/.. && pwd)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o $O/eval.s $R/gadget-2.0.7-ngravs_amd/csrc/kernels_eval.hip 2>&1 | grep -v "warning\|^$" | head
K=_Z11k_eval_ringILi2ELb1ELb1EE
a=$(grep -n "^$K" $O/eval.s | cut -d: -f1)
b=$(awk -v a=$a 'NR>a && /^\.Lfunc_end/{print NR; exit}' $O/eval.s)
sed -n "${a},${b}p" $O/eval.s > $O/k21.s
grep -A12 "\.name: *$K" $O/eval.s | grep "vgpr_count\|spill\|sgpr_count\|private_segment"
