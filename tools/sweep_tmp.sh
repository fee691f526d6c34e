mkdir -p gpurun_out/r2d
for t in "walk_sg=1" "walk_sg=2" "walk_sg=3" "walk_sg=4" "walk_sg=8"; do
  timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --tune $t > gpurun_out/r2d/b_$t.json 2>/dev/null
  python -c "
import json,sys
d=json.load(open('gpurun_out/r2d/b_$t.json'))
print('$t', round(d['ms_per_step'],1), d['roofline']['split_walk'], round(d['config']['walk_list_entries_per_group']))"
done
