#!/bin/bash
# Launch chains x hardware queues (run through gpurun): the 4 -> 5 chain cliff of round 2 / 3 was measured with 8 hardware
# queues; chains that share a queue serialise.  One line per (queues, chains): first window / further windows.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
out=$O/groups_sweep.txt; : > $out
for q in 8 16 24 32; do
  for g in 3 4 5 6 8; do
    line=$(GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --groups $g --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1)
    echo "queues $q chains $g: $(echo "$line" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), [round(w) for w in d.get('repeats', {}).get('values', [])], d['config'].get('step_path', {}).get('hw_queues'))" 2>/dev/null || echo "$line" | cut -c1-200)" >> $out
  done
done
cat $out
