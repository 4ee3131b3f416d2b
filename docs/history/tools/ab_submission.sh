#!/bin/bash
# graph or eager submission for a given number of steps?  alternating runs of bench.py on one box.  usage: ab_submission.sh [steps]
steps=${1:-20}
for i in 1 2 3 4; do
  for mode in "" "--eager"; do
    python3 bench.py --gpus 1 --steps $steps --warmup 5 --no-cpu-baseline --no-secondary $mode 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('%-8s host %.2f us  events %.2f us  frac %.3f  frac_host %.3f' % ('$mode' or 'graph', d['ms_per_step']*1e3, d['ms_per_step_events']*1e3, d['roofline']['frac'], d['roofline']['frac_host_clock']))
"
  done
done
