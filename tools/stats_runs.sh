#!/bin/bash
# so3_angle_stats under a kernel trace for four inputs: 10 classes / one class, uniform angles / K4's angles between random rotations
# (on the GPU box:  bash tools/stats_runs.sh > gpurun_out/stats_runs.txt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "10 uniform" "10 haar" "1 uniform" "1 haar"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_$1_$2 -- python3 tools/stats_loop.py 20 $1 $2 2>&1 < /dev/null | grep "so3_angle_stats"
  f=$(find gpurun_out/stats_$1_$2 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep "k_stats" "$f" < /dev/null | sed 's/(anonymous namespace):://g; s/([^)]*)"/"/' | cut -d, -f1-4
done
