set -e
export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "statistics or angle" > $O/tests_stats.log 2>&1 || { tail -40 $O/tests_stats.log; exit 1; }
tail -3 $O/tests_stats.log
timeout -k 10 120 python3 tools/stats_loop.py 50 10 | tee $O/stats_loop_10.txt
timeout -k 10 120 python3 tools/stats_loop.py 50 1 | tee $O/stats_loop_1.txt
true
true
true
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -4 $O/tests.log
