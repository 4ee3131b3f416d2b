#!/bin/bash
# The measurement pass behind profiles/rNN_*: usage  tools/profile_round.sh r03   (on the GPU box, from the repository root).
# Every rocprofv3 call has the program itself after `--`; counters are collected in their own passes (--pmc without trace domains).
set -e
tag=${1:-r03}
root=$(pwd)
P=$root/gpurun_out/prof_$tag
mkdir -p $P
export TMPDIR=/tmp
echo "== kernel trace of the default bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-secondary > $P/bench_under_trace.json 2> $P/kt.log
echo "== PMC passes (HBM traffic of K1)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2> $P/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2> $P/pmc_write.log
echo "== SQ counters of K1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $P/sq1 -- python3 tools/k1_loop.py 20 > /dev/null 2> $P/sq1.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $P/sq2 -- python3 tools/k1_loop.py 20 > /dev/null 2> $P/sq2.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $P/sq3 -- python3 tools/k1_loop.py 20 > /dev/null 2> $P/sq3.log
echo "== every kernel, traced and untraced"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_all -- python3 tools/bench_all.py > $P/bench_all_traced.txt 2> $P/kt_all.log
python3 tools/bench_all.py > $P/bench_all_untraced.txt 2>&1
echo "== bench lines"
python3 bench.py --steps 20 --warmup 5 > $P/bench_driver_command.json 2> $P/bench_driver_command.err
python3 bench.py > $P/bench_default.json 2> $P/bench_default.err
echo "== worst case, engine, mirror, certificate"
python3 -u tools/k1_hard_rows.py > $P/hard_rows.txt 2>&1
tools/ubench/k1_anatomy 1000 > $P/anatomy.txt 2>&1
python3 tools/mirror_modes.py > $P/mirror_modes.txt 2>&1
python3 tools/py_overhead.py > $P/py_overhead.txt 2>&1
python3 -m pytest tests/test_gpu_certificate_search.py -q -s -m gpu > $P/certificate_search.txt 2>&1
echo "== done"
