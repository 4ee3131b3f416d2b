#!/bin/bash
# The measurement pass behind profiles/rNN_*: usage  tools/profile_round.sh r04 [a|b]   (on the GPU box, from the repository root).
# Every rocprofv3 call has the program itself after `--`; counters are collected in their own passes (--pmc without trace domains).
# Part a: traces, counters, bench lines.  Part b: A/B against the previous round's library (build/variants/libso3proj_r03.so, built by
# tools/build_variant.sh from the previous round's sources), hard rows, engine anatomy, mirror, certificate search.
set -e
tag=${1:-r04}
part=${2:-ab}
root=$(pwd)
P=$root/gpurun_out/prof_$tag
mkdir -p $P
export TMPDIR=/tmp
if [[ $part == *a* ]]; then
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
fi
if [[ $part == *b* ]]; then
echo "== A/B against the previous round's library: one device, builds interleaved"
prev=build/variants/libso3proj_r03.so
if [ -f $prev ]; then
    cp poseestimation_amd/libso3proj.so build/variants/libso3proj_$tag.so
    AB_ROUNDS=6 python3 tools/ab_k1_graph.py $prev build/variants/libso3proj_$tag.so > $P/ab_k1_graph.txt 2>&1
    AB_HARD=1 AB_ROUNDS=5 python3 tools/ab_v2.py $prev build/variants/libso3proj_$tag.so > $P/ab_kernels_hard_rows.txt 2>&1
fi
echo "== engine anatomy, mirror, certificate"
[ -x tools/ubench/k1_anatomy ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -o tools/ubench/k1_anatomy tools/ubench/k1_anatomy.hip
tools/ubench/k1_anatomy 1000 > $P/anatomy.txt 2>&1 || echo "k1_anatomy failed (see $P/anatomy.txt)"
python3 tools/mirror_modes.py > $P/mirror_modes.txt 2>&1
python3 tools/py_overhead.py > $P/py_overhead.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_stats -- python3 tools/stats_loop.py > $P/stats_loop.txt 2>&1
python3 -m pytest tests/test_gpu_certificate_search.py -q -s -m gpu > $P/certificate_search.txt 2>&1
fi
echo "== done"
