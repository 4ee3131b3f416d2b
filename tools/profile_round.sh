#!/bin/bash
# The measurement pass behind profiles/rNN_*: ONE build, ONE device, one call.   usage: tools/profile_round.sh r05   (on the GPU box, repository root)
# Every rocprofv3 call has the program itself after `--`; counters are collected in their own passes (--pmc, no trace domains).
# A progress line per section; raw output under gpurun_out/prof_<tag>/, what is committed is copied / condensed into profiles/ by
# tools/collect_profiles.sh <tag> afterwards (on the build container).
set -e
tag=${1:-r05}
root=$(pwd)
P=$root/gpurun_out/prof_$tag
mkdir -p $P
export TMPDIR=/tmp
say() { echo "== $(date +%H:%M:%S) $*"; }
say "device"; (rocm-smi --showclocks --showpower 2>/dev/null || true) > $P/device.txt 2>&1
say "bench lines (the driver's command, then the default one)"
python3 bench.py --steps 20 --warmup 5 > $P/bench_driver_command.json 2> $P/bench_driver_command.err
python3 bench.py > $P/bench_default.json 2> $P/bench_default.err
say "kernel trace of the default bench command"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -- python3 bench.py --steps 1000 --warmup 50 --no-cpu-baseline --no-secondary > $P/bench_under_trace.json 2> $P/kt.log
say "HBM traffic of K1 (two PMC passes)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2> $P/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /dev/null 2> $P/pmc_write.log
say "SQ counters of K1 (three passes)"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $P/sq1 -- python3 tools/k1_loop.py 20 > /dev/null 2> $P/sq1.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $P/sq2 -- python3 tools/k1_loop.py 20 > /dev/null 2> $P/sq2.log
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $P/sq3 -- python3 tools/k1_loop.py 20 > /dev/null 2> $P/sq3.log
say "SQ counters of K2 / K3 / K1+K4 (two passes each)"
for k in k2 k3 k14; do
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $P/sqa_$k -- python3 tools/kernel_loop.py $k 12 > /dev/null 2> $P/sqa_$k.log
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES --output-format csv -d $P/sqb_$k -- python3 tools/kernel_loop.py $k 12 > /dev/null 2> $P/sqb_$k.log
done
say "every kernel: untraced table, kernel trace, HBM traffic"
python3 tools/bench_all.py > $P/bench_all_untraced.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_all -- python3 tools/bench_all.py > $P/bench_all_traced.txt 2> $P/kt_all.log
mkdir -p $P/pmc_all
SO3_BENCH_QUICK=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_all/fetch -- python3 tools/bench_all.py > /dev/null 2> $P/pmc_all_fetch.log
SO3_BENCH_QUICK=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_all/write -- python3 tools/bench_all.py > /dev/null 2> $P/pmc_all_write.log
say "A/B against rounds 3, 4 and 5: K1 as a 500-launch graph and eager, the projection kernels, hard rows"
cp poseestimation_amd/libso3proj.so build/variants/libso3proj_$tag.so
libs=""; for r in r03 r04 r05; do [ -f build/variants/libso3proj_$r.so ] && libs="$libs build/variants/libso3proj_$r.so"; done
AB_ROUNDS=6 python3 tools/ab_k1_graph.py $libs build/variants/libso3proj_$tag.so > $P/ab_k1_graph.txt 2>&1
AB_ROUNDS=5 python3 tools/ab_v2.py $libs build/variants/libso3proj_$tag.so > $P/ab_kernels.txt 2>&1
[ -f build/variants/libso3proj_r04.so ] && AB_HARD=1 AB_ROUNDS=3 AB_ONLY=K1 python3 tools/ab_v2.py build/variants/libso3proj_r04.so build/variants/libso3proj_$tag.so > $P/ab_hard_rows.txt 2>&1
say "time against batch size"
python3 tools/size_ramp.py > $P/size_ramp.txt 2>&1
say "engine anatomy, statistics, mirror, certificate search"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -o tools/ubench/k1_anatomy tools/ubench/k1_anatomy.hip
tools/ubench/k1_anatomy 1000 > $P/anatomy.txt 2>&1 || echo "k1_anatomy failed (see $P/anatomy.txt)"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt_stats -- python3 tools/stats_loop.py > $P/stats_loop.txt 2>&1
python3 tools/mirror_modes.py > $P/mirror_modes.txt 2>&1
python3 tools/k1_two_streams.py > $P/k1_two_streams.txt 2>&1
python3 -m pytest tests/test_gpu_certificate_search.py -q -s -m gpu > $P/certificate_search.txt 2>&1
if [ "${SKIP_SEEDS:-0}" != "1" ]; then      # (~5 minutes; SKIP_SEEDS=1 leaves it to a call of its own: gpurun's limit is 20 minutes per call)
say "the certificate's search under sixty seeds (the shipped constants)"
python3 tools/search_seeds.py poseestimation_amd/libso3proj.so $(seq 101 160) > $P/search_seeds.txt 2>&1
fi
say "done"
