#!/bin/bash
# What tools/profile_round.sh left under gpurun_out/prof_<tag>/ -> the small files committed under profiles/<tag>_*.   usage: tools/collect_profiles.sh r05
set -e
tag=${1:-r05}
P=gpurun_out/prof_$tag
python3 tools/summarize_profile.py $P $tag > /dev/null
python3 tools/summarize_sq.py $P $tag > /dev/null
python3 tools/summarize_sq_two_input.py $P $tag > /dev/null
python3 tools/summarize_pmc_all.py $P/pmc_all $tag > /dev/null
for f in bench_driver_command.json bench_default.json bench_under_trace.json; do cp $P/$f profiles/${tag}_$f; done
cp $P/bench_all_untraced.txt profiles/${tag}_bench_all_untraced.txt
cp $P/bench_all_traced.txt profiles/${tag}_bench_all.txt
cp $P/ab_k1_graph.txt profiles/${tag}_ab_k1.txt
{ echo "# K1 / K2 / K3 / K1+K4, builds of rounds 3, 4 and this one interleaved on ONE device (tools/ab_v2.py: best of 5 x 100 eager launches, us per 1M rows)"; cat $P/ab_kernels.txt; } > profiles/${tag}_ab_kernels.txt
[ -f $P/ab_hard_rows.txt ] && cp $P/ab_hard_rows.txt profiles/${tag}_hard_rows_ab.txt
cp $P/size_ramp.txt profiles/${tag}_size_ramp.txt
cp $P/anatomy.txt profiles/${tag}_k1_engine_copy.txt
cp $P/stats_loop.txt profiles/${tag}_angle_stats.txt
grep -h "k_stats" $P/kt_stats/*/*kernel_stats.csv >> profiles/${tag}_angle_stats.txt || true
cp $P/mirror_modes.txt profiles/${tag}_mirror_overhead.txt
[ -f $P/k1_two_streams.txt ] && cp $P/k1_two_streams.txt profiles/${tag}_k1_two_streams.txt
cp $P/certificate_search.txt profiles/${tag}_certificate_search.txt
[ -f $P/search_seeds.txt ] && cp $P/search_seeds.txt profiles/${tag}_search_seeds.txt
cp $P/device.txt profiles/${tag}_device.txt
ls profiles/${tag}_*
