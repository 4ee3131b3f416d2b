# usage: ab_bench_all.sh a.so b.so ...   -- tools/bench_all.py once per library build, the package's own library restored afterwards
set -e
cp poseestimation_amd/libso3proj.so /tmp/keep.so
for so in "$@"; do
  cp "$so" poseestimation_amd/libso3proj.so
  echo "== $so"; python3 tools/bench_all.py 2>/dev/null | grep -E "K2|K3|K1\+K4|fused|K1 " | cut -c1-130
done
cp /tmp/keep.so poseestimation_amd/libso3proj.so
