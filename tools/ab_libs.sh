# In-session A/B of prebuilt library variants build/ab/lib<X>.so (copied over the in-tree library one at a time).
# usage: bash tools/ab_libs.sh A B C ...     (the first one is restored at the end)
set -e
mkdir -p gpurun_out
for r in 1 2; do for v in "$@"; do
  cp build/ab/lib$v.so tacotron2_amd/libtacotron2_amd.so
  echo "variant $v" >> gpurun_out/ab.log
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode >> gpurun_out/ab.log 2>&1
done; done
cp build/ab/lib$1.so tacotron2_amd/libtacotron2_amd.so
