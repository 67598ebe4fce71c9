# usage: bash tools/ab_multi.sh "LIB ENV=VAL ..." "LIB ENV=VAL" ...   (each argument = one variant: library letter + environment)
set -e
mkdir -p gpurun_out
for r in 1 2; do for v in "$@"; do
  lib=${v%% *}; e=${v#* }
  cp build/ab/lib$lib.so tacotron2_amd/libtacotron2_amd.so
  echo "variant $v" >> gpurun_out/ab.log
  env $e timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode >> gpurun_out/ab.log 2>&1
done; done
cp build/ab/libA.so tacotron2_amd/libtacotron2_amd.so
