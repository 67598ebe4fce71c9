set -e
mkdir -p gpurun_out
cp build/ab/libB.so tacotron2_amd/libtacotron2_amd.so
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -m gpu -x -q > gpurun_out/t18.log 2>&1
timeout -k 10 120 python tools/stamps_bwd.py > gpurun_out/stamps18.log 2>&1
for v in A B A B; do cp build/ab/lib$v.so tacotron2_amd/libtacotron2_amd.so; echo "variant $v" >> gpurun_out/ab18.log; timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode >> gpurun_out/ab18.log 2>&1; done
