# End-of-round check on the GPU box: GPU tests, bench (with CPU baseline), build+smoke in one process, stamp tools,
# rocprofv3 kernel statistics of the bench, and the micro-benchmarks that back DESIGN.md's next levers.
set -e
mkdir -p gpurun_out
V=${1:-19}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t$V.log 2>&1
timeout -k 10 300 python bench.py > gpurun_out/bench$V.log 2>&1
timeout -k 10 100 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > gpurun_out/smoke$V.log 2>&1
timeout -k 10 120 python tools/ubench_attn.py > gpurun_out/ua$V.log 2>&1
timeout -k 10 120 python tools/stamps_bwd.py > gpurun_out/stamps$V.log 2>&1
if [ -x build/ubench_group_sync ]; then timeout -k 10 60 ./build/ubench_group_sync > gpurun_out/group_sync$V.log 2>&1; fi
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof$V -o r$V -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode > $GRAFT_REPO_ROOT/gpurun_out/prof$V.log 2>&1
