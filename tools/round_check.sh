set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/t18.log 2>&1
timeout -k 10 300 python bench.py > gpurun_out/bench18.log 2>&1
timeout -k 10 100 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" > gpurun_out/smoke18.log 2>&1
timeout -k 10 120 python tools/ubench_attn.py > gpurun_out/ua18.log 2>&1
timeout -k 10 120 python tools/stamps_bwd.py > gpurun_out/stamps18.log 2>&1
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof18 -o r18 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode > $GRAFT_REPO_ROOT/gpurun_out/prof18.log 2>&1
