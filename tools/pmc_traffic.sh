# HBM-side traffic of the forward frame-loop kernels (separate --pmc passes, no other trace domains); run on the GPU box.
set -e
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc/fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-high > $R/gpurun_out/pmc/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc/write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-high > $R/gpurun_out/pmc/write.log 2>&1
python3 $R/profiles/summarize_pmc.py --steps=2 --T=872 $(find $R/gpurun_out/pmc -name '*counter_collection.csv') > $R/gpurun_out/pmc/summary.txt
# keep only the summary (the raw per-dispatch CSVs are tens of MB)
find $R/gpurun_out/pmc -name '*.csv' -delete
