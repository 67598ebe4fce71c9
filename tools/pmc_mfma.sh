# MFMA-busy counters of one training step (north_star: "evidenced by rocprof HBM-GB/s and MFMA-busy counters"; SURVEY.md
# section 8d: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES).  One --pmc pass of its own (no other trace domains), program
# directly after `--`; run on the GPU box:  bash tools/pmc_mfma.sh [tag]
set -e
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; mkdir -p $R/gpurun_out/pmc_mfma; cd /tmp; export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv \
  -d $R/gpurun_out/pmc_mfma/run -o m -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-decode --no-high > $R/gpurun_out/pmc_mfma/run.log 2>&1
python3 $R/profiles/summarize_mfma.py $(find $R/gpurun_out/pmc_mfma/run -name '*counter_collection.csv') > $R/gpurun_out/pmc_mfma/${TAG}_pmc_mfma_busy.txt
find $R/gpurun_out/pmc_mfma -name '*.csv' -delete     # keep only the summary (raw per-dispatch CSVs are tens of MB)
tail -40 $R/gpurun_out/pmc_mfma/${TAG}_pmc_mfma_busy.txt
