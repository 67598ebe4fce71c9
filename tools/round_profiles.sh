# Round-2 profile set on the GPU box: rocprofv3 kernel statistics of the bench command, the decode-only statistics, PMC HBM
# traffic of the forward frame loop, MFMA-busy counters.  bash tools/round_profiles.sh <tag>
set -e
R=$GRAFT_REPO_ROOT; V=${1:-r02}; O=$R/gpurun_out/prof_$V; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -o k -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode --no-high > $O/train.log 2>&1
cp $(find $O/train -name '*kernel_stats.csv' | head -1) $O/${V}_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -o k -- python3 $R/tools/prof_decode.py > $O/dec.log 2>&1
cp $(find $O/dec -name '*kernel_stats.csv' | head -1) $O/${V}_decode_kernel_stats.csv
find $O -name '*kernel_trace.csv' -delete
cd $R
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 && cp gpurun_out/pmc/summary.txt $O/${V}_pmc_hbm_traffic.txt
bash tools/pmc_mfma.sh $V > $O/pmc_mfma.log 2>&1 && cp gpurun_out/pmc_mfma/${V}_pmc_mfma_busy.txt $O/
ls -la $O
