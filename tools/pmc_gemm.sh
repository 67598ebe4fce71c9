# counters of the GEMM kernel on one shape (separate --pmc passes); run on the GPU box: bash tools/pmc_gemm.sh M N K a_k b_k
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_gemm; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { timeout -k 10 200 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $O/$2 -o p -- python3 $R/tools/prof_gemm.py "${@:3}" > $O/$2.log 2>&1; }
run "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" sq "$@"
run "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" tcc "$@"
run "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" sq2 "$@"
python3 - $O <<'PY'
import csv, sys, glob, collections
agg = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:32s} n={len(v):3d} avg/launch {sum(v)/len(v):16.0f}")
PY
find $O -name '*.csv' -delete
