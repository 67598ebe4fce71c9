# In-session A/B of two settings of one environment knob (box-to-box spread is +-2 %, so variants are compared on one box).
# usage: bash tools/ab_session.sh KNOB VALUE_A VALUE_B [pytest-args...]
set -e
KNOB=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out
if [ $# -gt 0 ]; then timeout -k 10 500 python -m pytest "$@" -m gpu -x -q > gpurun_out/ab_tests.log 2>&1; fi
for v in $A $B $A $B; do
  echo "variant $KNOB=$v" >> gpurun_out/ab.log
  env $KNOB=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode >> gpurun_out/ab.log 2>&1
done
