#!/bin/bash
# Counters of the split-operand GEMM on one shape, f32-A kernel against the round-6 planes kernel (separate --pmc passes, --kernel-trace only).
#   tools/pmc_planes.sh [M N K] -> gpurun_out/r06_planes/pmc.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-9600}; N=${2:-1024}; K=${3:-1024}
OUT=$R/gpurun_out/r06_planes/pmc_${M}x${N}x${K}.txt
mkdir -p $(dirname $OUT)
: > $OUT
pass() {
  var=$1; shift
  rm -rf /tmp/pmcs
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcs -- python3 $R/tools/gemm_planes_pmc.py $var $M $N $K 6 > /tmp/pmcs.log 2>&1
  f=$(find /tmp/pmcs -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_split_kernel' in r['Kernel_Name'] or 'gemm_split_planes_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in acc.items():
    print('%-32s %14.5g   (launches %d)' % (c, sum(v) / len(v), len(v)))
PY
}
for var in f32 planes_f32out planes_all f32_both; do
  echo "# variant $var  $M x $N x $K (bias + residual + ReLU)" >> $OUT
  pass $var SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
  pass $var SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
  pass $var FETCH_SIZE
  pass $var WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
done
cat $OUT
