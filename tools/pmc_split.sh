#!/bin/bash
# Counters of the split-operand GEMM kernel (csrc/det_gemm_split.hip) on one shape: matrix-pipe busy share, wait / issue split, LDS
# conflicts, L2 hit rate, HBM traffic.  Separate --pmc passes (no tracing domains beside --kernel-trace).
#   tools/pmc_split.sh M N K [epilogue] [outfile]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-9600}; N=${2:-1024}; K=${3:-1024}; EPI=${4:-0}; OUT=${5:-$R/gpurun_out/pmc_split.txt}
mkdir -p $(dirname $OUT)
: > $OUT
pass() {
  rm -rf /tmp/pmcs
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcs -- python3 $R/tools/gemm_split_one.py $M $N $K 6 $EPI > /tmp/pmcs.log 2>&1
  f=$(find /tmp/pmcs -name "*counter_collection.csv" | head -1)
  python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_split_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in acc.items():
    print('%-32s %14.5g   (launches %d)' % (c, sum(v) / len(v), len(v)))
PY
}
echo "# gemm_split_kernel $M x $N x $K epilogue=$EPI" >> $OUT
pass SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
pass FETCH_SIZE
pass WRITE_SIZE TCC_EA0_RDREQ_sum
cat $OUT
