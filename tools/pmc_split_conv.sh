#!/bin/bash
# Counters of the split-operand kernel in convolution mode (implicit GEMM, MODE 1) on one shape - the same passes as tools/pmc_split.sh.
#   tools/pmc_split_conv.sh B C H W N [outfile]          (3x3, stride 1, pad 1; default: the box-head convolution 1000 256 7 7 256)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${1:-1000}; C=${2:-256}; H=${3:-7}; W=${4:-7}; N=${5:-256}; OUT=${6:-$R/gpurun_out/pmc_split_conv.txt}
mkdir -p $(dirname $OUT)
: > $OUT
pass() {
  rm -rf /tmp/pmcc
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/pmcc -- python3 $R/tools/conv_split_one.py $B $C $H $W $N > /tmp/pmcc.log 2>&1
  f=$(find /tmp/pmcc -name "*counter_collection.csv" | head -1)
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
echo "# gemm_split_kernel, conv mode: $B x $C x $H x $W -> $N (3x3, stride 1, pad 1): M = $((B*H*W)), K = $((9*C))" >> $OUT
pass SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pass TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
cat $OUT
