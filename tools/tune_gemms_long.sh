#!/bin/bash
# a longer TunableOp search than tools/tune_gemms.sh (WT_TUNE_MS / WT_TUNE_ITERS per candidate), e2e stage only; result -> gpurun_out/tunableop_long.csv,
# then the e2e bench with the packaged file and with the new one (A/B on the same box).  The packaged file is never moved or overwritten:
# every leg selects its file through WT_TUNABLEOP_IN (waymo_2d_tracking_amd/tuning/__init__.py; '' = no file).
set -eu
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
T=$(mktemp -d)
trap 'rm -rf "$T"' EXIT
line() { python3 bench.py --steps "$1" --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120; }
mkdir -p gpurun_out
line 6 > gpurun_out/tune_ab.txt                                                           # packaged selections
WT_TUNABLEOP_IN= WT_GEMM_TUNING_ONLINE=1 WT_TUNE_MS=${WT_TUNE_MS:-60} WT_TUNE_ITERS=${WT_TUNE_ITERS:-100} WT_TUNABLEOP_OUT=$T/wt_long.csv line 2 >> gpurun_out/tune_ab.txt
cp "$T/wt_long.csv" gpurun_out/tunableop_long.csv
WT_TUNABLEOP_IN=$T/wt_long.csv line 6 >> gpurun_out/tune_ab.txt                           # the long search's selections
line 6 >> gpurun_out/tune_ab.txt                                                          # packaged again (box drift)
cat gpurun_out/tune_ab.txt; wc -l gpurun_out/tunableop_long.csv
