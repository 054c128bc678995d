#!/bin/bash
# a longer TunableOp search than tools/tune_gemms.sh (WT_TUNE_MS / WT_TUNE_ITERS per candidate), e2e stage only; result -> gpurun_out/tunableop_long.csv,
# then the e2e bench with the packaged file and with the new one (A/B on the same box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120 > gpurun_out/tune_ab.txt
cp waymo_2d_tracking_amd/tuning/tunableop_gfx950.csv /tmp/packaged.csv
mv waymo_2d_tracking_amd/tuning/tunableop_gfx950.csv /tmp/old_tunableop.csv
WT_GEMM_TUNING_ONLINE=1 WT_TUNE_MS=${WT_TUNE_MS:-60} WT_TUNE_ITERS=${WT_TUNE_ITERS:-100} WT_TUNABLEOP_OUT=/tmp/wt_long.csv python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120 >> gpurun_out/tune_ab.txt
cp /tmp/wt_long.csv gpurun_out/tunableop_long.csv
cp /tmp/wt_long.csv waymo_2d_tracking_amd/tuning/tunableop_gfx950.csv
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120 >> gpurun_out/tune_ab.txt
cp /tmp/packaged.csv waymo_2d_tracking_amd/tuning/tunableop_gfx950.csv
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120 >> gpurun_out/tune_ab.txt
cat gpurun_out/tune_ab.txt; wc -l gpurun_out/tunableop_long.csv
