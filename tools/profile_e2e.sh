#!/bin/bash
# Round profile of the default bench command: rocprofv3 kernel stats + a separate PMC pass (HBM traffic of the roofline
# kernel).  Run on the GPU box; results land in gpurun_out/prof_e2e/ and are copied to profiles/ by hand.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_e2e
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /tmp/prof_stats.log
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# HBM traffic of the roofline kernel: raw TCC_EA0_* counters (the derived FETCH_SIZE / WRITE_SIZE names hang this rocprofv3
# build), collected by tools/pmc_traffic.sh on the res4 shape; it writes $OUT/pmc_traffic.json incl. traffic_bytes_per_launch
bash $R/tools/pmc_traffic.sh > /dev/null
cd $R && python3 bench.py --steps 5 --warmup 2 > $OUT/bench_line.json 2> /dev/null
tail -c 600 $OUT/bench_line.json
