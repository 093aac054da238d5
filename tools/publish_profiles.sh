#!/bin/bash
# Turns gpurun_out/prof (written by tools/collect_profiles.sh) into the files kept under profiles/<round>.
#   bash tools/publish_profiles.sh r01
set -e
R=${1:-r01}
P=gpurun_out/prof
mkdir -p profiles/$R
cp $P/trace/bench_kernel_stats.csv profiles/$R/bench_kernel_stats.csv
tail -1 $P/bench_default.json.log > profiles/$R/bench_default.json.log
if [ -d $P/fetch16 ]; then
  python tools/pmc_traffic.py $P/fetch $P/write $P/fetch16 $P/write16 > profiles/pmc_traffic.json
  python tools/pmc_summary.py $P/fetch16 > profiles/$R/bf16_pmc_fetch_size.csv
  python tools/pmc_summary.py $P/write16 > profiles/$R/bf16_pmc_write_size.csv
else
  python tools/pmc_traffic.py $P/fetch $P/write > profiles/pmc_traffic.json
fi
python tools/pmc_summary.py $P/fetch > profiles/$R/pmc_fetch_size.csv
python tools/pmc_summary.py $P/write > profiles/$R/pmc_write_size.csv
python tools/pmc_mfma.py $P/mfma > profiles/$R/pmc_mfma_util.csv
if [ -d $P/issue ]; then python tools/pmc_issue.py $P/issue > profiles/$R/pmc_issue.csv; fi
if [ -d $P/l2 ]; then python tools/pmc_l2.py $P/l2 > profiles/$R/pmc_l2_requests.csv; fi
if [ -d $P/trace16 ]; then
  cp $P/trace16/bench_kernel_stats.csv profiles/$R/bf16_bench_kernel_stats.csv
  tail -1 $P/bench_bf16.json.log > profiles/$R/bf16_bench.json.log
  python tools/pmc_mfma.py $P/mfma16 > profiles/$R/bf16_pmc_mfma_util.csv
fi
if [ -d $P/tracedef ]; then
  cp $P/tracedef/bench_kernel_stats.csv profiles/$R/refdef_bench_kernel_stats.csv
  tail -1 $P/bench_refdef.json.log > profiles/$R/refdef_bench.json.log
fi
python tools/profile_report.py profiles/$R > profiles/$R/kernel_report.md
