#!/bin/bash
# Collects the profile set of one round on the GPU box (run from the repository root through gpurun):
#   kernel trace + stats of the default bench command, five counter passes of a short bench run, and the
#   trace + matrix-core pass of the bfloat16 variant.
# Outputs land under gpurun_out/prof/; tools/pmc_traffic.py, pmc_mfma.py, pmc_summary.py and profile_report.py
# turn them into the files kept under profiles/.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 bench.py > $OUT/bench_default.json.log 2> $OUT/trace.err
echo "trace done"
timeout -s KILL 280 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o f -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > $OUT/fetch.log 2>&1
echo "fetch done"
timeout -s KILL 280 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write -o w -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > $OUT/write.log 2>&1
echo "write done"
timeout -s KILL 280 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/mfma -o m -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > $OUT/mfma.log 2>&1
echo "mfma done"
timeout -s KILL 280 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/issue -o i -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > $OUT/issue.log 2>&1
echo "issue done"
# L2 -> CU read requests (128 bytes each) and L2 hit rate per launch
timeout -s KILL 280 rocprofv3 --output-format csv --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/l2 -o l2 -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > $OUT/l2.log 2>&1
echo "l2 done"
# the bfloat16-tensor variant: kernel trace of the bench command, and the matrix-core busy cycles of a short run
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace16 -o bench -- python3 bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json.log 2> $OUT/trace16.err
echo "bf16 trace done"
timeout -s KILL 280 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/mfma16 -o m -- python3 bench.py --dtype bf16 --no-cpu-baseline --steps 1 --warmup 1 > $OUT/mfma16.log 2>&1
echo "bf16 mfma done"
timeout -s KILL 280 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/fetch16 -o f -- python3 bench.py --dtype bf16 --no-cpu-baseline --steps 1 --warmup 1 > $OUT/fetch16.log 2>&1
timeout -s KILL 280 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/write16 -o w -- python3 bench.py --dtype bf16 --no-cpu-baseline --steps 1 --warmup 1 > $OUT/write16.log 2>&1
echo "bf16 traffic done"
# the reference's default arguments (grouped lines, beam search with 10 beams): kernel trace of a short run
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tracedef -o bench -- python3 bench.py --reference-defaults --steps 10 --warmup 2 > $OUT/bench_refdef.json.log 2> $OUT/tracedef.err
echo "reference-defaults trace done"
find $OUT -name "*.csv" | head -30
