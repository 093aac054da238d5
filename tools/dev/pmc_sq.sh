#!/bin/bash
# Development aid: three shader-sequencer counter passes over the two network forwards (tools/bench_nets.py).
# Run from the repository root through gpurun; summaries: python3 tools/pmc_summary.py gpurun_out/sq/<pass>
export TMPDIR=/tmp
OUT=gpurun_out/sq
rm -rf $OUT && mkdir -p $OUT
ARGS="tools/bench_nets.py --batch 4 --reps 1 --crops 256 $@"
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY -d $OUT/a -o a -- python3 $ARGS > $OUT/a.log 2>&1 || exit 1
echo "a done"
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA -d $OUT/b -o b -- python3 $ARGS > $OUT/b.log 2>&1 || exit 1
echo "b done"
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC -d $OUT/c -o c -- python3 $ARGS > $OUT/c.log 2>&1 || exit 1
echo "c done"
