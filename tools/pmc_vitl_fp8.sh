#!/bin/bash
# on the GPU box: PMC write / fetch bytes per kernel of the ViT-L/14@336 fp8 tower (batch 128): is the MXFP8 LayerNorm's scale store wasteful?
set -e
R=$PWD; OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FAST="--model ViT-L-14-336 --batch 128 --precision fp8 --steps 3 --warmup 1 --no-cpu-baseline --no-full-forward --no-kernel-events --no-precisions --no-input-side --no-configs4 --no-sustained --no-batch-sweep"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $R/bench.py $FAST > /dev/null 2> $OUT/fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $R/bench.py $FAST > /dev/null 2> $OUT/write.err
cd $R
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic_vitl_fp8.json
rm -rf $OUT/fetch $OUT/write
# SQ counters of the same run (its own pass)
cd /tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o sq -- python3 $R/bench.py $FAST > /dev/null 2> $OUT/sq.err
cd $R
python3 tools/pmc_sq.py $OUT/sq $OUT/pmc_sq_vitl_fp8.json
rm -rf $OUT/sq
