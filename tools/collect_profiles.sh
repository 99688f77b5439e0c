#!/bin/bash
# Collects the round's profile set on the GPU box (run through gpurun from the repo root), in two parts (a gpurun call is
# limited to 20 minutes):
#   tools/collect_profiles.sh <tag> main    rocprofv3 kernel stats, PMC traffic (the fp16 headline AND the bf16 mode) and SQ
#                                           counters of the ViT-B/16 headline, then the bench line itself
#   tools/collect_profiles.sh <tag> vitl    BASELINE configs[4] at its per-GPU size in fp8 / bf16 / fp16 (seeded host weights)
# -> gpurun_out/prof_<tag>/..., summaries copied by hand into profiles/
set -e
TAG=${1:-r05}
PART=${2:-main}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FAST="--steps 3 --warmup 1 --no-cpu-baseline --no-full-forward --no-kernel-events --no-precisions --no-input-side --no-configs4 --no-sustained --no-batch-sweep"
if [ $PART = main ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/bench.py --no-configs4 --sustained-seconds 1 --no-batch-sweep > $OUT/bench_under_trace.json 2> $OUT/stats.err
echo "trace done"
for P in bf16 fp16; do
  S=""; [ $P = fp16 ] && S="_fp16"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$S -o fetch -- python3 $R/bench.py --precision $P $FAST > /dev/null 2> $OUT/fetch$S.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write$S -o write -- python3 $R/bench.py --precision $P $FAST > /dev/null 2> $OUT/write$S.err
  echo "pmc traffic $P done"
done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o sq -- python3 $R/bench.py $FAST > /dev/null 2> $OUT/sq.err
echo "sq done"
cd $R
python3 tools/trace_family.py $OUT/stats $OUT/gemm_family_trace_summary.json fp16
python3 tools/pmc_traffic.py $OUT/fetch $OUT/write $OUT/pmc_traffic_bench.json
python3 tools/pmc_traffic.py $OUT/fetch_fp16 $OUT/write_fp16 $OUT/pmc_traffic_bench_fp16.json
python3 tools/pmc_sq.py $OUT/sq $OUT/pmc_sq_bench.json
cp $(ls $OUT/stats/*kernel_stats.csv $OUT/stats/*/*kernel_stats.csv 2>/dev/null | head -n 1) $OUT/kernel_stats_bench.csv
# the bench line LAST, with this box's PMC summaries in place (in the box's scratch copy of the repo): `roofline.traffic` is only
# reported when the summary's recorded source hash equals the running library's, and so the line and the counters are of one box
cp $OUT/pmc_traffic_bench.json $R/profiles/${TAG}_pmc_traffic_bench.json
cp $OUT/pmc_traffic_bench_fp16.json $R/profiles/${TAG}_pmc_traffic_bench_fp16.json
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench done"
# the raw traces are large: keep the summaries only
rm -rf $OUT/stats/*/*kernel_trace.csv $OUT/stats/*kernel_trace.csv $OUT/fetch $OUT/write $OUT/fetch_fp16 $OUT/write_fp16 $OUT/sq
echo "summaries done"
else
# BASELINE configs[4] at its per-GPU size, with a kept record: ViT-L/14@336, batch 128, each precision, including full_forward
# and train_step (the reference loop being timed: train.py:95-105)
for P in fp8 bf16 fp16; do
  python3 $R/bench.py --model ViT-L-14-336 --batch 128 --precision $P --steps 20 --warmup 3 --no-cpu-baseline --no-input-side --no-precisions --no-configs4 \
    > $OUT/bench_${P}_vitl14_336_b128.json 2> $OUT/bench_${P}_vitl14_336_b128.err
  echo "vit-l $P done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_l14 -o stats -- python3 $R/bench.py --model ViT-L-14-336 --batch 128 --precision fp8 --steps 10 --warmup 2 --no-cpu-baseline --no-input-side --no-precisions --no-full-forward --no-configs4 > /dev/null 2> $OUT/stats_l14.err
cp $(ls $OUT/stats_l14/*kernel_stats.csv $OUT/stats_l14/*/*kernel_stats.csv 2>/dev/null | head -n 1) $OUT/kernel_stats_bench_fp8_vitl14_336_b128.csv
rm -rf $OUT/stats_l14
fi
