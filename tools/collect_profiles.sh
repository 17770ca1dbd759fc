#!/bin/bash
# copies the outputs of tools/measure_round.sh (gpurun_out/m) into profiles/ under the round's prefix: tools/collect_profiles.sh r04
set -e
R=${1:?round tag, e.g. r04}
M=gpurun_out/m; P=profiles
python tools/pmc_summarize.py $M/pmc_hbm $R | tail -3
cp $M/bench_default.json $P/${R}_bench_default.json
cp $M/bench_gpus2_rehearsal.json $P/${R}_bench_gpus2_rehearsal.json
cp $M/bench_r50.json $P/${R}_bench_r50_800_bs8.json
cp $M/decode.txt $P/${R}_decode.txt
cp $M/layer_table_f16x3.txt $P/${R}_layer_table_f16x3.txt
cp $M/layer_table_f16.txt $P/${R}_layer_table_f16.txt
for p in f16x3 f32 f16; do
  cp "$(find $M/prof_infer_$p -name '*kernel_stats.csv' | head -1)" $P/${R}_infer_${p}_kernel_stats.csv
  cp $M/bench_infer_${p}_under_rocprof.json $P/${R}_infer_${p}_under_rocprof.json
done
for p in f16x3 f16; do
  cp "$(find $M/prof_train_$p -name '*kernel_stats.csv' | head -1)" $P/${R}_train_${p}_kernel_stats.csv
  cp $M/bench_train_${p}_under_rocprof.json $P/${R}_train_${p}_under_rocprof.json
done
ls $P | grep "^${R}_"
