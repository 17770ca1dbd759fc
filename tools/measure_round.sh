#!/bin/bash
# The round's measurement set (GPU box, repo root): bash tools/measure_round.sh <part 1|2>; outputs under gpurun_out/m/.
# Part 1: default bench line, rocprofv3 kernel stats of the inference and of the (eager) training leg.
# Part 2: HBM counters per kernel (two --pmc passes), per-layer table, decode microbench, f32 / ResNet-50 bench lines.
set -e
ROOT=$(pwd)
M=$ROOT/gpurun_out/m
mkdir -p $M
if [ "$1" = "1" ]; then
  python bench.py > $M/bench_default.json 2> $M/bench_default.err
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $M/prof_infer -o t -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-f32 > $M/bench_infer_under_rocprof.json 2> $M/prof_infer.err
  CTDET_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $M/prof_train -o t -- python3 $ROOT/bench.py --task train --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $M/bench_train_under_rocprof.json 2> $M/prof_train.err
else
  bash tools/pmc_hbm.sh gpurun_out/m/pmc_hbm > $M/pmc_hbm.log 2>&1
  python tools/layer_table.py > $M/layer_table.txt 2>&1
  python tools/bench_decode.py > $M/decode.txt 2>&1
  python bench.py --precision f32 --no-train --no-cpu-baseline > $M/bench_f32_bs64.json 2> $M/bench_f32.err
  python bench.py --config r50 --no-cpu-baseline > $M/bench_r50.json 2> $M/bench_r50.err
fi
ls $M
