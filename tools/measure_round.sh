#!/bin/bash
# The round's measurement set (GPU box, repo root): bash tools/measure_round.sh <part 1|2|3>; outputs under gpurun_out/m/.
# Part 1: default bench line; rocprofv3 kernel stats of the f16x3 (headline), f32 and f16 inference legs.
# Part 2: rocprofv3 kernel stats of the (eager) f16x3 and f16 training legs; HBM counters per kernel (two --pmc passes) of the f16x3 forward.
# Part 3: per-layer tables (f16x3, f16), decode microbench, ResNet-50 bench line, two-rank rehearsal (two ranks share the GPU).
set -e
ROOT=$(pwd)
M=$ROOT/gpurun_out/m
mkdir -p $M
if [ "$1" = "1" ]; then
  python bench.py > $M/bench_default.json 2> $M/bench_default.err
  cd /tmp && export TMPDIR=/tmp
  for P in f16x3 f32 f16; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $M/prof_infer_$P -o t -- python3 $ROOT/bench.py --precision $P --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-f32 --no-f16 > $M/bench_infer_${P}_under_rocprof.json 2> $M/prof_infer_$P.err
  done
elif [ "$1" = "2" ]; then
  cd /tmp && export TMPDIR=/tmp
  for P in f16x3 f16; do
    CTDET_TRAIN_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $M/prof_train_$P -o t -- python3 $ROOT/bench.py --task train --precision $P --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $M/bench_train_${P}_under_rocprof.json 2> $M/prof_train_$P.err
  done
  cd $ROOT
  bash tools/pmc_hbm.sh gpurun_out/m/pmc_hbm f16x3 > $M/pmc_hbm.log 2>&1
else
  python tools/layer_table.py 64 f16x3 > $M/layer_table_f16x3.txt 2>&1
  python tools/layer_table.py 64 f16 > $M/layer_table_f16.txt 2>&1
  python tools/bench_decode.py > $M/decode.txt 2>&1
  python bench.py --config r50 --no-cpu-baseline > $M/bench_r50.json 2> $M/bench_r50.err
  python bench.py --gpus 2 --no-cpu-baseline --no-roofline > $M/bench_gpus2_rehearsal.json 2> $M/bench_gpus2_rehearsal.err
fi
ls $M
