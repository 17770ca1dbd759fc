#!/bin/bash
# HBM traffic per kernel of the eval forward: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; together they
# exceed the counter hardware) over the eager (no HIP graph) forward.  usage (GPU box, repo root): bash tools/pmc_hbm.sh <outdir> [f16x3|f16|f32]
set -e
OUT=${1:-gpurun_out/pmc_hbm}
ROOT=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PREC=${2:-f16x3}
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/fetch" -o run -- python3 "$ROOT/tools/eager_forward.py" 64 2 $PREC > "$ROOT/$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$ROOT/$OUT/write" -o run -- python3 "$ROOT/tools/eager_forward.py" 64 2 $PREC > "$ROOT/$OUT/write.log" 2>&1
ls "$ROOT/$OUT/fetch" "$ROOT/$OUT/write"
