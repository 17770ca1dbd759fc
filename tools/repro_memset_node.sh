#!/bin/bash
# Reproduces round 3's data-parallel corruption (profiles/r04_graph_memset_node.txt) on ONE GPU: two ranks share the device, the
# captured training step clears the heat-map target with hipMemsetAsync again (CTDET_TUNE_FLAGS=4096: a memset node), nothing
# runs between the steps.  Healthy final losses of rank 0: hm 93.7187 wh 3.00965 off 0.44334.
#   bash tools/repro_memset_node.sh [runs]
R=${1:-3}
mkdir -p gpurun_out/repro
for flags in 4096 0; do
  for i in $(seq 1 $R); do
    CTDET_TUNE_FLAGS=$flags timeout -k 10 200 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --no-f16 --no-roofline \
      > gpurun_out/repro/f${flags}_$i.json 2> gpurun_out/repro/f${flags}_$i.err
    echo "CTDET_TUNE_FLAGS=$flags run $i rc=$? $(python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/repro/f${flags}_$i.json").read().strip().splitlines()[-1])["train"]["config"]
    print(d["graph_nodes"], d["final_losses"])
except Exception:
    import re
    m = re.findall(r"loss_dict = (\{'hm_loss'.*?\})", open("gpurun_out/repro/f${flags}_$i.err", errors="replace").read())
    print("FloatingPointError", m[0] if m else "")
PY
)"
  done
done
