#!/bin/bash
# usage: tools/pmc_base.sh <tag>  (GPU box, repo root): SQ counter passes over tools/bench_base.py 64 512 f16x3, summarised
# per kernel into gpurun_out/pmc_base_<tag>.txt
TAG=${1:-base}; shift
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pmc_${TAG}_$n -o p --output-format csv -- python3 $R/tools/bench_base.py 64 512 f16x3 > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1 || exit 1
done
python3 - "$R/gpurun_out" "$TAG" <<'PY' > $R/gpurun_out/pmc_base_${TAG}.txt
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(f"{root}/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dla_base" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:32s} mean_per_dispatch={sum(v)/len(v):.4g}  n={len(v)}")
PY
