# A/B of the f16 DCNv2 window kernels on the DLA-34 layer shapes: bash tools/ab_dcn_f16.sh
set -e
for sh in "128 128 64 64" "64 64 128 64" "32 32 256 64"; do
  set -- $sh
  for t in 0 64; do
    for std in 0.5 2.0; do
    echo -n "tune=$t std=$std: "; timeout -k 10 120 python tools/bench_conv.py --dcn --B 64 --H $1 --W $2 --cin $3 --cout $4 --tune $t --off-std $std 2>/dev/null | tail -1
    done
  done
done
