# A/B of the f32 DCNv2 kernels (LDS window vs global gather) on the DLA-34 layer shapes: bash tools/ab_dcn_f32.sh
set -e
for sh in "128 128 64 64" "64 64 128 128" "32 32 256 256" "16 16 512 256" "64 64 128 64"; do
  set -- $sh
  for t in 0 32; do
    for std in 0.5 1.0; do
    echo -n "tune=$t std=$std: "; timeout -k 10 120 python tools/bench_conv.py --f32 --dcn --B 64 --H $1 --W $2 --cin $3 --cout $4 --tune $t --off-std $std 2>/dev/null | tail -1
    done
  done
done
