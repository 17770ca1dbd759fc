#!/bin/bash
# builds tools/_libctdet_abl{1,2}.so: the library with dla_base.hip compiled under -DCTDET_BASE_ABLATE=N (measurement only)
set -e
cd "$(dirname "$0")/../detectron2-centernet_amd/csrc"
for n in ${@:-1 2}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -Wno-inline-asm -DCTDET_BASE_ABLATE=$n -c dla_base.hip -o /tmp/dla_base_abl$n.o
  objs=$(ls ../lib/obj/*.o | grep -v dla_base.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_libctdet_abl$n.so $objs /tmp/dla_base_abl$n.o -ldl
done
