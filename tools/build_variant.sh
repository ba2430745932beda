#!/bin/bash
# diagnostic builds of the library with extra -D flags: tools/build_variant.sh <name> <flags...>  -> mamdr_amd/build/variants/lib<name>.so
N=$1; shift
D=mamdr_amd/csrc; O=/tmp/variant_$N; mkdir -p $O mamdr_amd/build/variants
for f in star_kernels fused_kernels mamdr_api graph_engine; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $D/$f.hip -o $O/$f.o &
done
for f in step_kernels tower4_kernels; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=14 "$@" -c $D/$f.hip -o $O/$f.o &
done
for f in emb_kernels outer_kernels; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off "$@" -c $D/$f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o mamdr_amd/build/variants/lib$N.so $O/*.o && echo built mamdr_amd/build/variants/lib$N.so
