#!/bin/bash
# usage: variants.sh "<flags1>" "<flags2>" ... : rebuild raster_bwd.hip with each extra flag set and profile cfg2 backward
cd /root/repo/3d-gaussian-splatting_amd/csrc
for fl in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -fno-slp-vectorize $fl -c ${SRC:-raster_bwd.hip} -o raster_bwd.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgs_amd.so cull_project.o binning.o radix_sort.o tile_sort.o tile_bin.o raster_fwd.o raster_bwd.o gs_frame.o adam.o loss.o densify.o
  echo "== $fl"; python /root/repo/tools/stage_profile.py ${CFGS:-cfg2} 2>&1 | tail -${NCFG:-1} | sed 's/.*bwd {/bwd {/'
done
