#!/bin/bash
# builds scratch/libs/libdie_stamps.so = the library with -DPIC_STAMPS (diagnostic; DIE_AMD_LIB selects it)
cd $(dirname $0)/..
objs=""
for f in die_agents die_env die_init die_sort die_pack die_ghost die_render die_pic die_pic_refresh die_nca; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=on -DPIC_STAMPS $STAMP_EXTRA -c die_amd/csrc/$f.hip -o scratch/libs/$f.stamps.o &
  objs="$objs scratch/libs/$f.stamps.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o scratch/libs/libdie_stamps.so && echo built
