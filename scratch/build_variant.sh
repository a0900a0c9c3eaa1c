#!/bin/bash
# usage: scratch/build_variant.sh NAME "-DFLAG ..."  → scratch/libs/libdie_NAME.so (DIE_AMD_LIB selects it)
cd $(dirname $0)/..
N=$1; shift
# every variant leaves its flags behind (tracked: scratch/variants.log) — a faulting A/B run must be attributable to its build
echo "$(date -u +%FT%TZ) $N: $* (die_pic.hip $(sha1sum die_amd/csrc/die_pic.hip | cut -c1-12))" >> scratch/variants.log
objs=""
for f in die_agents die_env die_init die_sort die_pack die_ghost die_render die_pic die_pic_refresh die_nca; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=on "$@" -c die_amd/csrc/$f.hip -o scratch/libs/$f.$N.o &
  objs="$objs scratch/libs/$f.$N.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o scratch/libs/libdie_$N.so && echo built $N
