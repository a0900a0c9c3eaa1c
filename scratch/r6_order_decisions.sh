#!/bin/bash
R=$GRAFT_REPO_ROOT
for mode in ${MODES:-0 1 always 0 1 always}; do
  unset DIE_AMD_LIB; export DIE_PIC_ORDER=$mode
  if [ $mode = always ]; then export DIE_PIC_ORDER=1 DIE_AMD_LIB=$R/scratch/libs/libdie_ord_always.so; fi
  if [ -f $R/scratch/libs/libdie_$mode.so ]; then export DIE_PIC_ORDER=1 DIE_AMD_LIB=$R/scratch/libs/libdie_$mode.so; fi
  echo "== order=$mode"; timeout -k 10 600 python3 $R/scratch/r6_order_decisions.py ${STEPS:-192 400 700 1000 3000} || exit 1
done
