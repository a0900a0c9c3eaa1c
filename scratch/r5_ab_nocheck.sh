#!/bin/bash
# like r5_ab.sh for variants that give WRONG results on purpose (timing experiments): the sticky error word is not checked (DIE_BENCH_NOCHECK=1)
export DIE_BENCH_NOCHECK=1
exec $(dirname $0)/r5_ab.sh "$@"
