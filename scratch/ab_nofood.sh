#!/bin/bash
# no food tile in LDS (31 KB per workgroup instead of 47): does a smaller workgroup then pay?
cd $GRAFT_REPO_ROOT
for t in 512 384 256; do echo "--- threads $t"; AB_EXTRA="--pic-threads $t" bash scratch/ab_libs.sh nofoodlds || exit 1; done
AB_EXTRA="--pic-threads 384" bash scratch/ab_libs.sh hip && bash scratch/ab_libs.sh hip
