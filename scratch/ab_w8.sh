#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/ab_libs.sh hip w8nf || exit 1
AB_EXTRA="--pic-threads 448" bash scratch/ab_libs.sh w7nf || exit 1
bash scratch/ab_libs.sh hip
