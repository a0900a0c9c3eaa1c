#!/bin/bash
cd $GRAFT_REPO_ROOT
AB_EXTRA="--no-pic" bash scratch/ab_libs.sh prev hip prev hip || exit 1
bash scratch/small_sizes.sh hip
