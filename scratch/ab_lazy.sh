#!/bin/bash
cd $GRAFT_REPO_ROOT
AB_EXTRA="--eager-actions" bash scratch/ab_libs.sh hip || exit 1
bash scratch/ab_libs.sh hip || exit 1
AB_EXTRA="--eager-actions" bash scratch/ab_libs.sh hip || exit 1
bash scratch/ab_libs.sh hip
