#!/bin/bash
cd $GRAFT_REPO_ROOT
for blocks in 256 512; do
  echo "== BLOCKS=$blocks"
  SPRK_WG_BLOCKS=$blocks python3 scratch/convbench.py 2>&1 | grep GFLOP
done
