#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 1000000000 512; do
  echo "== NW8_MIN=$m"
  SPRK_FWD_NW8_MIN=$m python3 scratch/convbench.py 2>&1 | grep GFLOP | grep -v cin | sed 's/| bw.*//'
done
