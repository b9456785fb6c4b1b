#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 0 1 2; do
  echo "== WG_XTAB=$m"
  SPRK_WG_XTAB=$m python3 scratch/convbench.py 2>&1 | grep GFLOP | grep -v cin | sed 's/.*| bw/bw/'
done
