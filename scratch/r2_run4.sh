#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
for d in 0 1 2 3 4 7 8 15; do echo "diag $d: $(SPRK_C16_DIAG=$d timeout -k 10 120 python scratch/convbench.py bf16 "dec1.2 96->96@64" 2>&1 | tail -1)"; done
for d in 0 3 4 ; do echo "diag $d: $(SPRK_C16_DIAG=$d timeout -k 10 120 python scratch/convbench.py bf16 "enc1.2 48->48@64" 2>&1 | tail -1)"; done
