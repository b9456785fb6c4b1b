#!/bin/bash
# scratch/r3/build_variant.sh <file.hip> <N> [extra flags]: libsprk with <file> compiled -DWINO_VARIANT=N -> scratch/r3/libsprk_v<N>.so
set -e
F=$1; V=$2; shift 2
cd "$(dirname "$0")/../../spr_pick_amd/csrc"
make -s
B=${F%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DWINO_VARIANT=$V -DWINO_STAMP_$V "$@" -c $F -o _build/${B}_v$V.o
OBJS=$(ls _build/*.o | grep -v "_v[0-9]*\.o" | grep -v "_build/$B.o" | grep -v -- "-hip-")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/r3/libsprk_v$V.so $OBJS _build/${B}_v$V.o
echo built scratch/r3/libsprk_v$V.so
