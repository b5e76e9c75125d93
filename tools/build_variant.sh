#!/bin/bash
# A/B and diagnostic builds of the library next to the shipped one:
#   tools/build_variant.sh NAME "SRC1.hip SRC2.hip" "-DFLAG=1 ..."
# recompiles the listed sources with the extra flags and links them with the shipped objects of the other sources into
# tools/libNAME.bin (git-ignored, travels to the GPU box; select it with BETA_CORES_LIB=tools/libNAME.bin).
set -e
cd "$(dirname "$0")/../beta_cores_amd/csrc"
NAME=$1; SRCS=$2; FLAGS=$3
make -s -j8
OBJS=""
for f in bc_core bc_upload bc_sweep bc_prefilter bc_snnls bc_project bc_gradx bc_gram bc_comm; do
  if [[ " $SRCS " == *" $f.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function $FLAGS -c $f.hip -o /tmp/${f}_$NAME.o
    OBJS="$OBJS /tmp/${f}_$NAME.o"
  else
    OBJS="$OBJS $f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lib$NAME.bin $OBJS -ldl -lpthread
echo built tools/lib$NAME.bin
