#!/bin/bash
# Build an alternative libxmhw_amd.so for A/B runs without touching the product build:
#   bash tools/mkvariant.sh NAME "-DFLAG ..."          kernels_sorted.hip recompiled with the flags, the other objects are the product's
#   bash tools/mkvariant.sh NAME "-DFLAG ..." full     the whole library rebuilt in /tmp (STATS=1 when NAME starts with "stats")
# Result: ab/NAME.so.  Use it with LD_PRELOAD=$PWD/ab/NAME.so (the pybind11 module binds the xmhw_* symbols of the preloaded
# library; nothing is copied over xmhw_amd/libxmhw_amd.so).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; MODE=${3:-sorted}
mkdir -p $R/ab /tmp/abv/$NAME
if [ "$MODE" = full ]; then
  rm -rf /tmp/abv/$NAME/tree; mkdir -p /tmp/abv/$NAME/tree/xmhw_amd /tmp/abv/$NAME/tree/tools
  cp -r $R/include /tmp/abv/$NAME/tree/; cp -r $R/xmhw_amd/csrc /tmp/abv/$NAME/tree/xmhw_amd/; rm -rf /tmp/abv/$NAME/tree/xmhw_amd/csrc/build
  cp -r $R/tools/experiments /tmp/abv/$NAME/tree/tools/
  S=0; case $NAME in stats*) S=1;; esac
  make -C /tmp/abv/$NAME/tree/xmhw_amd/csrc -j6 STATS=$S XFLAGS="$FLAGS" ../libxmhw_amd.so > /tmp/abv/$NAME/make.log 2>&1 || { tail -20 /tmp/abv/$NAME/make.log; exit 1; }
  cp /tmp/abv/$NAME/tree/xmhw_amd/libxmhw_amd.so $R/ab/$NAME.so
else
  C=$R/xmhw_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $FLAGS \
      -c $C/kernels_sorted.hip -o /tmp/abv/$NAME/kernels_sorted.o 2> /tmp/abv/$NAME/cc.log || { tail -20 /tmp/abv/$NAME/cc.log; exit 1; }
  OBJS=$(ls $C/build/*.o | grep -v kernels_sorted.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/$NAME.so $OBJS /tmp/abv/$NAME/kernels_sorted.o -ldl
fi
ls -la $R/ab/$NAME.so
