#!/bin/bash
# development helper: build tuning variants of librfx_hip.so into variants/ (git-ignored; they travel with gpurun).
#   ./build_variants.sh <kernel-file-stem> <name>:"<defines>" ...     e.g.  ./build_variants.sh k1_ssgi th4:"-DRFX_K1_TH=4" th8:"-DRFX_K1_TH=8"
# time them on the GPU box with tools/time_variants.sh
set -e
cd "$(dirname "$0")"
make -s
mkdir -p variants
stem=$1; shift
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function -Wno-unused-value -Wno-unused-result"
case $stem in k3_denoise|k4_compose) F="$F -ffp-contract=fast-honor-pragmas";; *) F="$F -ffp-contract=off";; esac
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  (
    /opt/rocm/bin/hipcc $F $defs -c $stem.hip -o variants/${stem}_$name.o 2>/dev/null
    objs=""
    for o in rfx_api rfx_comm rfx_peer k0_import k1_ssgi k2_temporal k3_denoise k4_compose; do
      if [ $o = $stem ]; then objs="$objs variants/${stem}_$name.o"; else objs="$objs $o.o"; fi
    done
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/librfx_${stem}_$name.so $objs 2>/dev/null
  ) &
done
wait
ls variants/*.so
