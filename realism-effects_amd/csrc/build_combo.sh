#!/bin/bash
# development helper: build tuning variants of librfx_hip.so that swap SEVERAL objects at once into variants/ (git-ignored; they travel with
# gpurun).  Complements build_variants.sh (one file, several define sets).
#   ./build_combo.sh obj  <objname> <source.hip> "<extra flags>"      compile one object into variants/<objname>.o
#   ./build_combo.sh link <libname> k1_ssgi=<objname> k3_denoise=<objname> ...   link variants/librfx_<libname>.so, the named objects swapped in
# contraction follows csrc/Makefile (K3 / K4 contract, the rest do not) unless the extra flags say otherwise.
set -e
cd "$(dirname "$0")"
mkdir -p variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -I. -Wno-unused-function -Wno-unused-value -Wno-unused-result"
case $1 in
obj)
  name=$2; src=$3; extra=$4
  case $(basename $src) in k3_denoise*|k4_compose*) C="-ffp-contract=fast-honor-pragmas";; *) C="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc $F $C $extra -c $src -o variants/$name.o 2>/dev/null
  ;;
link)
  lib=$2; shift 2
  objs=""
  for o in rfx_api rfx_comm rfx_peer k0_import k1_ssgi k2_temporal k3_denoise k4_compose; do
    r=$o.o
    for kv in "$@"; do [ "${kv%%=*}" = "$o" ] && r=variants/${kv#*=}.o; done
    objs="$objs $r"
  done
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/librfx_$lib.so $objs 2>/dev/null
  ;;
*) echo "usage: see the header"; exit 2;;
esac
