#!/bin/bash
# usage (here): tools/build_strip_ab.sh <bits>[s] ...   (suffix s: with s_memtime stamps, DSPH_STAMPS_DUMP=1 prints them)
#   -> build_ab/sp_<bits>.so (tuning builds of the strip kernel, DSPH_SP_ABL); SP_EXTRA='-D...' SP_TAG=_name adds defines
cd "$(dirname "$0")/.." || exit 1
C=deepsphere-cosmo-tf2_amd/csrc
for b in "$@"; do
  mkdir -p /tmp/ab_$b
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$C -fno-slp-vectorize -DDSPH_SP_ABL=${b%%s} $( [[ $b == *s ]] && echo -DDSPH_SP_STAMPS ) $SP_EXTRA -c $C/cheb_strip.hip -o /tmp/ab_$b/cheb_strip.o || exit 1
  objs=$(ls $C/build/*.o | grep -v cheb_strip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/sp_$b$SP_TAG.so $objs /tmp/ab_$b/cheb_strip.o || exit 1
  echo built build_ab/sp_$b$SP_TAG.so
done
