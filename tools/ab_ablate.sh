#!/bin/bash
# builds timing-only ablations of the structured-tile kernel into build_ab/v_*.so (results are wrong by construction;
# never shipped).  Time them with tools/ab2.sh on the GPU box.   usage: tools/ab_ablate.sh [set]   (set: dirs | skip)
cd "$(dirname "$0")/.." || exit 1
mkdir -p build_ab && rm -f build_ab/v_*.so
build() {  # name, extra flags
  make -C deepsphere-cosmo-tf2_amd/csrc -j8 OBJDIR=/tmp/sb/abl_$1 OUTDIR=/tmp/sb/abl_$1_lib STRUCT_FLAGS="-fno-slp-vectorize $2" 2>&1 | grep -E "error"
  cp /tmp/sb/abl_$1_lib/libdsphere_hip.so build_ab/v_$1.so
}
build a_full ""
if [ "${1:-dirs}" = dirs ]; then
  build b_dirs4 "-DDSPH_ST_ABL_DIRS=4"
  build c_dirs0 "-DDSPH_ST_ABL_DIRS=0"
  build d_reads "-DDSPH_ST_ABL_READS=1"
  build e_reads_dirs0 "-DDSPH_ST_ABL_READS=1 -DDSPH_ST_ABL_DIRS=0"
else
  build b_nogather "-DDSPH_ST_ABL_SKIP=1"
  build c_nocontract "-DDSPH_ST_ABL_SKIP=2"
  build d_nodma "-DDSPH_ST_ABL_SKIP=4"
  build e_nostore "-DDSPH_ST_ABL_SKIP=8"
  build f_nogather_nocontract "-DDSPH_ST_ABL_SKIP=3"
  build g_only_dma "-DDSPH_ST_ABL_SKIP=11"
  build h_skeleton "-DDSPH_ST_ABL_SKIP=15"
fi
ls build_ab/v_*.so
