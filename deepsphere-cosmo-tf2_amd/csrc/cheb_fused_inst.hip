// One (plane rows, ELL width) instantiation set of the fused kernel; built once per pair with
//   -DDSPH_FUSED_PR=<rows> -DDSPH_FUSED_WT=<width>   (see the Makefile).
#include "cheb_fused_kernel.h"

namespace dsph {

#define DSPH_CAT_(a, b, c, d) a##b##c##d
#define DSPH_CAT(a, b, c, d) DSPH_CAT_(a, b, c, d)

int DSPH_CAT(launch_fused_, DSPH_FUSED_PR, _, DSPH_FUSED_WT)(const FusedArgs& args, int nb, int prec, int grid,
                                                             size_t lds, hipStream_t stream) {
  // four lanes share a region row, 128 rows per pass: at most 512 of the 576 rows of the smallest
  // plane carry an ELL row (the outermost ring never does), otherwise as many as the plane has rows
  // (1,168: the 9-ring region of K = 10 -- rows with an ELL row are its rings 0 .. 8, 32 x 32 = 1,024 of them: plane_rows_for)
  constexpr int RPL = DSPH_FUSED_PR == 576 ? 4 : (DSPH_FUSED_PR == 1168 ? 8 : (DSPH_FUSED_PR + G_ROWS - 1) / G_ROWS);
  return dispatch_nb_prec<DSPH_FUSED_PR, DSPH_FUSED_WT, RPL>(args, nb, prec, grid, lds, stream);
}

}  // namespace dsph
