// The instantiations of cheb_istrip_kernel for one K (-DDSPH_IS_K=2 .. 5): one translation unit each, built in parallel.
#include "cheb_istrip_kernel.h"

namespace dsph {

template <int K, int CH>
static void (*istrip_pick_prec(int prec))(IStripArgs) {
  switch (prec) {
    case DSPH_PREC_FP32: return cheb_istrip_kernel<K, CH, DSPH_PREC_FP32>;
    case DSPH_PREC_BF16X6: return cheb_istrip_kernel<K, CH, DSPH_PREC_BF16X6>;
    default: return cheb_istrip_kernel<K, CH, DSPH_PREC_BF16X3>;
  }
}

#define DSPH_IS_CAT2(a, b) a##b
#define DSPH_IS_CAT(a, b) DSPH_IS_CAT2(a, b)
void (*DSPH_IS_CAT(istrip_kernel_k, DSPH_IS_K)(int ch, int prec))(IStripArgs) {
  if (ch == 1) {  // one or two input channels: the level-packed kernel
    switch (prec) {
      case DSPH_PREC_FP32: return cheb_istrip1_kernel<DSPH_IS_K, DSPH_PREC_FP32>;
      case DSPH_PREC_BF16X6: return cheb_istrip1_kernel<DSPH_IS_K, DSPH_PREC_BF16X6>;
      default: return cheb_istrip1_kernel<DSPH_IS_K, DSPH_PREC_BF16X3>;
    }
  }
  return ch == 4 ? istrip_pick_prec<DSPH_IS_K, 4>(prec) : istrip_pick_prec<DSPH_IS_K, 8>(prec);
}

}  // namespace dsph
