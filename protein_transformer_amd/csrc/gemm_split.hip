// ptamd_gemm in bf16x3 arithmetic: the 6- and 9-product instantiations of gemm_split_kernel.h (kernel description there).
#include "gemm_split_kernel.h"

namespace ptgemm {

int launch_split(const GemmParams &p, bool a_kmajor, bool b_kmajor, int splits, int products, hipStream_t st) {
  if (products == 9) return launch_layout<9, EPI_FULL>(p, a_kmajor, b_kmajor, splits, st);
  if (products == 3) return launch_split_f16x2(p, a_kmajor, b_kmajor, splits, st);  // gemm_f16x2.hip
  // the smallest epilogue that does the job (instruction-cache footprint, see gemm_common.h)
  const bool plain = p.slab != 0 || (!p.bias && !p.residual && !p.flags && p.dropout_p == 0.f);
  if (plain) return launch_layout<6, EPI_PLAIN>(p, a_kmajor, b_kmajor, splits, st);
  if (p.dropout_p == 0.f) return launch_layout<6, EPI_NODROP>(p, a_kmajor, b_kmajor, splits, st);
  return launch_layout<6, EPI_FULL>(p, a_kmajor, b_kmajor, splits, st);
}

}  // namespace ptgemm
