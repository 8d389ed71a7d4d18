// Register/scalar-operand specialisation of the dense-LMI Schur assembly for small orders
// (the benchmark shape n = 20, m = 20).  See DESIGN.md "lmi_schur_fused".
#pragma once
#include "kernels_lmi.hip.h"

namespace cxk {

inline bool LmiFusedSupports(int n, int m) {
  (void)n;
  (void)m;
  return false;
}

inline hipError_t LaunchLmiSchurFused(const LmiGroup& g, const Arena& ar, hipStream_t stream) {
  (void)g;
  (void)ar;
  (void)stream;
  return hipErrorNotSupported;
}

}  // namespace cxk
