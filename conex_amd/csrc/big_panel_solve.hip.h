// The register forward substitution with a factored 32 x 32 diagonal block that the blocked
// factorizations of big supernodes share (kernels_kkt_big.hip.h: big_panel; big_chol.hip).
#pragma once
#include "kernels_kkt.hip.h"

namespace cxk {

// Steps I .. NB-1 of the register forward substitution of big_panel (compile-time recursion:
// register indices and DPP controls are immediates).
template <int NB, int I>
struct BigPanelSolve {
  static __device__ __forceinline__ void run(const double (&a)[NB + 1], double (&x)[NB], double dinv) {
    if constexpr (I < NB) {
      x[I] *= ReadLane(dinv, I);
      if constexpr (I + 1 < NB) {
        const RowPair cp = Swap16(a[I]);  // a = rows 0/2 everywhere (L[0..15][I]), b = rows 1/3 (L[16..31][I])
        double c0 = cp.a, c1 = cp.b;
        double nx = -x[I];
        DppOperandFence(c0, c1, nx);
        constexpr int kLo0 = (I + 1 < 16) ? I + 1 : 16;
        constexpr int kHi0 = (I + 1 > 16) ? I + 1 : 16;
        DppColumns<NB, kLo0, 16, 0>::run(x, c0, nx);
        DppColumns<NB, kHi0, NB, 16>::run(x, c1, nx);
      }
      BigPanelSolve<NB, I + 1>::run(a, x, dinv);
    }
  }
};

}  // namespace cxk
