// Host interface of the MFMA dense-LMI Schur kernel (lmi_fused_mfma.hip): its own translation
// unit, so the kernel rebuilds in seconds and this header stays free of device code.
#pragma once
#include <hip/hip_runtime.h>

#include "lmi_types.h"

namespace cxk {

// Shapes the persistent producer / consumer kernel covers: instances of order 8, 12, 16, 20, 24;
// any other order 2 <= n <= 24 runs on the next instance up (LmiMfmaPaddedOrder) -- the caller then
// passes zero-padded copies of [A_1 .. A_m | C] in g.A / g.a_stride while g.n and g.W keep the
// order itself (the kernel masks its W loads).  Any number of variables m with m + 1 <= 24
// matrices (<= 32 for instances <= 16) and at most 512 stacked rows whose P image fits LDS -- twice
// (stage 2 of a constraint overlaps stage 1 of the next) or, failing that, once (they take turns).
// herm_d == 2 (complex Hermitian cones in their real representation of order n = 24): the folded
// form that reads and keeps the top half of every matrix only (any m <= 15 or 24 <= m <= 31).
int LmiMfmaPaddedOrder(int n);
bool LmiMfmaSupports(int n, int m, int herm_d = 0);

// ConstructSchurComplementSystem(DenseLMIConstraint*) for every member of the group
// (dense_lmi_constraint.cc:72-103); `cus` = multiprocessors of the device the stream runs on.
// ev_start / ev_stop (both or neither): HIP events attached to the dispatch (hipExtLaunchKernel), whose
// elapsed time is the kernel's own duration.
hipError_t LaunchLmiSchurMfma(const LmiGroup& g, const Arena& ar, int cus, hipStream_t stream,
                              hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

}  // namespace cxk
