// Blocked Cholesky of ONE big supernode (panel beyond LDS) in a single launch: host-side interface
// of big_chol.hip.  Reference semantics: BlockCholeskyInPlace for one supernode and the forward
// substitution that rides in it (block_triangular_operations.cc:184-219, :114-147).
#pragma once
#include <hip/hip_runtime.h>

namespace cxk {

struct BigCholArgs {
  double* D;    // ns x ns column-major (lower triangle): A in, L out
  double* B;    // ns x s column-major off block: in, L^-1 B out (s may be 0)
  double* b;    // ns right-hand side or nullptr: in, L^-1 b out
  int ns, s;
  int* flags;   // 2 * kBigCholMaxBlocks words, never reset: a block column is done when its word == gen
  int gen;      // > 0, different from the previous launch on the same flags
  int* fail;    // *fail = 1 on a pivot that is not positive or a wait that ran out
};

constexpr int kBigCholMaxBlocks = 256;  // block columns = workgroups, all resident (one per CU)
constexpr int kBigCholMaxRows = 928;    // ns + s + 1: rows a workgroup keeps in registers (see big_chol.hip)

inline bool BigCholSupports(int ns, int s) {
  return ns + s + 1 <= kBigCholMaxRows && (ns + 31) / 32 <= kBigCholMaxBlocks;
}
hipError_t LaunchBigChol(const BigCholArgs& a, hipStream_t stream);

}  // namespace cxk
