// big_chol_dataflow: right-looking blocked Cholesky of one big supernode (more columns than the
// LDS kernels hold) with its off block and right-hand side, ONE launch, one workgroup per
// 32-column block column -- the device-side panel loop of DESIGN 8 item 1.
//
// Reference semantics: BlockCholeskyInPlace for one supernode, L^-1 applied to its separator
// columns and to the right-hand side (block_triangular_operations.cc:184-219, :114-147).
//
// The extended matrix E = [D; B^T; b^T] ((ns + s + 1) x ns): the rows under D transform exactly like
// rows of D below the diagonal block (x <- x L11^-T, then x_rest -= x_k L_rest,k^T), so L^-1 B and
// L^-1 b fall out of the factorization.  Workgroup j owns block column j (rows 32 j .. of E):
//
//   for i < j, as block column i is published:   X_j -= L_i[rows >= 32 j] (L_i[32 j .. 32 j + 31])^T
//        -- 16 x 16 x 4 fp64 MFMA tiles, X_j^T held in the accumulators, the operands straight from
//           memory (sc1 loads: block column i was written by another CU, possibly another XCD)
//   wave 0:  the 32 x 32 diagonal block, row per lane, the register elimination of the small
//            supernodes (ElimSteps)                                     ~4 us, THE dependent chain
//   waves 1 .. 7:  their rows times L11^-T (BigPanelSolve, 64 rows per pass), stored write-through
//   flag A (rows 32 j + 32 .. 32 j + 95 are final: what the next two diagonal blocks need) and
//   flag B (the whole block column is final): one word each, == the launch's generation.
//
// So the chain of diagonal blocks runs through wave 0 of consecutive workgroups with one hand-off
// (~1 us), one 32 x 32 x 32 update and one 64-row solve between two eliminations, while the other
// waves of every later workgroup apply each finished block column to their rows as it appears:
// ~8 us per block column where the host-driven loop (big_panel + a GEMM launch or two per panel)
// takes ~19.  Waits are bounded (kSpin polls, then *fail = 1) and a workgroup only ever waits for
// lower-numbered ones, which the dispatcher starts first.
#include <hip/hip_runtime.h>

#define CXK_DEVICE_FUNCTIONS_ONLY
#include "big_chol.h"
#include "big_panel_solve.hip.h"

namespace cxk {
namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int NB = 32, WAVES = 8, GROUPS = 2, LDW = 33;
constexpr int kSpin = 1 << 20;
constexpr size_t kLdsDoubles = (size_t)(WAVES - 1) * 64 * LDW + 2 * NB * LDW;
static_assert(32 + 64 * (WAVES - 1) * GROUPS == kBigCholMaxRows, "rows a workgroup covers");

// Diagnostic build (-DCXK_BIGCHOL_STAMPS, `make dbg`): s_memrealtime stamps (100 MHz, one clock for
// the chip) of waves 0 and 1 of every workgroup, tools/big_chol_stamps.py.
#ifdef CXK_BIGCHOL_STAMPS
__device__ long long g_big_chol_stamp[kBigCholMaxBlocks * 2 * 8];
#define BC_STAMP(i)                                                                                   \
  do {                                                                                                \
    if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) < 2)                                            \
      g_big_chol_stamp[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define BC_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ double LdA(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void StA(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Row r of E: element (r, c) is base[c * st].  Rows past the end alias row 0 (loads stay in bounds,
// values are masked, nothing is stored).
struct Row {
  double* base;
  int64_t st;
  bool ok;
};
__device__ __forceinline__ Row RowOf(const BigCholArgs& A, int r, int R) {
  Row x;
  x.ok = r < R;
  const int rc = x.ok ? r : 0;
  if (rc < A.ns) {
    x.base = A.D + rc;
    x.st = A.ns;
  } else if (rc < A.ns + A.s) {
    x.base = A.B + (size_t)(rc - A.ns) * A.ns;
    x.st = 1;
  } else {
    x.base = A.b;
    x.st = 1;
  }
  return x;
}

__device__ __forceinline__ void WaitFlag(const int* f, int gen, int* fail) {
  for (int spin = 0; spin < kSpin; spin++) {
    int v = 0;
    if ((threadIdx.x & 63) == 0) v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v = __builtin_amdgcn_readfirstlane(v);
    if (v == gen) return;
    __builtin_amdgcn_s_sleep(2);
  }
  if ((threadIdx.x & 63) == 0) atomicExch(fail, 1);  // (carries on with what is there: the grid must drain)
}

// X^T tiles of RT 16-row tiles x the 32 columns of the block column: C[rt][ct][e] = X[r][c],
// r = row0 + 16 rt + (lane & 15), c = 16 ct + (lane >> 4) + 4 e  (the MFMA's C / D layout).
template <int RT>
__device__ __forceinline__ void LoadTiles(const BigCholArgs& A, d4_t (&C)[RT][2], int row0, int R, int k0, int nb,
                                          int lane) {
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    const Row rw = RowOf(A, row0 + 16 * rt + (lane & 15), R);
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int c = 16 * ct + (lane >> 4) + 4 * e;
        const bool ok = rw.ok && c < nb;
        const double v = rw.base[(int64_t)(k0 + (c < nb ? c : 0)) * rw.st];
        C[rt][ct][e] = ok ? v : 0.0;
      }
  }
}

// X^T -= L_i[32 j .. 32 j + 31] (L_i[rows])^T, K = the 32 columns of block column i.
// A operand (16 c x 4 k): lane (k = lane >> 4, c = lane & 15); B operand (4 k x 16 r) the same.
// KB k-steps per batch of operand loads: 8 = the whole update behind ONE round trip (48 operand
// registers at RT = 4), 2 = four round trips with 12.
template <int RT, int KB, bool LOWER = false>  // LOWER: the diagonal block (RT = 2): its upper tile is never read
__device__ __forceinline__ void Update(const BigCholArgs& A, d4_t (&C)[RT][2], int row0, int R, int k0, int nb,
                                       int i, int lane) {
  const int kq = lane >> 4, il = lane & 15;
  const int64_t col0 = 32 * i + kq;
  const double* pa[2];
  bool oka[2];
#pragma unroll
  for (int ct = 0; ct < 2; ct++) {
    oka[ct] = 16 * ct + il < nb;
    pa[ct] = A.D + (k0 + (oka[ct] ? 16 * ct + il : 0)) + col0 * A.ns;
  }
  const double* pb[RT];
  int64_t stb[RT];
  bool okb[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) {
    const Row rw = RowOf(A, row0 + 16 * rt + il, R);
    pb[rt] = rw.base + col0 * rw.st;
    stb[rt] = rw.st;
    okb[rt] = rw.ok;
  }
#pragma unroll
  for (int h = 0; h < 8 / KB; h++) {
    double av[2][KB], bv[RT][KB];
#pragma unroll
    for (int q = 0; q < KB; q++) {
      const int kk = KB * h + q;
#pragma unroll
      for (int ct = 0; ct < 2; ct++) av[ct][q] = LdA(pa[ct] + (int64_t)(4 * kk) * A.ns);
#pragma unroll
      for (int rt = 0; rt < RT; rt++) bv[rt][q] = LdA(pb[rt] + (int64_t)(4 * kk) * stb[rt]);
    }
#pragma unroll
    for (int q = 0; q < KB; q++) {
#pragma unroll
      for (int ct = 0; ct < 2; ct++) av[ct][q] = oka[ct] ? -av[ct][q] : 0.0;
#pragma unroll
      for (int rt = 0; rt < RT; rt++) bv[rt][q] = okb[rt] ? bv[rt][q] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < KB; q++)
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
          if (!(LOWER && rt == 0 && ct == 1))
            C[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ct][q], bv[rt][q], C[rt][ct], 0, 0, 0);
  }
}

// A group's tiles parked in LDS between updates (the register layout as it is: entry q of lane l at
// [64 q + l]): the second group of a worker wave, which only very tall block columns have.
__device__ __forceinline__ void Park(double* my, const d4_t (&C)[4][2], int lane) {
#pragma unroll
  for (int rt = 0; rt < 4; rt++)
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int e = 0; e < 4; e++) my[64 * ((rt * 2 + ct) * 4 + e) + lane] = C[rt][ct][e];
}
__device__ __forceinline__ void Unpark(const double* my, d4_t (&C)[4][2], int lane) {
#pragma unroll
  for (int rt = 0; rt < 4; rt++)
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int e = 0; e < 4; e++) C[rt][ct][e] = my[64 * ((rt * 2 + ct) * 4 + e) + lane];
}

template <int RT>
__device__ __forceinline__ void TilesToLds(double* my, const d4_t (&C)[RT][2], int lane) {
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
      for (int e = 0; e < 4; e++) my[(16 * rt + (lane & 15)) * LDW + 16 * ct + (lane >> 4) + 4 * e] = C[rt][ct][e];
}

// BigPanelSolve with the factored block read from LDS one column ahead of its use (lrow = row l32 of
// the image) instead of held in 64 registers: the worker waves keep two groups of tiles (128
// registers) through the solve of the first.  Same operations in the same order.
template <int I>
struct SolveFromLds {
  static __device__ __forceinline__ void run(const double* lrow, double (&x)[NB], double dinv, double aI) {
    if constexpr (I < NB) {
      x[I] *= ReadLane(dinv, I);
      if constexpr (I + 1 < NB) {
        const double anext = lrow[I + 1];
        const RowPair cp = Swap16(aI);
        double c0 = cp.a, c1 = cp.b;
        double nx = -x[I];
        DppOperandFence(c0, c1, nx);
        constexpr int kLo0 = (I + 1 < 16) ? I + 1 : 16;
        constexpr int kHi0 = (I + 1 > 16) ? I + 1 : 16;
        DppColumns<NB, kLo0, 16, 0>::run(x, c0, nx);
        DppColumns<NB, kHi0, NB, 16>::run(x, c1, nx);
        SolveFromLds<I + 1>::run(lrow, x, dinv, anext);
      }
    }
  }
};

__global__ void __launch_bounds__(64 * WAVES) big_chol_dataflow(BigCholArgs A) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l32 = lane & 31;
  const int j = blockIdx.x, k0 = NB * j;
  const int ns = A.ns, nb = ns - k0 < NB ? ns - k0 : NB;
  const int R = ns + A.s + (A.b ? 1 : 0);
  double* l11 = lds;                 // the factored diagonal block, row-major, LDW apart
  double* my0 = lds + NB * LDW;      // wave 0's transposition image
  double* myw = lds + 2 * NB * LDW + (size_t)(wave > 0 ? wave - 1 : 0) * 64 * LDW;
  const int* flagA = A.flags;        // [2 i]: rows 32 i + 32 .. 32 i + 95 of block column i are final
  const int gen = A.gen;

  if (wave == 0) {
    // ---------------------------------------------------------------- the diagonal block
    __builtin_amdgcn_s_setprio(3);  // THE dependent chain: ahead of the wave that shares its SIMD
    d4_t C[2][2];
    BC_STAMP(0);
    LoadTiles<2>(A, C, k0, R, k0, nb, lane);
    for (int i = 0; i < j; i++) {
      if (i == j - 1) BC_STAMP(1);
      // rows 32 j .. 32 j + 31 of block column i: inside its early part for i >= j - 2
      WaitFlag(flagA + 2 * i + (i + 2 >= j ? 0 : 1), gen, A.fail);
      if (i == j - 1) BC_STAMP(2);
      Update<2, 8, true>(A, C, k0, R, k0, nb, i, lane);
    }
    BC_STAMP(3);
    TilesToLds<2>(my0, C, lane);
    WaveSync();
    double a[NB + 1];
#pragma unroll
    for (int c = 0; c < NB; c++) {
      const double v = my0[l32 * LDW + c];
      a[c] = (l32 < nb && c <= l32) ? v : 0.0;
      if (c >= nb && l32 == c) a[c] = 1.0;  // padding pivots
    }
    a[NB] = 0.0;
    bool bad = false;
    BC_STAMP(4);
    ElimSteps<NB, 0, 0>::run(a, l32, bad, nb);
    BC_STAMP(5);
    if (bad && lane == 0) atomicExch(A.fail, 1);
    if (lane < NB) {
#pragma unroll
      for (int c = 0; c < NB; c++) l11[lane * LDW + c] = a[c];
    }
    __syncthreads();  // (A) the factored block is in LDS
    BC_STAMP(6);
    // (its copy in memory is only needed by flag B: 32 store instructions a lone wavefront takes
    // ~1 us to issue, off the chain of diagonal blocks)
    if (lane < nb) {
      double* dst = A.D + (k0 + lane) + (size_t)k0 * ns;
#pragma unroll
      for (int c = 0; c < NB; c++)
        if (c <= lane) StA(dst + (size_t)c * ns, a[c]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // (B) every wave's rows are stored
    BC_STAMP(7);
    if (lane == 0) __hip_atomic_store(A.flags + 2 * j + 1, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }

  // ------------------------------------------------------------------ the rows below it
  // group 0 (64 rows) lives in the accumulators from the first update to the solve; group 1 -- only
  // block columns with more than 32 + 7 x 64 rows have one -- is parked in LDS between updates
  int row0[GROUPS];
#pragma unroll
  for (int g = 0; g < GROUPS; g++)
    row0[g] = k0 + nb + 64 * ((wave - 1) + (WAVES - 1) * g);  // (a ragged last block: the rows of B^T and b^T follow at once)
  const bool has0 = row0[0] < R, has1 = row0[1] < R;
  d4_t C0[4][2];
  if (has0) LoadTiles<4>(A, C0, row0[0], R, k0, nb, lane);
  if (has1) {
    d4_t C1[4][2];
    LoadTiles<4>(A, C1, row0[1], R, k0, nb, lane);
    Park(myw, C1, lane);
  }
  BC_STAMP(0);
  if (has0)
    for (int i = 0; i < j; i++) {
      if (i == j - 1) BC_STAMP(1);
      WaitFlag(flagA + 2 * i + 1, gen, A.fail);
      if (i == j - 1) BC_STAMP(2);
      Update<4, 8>(A, C0, row0[0], R, k0, nb, i, lane);
      if (has1) {
        d4_t C1[4][2];
        Unpark(myw, C1, lane);
        Update<4, 4>(A, C1, row0[1], R, k0, nb, i, lane);
        Park(myw, C1, lane);
      }
    }
  BC_STAMP(3);
  // group 0's rows, row per lane, while wave 0 still eliminates (group 1 comes out of LDS first)
  d4_t C1[4][2];
  double x[NB];
  if (has0) {
    if (has1) {
      Unpark(myw, C1, lane);
      WaveSync();
    }
    TilesToLds<4>(myw, C0, lane);
    WaveSync();
#pragma unroll
    for (int c = 0; c < NB; c++) x[c] = myw[lane * LDW + c];
    WaveSync();
  }
  if (wave == 1) __builtin_amdgcn_s_setprio(2);  // its first rows are what the next diagonal block waits for
  __syncthreads();  // (A)
  BC_STAMP(4);
  if (has0) {
    const double* lrow = l11 + l32 * LDW;
    const double dinv = 1.0 / lrow[l32];  // lane c: 1 / L[c][c]
#pragma unroll
    for (int g = 0; g < GROUPS; g++) {
      if (g == 0 || has1) {
        if (g == 1) {
          TilesToLds<4>(myw, C1, lane);
          WaveSync();
#pragma unroll
          for (int c = 0; c < NB; c++) x[c] = myw[lane * LDW + c];
          WaveSync();
        }
        SolveFromLds<0>::run(lrow, x, dinv, lrow[0]);
        const Row rw = RowOf(A, row0[g] + lane, R);
        if (rw.ok) {
#pragma unroll
          for (int c = 0; c < NB; c++)
            if (c < nb) StA(rw.base + (int64_t)(k0 + c) * rw.st, x[c]);
        }
      }
      if (g == 0 && wave == 1) {
        BC_STAMP(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(A.flags + 2 * j, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        BC_STAMP(6);
        __builtin_amdgcn_s_setprio(0);
      }
    }
  } else if (wave == 1) {
    if (lane == 0) __hip_atomic_store(A.flags + 2 * j, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // (B)
}

}  // namespace

hipError_t LaunchBigChol(const BigCholArgs& a, hipStream_t stream) {
  if (!BigCholSupports(a.ns, a.s) || a.gen <= 0) return hipErrorInvalidValue;
  const size_t lds = sizeof(double) * kLdsDoubles;
  static PerDeviceOnce once;
  hipError_t e = once.run([] {
    const size_t lds = sizeof(double) * kLdsDoubles;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&big_chol_dataflow),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  });
  if (e != hipSuccess) return e;
  big_chol_dataflow<<<(a.ns + NB - 1) / NB, 64 * WAVES, lds, stream>>>(a);
  return hipGetLastError();
}

}  // namespace cxk

#ifdef CXK_BIGCHOL_STAMPS
extern "C" int cxk_debug_big_chol_stamps(long long* out, int blocks) {
  if (blocks > cxk::kBigCholMaxBlocks) blocks = cxk::kBigCholMaxBlocks;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cxk::g_big_chol_stamp), sizeof(long long) * 16 * (size_t)blocks) == hipSuccess ? 0 : 1;
}
#endif
