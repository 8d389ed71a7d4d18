// Small HIP helpers shared by the kernel files (gfx950 only: 64-lane wavefronts).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

namespace cxk {

constexpr int kWave = 64;

// Function attributes (hipFuncSetAttribute: dynamic LDS above 64 KB) are PER DEVICE, and a process
// may hold contexts on several GPUs and call from several threads: a launcher keeps one of these
// per kernel and configures the kernel once on every device it meets.  (Setting an attribute twice
// from two racing threads is harmless; launching before it is set is not.)
struct PerDeviceOnce {
  std::atomic<unsigned long long> done{0};
  template <typename F>
  hipError_t run(F&& configure) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = configure();
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
  }
};

// Cross-lane exchange steps of the 64-lane butterfly without LDS traffic (ds_bpermute costs a
// full LDS round trip per step): xor 1 / xor 2 are DPP quad permutes, the 4- and 8-lane steps use
// row_half_mirror / row_mirror (after the previous steps every lane of a quad / half-row already
// holds the same partial, so any quad-to-quad / half-to-half exchange is the butterfly partner),
// the 16- and 32-lane steps use gfx950's v_permlane16_swap / v_permlane32_swap.  The operand pairs
// are exactly those of the xor butterfly, so sums are bit-identical to it.
template <int CTRL>
__device__ __forceinline__ double DppMove(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
struct RowPair {
  double a, b;
};
__device__ __forceinline__ RowPair Swap16(double v) {  // a = [r0 r0 r2 r2], b = [r1 r1 r3 r3]
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const u2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const u2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return {__hiloint2double(h.x, l.x), __hiloint2double(h.y, l.y)};
}
__device__ __forceinline__ RowPair Swap32(double v) {  // a = [lower lower], b = [upper upper]
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const u2 l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const u2 h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return {__hiloint2double(h.x, l.x), __hiloint2double(h.y, l.y)};
}

// Sum over the 64 lanes of a wavefront; every lane returns the total.
__device__ __forceinline__ double WaveSum(double v) {
  v += DppMove<0xB1>(v);   // quad_perm [1,0,3,2]
  v += DppMove<0x4E>(v);   // quad_perm [2,3,0,1]
  v += DppMove<0x141>(v);  // row_half_mirror
  v += DppMove<0x140>(v);  // row_mirror
  RowPair p = Swap16(v);
  v = p.a + p.b;
  p = Swap32(v);
  return p.a + p.b;
}

__device__ __forceinline__ double WaveMax(double v) {
  v = fmax(v, DppMove<0xB1>(v));
  v = fmax(v, DppMove<0x4E>(v));
  v = fmax(v, DppMove<0x141>(v));
  v = fmax(v, DppMove<0x140>(v));
  RowPair p = Swap16(v);
  v = fmax(p.a, p.b);
  p = Swap32(v);
  return fmax(p.a, p.b);
}

// Block-wide sum in a fixed (deterministic) order. `scratch` holds >= blockDim/64 doubles.
__device__ __forceinline__ double BlockSum(double v, double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = WaveSum(v);
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double t = 0;
  const int nw = (blockDim.x + 63) >> 6;
  for (int w = 0; w < nw; w++) t += scratch[w];
  return t;
}

// Orders LDS traffic between the lanes of ONE wavefront (program order is execution order
// inside a wave; this only stops the compiler from reordering across it).
__device__ __forceinline__ void WaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// value of `v` held by lane `src` (a compile-time constant after unrolling: v_readlane_b32 x2)
__device__ __forceinline__ double ReadLane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

}  // namespace cxk
