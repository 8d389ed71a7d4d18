// Small HIP helpers shared by the kernel files (gfx950 only: 64-lane wavefronts).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cxk {

constexpr int kWave = 64;

// Sum over the 64 lanes of a wavefront; every lane returns the total.
__device__ __forceinline__ double WaveSum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

__device__ __forceinline__ double WaveMax(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, kWave));
  return v;
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
