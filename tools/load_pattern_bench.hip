// How fast can one CU pull a constraint's 16-row x 20-column fp64 tiles (2560 contiguous bytes)
// straight into the MFMA A-operand layout (lane = 16 q + row: 5 contiguous doubles of row `row`
// starting at column 5 q) compared with a fully coalesced dwordx4 stream of the same bytes?
// 256 workgroups x WAVES wavefronts, each wavefront walks its own tiles of a 64 MB array (the C4
// A matrices: Infinity-Cache resident on repeat), DEPTH tiles in flight per wavefront.
// Prints GB/s per variant.  Build: hipcc --offload-arch=gfx950 -O3 tools/load_pattern_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int kTileDoubles = 320;  // 16 rows x 20 columns

// MODE 0: coalesced dwordx4 (each tile = 2.5 wave loads; done as 5 loads per 2 tiles)
// MODE 1: MFMA layout, 5 x dwordx2 per lane per tile
// MODE 2: MFMA layout, 3 x dwordx4 per lane per tile (16-byte aligned, 20 % over-fetch from L1)
// MODE 3: "4x4 quarter" layout: lane = 16 b + 4 kg + i -> row 4 b + i, columns 5 kg..: a quarter
//         wave covers 4 whole rows (640 contiguous bytes), 5 x dwordx2
template <int MODE, int DEPTH>
__global__ void __launch_bounds__(512) pull(const double* __restrict__ src, size_t tiles, double* out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const size_t gw = (size_t)blockIdx.x * nw + wave, tw = (size_t)gridDim.x * nw;
  double acc = 0;
  if constexpr (MODE == 0) {
    for (size_t t = gw * 2; t + 1 < tiles; t += tw * 2 * DEPTH) {
      double2 v[DEPTH][5];
#pragma unroll
      for (int d = 0; d < DEPTH; d++) {
        const size_t tt = t + (size_t)d * tw * 2;
        const double2* p = reinterpret_cast<const double2*>(src + (tt < tiles - 1 ? tt : 0) * kTileDoubles);
#pragma unroll
        for (int u = 0; u < 5; u++) v[d][u] = p[u * 64 + lane];
      }
#pragma unroll
      for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int u = 0; u < 5; u++) acc += v[d][u].x + v[d][u].y;
    }
  } else {
    int row, q;
    if constexpr (MODE == 3) {
      row = 4 * (lane >> 4) + (lane & 3);
      q = (lane >> 2) & 3;
    } else {
      row = lane & 15;
      q = lane >> 4;
    }
    for (size_t t = gw; t < tiles; t += tw * DEPTH) {
      if constexpr (MODE == 2) {
        // aligned 48-byte windows: q=0: doubles 0..5, q=1: 4..9, q=2: 10..15, q=3: 14..19
        const int off = q == 0 ? 0 : q == 1 ? 4 : q == 2 ? 10 : 14;
        double2 v[DEPTH][3];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
          const size_t tt = t + (size_t)d * tw;
          const double2* p = reinterpret_cast<const double2*>(src + (tt < tiles ? tt : 0) * kTileDoubles + row * 20 + off);
#pragma unroll
          for (int u = 0; u < 3; u++) v[d][u] = p[u];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
          for (int u = 0; u < 3; u++) acc += v[d][u].x + v[d][u].y;
      } else {
        double v[DEPTH][5];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
          const size_t tt = t + (size_t)d * tw;
          const double* p = src + (tt < tiles ? tt : 0) * kTileDoubles + row * 20 + q * 5;
#pragma unroll
          for (int u = 0; u < 5; u++) v[d][u] = p[u];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
#pragma unroll
          for (int u = 0; u < 5; u++) acc += v[d][u];
      }
    }
  }
  if (acc == 12345.678) out[0] = acc;  // keep the loads alive
}

template <int MODE, int DEPTH>
void run(const char* name, const double* src, size_t tiles, double* out, int waves) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; w++) pull<MODE, DEPTH><<<256, waves * 64>>>(src, tiles, out);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int w = 0; w < reps; w++) pull<MODE, DEPTH><<<256, waves * 64>>>(src, tiles, out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)tiles * kTileDoubles * 8;
  printf("{\"pattern\": \"%s\", \"waves_per_cu\": %d, \"tiles_in_flight_per_wave\": %d, \"us\": %.2f, \"GBps\": %.0f}\n",
         name, waves, DEPTH, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9);
  fflush(stdout);
}

int main() {
  const size_t tiles = 26250;  // 1000 constraints x 26.25 tiles = 67.2 MB
  double *src, *out;
  CHECK(hipMalloc(&src, tiles * kTileDoubles * 8));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(src, 0, tiles * kTileDoubles * 8));
  for (int waves : {4, 8}) {
    run<0, 2>("coalesced dwordx4", src, tiles, out, waves);
    run<0, 4>("coalesced dwordx4", src, tiles, out, waves);
    run<1, 4>("mfma layout 5 x dwordx2", src, tiles, out, waves);
    run<1, 7>("mfma layout 5 x dwordx2", src, tiles, out, waves);
    run<2, 4>("mfma layout 3 x dwordx4 aligned windows", src, tiles, out, waves);
    run<2, 7>("mfma layout 3 x dwordx4 aligned windows", src, tiles, out, waves);
    run<3, 4>("quarter = 4 whole rows, 5 x dwordx2", src, tiles, out, waves);
    run<3, 7>("quarter = 4 whole rows, 5 x dwordx2", src, tiles, out, waves);
  }
  return 0;
}
