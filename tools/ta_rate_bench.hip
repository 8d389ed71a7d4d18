// Address-processing cost of the candidate operand gathers, HBM taken out of the picture: every
// workgroup (8 wavefronts) re-reads its own 80 KB window (L2-resident after the first pass) and
// reports shader cycles per wave-level load instruction per CU.
//   MODE 0  row gather, the A-operand layout of lmi_schur_mfma: lane (s, q) reads the 40 bytes
//           q*40.. of row s (row stride 160 B) as dwordx4, dwordx4, dwordx2
//   MODE 1  column gather, the B-operand layout: lane (q, j) reads element (k = 5 q + e, column j)
//           for e = 0..4 as five dwordx2: 16 neighbouring lanes read 128 contiguous bytes
//   MODE 2  fully coalesced dwordx4 (1 KB per instruction)
//   MODE 3  like 0 but five dwordx2
// Build: hipcc --offload-arch=gfx950 -O3 tools/ta_rate_bench.hip -o /tmp/ta && /tmp/ta
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x)                                                                \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int kTile = 320;        // doubles: 16 rows x 20 columns
constexpr int kTilesPerWg = 32;   // 80 KB window
constexpr int kPasses = 64;

template <int MODE>
__global__ void __launch_bounds__(512) gather(const double* __restrict__ src, double* out, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double* win = src + (size_t)blockIdx.x * kTilesPerWg * kTile;
  const int s = lane & 15, q = lane >> 4;
  double acc = 0;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int pass = 0; pass < kPasses; pass++) {
#pragma unroll
    for (int tt = 0; tt < 4; tt++) {
      const double* tile = win + (wave + 8 * tt) * kTile;
      if constexpr (MODE == 0) {
        const double* p = tile + s * 20 + q * 5;
        double2 a = *reinterpret_cast<const double2*>(p), b = *reinterpret_cast<const double2*>(p + 2);
        asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y));
        acc += a.x + a.y + b.x + b.y + p[4];
      } else if constexpr (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 5; e++) acc += tile[(5 * q + e) * 16 + s];
      } else if constexpr (MODE == 2) {
        const double2* p = reinterpret_cast<const double2*>(tile);
        const double2 a = p[lane], b = p[64 + lane];
        acc += a.x + a.y + b.x + b.y;
        if (lane < 32) acc += tile[256 + 2 * lane] + tile[257 + 2 * lane];
      } else {
        const double* p = tile + s * 20 + q * 5;
#pragma unroll
        for (int e = 0; e < 5; e++) acc += p[e];
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (acc == 12345.678) out[0] = acc;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int insts_per_tile, const double* src, double* out, long long* cyc) {
  for (int w = 0; w < 2; w++) gather<MODE><<<256, 512>>>(src, out, cyc);
  CHECK(hipDeviceSynchronize());
  std::vector<long long> h(256 * 8);
  CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  const double insts_cu = 8.0 * 4 * kPasses * insts_per_tile;  // per CU
  printf("{\"pattern\": \"%s\", \"cycles_per_load_instruction_per_cu\": %.1f, \"bytes_per_cycle_per_cu\": %.1f}\n", name,
         med / insts_cu, 8.0 * 4 * kPasses * kTile * 8 / med);
}

int main() {
  double *src, *out;
  long long* cyc;
  const size_t n = (size_t)256 * kTilesPerWg * kTile;
  CHECK(hipMalloc(&src, n * 8));
  CHECK(hipMemset(src, 0, n * 8));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMalloc(&cyc, sizeof(long long) * 256 * 8));
  run<0>("row gather x4 x4 x2 (A-operand layout)", 3, src, out, cyc);
  run<3>("row gather 5 x dwordx2", 5, src, out, cyc);
  run<1>("column gather 5 x dwordx2 (B-operand layout)", 5, src, out, cyc);
  run<2>("coalesced dwordx4", 3, src, out, cyc);
  return 0;
}
