// Does a line survive in an XCD's L2 across a kernel boundary on gfx950 (same stream, same
// process)?  A single wavefront chases pointers through a small buffer:
//   kernel 1: pass A (cold: HBM / Infinity Cache), pass B (same kernel: L2 hit)
//   kernel 2 (launched right after, same workgroup index = same XCD): pass C
// If C costs what B costs the L2 keeps clean lines across launches; if it costs what A costs it
// is invalidated at kernel start and any "warm-up" of the next kernel's data can only reach the
// Infinity Cache.  A streaming kernel over 256 MB runs first to empty every cache.
// Build: hipcc --offload-arch=gfx950 -O3 tools/l2_survival_bench.hip -o /tmp/l2s && /tmp/l2s
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kHops = 64;
__global__ void chase(const int* __restrict__ next, int passes, long long* out, int* sink) {
  if (threadIdx.x != 0) return;
  int p = 0;
  for (int pass = 0; pass < passes; pass++) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int h = 0; h < kHops; h++) p = next[p];
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[pass] = t1 - t0 + (p == -1);
  }
  *sink = p;
}
__global__ void flush(const double* __restrict__ a, size_t n, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 1.2345) out[0] = s;
}
int main() {
  const int lines = 4096;                       // 4096 lines x 128 B = 512 KB, one hop per line
  std::vector<int> h(lines * 32, 0);
  unsigned z = 12345;
  std::vector<int> perm(lines);
  for (int i = 0; i < lines; i++) perm[i] = i;
  for (int i = lines - 1; i > 0; i--) { z = z * 1664525u + 1013904223u; const int j = z % (i + 1); std::swap(perm[i], perm[j]); }
  for (int i = 0; i < lines; i++) h[perm[i] * 32] = perm[(i + 1) % lines] * 32;
  int* d; long long* out; int* sink; double* big; double* o2;
  CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&out, 64 * 8)); CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&o2, 8));
  const size_t nbig = (size_t)48 << 20;          // 384 MB
  CHECK(hipMalloc(&big, nbig * 8)); CHECK(hipMemset(big, 0, nbig * 8));
  for (int rep = 0; rep < 3; rep++) {
    flush<<<2048, 256>>>(big, nbig, o2);
    chase<<<1, 64>>>(d, 2, out, sink);          // passes A, B
    chase<<<1, 64>>>(d, 1, out + 8, sink);      // pass C: same chain from the start, next kernel
    CHECK(hipDeviceSynchronize());
    long long r[16]; CHECK(hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost));
    printf("{\"rep\": %d, \"cold_ticks_per_hop\": %.0f, \"same_kernel_again\": %.0f, \"next_kernel\": %.0f}\n", rep,
           r[0] / (double)kHops, r[1] / (double)kHops, r[8] / (double)kHops);
  }
  return 0;
}
