// Operand / result lane layout of v_mfma_f64_4x4x4 (4 blocks) on gfx950, found empirically:
// A is one-hot at lane la, B holds lane + 1; every nonzero D lane then names the B lane that
// met A's lane.  Prints for each A lane the (D lane <- B lane) pairs.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(double* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++) {
    const double a = lane == la ? 1.0 : 0.0;
    const double b = lane + 1.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[la * 64 + lane] = d;
  }
}

int main() {
  double* d;
  hipMalloc(&d, sizeof(double) * 64 * 64);
  probe<<<1, 64>>>(d);
  static double h[64 * 64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; la++) {
    printf("A lane %2d:", la);
    for (int l = 0; l < 64; l++)
      if (h[la * 64 + l] != 0.0) printf(" D%d<-B%d", l, (int)h[la * 64 + l] - 1);
    printf("\n");
  }
  return 0;
}
