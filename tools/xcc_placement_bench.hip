// Where does the dispatcher put workgroup b of consecutive launches on gfx950?  A sequence of
// kernels shaped like one KKT solve (256 x 768 threads, 219 / 28 / 4 x 256, 1 x 1024, ...) is
// launched back to back for several steps; every workgroup records HW_REG_XCC_ID.  Printed per
// launch: the XCD of block 0 and whether block b sits on XCD (xcc(0) + b) % 8 for all b.
// Speed-only knowledge (an XCD-affine order of the supernodes in a level): never correctness.
// Build: hipcc --offload-arch=gfx950 -O3 tools/xcc_placement_bench.hip -o /tmp/xcc && /tmp/xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void record(int* out, int spin) {
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    out[blockIdx.x] = (int)(id & 15);
  }
  // a little work so that launches overlap the way real ones do
  double s = threadIdx.x;
  for (int i = 0; i < spin; i++) s = s * 1.0000001 + 1e-9;
  if (s == 12345.678) out[0] = -1;
}

int main() {
  const int grids[] = {256, 219, 28, 4, 1, 4, 28, 219};
  const int blocks[] = {768, 256, 256, 256, 1024, 256, 256, 256};
  const int nk = 8, steps = 12;
  int* d;
  CHECK(hipMalloc(&d, sizeof(int) * nk * steps * 256));
  CHECK(hipMemset(d, 0xff, sizeof(int) * nk * steps * 256));
  hipStream_t st;
  CHECK(hipStreamCreate(&st));
  for (int s = 0; s < steps; s++)
    for (int k = 0; k < nk; k++) record<<<grids[k], blocks[k], 0, st>>>(d + (s * nk + k) * 256, 2000);
  CHECK(hipStreamSynchronize(st));
  std::vector<int> h(nk * steps * 256);
  CHECK(hipMemcpy(h.data(), d, sizeof(int) * h.size(), hipMemcpyDeviceToHost));
  for (int s = 0; s < steps; s++) {
    printf("step %2d:", s);
    for (int k = 0; k < nk; k++) {
      const int* o = &h[(s * nk + k) * 256];
      int bad = 0;
      for (int b = 0; b < grids[k]; b++) bad += o[b] != (o[0] + b) % 8;
      printf("  g%-3d x0=%d bad=%-3d", grids[k], o[0], bad);
    }
    printf("\n");
  }
  // the full placement of one mid-run launch of 28 workgroups
  const int* o = &h[(6 * nk + 2) * 256];
  printf("28 blocks:");
  for (int b = 0; b < 28; b++) printf(" %d", o[b]);
  printf("\n");
  return 0;
}
