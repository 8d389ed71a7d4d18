// Measured fp64 issue rates on gfx950: back-to-back v_mfma_f64_16x16x4, v_mfma_f64_4x4x4 (4 blocks),
// v_fma_f64 and v_fmac_f64_dpp row_newbcast, at 1 / 2 / 4 wavefronts per SIMD, every CU busy.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/peak && /tmp/peak
// Prints one JSON line per variant: cycles per instruction per SIMD (s_memtime) and chip TFLOP/s
// (hipEvent wall).  The 16x16x4 line is the measured peak the SYRK / GEMM roofline is priced against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double d4_t __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

constexpr int kIters = 16384;   // loop trips; each trip issues kUnroll instructions

template <int MODE>
__global__ void __launch_bounds__(1024) rate_kernel(double* out, long long* cyc, double seed) {
  __shared__ double lds_buf[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds_buf[i] = seed * i;
  const double* lp = lds_buf + lane + 64 * (threadIdx.x >> 6);
  double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  double2 rr = make_double2(0, 0);
  int iv[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  double ra[8], rb[8];
  {
    unsigned long long z = 0x9E3779B97F4A7C15ull * (blockIdx.x * 1024 + threadIdx.x + 1);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      ra[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      rb[i] = ((double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0) * 1e-3;
    }
  }
  int sv = 0;
  double a = seed + lane * 1e-3, b = seed * 0.5 - lane * 1e-3;
  d4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
  double s[16];
#pragma unroll
  for (int i = 0; i < 16; i++) s[i] = seed * i;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; it++) {
    if constexpr (MODE == 0) {          // 8 independent 16x16x4 accumulators
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    } else if constexpr (MODE == 1) {   // one dependent 16x16x4 chain (8 per trip)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
    } else if constexpr (MODE == 2) {   // 8 independent 4x4x4 (4 blocks) accumulators
#pragma unroll
      for (int i = 0; i < 8; i++) s[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s[i], 0, 0, 0);
    } else if constexpr (MODE == 3) {   // dependent 4x4x4 chain
#pragma unroll
      for (int i = 0; i < 8; i++) s[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s[0], 0, 0, 0);
    } else if constexpr (MODE == 4) {   // 16 independent v_fma_f64
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
    } else if constexpr (MODE == 5) {   // 16 independent v_fmac_f64_dpp row_newbcast
#pragma unroll
      for (int i = 0; i < 16; i++)
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(a), "v"(b));
    } else if constexpr (MODE == 6) {   // dependent v_fmac_f64 chain
#pragma unroll
      for (int i = 0; i < 16; i++) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(s[0]) : "v"(a), "v"(b));
    } else if constexpr (MODE == 7) {   // MFMA 16x16x4 interleaved with 4 DPP FMAs each (co-issue test)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++)
          asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(s[4 * i + j]) : "v"(a), "v"(b));
      }
    } else if constexpr (MODE == 9) {   // MFMA 16x16x4 + 2 ds_read_b64 in its shadow (results unused until the end)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 1], 0, 0, 0);
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512" : "=v"(r0), "=v"(r1) : "v"((unsigned)(size_t)lp + 64 * i));
      }
    } else if constexpr (MODE == 10) {  // MFMA 16x16x4 + 4 ds_read_b64
#pragma unroll
      for (int i = 0; i < 8; i++) {
        acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 1], 0, 0, 0);
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536"
                     : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"((unsigned)(size_t)lp + 64 * i));
      }
    } else if constexpr (MODE == 11) {  // MFMA 16x16x4 + 1 ds_read_b128
#pragma unroll
      for (int i = 0; i < 8; i++) {
        acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 1], 0, 0, 0);
        asm volatile("ds_read_b128 %0, %1" : "=v"(rr) : "v"((unsigned)(size_t)(lds_buf + 2 * lane) + 128 * i));
      }
    } else if constexpr (MODE == 12) {  // 2 ds_read_b64 alone (no MFMA): the wave's own LDS issue rate
#pragma unroll
      for (int i = 0; i < 8; i++)
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512" : "=v"(r0), "=v"(r1) : "v"((unsigned)(size_t)lp + 64 * i));
    } else if constexpr (MODE == 13) {  // MFMA 16x16x4 + 2 ds_write_b64
#pragma unroll
      for (int i = 0; i < 8; i++) {
        acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 1], 0, 0, 0);
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %1 offset:512" : : "v"((unsigned)(size_t)lp + 64 * i), "v"(a) : "memory");
      }
    } else if constexpr (MODE == 14) {  // MFMA 16x16x4 + 8 integer VALU ops
#pragma unroll
      for (int i = 0; i < 8; i++) {
        acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 1], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[j]) : "v"(lane));
      }
    } else if constexpr (MODE == 15) {  // MFMA 16x16x4 + 16 integer VALU ops
#pragma unroll
      for (int i = 0; i < 8; i++) {
        acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 1], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 16; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[j & 7]) : "v"(lane));
      }
    } else if constexpr (MODE == 16) {  // MFMA 16x16x4 + 8 scalar ALU ops
#pragma unroll
      for (int i = 0; i < 8; i++) {
        acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 1], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv));
      }
    } else if constexpr (MODE == 17) {  // 8 independent 16x16x4 accumulators, RANDOM operands (power / clock under load)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[i], rb[(i + it) & 7], acc[i], 0, 0, 0);
    } else if constexpr (MODE == 8) {   // MFMA 16x16x4 interleaved with 8 plain FMAs each
#pragma unroll
      for (int i = 0; i < 2; i++) {
        acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(s[8 * i + j]) : "v"(a), "v"(b));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  double r = r0 + r1 + r2 + r3 + rr.x + rr.y + sv;
#pragma unroll
  for (int j = 0; j < 8; j++) r += iv[j];
#pragma unroll
  for (int i = 0; i < 8; i++) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; i++) r += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (lane == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

struct Variant {
  const char* name;
  int mode;
  int insts_per_trip;
  double flop_per_inst;   // per wave instruction
};

template <int MODE>
void run(const Variant& v, int waves_per_simd, double* out, long long* cyc, int ncu) {
  const int threads = 256 * waves_per_simd;  // 4 SIMDs x waves_per_simd wavefronts
  const int grid = ncu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; w++) rate_kernel<MODE><<<grid, threads>>>(out, cyc, 1.0 + 1e-9 * w);
  CHECK(hipDeviceSynchronize());
  const int reps = 5;
  CHECK(hipEventRecord(e0));
  for (int w = 0; w < reps; w++) rate_kernel<MODE><<<grid, threads>>>(out, cyc, 1.0 + 1e-9 * w);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const int nw = grid * threads / 64;
  std::vector<long long> h(nw);
  CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double med = (double)h[nw / 2];
  const double insts = (double)kIters * v.insts_per_trip;
  // cycles per instruction per SIMD: the waves_per_simd wavefronts of a SIMD share it
  const double cyc_per_inst_simd = med / insts / waves_per_simd;
  const double flops = insts * v.flop_per_inst * nw * reps;
  printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_per_simd\": %.2f, "
         "\"cycles_per_inst_per_wave\": %.2f, \"chip_tflops\": %.2f, \"ms\": %.4f}\n",
         v.name, waves_per_simd, cyc_per_inst_simd, med / insts, flops / (ms * 1e-3) / 1e12, ms / reps);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.gcnArchName, ncu, p.clockRate / 1000);
  double* out;
  long long* cyc;
  CHECK(hipMalloc(&out, sizeof(double) * ncu * 1024));
  CHECK(hipMalloc(&cyc, sizeof(long long) * ncu * 16));
  const Variant vs[] = {
      {"mfma_f64_16x16x4 independent", 0, 8, 2.0 * 16 * 16 * 4},
      {"mfma_f64_16x16x4 dependent", 1, 8, 2.0 * 16 * 16 * 4},
      {"mfma_f64_4x4x4_4b independent", 2, 8, 2.0 * 4 * 4 * 4 * 4},
      {"mfma_f64_4x4x4_4b dependent", 3, 8, 2.0 * 4 * 4 * 4 * 4},
      {"v_fmac_f64 independent", 4, 16, 128.0},
      {"v_fmac_f64_dpp row_newbcast independent", 5, 16, 128.0},
      {"v_fmac_f64 dependent", 6, 16, 128.0},
      {"mfma16 + 4 dpp fma interleaved (per group of 5)", 7, 20, (2048.0 + 4 * 128.0) / 5},
      {"mfma16 + 8 fma interleaved (per group of 9)", 8, 18, (2048.0 + 8 * 128.0) / 9},
      {"mfma16 + 2 ds_read_b64 each (per mfma)", 9, 8, 2048.0},
      {"mfma16 + 4 ds_read_b64 each (per mfma)", 10, 8, 2048.0},
      {"mfma16 + 1 ds_read_b128 each (per mfma)", 11, 8, 2048.0},
      {"2 ds_read_b64 alone (per pair)", 12, 8, 0.0},
      {"mfma16 + 2 ds_write_b64 each (per mfma)", 13, 8, 2048.0},
      {"mfma16 + 8 v_add_u32 each (per mfma)", 14, 8, 2048.0},
      {"mfma16 + 16 v_add_u32 each (per mfma)", 15, 8, 2048.0},
      {"mfma16 + 8 s_add_u32 each (per mfma)", 16, 8, 2048.0},
      {"mfma_f64_16x16x4 independent, random operands", 17, 8, 2.0 * 16 * 16 * 4},
  };
  for (int w : {1, 2, 4}) {
    run<0>(vs[0], w, out, cyc, ncu);
    run<1>(vs[1], w, out, cyc, ncu);
    run<2>(vs[2], w, out, cyc, ncu);
    run<3>(vs[3], w, out, cyc, ncu);
    run<4>(vs[4], w, out, cyc, ncu);
    run<5>(vs[5], w, out, cyc, ncu);
    run<6>(vs[6], w, out, cyc, ncu);
    run<7>(vs[7], w, out, cyc, ncu);
    run<8>(vs[8], w, out, cyc, ncu);
    run<9>(vs[9], w, out, cyc, ncu);
    run<10>(vs[10], w, out, cyc, ncu);
    run<11>(vs[11], w, out, cyc, ncu);
    run<12>(vs[12], w, out, cyc, ncu);
    run<13>(vs[13], w, out, cyc, ncu);
    run<14>(vs[14], w, out, cyc, ncu);
    run<15>(vs[15], w, out, cyc, ncu);
    run<16>(vs[16], w, out, cyc, ncu);
    run<17>(vs[17], w, out, cyc, ncu);
  }
  return 0;
}
