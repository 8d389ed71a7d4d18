// lmi_schur_mfma<N>: dense-LMI Schur assembly on the fp64 matrix pipe, persistent producer /
// consumer workgroups.  Reference semantics: ConstructSchurComplementSystem(DenseLMIConstraint*),
// dense_lmi_constraint.cc:72-103 (G(i,j) = <W A_i W, A_j>, AW(i) = tr(A_i W), AQc(i) = <C, W A_i W>,
// <w,c> = <C, W>, <c,Qc> = <C, W C W>), A_i and C symmetric (the host routes anything else to the
// literal kernel).
//
// Why this shape.  On gfx950 fp64 FMAs and fp64 MFMAs drain the SAME pipe (measured,
// profiles/r02/mfma_f64_peak.jsonl: 16 multiply-adds per clock per SIMD either way, no overlap
// between the two), so the kernel is priced by multiply-adds executed, and at the benchmark shape
// (n = m = 20) the 72 KB of A per constraint need as long to arrive (6.3 TB/s) as the pipe needs for
// them: the design goal is "both busy all the time".
//
//   stage 1  P = S W,  S = [A_1; ..; A_m; C] stacked (M1 N rows x N).  A 16-row tile is ONE MFMA
//            A-operand per k-step in both shapes used: v_mfma_f64_16x16x4 against W[:, 0..15] and
//            v_mfma_f64_4x4x4 (4 blocks) against W[:, 16..] -- the two instructions share the
//            A-operand layout (lane = 16 k + row; layout of the 4x4x4 form measured,
//            profiles/r02/mfma_f64_4x4x4_layout.txt), so N = 20 costs 5 x (64 + 16) cycles per tile
//            with no padded columns.  Operands come STRAIGHT from HBM in that layout: lane (row, q)
//            loads the N/4 consecutive doubles q N/4 .. of its row (the k-steps take k = q N/4 + e,
//            W's operand rows are permuted to match); a tile is 16 N contiguous doubles and this
//            gather streams as fast as a coalesced copy (tools/load_pattern_bench.hip: 6.3 TB/s).
//            No LDS staging of A, no VALU work on the data.
//   stage 2  G(i,j) = tr(P_i P_j) (A symmetric: the second product W (A_i W) is never formed):
//            a (M1 x N^2)(N^2 x M1) product over the P image in LDS.  For 17 <= M1 <= 24 the lower
//            triangle is covered by ONE 16 x 16 v_mfma_f64_16x16x4 tile (rows R..M1-1 x columns
//            0..15, R = M1 - 16) plus the two R x R triangles it leaves out, on v_mfma_f64_4x4x4
//            blocks (Triangles below); for M1 <= 16 by one tile.
//
// Roles.  One 768-thread workgroup per CU, persistent over its constraints.  Waves 0-7 (two per
// SIMD) are PRODUCERS: wave w owns the tiles t = w (mod 8), keeps the NEXT constraint's tiles in
// flight in registers (a tile's registers are reloaded as soon as its MFMAs have issued: a whole
// constraint of prefetch distance) and writes P into one of two LDS images.  Waves 8-11 (one per
// SIMD) are CONSUMERS, three iterations deep in all:
//
//   iteration it    producers            consumers
//                   stage 1 of c_it      contraction of c_{it-1} (K split over the waves -> partial
//                                        tiles in LDS buffer (it-1) & 1), traces of c_{it-1}, then the
//                                        epilogue of c_{it-2}: partial tiles of buffer it & 1 summed
//                                        in a fixed order -> G / AQc / <c,Qc>
//                   --------------------- one workgroup barrier ---------------------
//
// The pipe is the bound: with real operands a 16x16x4 instruction holds a SIMD for 64-80 cycles
// (profiles/r02/mfma_f64_peak.jsonl, random operands), an iteration at n = m = 20 issues ~6.3 k
// cycles of them per SIMD and takes ~8.5 k (in-kernel timeline: profiles/r02/lmi_schur_mfma_stamps.txt).
//
// All sums run in a fixed order: results are bit-reproducible run to run.  Differences to the
// reference's summation order are rounding-level (tests: <= 1e-13 on every Schur block).
#include "lmi_fused_mfma.h"

#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "device_utils.h"

#ifndef CXK_RELOAD_MODE
#define CXK_RELOAD_MODE 3  // when a producer asks for the next constraint's operands (producer loop below): 0 late, 1 early, 3 early from the second iteration on
#endif

namespace cxk {
namespace {

typedef double d4_t __attribute__((ext_vector_type(4)));

constexpr int kDestSlots = 64;  // per-workgroup table of output offsets (LDS)

#ifdef CXK_MFMA_STAMPS
// diagnostic build only (make dbg): s_memtime stamps of workgroups 0 and 200, per wave
__device__ long long g_mfma_stamp[2 * 16 * 64];
#define MSTAMP(slot)                                                                               \
  do {                                                                                             \
    if ((blockIdx.x == 0 || blockIdx.x == 200) && (threadIdx.x & 63) == 0 && (slot) < 64)          \
      g_mfma_stamp[((blockIdx.x ? 1 : 0) * 16 + (threadIdx.x >> 6)) * 64 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define MSTAMP(slot) \
  do {               \
  } while (0)
#endif

// H: the matrices are real representations [[Re, -Im], [Im, Re]] of complex Hermitian ones
// (hermitian_psd.cc: d = 2).  Products of such matrices have the same form, so only the TOP half of
// the rows of every A_i is read and only the top half of every P_i is formed and kept; the
// contraction over the full index space folds onto the top rows,
//   tr(P_x P_y) = 2 [ sum_{r,c < N/2} P_x[r][c] P_y[c][r]  -  sum_{r,c < N/2} P_x[r][N/2+c] P_y[c][N/2+r] ],
// (the bottom-left block is minus the top-right one): half the bytes, half the multiply-adds.
template <int N, bool H = false>
struct MfmaCfg {
  static_assert(N % 4 == 0 && N >= 8 && N < 32, "one 16-column tile (part of it for N < 16) plus 4-column blocks");
  static_assert(!H || N % 8 == 0, "complex order N/2, whose k-steps split evenly between the two sums");
  static constexpr int RPM = H ? N / 2 : N;   // rows of a matrix that are read / formed / kept
  static constexpr int NK = N / 4;            // stage-1 k-steps = doubles of its row a lane holds
  static constexpr int NC4 = N > 16 ? (N - 16) / 4 : 0;  // 4-column blocks beside the 16-column tile
  static constexpr int NC4A = NC4 > 0 ? NC4 : 1;         // (array extent)
  static constexpr int PIECES = 4 + NC4;      // stores per finished tile (StorePiece)
  static constexpr bool NARROW = N < 16;      // the 16-column tile has columns past N: every store is masked
  static constexpr bool THREE_TILES = N <= 16 || H;  // more than 24 matrices fit LDS (twice) only then
  static constexpr int LD = N + 1;            // odd row stride of the P image
  static constexpr int MS = RPM * LD + (6 - (RPM * LD) % 4) % 4;  // matrix stride = 2 (mod 4) doubles
  static constexpr int KSTEPS = N * N / 4;    // stage-2 k-steps
  static constexpr int PROD = 8, CONS = 4;    // producer / consumer wavefronts (two / one per SIMD)
  static constexpr int TPW = 4;               // tiles per producer wave (<= 32 tiles of 16 rows)
  static constexpr int YSLOTS = 1;            // tile slots a producer keeps for phase Y
  static constexpr bool DRAIN_HELP = N <= 20; // the producers take epilogues off the consumers in the drain
  static constexpr int THREADS = 64 * (PROD + CONS);
  static constexpr int CUT_A = (3 * RPM + 2) / 4;  // three tiles: the 3/4 cut of the K range (rows of a matrix)
  static constexpr int PARTS = THREE_TILES ? 6 : CONS;  // partial tiles of a contraction (three tiles: each in two K parts)
  static_assert(N % CONS == 0, "stage-2 rows (N/4 k-steps each) are dealt evenly to the consumer waves");
  // offset of the B-side operand of k-step bi inside the row block of P_y (doubles): P_y[4 bi + kq][r],
  // or for the second sum of the Hermitian form P_y[4 (bi - NK/2) + kq][N/2 + r]
  static constexpr int BOfs(int bi) { return H && bi >= NK / 2 ? 4 * (bi - NK / 2) * LD + RPM : 4 * bi * LD; }
  static_assert(MS % 4 == 2, "both stage-2 operand reads are bank-conflict free only then");
};

// LDS traffic of this wave has landed, then the workgroup barrier.  Not __syncthreads(): its
// release fence would also wait for the producers' prefetch loads (vmcnt).
__device__ __forceinline__ void LdsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int N, bool H>
__device__ __forceinline__ int PAddr(int rho) {  // stacked row -> offset of its row in the P image
  using Cfg = MfmaCfg<N, H>;
  const int mat = rho / Cfg::RPM;
  return mat * Cfg::MS + (rho - mat * Cfg::RPM) * Cfg::LD;
}

// Per-lane geometry of the producer's tiles: the same for every constraint, computed once.
// The host stores C behind the m matrices A_i of its constraint (LmiGroup::a_stride = (m+1) n^2),
// so the stacked rows [A_1; ..; A_m; C] are ONE contiguous (M1 N) x N row-major array.
template <int N>
struct TileGeom {
  unsigned src[MfmaCfg<N>::TPW];  // byte offset of the lane's NK operands inside the constraint's block
  int p16[MfmaCfg<N>::TPW][4];    // P-image offsets (doubles) of the four 16x16x4 result elements
  int p4[MfmaCfg<N>::TPW];        // P-image offset of the 4x4x4 result (first 4-column block)
  unsigned keep;                  // bit 8 tt + e (e < 4): 16x16x4 element e is a real row; bit 8 tt + 4: the 4x4x4 row is
};

template <int N, bool H>
__device__ __forceinline__ void MakeGeom(TileGeom<N>& gm, int wave, int lane, int rows, int nt1) {
  using Cfg = MfmaCfg<N, H>;
  const int s = lane & 15, q = lane >> 4;
  gm.keep = 0;
#pragma unroll
  for (int tt = 0; tt < Cfg::TPW; tt++) {
    int t = wave + Cfg::PROD * tt;
    const bool real = t < nt1;
    t = real ? t : nt1 - 1;             // a slot past the last tile re-reads the last tile (results unused)
#pragma unroll
    for (int e = 0; e < 4; e++)
      gm.keep |= (real && 16 * t + q + 4 * e < rows && (!Cfg::NARROW || s < N) ? 1u : 0u) << (8 * tt + e);
    gm.keep |= (real && 16 * t + 4 * ((lane >> 2) & 3) + q < rows ? 1u : 0u) << (8 * tt + 4);
    int rho = 16 * t + s;
    rho = rho < rows ? rho : rows - 1;  // rows past the last matrix: any valid address, results unused
    const int mat = rho / Cfg::RPM;  // (H: the kept rows of a matrix are its first RPM, matrices stay N x N apart)
    gm.src[tt] = (unsigned)(mat * N * N + (rho - mat * Cfg::RPM) * N + q * Cfg::NK) * 8u;
    // 16x16x4 result: element e of lane l is (row (l >> 4) + 4 e, column l & 15)
#pragma unroll
    for (int e = 0; e < 4; e++) gm.p16[tt][e] = PAddr<N, H>(16 * t + q + 4 * e) + s;
    // 4x4x4 result: lane l holds (row 4 ((l >> 2) & 3) + (l >> 4), column l & 3) of each block
    gm.p4[tt] = PAddr<N, H>(16 * t + 4 * ((lane >> 2) & 3) + q) + 16 + (lane & 3);
  }
}

// Lane (row s, k-group q) fetches its NK doubles of the stacked row it owns in tile slot tt:
// uniform base + per-lane 32-bit offset (no address arithmetic on the vector ALU, which the
// fp64 matrix instructions keep busy: measured, every VALU instruction costs pipe time).
template <int N>
__device__ __forceinline__ void LoadTile(double (&a)[MfmaCfg<N>::NK], const TileGeom<N>& gm, int tt,
                                         const double* __restrict__ block) {
  constexpr int NK = MfmaCfg<N>::NK;
  const double* src = reinterpret_cast<const double*>(reinterpret_cast<const char*>(block) + gm.src[tt]);
#pragma unroll
  for (int e = 0; e < NK; e++) a[e] = src[e];
}

template <int N>
struct WOps {  // stage-1 B operands of one constraint: k-step e uses W row q NK + e
  double w16[MfmaCfg<N>::NK];
  double w4[MfmaCfg<N>::NC4A][MfmaCfg<N>::NK];
};

// nr = the order W is stored in (nr x nr, nr <= N): an order that is not a multiple of four runs on
// the next instance up, its A_i zero-padded by the host and W zero-padded here (P = A W then
// carries zero rows and columns, which add nothing to any trace).
template <int N>
__device__ __forceinline__ void LoadW(WOps<N>& w, const double* __restrict__ Wg, int lane, int nr) {
  using Cfg = MfmaCfg<N>;
  const int s = lane & 15, q = lane >> 4, j = lane & 3;
  if (nr != N) {  // uniform
#pragma unroll
    for (int e = 0; e < Cfg::NK; e++) {
      const int r = q * Cfg::NK + e;
      const double* row = Wg + (size_t)(r < nr ? r : nr - 1) * nr;
      const double v = row[s < nr ? s : nr - 1];
      w.w16[e] = (r < nr && s < nr) ? v : 0.0;
#pragma unroll
      for (int cb = 0; cb < Cfg::NC4; cb++) {
        const int c = 16 + 4 * cb + j;
        const double u = row[c < nr ? c : nr - 1];
        w.w4[cb][e] = (r < nr && c < nr) ? u : 0.0;
      }
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < Cfg::NK; e++) {
    const double* row = Wg + (size_t)(q * Cfg::NK + e) * N;
    w.w16[e] = row[Cfg::NARROW && s >= N ? N - 1 : s];
#pragma unroll
    for (int cb = 0; cb < Cfg::NC4; cb++) w.w4[cb][e] = row[16 + 4 * cb + j];
  }
  if constexpr (Cfg::NARROW) {  // columns past N of the 16-column tile: zero operand (their results are never stored)
#pragma unroll
    for (int e = 0; e < Cfg::NK; e++) w.w16[e] = s < N ? w.w16[e] : 0.0;
  }
}

template <int N>
struct TileAcc {  // results of one tile: the 16-column MFMA tile and the 4-column blocks
  d4_t a16;
  double a4[MfmaCfg<N>::NC4A];
};

// Piece e of a finished tile: pieces 0..3 are the 16x16x4 elements (rows (l >> 4) + 4 e, column
// l & 15), pieces 4.. the 4x4x4 blocks.
template <int N, bool MASKED>
__device__ __forceinline__ void StorePiece(const TileAcc<N>& r, int e, double* __restrict__ Pb, const TileGeom<N>& gm, int tp) {
  if (e < 4) {
    if (!MASKED || ((gm.keep >> (8 * tp + e)) & 1)) Pb[gm.p16[tp][e]] = r.a16[e];
  } else {
    if (!MASKED || ((gm.keep >> (8 * tp + 4)) & 1)) Pb[gm.p4[tp] + 4 * (e - 4)] = r.a4[e - 4];
  }
}

// Tile slot tt's MFMAs issue back to back; the stores of the PREVIOUS slot's results are dealt
// into the gaps between them (an MFMA occupies the pipe for 64 / 16 cycles during which the wave
// may issue LDS work), pinned there by the scheduling barriers.  A slot that still has MFMAs to
// issue is never preceded by the ragged last tile, so these stores need no lane mask (orders below
// 16 mask the tile's unused columns everywhere).
template <int N, bool HAS_PREV>
__device__ __forceinline__ void FullStep(TileAcc<N>& cur, const double (&a)[MfmaCfg<N>::NK], const WOps<N>& w,
                                         const TileAcc<N>& prev, double* __restrict__ Pb, const TileGeom<N>& gm, int tt) {
  using Cfg = MfmaCfg<N>;
  const d4_t zero4 = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int e = 0; e < Cfg::NK; e++) {
    cur.a16 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[e], w.w16[e], e == 0 ? zero4 : cur.a16, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (HAS_PREV && e < Cfg::PIECES) StorePiece<N, Cfg::NARROW>(prev, e, Pb, gm, tt - 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int cb = 0; cb < Cfg::NC4; cb++)
      cur.a4[cb] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[e], w.w4[cb][e], e == 0 ? 0.0 : cur.a4[cb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (HAS_PREV && Cfg::PIECES > Cfg::NK) {  // fewer k-steps than pieces (N < 16): the rest follow
#pragma unroll
    for (int e = Cfg::NK; e < Cfg::PIECES; e++) StorePiece<N, Cfg::NARROW>(prev, e, Pb, gm, tt - 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

struct ConstraintPtrs {
  const double* block;  // [A_1 .. A_m | C] of the constraint
  const double* Wg;
};
__device__ __forceinline__ ConstraintPtrs Member(const LmiGroup& g, int mem, int nn) {
  ConstraintPtrs p;
  p.block = g.A + (size_t)mem * g.a_stride;
  p.Wg = g.W + (size_t)mem * nn;
  return p;
}

// One consumer wave's share of the stage-2 contraction: ROWS rows rr0 .. of the N x N index space
// (NK k-steps each) of the tile whose A-operand rows are the matrices at ra(lane) and whose
// B-operand columns are the matrices at ca(lane).  The operands of the next row are read in the
// gaps between this row's MFMAs (a lone wavefront issues a ds_read_b64 every ~17 cycles: a row's
// reads up front would idle the pipe).
template <int N, int ROWS, bool H = false>
__device__ __forceinline__ d4_t Contract(const double* __restrict__ Pb, int ra, int ca, int rr0, int kq, int nrows = ROWS) {
  using Cfg = MfmaCfg<N, H>;
  constexpr int NK = Cfg::NK, LD = Cfg::LD;
  d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
  d4_t acc2 = (d4_t){0.0, 0.0, 0.0, 0.0};       // H: the second (subtracted) sum of the Hermitian form
  double a0[2][NK], b0[2][NK];
  const double* pa = Pb + ra + rr0 * LD + kq;   // A side: P_x[r][4 bi + kq]
  const double* pb = Pb + ca + kq * LD + rr0;   // B side: P_y[4 bi + kq][r]  (Cfg::BOfs)
#pragma unroll
  for (int bi = 0; bi < NK; bi++) {
    a0[0][bi] = pa[4 * bi];
    b0[0][bi] = pb[Cfg::BOfs(bi)];
  }
#pragma unroll
  for (int r = 0; r < ROWS; r++) {
    const int cb_ = r & 1, nb_ = cb_ ^ 1;
    const bool more = r + 1 < ROWS && r + 1 < nrows;
    if (r < nrows)  // wave-uniform (N rows do not always divide evenly)
#pragma unroll
    for (int bi = 0; bi < NK; bi++) {
      __builtin_amdgcn_sched_barrier(0);
      if (H && bi >= NK / 2)
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[cb_][bi], b0[cb_][bi], acc2, 0, 0, 0);
      else
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[cb_][bi], b0[cb_][bi], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // next row's operands, two per LDS instruction (ds_read2_b64: neighbours share a base
      // register): the A-side pair after an even k-step, the B-side pair after an odd one
      if (more) {
        if ((bi & 1) == 0) {
          a0[nb_][bi] = pa[(r + 1) * LD + 4 * bi];
          if (bi + 1 < NK) a0[nb_][bi + 1] = pa[(r + 1) * LD + 4 * (bi + 1)];
          if (bi + 1 == NK) b0[nb_][bi] = pb[Cfg::BOfs(bi) + r + 1];
        } else {
          b0[nb_][bi - 1] = pb[Cfg::BOfs(bi - 1) + r + 1];
          b0[nb_][bi] = pb[Cfg::BOfs(bi) + r + 1];
        }
      }
    }
  }
  if constexpr (H) acc -= acc2;
  return acc;
}

// The two R x R triangles (R = M1 - 16 <= 8) the 16 x 16 tile leaves out -- G over the matrices
// {0..R-1} and over {16..M1-1} -- on v_mfma_f64_4x4x4 (four independent 4 x 4 x 4 blocks per
// instruction, a quarter of the 16 x 16 x 4 instruction's pipe time).  Block of a lane
// ((lane >> 2) & 3) = 2 t + h: triangle t, K parity h (row r = 2 p + h of the N x N index space);
// instruction u covers the 4 x 4 sub-block (brow, bcol) = (0,0), (1,0), (1,1) of both triangles
// (u = 0 alone when R <= 4; slots past R alias R - 1).  A second 16 x 16 tile for these 2 x 15
// entries would double the contraction's pipe time for 12 % of its results.
// Layouts (measured, profiles/r02/mfma_f64_4x4x4_layout.txt): A[blk][i][k] and B[blk][k][j] sit in
// lane 16 k + 4 blk + (i or j), D[blk][i][j] in lane 16 i + 4 blk + j.
template <int N, bool BIG>  // BIG: R > 4
__device__ __forceinline__ void Triangles(const double* __restrict__ Pb, double* __restrict__ out, int lane, int M1) {
  using Cfg = MfmaCfg<N>;
  constexpr int NK = Cfg::NK, LD = Cfg::LD, MS = Cfg::MS;
  const int R = M1 - 16;
  const int blk = (lane >> 2) & 3, i = lane & 3, t = blk >> 1, h = blk & 1;
  // which of a step's four columns a lane supplies: the second triangle's blocks take them rotated by
  // two, or its matrices (16 further on: a multiple of 64 LDS banks) would alias the first's
  const int kq = ((lane >> 4) + 2 * t) & 3;
  const int lo = (t ? 16 : 0) + (i < R ? i : R - 1), hi = (t ? 16 : 0) + (4 + i < R ? 4 + i : R - 1);
  // operand offsets: A-side P_x[r][4 bi + kq], B-side P_y[4 bi + kq][r], r = 2 p + h
  const int a_lo = lo * MS + h * LD + kq, a_hi = hi * MS + h * LD + kq;
  const int b_lo = lo * MS + kq * LD + h, b_hi = hi * MS + kq * LD + h;
  constexpr bool big = BIG;
  double acc0 = 0, acc1 = 0, acc2 = 0;
  constexpr int STEPS = (N / 2) * NK;  // step st: rows 2 (st / NK) + h, columns 4 (st % NK) + kq
  static_assert(STEPS % 2 == 0, "operands are fetched two steps at a time");
  auto aoff = [](int st) { return 2 * (st / NK) * LD + 4 * (st % NK); };
  auto boff = [](int st) { return 4 * (st % NK) * LD + 2 * (st / NK); };
  const double *pal = Pb + a_lo, *pbl = Pb + b_lo, *pah = Pb + a_hi, *pbh = Pb + b_hi;
  double al[2][2], ah[2][2], bl[2][2], bh[2][2];  // [buffer][step of the pair]
#pragma unroll
  for (int q = 0; q < 2; q++) {
    al[0][q] = pal[aoff(q)];
    bl[0][q] = pbl[boff(q)];
    ah[0][q] = big ? pah[aoff(q)] : 0.0;
    bh[0][q] = big ? pbh[boff(q)] : 0.0;
  }
#pragma unroll
  for (int st = 0; st < STEPS; st += 2) {
    const int cb_ = (st >> 1) & 1, nb_ = cb_ ^ 1;
    const bool more = st + 2 < STEPS;
    // six instructions per pair of steps; the next pair's operands arrive as four ds_read2_b64
    // dealt into the gaps
    __builtin_amdgcn_sched_barrier(0);
    acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(al[cb_][0], bl[cb_][0], acc0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      al[nb_][0] = pal[aoff(st + 2)];
      al[nb_][1] = pal[aoff(st + 3)];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (big) {
      acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(ah[cb_][0], bl[cb_][0], acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(ah[cb_][0], bh[cb_][0], acc2, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
      bl[nb_][0] = pbl[boff(st + 2)];
      bl[nb_][1] = pbl[boff(st + 3)];
    }
    __builtin_amdgcn_sched_barrier(0);
    acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(al[cb_][1], bl[cb_][1], acc0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (big) {
      if (more) {
        ah[nb_][0] = pah[aoff(st + 2)];
        ah[nb_][1] = pah[aoff(st + 3)];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(ah[cb_][1], bl[cb_][1], acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        bh[nb_][0] = pbh[boff(st + 2)];
        bh[nb_][1] = pbh[boff(st + 3)];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(ah[cb_][1], bh[cb_][1], acc2, 0, 0, 0);
    }
  }
  out[lane] = acc0;
  if (big) {
    out[64 + lane] = acc1;
    out[128 + lane] = acc2;
  }
}

// Epilogue of constraint c, one iteration after its contraction: the K-split partial tiles (sb) are
// summed in a fixed order and written out; entry e of the lower triangle by thread t0 + k nt.
template <int N, bool H>
__device__ __forceinline__ void Epilogue(const double* __restrict__ sb, const int64_t* __restrict__ dest, const int* __restrict__ etab,
                                         int c, const Arena& ar, int nout, bool two, bool three, double osc, int t0, int nt) {
  const int id = (int)dest[3 * c];
  double* G = ar.G + dest[3 * c + 1];
  double* AQc = ar.AQcc + dest[3 * c + 2];
  for (int e = t0; e < nout; e += nt) {
    const int code = etab[e];
    const int off = code & 2047, kind = (code >> 11) & 3, dst = code >> 13;
    double sum;
    if (three) {
      sum = sb[off] + sb[off + 256];  // (every tile in two K parts: the consumers' work split)
    } else {
      if (two && off >= 768)
        sum = sb[off] + sb[off + 4];
      else
        sum = (sb[off] + sb[off + 256]) + sb[off + 512];
      if (!two) sum += sb[off + 768];
    }
    sum *= osc;
    if (kind == 0)
      G[dst] = sum;
    else if (kind == 1)
      AQc[dst] = sum;
    else
      ar.sc[2 * id + 1] = sum;
  }
}

template <int N, bool H>
// single != 0: ONE P image instead of two (more matrices than fit LDS twice): the producers then wait
// at the top of an iteration until the consumers have contracted the image they are about to
// overwrite -- a second barrier per constraint, stages 1 and 2 no longer overlap.
// (Tried on top of it: one buffer of partial tiles as well, which halves a workgroup's LDS, and TWO
// workgroups per CU -- 26.9 us instead of 22.2 at C4: symmetric workgroups fill and drain together,
// so neither hides the other's waits, and the first loads double.)
__global__ void __launch_bounds__(MfmaCfg<N>::THREADS) lmi_schur_mfma(LmiGroup g, Arena ar, int single) {
  using Cfg = MfmaCfg<N, H>;
  constexpr int NK = Cfg::NK, LD = Cfg::LD, MS = Cfg::MS, TPW = Cfg::TPW, RPM = Cfg::RPM;
  extern __shared__ double lds[];
  const int M = g.m, M1 = M + 1, rows = M1 * RPM;
  const int nt1 = (rows + 15) >> 4;
  const int pbuf = M1 * MS;
  // contraction cover of the M1 x M1 lower triangle: one tile (M1 <= 16), one tile + two corner
  // triangles (17..24), three tiles (25..32)
  const bool two = M1 > 16 && M1 <= 24, three = Cfg::THREE_TILES && M1 > 24;
  double* P0 = lds;
  double* scratch = lds + (single ? 1 : 2) * (size_t)pbuf;  // 2 buffers x CONS partial tiles of 256 entries
  constexpr int SB = Cfg::PARTS * 256;
  // where this workgroup's constraints write (kDestSlots x {id, g_off, r_off}) and the epilogue
  // table, both filled by the consumers in iteration 0
  int64_t* dest = reinterpret_cast<int64_t*>(scratch + 2 * SB);
  int* etab = reinterpret_cast<int*>(dest + 3 * kDestSlots);
  const int nout = M1 * (M1 + 1) / 2;
  // Hermitian cones: tr over the real representation, divided by d; the folded form (H) sums half of it
  const double osc = (g.herm_d > 1 ? 1.0 / g.herm_d : 1.0) * (H ? 2.0 : 1.0);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int first = blockIdx.x, stride = gridDim.x;
  const int cnt = (g.count - first + stride - 1) / stride;  // constraints of this workgroup (>= 1)

  if (wave < Cfg::PROD) {
    // ------------------------------------------------------------------ producers
    double a[TPW][NK];
    WOps<N> w, wn;
    TileGeom<N> gm;
    MakeGeom<N, H>(gm, wave, lane, rows, nt1);
    MSTAMP(0);
    {
      const ConstraintPtrs c0 = Member(g, first, g.n * g.n);
      // Issue order matters: waits count loads in flight in issue order, and the loop below
      // reloads the slots in the order 0, 1, ...; the scheduling barriers keep the compiler from
      // permuting these independent loads (it emitted them last-slot-first, which made every
      // iteration wait for ALL of its operands before its first MFMA).
      LoadW<N>(wn, c0.Wg, lane, g.n);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tt = 0; tt < TPW; tt++) {
        LoadTile<N>(a[tt], gm, tt, c0.block);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    auto iteration = [&](int it, auto early_c) {
      // Every load below is unconditional (a tile slot past the last tile re-reads the last tile,
      // the last iteration re-reads its own constraint: cache hits that nothing waits for), so the
      // count of loads in flight is the same on every path and each wait can be exact.
      MSTAMP(1 + 4 * it);
      const int itn = it + 1 < cnt ? it + 1 : it;
      const ConstraintPtrs nx = Member(g, first + itn * stride, g.n * g.n);
      w = wn;
      // the next constraint's W operands: a whole iteration ahead where the registers allow it
      // (N <= 20), behind this iteration's tiles otherwise (the order-24 instances spill with both
      // sets live; W is 4.6 KB per constraint and shared by the eight waves: an L2 hit)
      if constexpr (N <= 20) LoadW<N>(wn, nx.Wg, lane, g.n);
      __builtin_amdgcn_sched_barrier(0);
      double* Pb = P0 + (single ? 0 : (it & 1) * pbuf);
      if (single && it > 0) LdsBarrier();  // the consumers are done with the image
      TileAcc<N> res[2];
      // When a producer asks for the next constraint's operands.  EARLY: right behind the MFMAs of the tile that
      // held the registers (they read their operands at issue) -- a tile's pipe time, 800 - 2000 cycles of this
      // wave's schedule, earlier than behind the NEXT tile's MFMAs, where the request sat until round 4.  The
      // steady state is bound by WHEN the requests go out, not by how many bytes are in flight: C4 step
      // 47.7 -> 46.7 us on one box (asking again k-step by k-step inside the tile: 48.1 -- more instructions
      // between the MFMAs, two spilled registers).  The FIRST iteration keeps the late placement: its own
      // operands are still on their way from every CU of the chip at once, and early requests for the second
      // constraint lengthened that fill from 11.7 k to 13.9 k cycles.  (The first iteration is peeled off the
      // loop, not a branch inside it: the compiler's load counting would merge the two placements at every tile
      // and wait for the younger loads.)
      constexpr bool EARLY = decltype(early_c)::value;
      {
#pragma unroll
        for (int tt = 0; tt <= TPW; tt++) {
          const bool cur_ok = tt < TPW && wave + Cfg::PROD * tt < nt1;
          const bool prev_ok = tt > 0 && wave + Cfg::PROD * (tt - 1) < nt1;
          if (it == 2) MSTAMP(48 + 2 * tt);
          if (cur_ok) {
            if (tt == 0)
              FullStep<N, false>(res[0], a[0], w, res[1], Pb, gm, 0);
            else
              FullStep<N, true>(res[tt & 1], a[tt < TPW ? tt : 0], w, res[(tt & 1) ^ 1], Pb, gm, tt);
          } else if (prev_ok) {
            // the wave's last tile (possibly the ragged last tile of the constraint: masked stores)
#pragma unroll
            for (int e = 0; e < Cfg::PIECES; e++) StorePiece<N, true>(res[(tt & 1) ^ 1], e, Pb, gm, tt - 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (it == 2) MSTAMP(49 + 2 * tt);
          if constexpr (EARLY) {
            if (tt < TPW) LoadTile<N>(a[tt], gm, tt, nx.block);
          } else {
            if (tt > 0) LoadTile<N>(a[tt - 1], gm, tt - 1, nx.block);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (N > 20) {
        LoadW<N>(wn, nx.Wg, lane, g.n);
        __builtin_amdgcn_sched_barrier(0);
      }
      MSTAMP(2 + 4 * it);
      LdsBarrier();
    };
    if (CXK_RELOAD_MODE == 1) {
      for (int it = 0; it < cnt; it++) iteration(it, std::true_type{});
    } else if (CXK_RELOAD_MODE == 0) {
      for (int it = 0; it < cnt; it++) iteration(it, std::false_type{});
    } else {
      iteration(0, std::false_type{});
      for (int it = 1; it < cnt; it++) iteration(it, std::true_type{});
    }
    MSTAMP(1 + 4 * cnt);
    // The drain: the consumers contract the last constraint.  Its predecessor's epilogue -- theirs in
    // every other iteration, in the slack behind their contraction -- is taken off their hands here (the
    // partial tiles of c_{cnt-2} were complete at the barrier just passed), and the last epilogue is
    // shared by all twelve waves.
    // (orders up to 20: the order-24 instances have no registers to spare for it -- they spilled)
    if constexpr (Cfg::DRAIN_HELP) {
      if (cnt >= 2) Epilogue<N, H>(scratch + ((cnt - 2) & 1) * SB, dest, etab, cnt - 2, ar, nout, two, three, osc, threadIdx.x, 64 * Cfg::PROD);
    }
    LdsBarrier();
    if constexpr (Cfg::DRAIN_HELP)
      Epilogue<N, H>(scratch + ((cnt - 1) & 1) * SB, dest, etab, cnt - 1, ar, nout, two, three, osc, threadIdx.x, Cfg::THREADS);
    return;
  }

  // -------------------------------------------------------------------- consumers
  // The consumers' instructions win the SIMD's issue arbitration (measured: priority 1 and 2 alike
  // shorten an iteration by ~6 % against equal priorities).
  __builtin_amdgcn_s_setprio(1);
  const int cw = wave - Cfg::PROD;
  const int ct = threadIdx.x - 64 * Cfg::PROD;  // 0 .. 255
  const int il = lane & 15, kq = lane >> 4;
  const int R = M1 - 16;
  // Work split.  17 <= M1 <= 24: waves 0-2 take a third of the K range each of the 16 x 16 tile
  // (rows = matrices R .., columns = matrices 0 .. 15), wave 3 the two R x R triangles that tile
  // leaves out.  M1 <= 16: the four waves take the K quarters of the single tile (rows past M1
  // alias M1 - 1).  25 <= M1 <= 32: three tiles -- T00 (matrices 0..15 squared) on wave 0, T10
  // (rows 16.., columns 0..15) in two K halves on waves 1 and 2, T11 (16.. squared) on wave 3.
  // (Round 4: T00 and T11 whole on one wave each and T10 in halves gave the four waves 4 : 2 : 2 : 4 of the
  // K range -- the contraction, not the producers, set the pace of config 5's folded Hermitian form.  Now every
  // tile goes in two K parts, cut so that each wave takes 3/4 of a tile's range: wave 0 the first 3/4 of T00;
  // wave 1 its last 1/4 and the first half of T10; wave 2 the second half of T10 and the first 1/4 of T11;
  // wave 3 its last 3/4.  Six partial tiles, each result the sum of two.)
  int rr0, nrows, ra, ca;
  int rr0b = 0, nrowsb = 0, rab = 0, cab = 0;  // three tiles: a wave's second piece (none on waves 0 and 3)
  if (three) {
    const int hi = (16 + il < M1 ? 16 + il : M1 - 1) * MS, lo = il * MS;
    constexpr int QA = Cfg::CUT_A, QB = RPM / 2;
    ra = cw <= 1 ? lo : hi;                 // first piece: T00 (waves 0, 1), T10 (wave 2), T11 (wave 3)
    ca = cw == 3 ? hi : lo;
    rr0 = cw == 0 ? 0 : (cw == 1 ? QA : (cw == 2 ? QB : RPM - QA));
    nrows = cw == 0 ? QA : (cw == 1 ? RPM - QA : (cw == 2 ? RPM - QB : QA));
    rab = hi;                               // second piece: T10 first half (wave 1), T11 first quarter (wave 2)
    cab = cw == 1 ? lo : hi;
    nrowsb = cw == 1 ? QB : (cw == 2 ? RPM - QA : 0);
  } else {
    const int parts = two ? Cfg::CONS - 1 : Cfg::CONS;
    const int part = cw < parts ? cw : 0;
    rr0 = part * RPM / parts;
    nrows = (part + 1) * RPM / parts - rr0;
    ra = two ? (R + il) * MS : (il < M1 ? il : M1 - 1) * MS;
    ca = two ? il * MS : ra;
  }
  // Iteration 0 has nothing to consume: look up where this workgroup's constraints write (two
  // dependent loads each) and park the answers in LDS, off every later critical path.
  if (ct < kDestSlots && ct < cnt) {
    const int id = g.ids[first + ct * stride];
    dest[3 * ct] = id;
    dest[3 * ct + 1] = ar.g_off[id];
    dest[3 * ct + 2] = ar.r_off[id];
  }
  // ... and tabulate the epilogue: entry e = (ii, jj <= ii) of the lower triangle of the M1 x M1
  // result -> where the first of its partial sums sits in a buffer of partial tiles (tile: the
  // others follow at multiples of 256; triangle: the other K parity 4 further) and where the total goes (kind 0: G[dst], 1: AQc[dst],
  // 2: <c,Qc>), packed  offset | kind << 10 | dst << 12.
  for (int idx = ct; idx < M1 * M1; idx += 64 * Cfg::CONS) {
    const int ii = idx / M1, jj = idx - ii * M1;
    if (jj > ii) continue;
    int off;
    if (two) {
      if (ii >= R && jj < 16) {
        off = (ii - R) * 16 + jj;  // the tile: waves 0 - 2
      } else {                     // a triangle: wave 3, instruction u, lane 16 i + 4 (2 t + h) + j
        const int t = ii < R ? 0 : 1, a = ii - 16 * t, b = jj - 16 * t;
        const int u = a < 4 ? 0 : (b < 4 ? 1 : 2);
        off = 768 + 64 * u + 16 * (a & 3) + 8 * t + (b & 3);
      }
    } else if (three) {
      off = ii < 16 ? ii * 16 + jj : (jj < 16 ? 512 + (ii - 16) * 16 + jj : 1024 + (ii - 16) * 16 + (jj - 16));
    } else {
      off = ii * 16 + jj;
    }
    const int kind = ii < M ? 0 : (jj < M ? 1 : 2);
    const int dst = ii < M ? ii + jj * M : (jj < M ? jj : 0);
    etab[ii * (ii + 1) / 2 + jj] = off | kind << 11 | dst << 13;
  }
  // AW(i) = tr(P_i), <w,c> = tr(P_C) from the image being contracted: one lane per matrix.
  auto traces = [&](int c) {
    if (ct >= M1) return;
    const double* Pc = P0 + (single ? 0 : (c & 1) * pbuf) + ct * MS;
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int r = 0; r < RPM; r += 2) {
      s0 += Pc[r * LD + r];
      s1 += Pc[(r + 1) * LD + r + 1];
    }
    const double sum = (s0 + s1) * osc;
    if (ct < M)
      ar.AWc[dest[3 * c + 2] + ct] = sum;
    else
      ar.sc[2 * dest[3 * c]] = sum;
  };
  auto epilogue = [&](int c) {
    Epilogue<N, H>(scratch + (c & 1) * SB, dest, etab, c, ar, nout, two, three, osc, ct, 64 * Cfg::CONS);
  };
  LdsBarrier();
  MSTAMP(0);
  constexpr int MAXROWS = (RPM + Cfg::CONS - 2) / (Cfg::CONS - 1);
  for (int it = 1; it <= cnt; it++) {
    MSTAMP(1 + 4 * it);
    const double* Pb = P0 + (single ? 0 : ((it - 1) & 1) * pbuf);
    double* sb = scratch + ((it - 1) & 1) * SB;
    if (!H && two && cw == Cfg::CONS - 1) {  // (the folded Hermitian form has no 17 .. 24 instance: SupportsT)
      if (R > 4)
        Triangles<N, true>(Pb, sb + 256 * (Cfg::CONS - 1), lane, M1);
      else
        Triangles<N, false>(Pb, sb + 256 * (Cfg::CONS - 1), lane, M1);
    } else {
      const d4_t acc = three ? Contract<N, Cfg::CUT_A, H>(Pb, ra, ca, rr0, kq, nrows) : Contract<N, MAXROWS, H>(Pb, ra, ca, rr0, kq, nrows);
      // (three tiles: partial tiles T00a T00b T10a T10b T11a T11b; wave 0 -> 0, wave 1 -> 1 and 2, wave 2 -> 3 and 4, wave 3 -> 5)
      const int slot = three ? (cw == 0 ? 0 : (cw == 1 ? 1 : (cw == 2 ? 3 : 5))) : cw;
#pragma unroll
      for (int e = 0; e < 4; e++)  // C/D layout: column = lane & 15, row = (lane >> 4) + 4 e
        sb[slot * 256 + (kq + 4 * e) * 16 + il] = acc[e];
      if (three && nrowsb > 0) {  // (wave-uniform)
        const d4_t accb = Contract<N, RPM / 2, H>(Pb, rab, cab, rr0b, kq, nrowsb);
#pragma unroll
        for (int e = 0; e < 4; e++) sb[(slot + 1) * 256 + (kq + 4 * e) * 16 + il] = accb[e];
      }
    }
    MSTAMP(2 + 4 * it);
    // The consumers finish their contraction well before the producers their tiles: the results
    // leave in that slack.
    traces(it - 1);
    if (single && it < cnt) LdsBarrier();  // the image is free for the next constraint
    if (it >= 2 && (it < cnt || !Cfg::DRAIN_HELP)) epilogue(it - 2);  // (it == cnt: the producers, idle by then, take it)
    MSTAMP(4 + 4 * it);
    LdsBarrier();
  }
  if constexpr (Cfg::DRAIN_HELP)  // (with the producers: all twelve waves)
    Epilogue<N, H>(scratch + ((cnt - 1) & 1) * SB, dest, etab, cnt - 1, ar, nout, two, three, osc, threadIdx.x, Cfg::THREADS);
  else
    epilogue(cnt - 1);
  MSTAMP(1 + 4 * (cnt + 1));
}

template <int N, bool H>
size_t MfmaLds(int m, int single = 0) {
  using Cfg = MfmaCfg<N, H>;
  const int m1 = m + 1;
  return sizeof(double) * ((single ? 1 : 2) * (size_t)m1 * Cfg::MS + 2 * (size_t)Cfg::PARTS * 256 + 3 * kDestSlots) +
         sizeof(int) * (size_t)(m1 * (m1 + 1) / 2);
}

constexpr size_t kLdsPerCu = 160 * 1024;

template <int N, bool H>
bool SupportsT(int m) {
  using Cfg = MfmaCfg<N, H>;
  const int m1 = m + 1;
  if (H && m1 > 16 && m1 <= 24) return false;  // the corner-triangle cover exists for the full form only
  // (two P images where they fit, one otherwise: LaunchT)
  return m >= 1 && m1 <= (Cfg::THREE_TILES ? 32 : 24) && (m1 * Cfg::RPM + 15) / 16 <= Cfg::TPW * Cfg::PROD &&
         MfmaLds<N, H>(m, 1) <= kLdsPerCu;
}

template <int N, bool H>
hipError_t LaunchT(const LmiGroup& g, const Arena& ar, int cus, hipStream_t stream, hipEvent_t ev_start,
                   hipEvent_t ev_stop) {
  static PerDeviceOnce once;
  const hipError_t ec = once.run([] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&lmi_schur_mfma<N, H>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsPerCu);
  });
  if (ec != hipSuccess) return ec;
  int grid = g.count < cus ? g.count : cus;
  const int need = (g.count + kDestSlots - 1) / kDestSlots;  // at most kDestSlots constraints per workgroup
  if (grid < need) grid = need;
  const int single = MfmaLds<N, H>(g.m) > kLdsPerCu ? 1 : 0;
  const size_t lds = MfmaLds<N, H>(g.m, single);
  if (ev_start && ev_stop)
    // the events ride on the dispatch itself (its own begin / end time stamps, what rocprofv3
    // reports): no marker packets around the kernel, no ~5.7 us bubble behind a bracketed launch
    hipExtLaunchKernelGGL((lmi_schur_mfma<N, H>), dim3(grid), dim3(MfmaCfg<N>::THREADS), (uint32_t)lds, stream, ev_start,
                          ev_stop, 0, g, ar, single);
  else
    lmi_schur_mfma<N, H><<<grid, MfmaCfg<N>::THREADS, lds, stream>>>(g, ar, single);
  return hipGetLastError();
}

}  // namespace

// Complex Hermitian cones (herm_d == 2) whose real representation has order n take the folded form
// when it has an instance; everything else the full form.
static bool Folded(int n, int m, int herm_d) { return herm_d == 2 && n == 24 && SupportsT<24, true>(m); }

// The instance an order runs on: itself when it has one, the next multiple of four (at least 8)
// otherwise -- the host then hands the kernel zero-padded copies of the A_i (LmiMfmaPaddedOrder).
int LmiMfmaPaddedOrder(int n) { return n <= 8 ? 8 : (n + 3) / 4 * 4; }

bool LmiMfmaSupports(int n, int m, int herm_d) {
  if (Folded(n, m, herm_d)) return true;
  if (n < 2 || n > 24) return false;
  switch (LmiMfmaPaddedOrder(n)) {
    case 8: return SupportsT<8, false>(m);
    case 12: return SupportsT<12, false>(m);
    case 16: return SupportsT<16, false>(m);
    case 20: return SupportsT<20, false>(m);
    case 24: return SupportsT<24, false>(m);
  }
  return false;
}

// g.A / g.a_stride: [A_1 .. A_m | C] per member at the PADDED order; g.n, g.W: the order itself.
hipError_t LaunchLmiSchurMfma(const LmiGroup& g, const Arena& ar, int cus, hipStream_t stream, hipEvent_t ev_start,
                              hipEvent_t ev_stop) {
  if (g.count <= 0) return hipSuccess;
  if (Folded(g.n, g.m, g.herm_d)) return LaunchT<24, true>(g, ar, cus, stream, ev_start, ev_stop);
  switch (LmiMfmaPaddedOrder(g.n)) {
    case 8: return LaunchT<8, false>(g, ar, cus, stream, ev_start, ev_stop);
    case 12: return LaunchT<12, false>(g, ar, cus, stream, ev_start, ev_stop);
    case 16: return LaunchT<16, false>(g, ar, cus, stream, ev_start, ev_stop);
    case 20: return LaunchT<20, false>(g, ar, cus, stream, ev_start, ev_stop);
    case 24: return LaunchT<24, false>(g, ar, cus, stream, ev_start, ev_stop);
  }
  return hipErrorNotSupported;
}

}  // namespace cxk

#ifdef CXK_MFMA_STAMPS
extern "C" __attribute__((visibility("default"))) int cxk_debug_mfma_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(cxk::g_mfma_stamp), sizeof(long long) * 2 * 16 * 64) == hipSuccess ? 0 : 1;
}
#endif
