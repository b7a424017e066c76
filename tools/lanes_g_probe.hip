// lanes_g_probe.hip -- feasibility probe (round 5): config-3 nlp_g with one LANE per evaluation point.
// Van der Pol 2000 x [3,30,3] shapes (N = 24010 nodes, z = [X0 | X1 | U | t0 tf], g = [F0 | F1 | mU]), B = 512 = 8 blocks of 64 points.
// One wavefront = (block of 64 points, one item): a degree-30 segment (31 columns of each array -> LDS tile -> registers; D.X and
// C_mid.U with the table entries as SCALAR operands; 30 rows of each output array -> tile -> stores) or a pair of degree-3 segments.
// The question it answers before anything is built into libmpx: can this data movement + arithmetic beat the matrix-core kernel's
// 138 us (profiles/r5_c3_light)?  hipcc --offload-arch=gfx950 -O3 -o lanes_g_probe tools/lanes_g_probe.hip && ./lanes_g_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#ifndef NO_CONTRACTION
#define NO_CONTRACTION 0  // 1: one term per row instead of 31 -- the data movement and the node functions alone
#endif
constexpr int LDW = 65, P = 30, P1 = 31, PF = 3;
#define SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

struct Args {
  const double* z; double* g; int64_t zs, gs; int N, B, n_blocks, n_hi, n_lo;
  const int* hi_start;   // point 0 of the segment (node index)
  const int* lo_start;   // first node of a pair of low segments (point 0 of the first)
  const double *D, *C, *D3, *C3;  // [31][31], [30][31], [4][4], [3][4]
};

// piece of 2^K columns starting at column c0 (runtime, uniform) of the row-array at `base` (uniform, block's first point): -> v[i]
template <int K>
__device__ __forceinline__ void ld_piece(const double* base, int64_t stride, int c0, int lane, double* v) {
  constexpr int PPI = 64 >> K;
  const unsigned lo = 8u * (unsigned)((lane >> K) * stride + (lane & ((1 << K) - 1)));
#pragma unroll
  for (int i = 0; i < (1 << K); ++i) {
    const char* b = (const char*)(base + (int64_t)i * PPI * stride + c0);  // uniform
    v[i] = *(const double*)(b + lo);
  }
}
template <int K>
__device__ __forceinline__ void put_piece(double* T, int e0, int lane, const double* v) {
  constexpr int PPI = 64 >> K;
  const int lo = (lane & ((1 << K) - 1)) * LDW + (lane >> K);
#pragma unroll
  for (int i = 0; i < (1 << K); ++i) T[lo + e0 * LDW + i * PPI] = v[i];
}
template <int K>
__device__ __forceinline__ void st_piece(double* base, int64_t stride, int c0, const double* T, int e0, int lane) {
  constexpr int PPI = 64 >> K;
  const unsigned lo = 8u * (unsigned)((lane >> K) * stride + (lane & ((1 << K) - 1)));
  const int lt = (lane & ((1 << K) - 1)) * LDW + (lane >> K);
  double w[1 << K];
#pragma unroll
  for (int i = 0; i < (1 << K); ++i) w[i] = T[lt + e0 * LDW + i * PPI];
#pragma unroll
  for (int i = 0; i < (1 << K); ++i) {
    char* b = (char*)(base + (int64_t)i * PPI * stride + c0);
    *(double*)(b + lo) = w[i];
  }
}
// 31 columns in, pieces 16 + 8 + 4 + 2 + 1
__device__ __forceinline__ void load31(const double* base, int64_t zs, int c0, int lane, double* T, double* x) {
  double v[31];
  ld_piece<4>(base, zs, c0, lane, v); ld_piece<3>(base, zs, c0 + 16, lane, v + 16); ld_piece<2>(base, zs, c0 + 24, lane, v + 24);
  ld_piece<1>(base, zs, c0 + 28, lane, v + 28); ld_piece<0>(base, zs, c0 + 30, lane, v + 30);
  SYNC();
  put_piece<4>(T, 0, lane, v); put_piece<3>(T, 16, lane, v + 16); put_piece<2>(T, 24, lane, v + 24); put_piece<1>(T, 28, lane, v + 28); put_piece<0>(T, 30, lane, v + 30);
  SYNC();
#pragma unroll
  for (int j = 0; j < 31; ++j) x[j] = T[j * LDW + lane];
}
// 30 rows out, pieces 16 + 8 + 4 + 2
__device__ __forceinline__ void store30(double* base, int64_t gs, int c0, int lane, double* T, const double* r) {
  SYNC();
#pragma unroll
  for (int j = 0; j < 30; ++j) T[j * LDW + lane] = r[j];
  SYNC();
  st_piece<4>(base, gs, c0, T, 0, lane); st_piece<3>(base, gs, c0 + 16, T, 16, lane); st_piece<2>(base, gs, c0 + 24, T, 24, lane); st_piece<1>(base, gs, c0 + 28, T, 28, lane);
}

extern "C" __global__ __launch_bounds__(64) void probe(const Args A) {
  __shared__ double T[32 * LDW];
  const int lane = threadIdx.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, n_items = A.n_hi + A.n_lo;
  const int it = idx % n_items, blk = (idx / n_items) * 8 + xcd;
  if (blk >= A.n_blocks) return;
  const int64_t b0 = (int64_t)blk * 64;
  const double* zb = A.z + b0 * A.zs;
  double* gb = A.g + b0 * A.gs;
  const int N = A.N;
  const double* zt = A.z + (b0 + lane) * A.zs + 3 * (int64_t)N;
  const double t0 = zt[0], tf = zt[1];
  const double kap = (tf - t0) * (1.0 / 2000.0) * 0.5;
  if (it < A.n_hi) {
    const int st = A.hi_start[it];
    double x0[31], x1[31], u[31];
    load31(zb, A.zs, st, lane, T, x0);
    load31(zb + N, A.zs, st, lane, T, x1);
    load31(zb + 2 * (int64_t)N, A.zs, st, lane, T, u);
    double r[30];
    // F0 = D.X0 - kap f0(x), f0 = (1 - x1^2) x0 - x1 + u
#pragma unroll
    for (int k = 1; k <= P; ++k) {
      double acc = 0;
#pragma unroll
      for (int j = 0; j < (NO_CONTRACTION ? 1 : P1); ++j) acc = __builtin_fma(A.D[k * P1 + j], x0[j], acc);
      r[k - 1] = acc - kap * ((1.0 - x1[k] * x1[k]) * x0[k] - x1[k] + u[k]);
    }
    store30(gb, A.gs, st + 1, lane, T, r);
#pragma unroll
    for (int k = 1; k <= P; ++k) {
      double acc = 0;
#pragma unroll
      for (int j = 0; j < (NO_CONTRACTION ? 1 : P1); ++j) acc = __builtin_fma(A.D[k * P1 + j], x1[j], acc);
      r[k - 1] = acc - kap * x0[k];
    }
    store30(gb + N, A.gs, st + 1, lane, T, r);
#pragma unroll
    for (int k = 1; k <= P; ++k) {
      double acc = 0;
#pragma unroll
      for (int j = 0; j < (NO_CONTRACTION ? 1 : P1); ++j) acc = __builtin_fma(A.C[(k - 1) * P1 + j], u[j], acc);
      r[k - 1] = acc;
    }
    store30(gb + 2 * (int64_t)N, A.gs, st, lane, T, r);
  } else {
    // two degree-3 segments: 7 columns of each array (pieces 4 + 2 + 1), 6 rows of each output (pieces 4 + 2)
    const int st = A.lo_start[it - A.n_hi];
    double xs[3][7];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double v[7];
      const double* base = zb + a * (int64_t)N;
      ld_piece<2>(base, A.zs, st, lane, v); ld_piece<1>(base, A.zs, st + 4, lane, v + 4); ld_piece<0>(base, A.zs, st + 6, lane, v + 6);
      SYNC();
      put_piece<2>(T, 0, lane, v); put_piece<1>(T, 4, lane, v + 4); put_piece<0>(T, 6, lane, v + 6);
      SYNC();
#pragma unroll
      for (int j = 0; j < 7; ++j) xs[a][j] = T[j * LDW + lane];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double r[6];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 1; k <= PF; ++k) {
          double acc = 0;
#pragma unroll
          for (int j = 0; j <= PF; ++j) acc = __builtin_fma(a < 2 ? A.D3[k * 4 + j] : A.C3[(k - 1) * 4 + j], xs[a][3 * s + j], acc);
          const int n = 3 * s + k;
          r[3 * s + k - 1] = a == 0 ? acc - kap * ((1.0 - xs[1][n] * xs[1][n]) * xs[0][n] - xs[1][n] + xs[2][n]) : (a == 1 ? acc - kap * xs[0][n] : acc);
        }
      SYNC();
#pragma unroll
      for (int j = 0; j < 6; ++j) T[j * LDW + lane] = r[j];
      SYNC();
      double* base = gb + a * (int64_t)N;
      const int c0 = a < 2 ? st + 1 : st;
      st_piece<2>(base, A.gs, c0, T, 0, lane); st_piece<1>(base, A.gs, c0 + 4, T, 4, lane);
    }
  }
}

int main() {
  const int S = 2000, B = 512;
  std::vector<int> start(S + 1, 0), hi, lo;
  for (int s = 0; s < S; ++s) start[s + 1] = start[s] + (s % 3 == 1 ? 30 : 3);
  const int N = start[S] + 1;
  for (int s = 0; s < S; ++s)
    if (s % 3 == 1) hi.push_back(start[s]);
  for (int s = 2; s + 1 < S; s += 3) lo.push_back(start[s]);  // pairs (s, s + 1) of low segments: s % 3 == 2, 0
  const int64_t zs = 3 * (int64_t)N + 2, gs = 3 * (int64_t)N - 1;
  printf("N %d, n_z %lld, n_g %lld, high items %zu, low items %zu, B %d\n", N, (long long)zs, (long long)gs, hi.size(), lo.size(), B);
  double *z, *g, *D, *C, *D3, *C3; int *dhi, *dlo;
  CK(hipMalloc(&z, zs * B * 8)); CK(hipMalloc(&g, gs * B * 8));
  CK(hipMalloc(&D, 31 * 31 * 8)); CK(hipMalloc(&C, 30 * 31 * 8)); CK(hipMalloc(&D3, 128)); CK(hipMalloc(&C3, 96));
  CK(hipMalloc(&dhi, hi.size() * 4)); CK(hipMalloc(&dlo, lo.size() * 4));
  std::vector<double> hz(zs * B);
  for (auto& v : hz) v = rand() / (double)RAND_MAX;
  CK(hipMemcpy(z, hz.data(), zs * B * 8, hipMemcpyHostToDevice));
  std::vector<double> tab(31 * 31);
  for (auto& v : tab) v = rand() / (double)RAND_MAX - 0.5;
  CK(hipMemcpy(D, tab.data(), 31 * 31 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(C, tab.data(), 30 * 31 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(D3, tab.data(), 128, hipMemcpyHostToDevice)); CK(hipMemcpy(C3, tab.data(), 96, hipMemcpyHostToDevice));
  CK(hipMemcpy(dhi, hi.data(), hi.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dlo, lo.data(), lo.size() * 4, hipMemcpyHostToDevice));
  Args A{z, g, zs, gs, N, B, B / 64, (int)hi.size(), (int)lo.size(), dhi, dlo, D, C, D3, C3};
  const unsigned grid = 8u * (unsigned)(hi.size() + lo.size()) * ((B / 64 + 7) / 8);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, A);
    CK(hipEventRecord(e0));
    for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, A);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 50, bytes = 8.0 * B * (zs + gs);
    printf("probe: %.1f us per pass  (%.2f TB/s of z + g = %.0f MB; the matrix-core kernel: 138 us)\n", us, bytes / us / 1e6, bytes / 1e6);
  }
  hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)probe));
  printf("registers %d, LDS %zu B, scratch %zu B\n", fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
  return 0;
}
