// Scratch: is replaying a captured 3-kernel + 2-copy sequence cheaper than issuing it call by call? (B=1 oracle call shape)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void k(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
int main() {
  const int n = 15000, nj = 120000;
  double *d, *dj, *h, *hj;
  CHK(hipMalloc(&d, n * 8)); CHK(hipMalloc(&dj, nj * 8)); CHK(hipHostMalloc(&h, n * 8)); CHK(hipHostMalloc(&hj, nj * 8));
  hipStream_t s; CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto seq = [&]() {
    hipMemcpyAsync(d, h, n * 8, hipMemcpyHostToDevice, s);
    k<<<(n + 255) / 256, 256, 0, s>>>(d, n);
    k<<<(nj + 255) / 256, 256, 0, s>>>(dj, nj);
    k<<<1, 256, 0, s>>>(d, 256);
    hipMemcpyAsync(h, d, n * 8, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(hj, dj, nj * 8, hipMemcpyDeviceToHost, s);
  };
  for (int i = 0; i < 20; ++i) { seq(); CHK(hipStreamSynchronize(s)); }
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 500; ++i) { seq(); hipStreamSynchronize(s); }
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 500;
  printf("direct: %.1f us per call\n", us);
  hipGraph_t g; hipGraphExec_t ge;
  CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); seq(); CHK(hipStreamEndCapture(s, &g));
  CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 20; ++i) { CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s)); }
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 500; ++i) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
  us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 500;
  printf("graph : %.1f us per call\n", us);
  // kernels only
  auto seqk = [&]() { k<<<(n + 255) / 256, 256, 0, s>>>(d, n); k<<<(nj + 255) / 256, 256, 0, s>>>(dj, nj); k<<<1, 256, 0, s>>>(d, 256); };
  for (int i = 0; i < 20; ++i) { seqk(); hipStreamSynchronize(s); }
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 500; ++i) { seqk(); hipStreamSynchronize(s); }
  printf("direct kernels only: %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 500);
  hipGraph_t g2; hipGraphExec_t ge2;
  CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); seqk(); CHK(hipStreamEndCapture(s, &g2));
  CHK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  for (int i = 0; i < 20; ++i) { hipGraphLaunch(ge2, s); hipStreamSynchronize(s); }
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 500; ++i) { hipGraphLaunch(ge2, s); hipStreamSynchronize(s); }
  printf("graph kernels only : %.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 500);
  return 0;
}
