// Scratch: primitive costs for the single-evaluation (IPOPT) regime on this box: launch+sync floor, pinned H2D / D2H copies,
// and zero-copy (kernel stores straight into page-locked host memory, kernel loads straight from it).
// hipcc --offload-arch=gfx950 -O3 -o zc_probe zc_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void k_empty() {}
__global__ void k_write(d2* out, size_t n, double v) {  // n = number of 16-B pieces
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = d2{v + i, v};
}
__global__ void k_read_write(const double* in, size_t nin, d2* out, size_t n) {
  double s = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nin; i += (size_t)gridDim.x * blockDim.x) s += in[i];
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = d2{s, s + i};
}
template <class F> double timeit(F f, int n = 300) {
  for (int i = 0; i < 20; ++i) f();
  auto t = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f();
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count() / n;
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const size_t sizes[] = {1500, 15000, 120000, 1300000};  // bytes: config-1 outputs, z of config 2, g of config 2, all outputs of config 2
  printf("launch + sync (empty kernel): %.1f us\n", timeit([&] { k_empty<<<1, 64, 0, s>>>(); (void)hipStreamSynchronize(s); }));
  printf("2 launches + sync: %.1f us\n", timeit([&] { k_empty<<<1, 64, 0, s>>>(); k_empty<<<1, 64, 0, s>>>(); (void)hipStreamSynchronize(s); }));
  for (size_t bytes : sizes) {
    void *h, *hm, *d; CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); CK(hipMalloc(&d, bytes));
    CK(hipHostMalloc(&hm, bytes, hipHostMallocMapped)); void* hmd; CK(hipHostGetDevicePointer(&hmd, hm, 0));
    std::vector<char> pg(bytes); void* pgd = nullptr;
    const size_t n16 = bytes / 16; const int blocks = (int)std::min<size_t>((n16 + 255) / 256, 1024);
    printf("--- %zu bytes (%d blocks)\n", bytes, blocks);
    printf("  H2D pinned copy + sync            %.1f us\n", timeit([&] { (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); }));
    printf("  D2H pinned copy + sync            %.1f us\n", timeit([&] { (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }));
    printf("  kernel(write device) + D2H + sync %.1f us\n", timeit([&] { k_write<<<blocks, 256, 0, s>>>((d2*)d, n16, 1.0); (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }));
    printf("  kernel(write device) + sync       %.1f us\n", timeit([&] { k_write<<<blocks, 256, 0, s>>>((d2*)d, n16, 1.0); (void)hipStreamSynchronize(s); }));
    printf("  kernel(write HOST zero-copy)+sync %.1f us\n", timeit([&] { k_write<<<blocks, 256, 0, s>>>((d2*)hmd, n16, 1.0); (void)hipStreamSynchronize(s); }));
    printf("  kernel(read 120KB HOST, write HOST)+sync %.1f us\n", timeit([&] { k_read_write<<<blocks, 256, 0, s>>>((const double*)hmd, std::min<size_t>(bytes / 8, 15000), (d2*)hmd, n16); (void)hipStreamSynchronize(s); }));
    printf("  kernel(read 120KB DEV, write HOST)+sync  %.1f us\n", timeit([&] { k_read_write<<<blocks, 256, 0, s>>>((const double*)d, std::min<size_t>(bytes / 8, 15000), (d2*)hmd, n16); (void)hipStreamSynchronize(s); }));
    // registered caller memory (what mpx_host_register does)
    CK(hipHostRegister(pg.data(), bytes, hipHostRegisterMapped)); CK(hipHostGetDevicePointer(&pgd, pg.data(), 0));
    printf("  kernel(write REGISTERED host)+sync %.1f us\n", timeit([&] { k_write<<<blocks, 256, 0, s>>>((d2*)pgd, n16, 2.0); (void)hipStreamSynchronize(s); }));
    double chk = ((double*)pg.data())[1];
    printf("  (visible on host after sync: %s)\n", chk == 2.0 ? "yes" : "NO");
    CK(hipHostUnregister(pg.data()));
    (void)hipHostFree(h); (void)hipHostFree(hm); (void)hipFree(d);
  }
  return 0;
}
