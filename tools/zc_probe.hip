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
__global__ void k_flag(volatile unsigned long long* flag, unsigned long long seq) {
  __threadfence_system();
  __hip_atomic_store((unsigned long long*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_write_flag(d2* out, size_t n, double v, unsigned long long* flag, unsigned long long seq, unsigned int* cnt) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = d2{v + i, v};
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    if (atomicAdd(cnt, 1u) == gridDim.x - 1) { *cnt = 0; __threadfence_system(); __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}
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
  {
    unsigned long long* flag; CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped)); *flag = 0;
    unsigned long long seq = 0;
    printf("launch(flag kernel) + host spin on mapped flag: %.1f us\n", timeit([&] { ++seq; k_flag<<<1, 64, 0, s>>>(flag, seq); while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {} }));
    printf("2 launches (empty + flag kernel) + host spin: %.1f us\n", timeit([&] { ++seq; k_empty<<<1, 64, 0, s>>>(); k_flag<<<1, 64, 0, s>>>(flag, seq); while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {} }));
    unsigned int* cnt; CK(hipMalloc((void**)&cnt, 4)); CK(hipMemset(cnt, 0, 4));
    void* hm; CK(hipHostMalloc(&hm, 1300000, hipHostMallocMapped));
    for (size_t bytes : {15000ul, 120000ul, 1300000ul}) {
      const size_t n16 = bytes / 16; const int blocks = (int)std::min<size_t>((n16 + 255) / 256, 1024);
      printf("kernel(write %zu B HOST + last-block flag) + host spin: %.1f us   (same with hipStreamSynchronize: %.1f us)\n", bytes,
             timeit([&] { ++seq; k_write_flag<<<blocks, 256, 0, s>>>((d2*)hm, n16, 1.0, flag, seq, cnt); while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {} }),
             timeit([&] { ++seq; k_write_flag<<<blocks, 256, 0, s>>>((d2*)hm, n16, 1.0, flag, seq, cnt); (void)hipStreamSynchronize(s); }));
      // check data visibility at flag time
      int bad = 0;
      for (int rep = 0; rep < 200; ++rep) {
        ++seq; double v = (double)rep;
        k_write_flag<<<blocks, 256, 0, s>>>((d2*)hm, n16, v, flag, seq, cnt);
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
        const double* h = (const double*)hm;
        for (size_t i = 0; i < n16; i += 97) if (h[2 * i + 1] != v) { ++bad; break; }
      }
      printf("   stale reads after the flag: %d of 200\n", bad);
      (void)hipStreamSynchronize(s);
    }
  }
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
