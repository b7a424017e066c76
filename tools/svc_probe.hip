// Scratch: round-trip latency of a RESIDENT kernel fed through page-locked memory (doorbell + completion flag) against launching
// a kernel per request, for the shape of one nlp_g call at config 2 (21 workgroups, 120 KB read from host memory, 120 KB written
// to host memory).  Every wait in the kernel is bounded (idle timeout, lifetime limit): it cannot hang the GPU.
// hipcc --offload-arch=gfx950 -O3 -o svc_probe svc_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Cmd { unsigned long long seq; const double* in; double* out; int n; int quit; };

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ __launch_bounds__(256) void service(const Cmd* cmd, unsigned long long* done, unsigned int* tickets, unsigned long long* alive, long long idle_ticks, long long life_ticks) {
  __shared__ unsigned long long s_seq;
  __shared__ int s_quit;
  unsigned long long have = 0;
  const long long t_start = wall_clock64();
  for (;;) {
    if (threadIdx.x == 0) {
      long long t_idle = wall_clock64();
      s_quit = 0;
      for (;;) {
        const unsigned long long s = ld_sys(&cmd->seq);
        if (s != have) { s_seq = s; s_quit = cmd->quit; break; }
        const long long now = wall_clock64();
        if (now - t_idle > idle_ticks || now - t_start > life_ticks) { s_quit = 1; s_seq = have; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    if (s_quit) break;
    have = s_seq;
    const double* in = cmd->in; double* out = cmd->out; const int n = cmd->n;
    const int per = (n + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(n, i0 + per);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) out[i] = in[i] * 2.0 + (double)have;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence_system();
      if (atomicAdd(tickets, 1u) == gridDim.x - 1) {
        *tickets = 0;
        __threadfence_system();
        __hip_atomic_store(done, have, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(alive, 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void oneshot(const double* in, double* out, int n, unsigned long long seq, unsigned long long* done, unsigned int* tickets) {
  const int per = (n + gridDim.x - 1) / gridDim.x, i0 = blockIdx.x * per, i1 = min(n, i0 + per);
  for (int i = i0 + threadIdx.x; i < i1; i += 256) out[i] = in[i] * 2.0 + (double)seq;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    if (atomicAdd(tickets, 1u) == gridDim.x - 1) { *tickets = 0; __threadfence_system(); __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int n = 15000, WG = 21;
  Cmd* cmd; unsigned long long *done, *alive; double *in, *out; unsigned int* tickets;
  CK(hipHostMalloc((void**)&cmd, 256, hipHostMallocMapped)); CK(hipHostMalloc((void**)&done, 64, hipHostMallocMapped)); CK(hipHostMalloc((void**)&alive, 64, hipHostMallocMapped));
  CK(hipHostMalloc((void**)&in, n * 8, hipHostMallocMapped)); CK(hipHostMalloc((void**)&out, n * 8, hipHostMallocMapped)); CK(hipMalloc((void**)&tickets, 4)); CK(hipMemset(tickets, 0, 4));
  for (int i = 0; i < n; ++i) in[i] = i;
  *done = 0; cmd->seq = 0; cmd->in = in; cmd->out = out; cmd->n = n; cmd->quit = 0; *alive = 1;
  unsigned long long seq = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  // per-request launches
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = now(); int N = 2000; int bad = 0;
    for (int i = 0; i < N; ++i) {
      ++seq; in[7] = (double)seq;
      oneshot<<<WG, 256, 0, s>>>(in, out, n, seq, done, tickets);
      while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != seq) {}
      if (out[7] != 2.0 * (double)seq + (double)seq) ++bad;
    }
    printf("launch per request: %.2f us per round trip (bad %d)\n", std::chrono::duration<double, std::micro>(now() - t0).count() / N, bad);
  }
  (void)hipStreamSynchronize(s);
  // resident kernel: 50 ms idle timeout, 1.5 s lifetime (wall_clock64 ticks at 100 MHz)
  service<<<WG, 256, 0, s>>>(cmd, done, tickets, alive, 5000000LL, 150000000LL);
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = now(); int N = 2000; int bad = 0;
    for (int i = 0; i < N; ++i) {
      ++seq; in[7] = (double)seq;
      __atomic_store_n(&cmd->seq, seq, __ATOMIC_RELEASE);
      while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != seq) {}
      if (out[7] != 2.0 * (double)seq + (double)seq) ++bad;
    }
    printf("resident kernel:    %.2f us per round trip (bad %d)\n", std::chrono::duration<double, std::micro>(now() - t0).count() / N, bad);
  }
  cmd->quit = 1; __atomic_store_n(&cmd->seq, ++seq, __ATOMIC_RELEASE);
  auto t0 = now();
  CK(hipStreamSynchronize(s));
  printf("quit -> stream idle: %.1f us, alive flag %llu\n", std::chrono::duration<double, std::micro>(now() - t0).count(), *alive);
  // idle timeout check: start again and just wait
  *alive = 1; cmd->quit = 0;
  service<<<WG, 256, 0, s>>>(cmd, done, tickets, alive, 5000000LL, 150000000LL);
  t0 = now(); CK(hipStreamSynchronize(s));
  printf("idle timeout exit after %.1f ms, alive flag %llu\n", std::chrono::duration<double, std::milli>(now() - t0).count(), *alive);
  return 0;
}
