// exp_barrier.hip -- what a device-wide barrier costs on this part, against the cost of a dependent kernel launch.
//
// Result (profiles/r02ad_barrier_vs_launch.txt): launch 2.6 us, barrier 10.8 us, barrier + fences + 20 KB read 71 us.
// It decided whether the two launch-bound loops of the eigensolvers -- one launch per Householder column
// (csrc/eig.hip: 2,504 launches x ~11.5 us, of which ~5-7 us are fixed), five launches per Lanczos step
// (csrc/eig_lanczos.hip) -- can be replaced by persistent cooperative kernels (they cannot): a barrier would have to cost clearly less
// than the ~4.5 us floor of a dependent launch, including making the data written before it visible to the other
// XCDs (8 L2s that are not coherent with each other: release = write back, acquire = invalidate).
//
//   (a) K dependent launches of an almost empty kernel, 256 x 512 threads          -> us per launch
//   (b) one cooperative launch, K barriers (atomic counter at agent scope + spin)   -> us per barrier
//   (c) as (b), every workgroup also publishes 10 doubles before the barrier and reads all 2,560 after it
//       (the q vector of a Householder step), with __threadfence() on both sides    -> us per step, and a checksum
//
// build: make -C tools exp_barrier      run: tools/exp_barrier [steps = 2000]
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(expr)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      std::exit(1);                                                                               \
    }                                                                                             \
  } while (0)

constexpr int kThreads = 512;

__global__ __launch_bounds__(kThreads) void tiny_kernel(double* __restrict__ x, int step) {
  if (threadIdx.x == 0) x[blockIdx.x] += (double)step;   // something that depends on the previous launch
}

// one barrier: every workgroup adds 1 to a monotonically growing counter and waits until all of this round's arrivals
// are in.  Thread 0 does the atomics, the workgroup waits for it at __syncthreads().
__device__ __forceinline__ void grid_barrier(unsigned int* ctr, unsigned int target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    // bounded: a barrier that cannot complete (a workgroup that never became resident) must not hang the GPU -- after
    // ~50 ms of spinning the workgroup gives up and the counter's overflow word records it (ctr[1])
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > 5000000LL) { atomicAdd(ctr + 1, 1u); break; }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(kThreads) void barrier_only_kernel(unsigned int* ctr, int steps, long long* ticks) {
  const long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) grid_barrier(ctr, (unsigned)(s + 1) * gridDim.x);
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = wall_clock64() - t0;
}

// (c): a Householder-like exchange.  q has two halves (step parity) so that a fast workgroup writing step s + 1 never
// overwrites what a slow one still reads of step s.
__global__ __launch_bounds__(kThreads) void exchange_kernel(unsigned int* ctr, int steps, double* __restrict__ q, int n,
                                                            long long* ticks, double* checksum) {
  __shared__ double red[kThreads / 64];
  const int per = (n + gridDim.x - 1) / gridDim.x;
  double acc = 0.0, mine = 1.0 + blockIdx.x;
  const long long t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
    double* qs = q + (size_t)(s & 1) * n;
    if ((int)threadIdx.x < per && (int)(blockIdx.x * per + threadIdx.x) < n) qs[blockIdx.x * per + threadIdx.x] = mine + s;
    __threadfence();                                   // release: the stores leave this XCD's L2
    grid_barrier(ctr, (unsigned)(s + 1) * gridDim.x);
    __threadfence();                                   // acquire: stale lines of the other XCDs' data are dropped
    double part = 0.0;
    for (int i = threadIdx.x; i < n; i += kThreads) part += qs[i];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) tot += red[w];
    acc += tot;
    mine = 1.0 + blockIdx.x + 1e-9 * tot;              // the next step depends on what was read
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (blockIdx.x == 0) ticks[0] = wall_clock64() - t0;
    checksum[blockIdx.x] = acc;
  }
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? std::atoi(argv[1]) : 2000;
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount;            // one workgroup per CU
  std::printf("device %s, %d CUs, cooperative launch %s\n", prop.gcnArchName, grid, prop.cooperativeLaunch ? "yes" : "NO");
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double* x = nullptr; unsigned int* ctr = nullptr; long long* ticks = nullptr; double* q = nullptr; double* sum = nullptr;
  const int n = 2560;
  CK(hipMalloc((void**)&x, sizeof(double) * grid));
  CK(hipMalloc((void**)&ctr, 256));
  CK(hipMalloc((void**)&ticks, 64));
  CK(hipMalloc((void**)&q, sizeof(double) * 2 * n));
  CK(hipMalloc((void**)&sum, sizeof(double) * grid));
  CK(hipMemset(x, 0, sizeof(double) * grid));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // (a)
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(grid), dim3(kThreads), 0, s, x, i);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("(a) %d dependent launches of %d x %d threads: %.3f us per launch\n", steps, grid, kThreads, 1e3 * ms / steps);
  }
  // (b), (c): cooperative launches (co-residency of the whole grid is what makes the spin safe)
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemsetAsync(ctr, 0, 256, s));
      CK(hipMemsetAsync(q, 0, sizeof(double) * 2 * n, s));
      int st = steps, nn = n;
      void* args_b[] = {&ctr, &st, &ticks};
      void* args_c[] = {&ctr, &st, &q, &nn, &ticks, &sum};
      CK(hipEventRecord(e0, s));
      if (mode == 0)
        CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(barrier_only_kernel), dim3(grid), dim3(kThreads), args_b, 0, s));
      else
        CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(exchange_kernel), dim3(grid), dim3(kThreads), args_c, 0, s));
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      long long t = 0;
      CK(hipMemcpy(&t, ticks, sizeof(t), hipMemcpyDeviceToHost));
      unsigned int gave_up = 0;
      CK(hipMemcpy(&gave_up, ctr + 1, sizeof(gave_up), hipMemcpyDeviceToHost));
      if (gave_up) std::printf("    !! %u barrier waits timed out: the grid was not co-resident, the figures below mean nothing\n", gave_up);
      if (mode == 0) {
        std::printf("(b) %d barriers over %d workgroups: %.3f us per barrier (events), %.3f us (s_memrealtime in workgroup 0)\n",
                    steps, grid, 1e3 * ms / steps, 1e-2 * (double)t / steps);
      } else {
        std::vector<double> h(grid);
        CK(hipMemcpy(h.data(), sum, sizeof(double) * grid, hipMemcpyDeviceToHost));
        bool same = true;
        for (int b = 1; b < grid; ++b) same = same && (h[b] == h[0]);
        std::printf("(c) %d steps of publish 10 doubles / barrier / read %d doubles: %.3f us per step (events), %.3f us "
                    "(workgroup 0); every workgroup read the same values: %s (checksum %.6e)\n",
                    steps, n, 1e3 * ms / steps, 1e-2 * (double)t / steps, same ? "yes" : "NO -- VISIBILITY BUG", h[0]);
      }
    }
  std::printf("done\n");
  return 0;
}
