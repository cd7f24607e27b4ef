// exp_vmm2.hip -- is memory mapped through the HIP virtual-memory API coherent the way hipMalloc memory is?
// (the guard-page sweeps of tests/guard_sweep.py came back with wrong counts, not faults).  Each check runs on a VMM mapping
// and on a hipMalloc buffer: (1) kernel A writes, kernel B (other block -> XCD mapping) reads; (2) hipMemsetAsync then a
// kernel reads; (3) hipMemcpy H2D then a kernel reads; (4) a kernel writes, D2H reads; (5) atomics from all XCDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { std::printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void wr(uint32_t* p, int64_t n, uint32_t salt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i * 2654435761u ^ salt;
}
__global__ void rd(const uint32_t* p, int64_t n, uint32_t salt, unsigned long long* bad) {
  unsigned long long b = 0;
  // reversed block order: block b reads what block gridDim - 1 - b wrote
  for (int64_t i = (int64_t)(gridDim.x - 1 - blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    b += p[i] != ((uint32_t)i * 2654435761u ^ salt);
  if (b) atomicAdd(bad, b);
}
__global__ void rdconst(const uint32_t* p, int64_t n, uint32_t want, unsigned long long* bad) {
  unsigned long long b = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b += p[i] != want;
  if (b) atomicAdd(bad, b);
}
__global__ void atom(uint32_t* p) { atomicAdd(p + (threadIdx.x & 3), 1u); }
static int run(const char* name, uint32_t* buf, int64_t n, hipStream_t s, unsigned long long* bad) {
  unsigned long long h = 0, tot = 0;
  for (int it = 0; it < 50; ++it) {
    CK(hipMemsetAsync(bad, 0, 8, s));
    hipLaunchKernelGGL(wr, dim3(1024), dim3(256), 0, s, buf, n, (uint32_t)it);
    hipLaunchKernelGGL(rd, dim3(1024), dim3(256), 0, s, buf, n, (uint32_t)it, bad);
    CK(hipMemcpyAsync(&h, bad, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); tot += h;
  }
  std::printf("%-10s (1) kernel writes -> kernel reads, 50 rounds: %llu wrong words\n", name, tot);
  tot = 0;
  for (int it = 0; it < 50; ++it) {
    CK(hipMemsetAsync(bad, 0, 8, s));
    CK(hipMemsetAsync(buf, it & 0xff, (size_t)n * 4, s));
    const uint32_t w = (uint32_t)(it & 0xff) * 0x01010101u;
    hipLaunchKernelGGL(rdconst, dim3(1024), dim3(256), 0, s, buf, n, w, bad);
    CK(hipMemcpyAsync(&h, bad, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); tot += h;
  }
  std::printf("%-10s (2) hipMemsetAsync -> kernel reads, 50 rounds: %llu wrong words\n", name, tot);
  std::vector<uint32_t> host((size_t)n);
  tot = 0;
  for (int it = 0; it < 20; ++it) {
    for (int64_t i = 0; i < n; ++i) host[(size_t)i] = (uint32_t)i * 2654435761u ^ (uint32_t)(it + 77);
    CK(hipMemcpy(buf, host.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    CK(hipMemsetAsync(bad, 0, 8, s));
    hipLaunchKernelGGL(rd, dim3(1024), dim3(256), 0, s, buf, n, (uint32_t)(it + 77), bad);
    CK(hipMemcpyAsync(&h, bad, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); tot += h;
  }
  std::printf("%-10s (3) hipMemcpy H2D -> kernel on a non-blocking stream reads, 20 rounds: %llu wrong words\n", name, tot);
  tot = 0;
  for (int it = 0; it < 20; ++it) {
    hipLaunchKernelGGL(wr, dim3(1024), dim3(256), 0, s, buf, n, (uint32_t)(it + 5));
    CK(hipMemcpyAsync(host.data(), buf, (size_t)n * 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i) tot += host[(size_t)i] != ((uint32_t)i * 2654435761u ^ (uint32_t)(it + 5));
  }
  std::printf("%-10s (4) kernel writes -> D2H, 20 rounds: %llu wrong words\n", name, tot);
  CK(hipMemsetAsync(buf, 0, 16, s));
  hipLaunchKernelGGL(atom, dim3(4096), dim3(256), 0, s, buf);
  CK(hipMemcpyAsync(host.data(), buf, 16, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
  std::printf("%-10s (5) atomics from 4096 blocks: %u %u %u %u (expected %u each)\n", name, host[0], host[1], host[2], host[3], 4096u * 64u);
  return 0;
}
int main() {
  CK(hipSetDevice(0));
  const int64_t n = 1 << 22;  // 16 MiB
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long* bad;
  CK(hipMalloc(&bad, 8));
  uint32_t* plain;
  CK(hipMalloc(&plain, (size_t)n * 4));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  const size_t sz = ((size_t)n * 4 + gran - 1) / gran * gran;
  void* va;
  CK(hipMemAddressReserve(&va, sz + 2 * gran, 0, nullptr, 0));
  hipMemGenericAllocationHandle_t h;
  CK(hipMemCreate(&h, sz, &prop, 0));
  CK(hipMemMap((char*)va + gran, sz, 0, h, 0));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess((char*)va + gran, sz, &acc, 1));
  uint32_t* vmm = (uint32_t*)((char*)va + gran);
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, vmm) == hipSuccess) std::printf("vmm pointer attributes: type %d device %d managed %d\n", (int)at.type, at.device, at.isManaged);
  else { (void)hipGetLastError(); std::printf("hipPointerGetAttributes does not know the vmm pointer\n"); }
  if (run("hipMalloc", plain, n, s, bad)) return 2;
  if (run("vmm", vmm, n, s, bad)) return 2;
  return 0;
}
