// d2h_floor_probe.hip -- what ANY one-shot kernel that ends with N bytes in pinned host memory costs on this box: the floor
// under the batching turn's advance list.  An empty kernel; a kernel whose workgroups write N bytes of
// host-mapped memory with lane-consecutive 16-byte stores straight from registers (no loads: pure store + drain); the same
// behind one dependent device-memory load per lane (a kernel that has to look something up first); 64 / 256 / 1024
// workgroups; 350 KB (the turn's 21,845 16-byte advances), 524 KB (24-byte ones) and 2.6 MB (a Step batch's results).
// Events around 200 back-to-back launches each; kernels are dependent (same stream), as a turn's are.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\":\"%s -> %s\"}\n", #x, hipGetErrorString(e_)); return 3; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void empty_kernel() {}
__global__ void store_kernel(u32x4* out, uint64_t n16, const uint32_t* dep) {
  uint32_t seed = dep ? dep[(blockIdx.x * blockDim.x + threadIdx.x) & 0xffff] : 7u;  // one dependent load first, or none
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
    u32x4 v;
    v.x = seed; v.y = (uint32_t)i; v.z = 3; v.w = 4;
    out[i] = v;
  }
}

int main() {
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  void* host = nullptr;
  CK(hipHostMalloc(&host, 4 << 20, hipHostMallocMapped | hipHostMallocCoherent));
  u32x4* hd = nullptr;
  CK(hipHostGetDevicePointer((void**)&hd, host, 0));
  uint32_t* dep = nullptr;
  CK(hipMalloc((void**)&dep, 65536 * 4));
  CK(hipMemset(dep, 1, 65536 * 4));
  const int reps = 200;
  auto time_us = [&](auto launch) -> double {
    for (int i = 0; i < 10; ++i) launch();
    (void)hipStreamSynchronize(s);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
  };
  printf("{\"empty_kernel_us\": %.2f", time_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); }));
  const uint64_t sizes[3] = {21845ull * 16, 21845ull * 24, 65536ull * 40};
  const int wgs[3] = {64, 256, 1024};
  for (uint64_t bytes : sizes)
    for (int w : wgs)
      for (int d = 0; d < 2; ++d) {
        const double us = time_us([&] { hipLaunchKernelGGL(store_kernel, dim3(w), dim3(256), 0, s, hd, bytes / 16, d ? dep : (const uint32_t*)nullptr); });
        printf(",\n \"store_%lluB_%dwg%s_us\": %.2f", (unsigned long long)bytes, w, d ? "_after_one_load" : "", us);
      }
  printf("}\n");
  return 0;
}
