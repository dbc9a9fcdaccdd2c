// sort_check.hip -- the hand-written radix sort (raftsql_amd/csrc/raftq_sort_kernels.hpp) against std::stable_sort
// on the host: sizes from 1 pair to a few million (both the self-scanning and the scanned form, and tiles that grow),
// key widths of 1..40 bits, uniform keys, one hot key, a few hot keys, already sorted and reversed input.
// Built and run by tests/test_sort_gpu.py; prints one line per case and "ALL OK" at the end.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "raftq_sort_kernels.hpp"

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
  uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

int main() {
  struct Case { uint64_t n; int bits; int dist; };  // dist: 0 uniform, 1 one key, 2 three hot keys + noise, 3 sorted, 4 reversed
  std::vector<Case> cases;
  const uint64_t sizes[] = {1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 65536, 100003, (uint64_t)128 * 2048, (uint64_t)128 * 2048 + 1,
                            1000003, (uint64_t)4096 * 2048 + 5, 9000001};
  for (uint64_t n : sizes)
    for (int dist = 0; dist < 5; ++dist) {
      const int bits = n > 4000000 ? 24 : (dist == 0 ? 20 : 15);
      cases.push_back({n, bits, dist});
    }
  for (int bits : {1, 7, 8, 9, 16, 17, 31, 33, 40}) cases.push_back({50000, bits, 0});
  int bad = 0;
  for (const Case& c : cases) {
    const uint64_t n = c.n, mask = c.bits >= 64 ? ~0ull : ((1ull << c.bits) - 1);
    std::vector<uint64_t> k(n);
    std::vector<uint32_t> v(n);
    for (uint64_t i = 0; i < n; ++i) {
      v[i] = (uint32_t)i;
      switch (c.dist) {
        case 0: k[i] = rnd() & mask; break;
        case 1: k[i] = 12345 & mask; break;
        case 2: { const uint64_t r = rnd(); k[i] = (r % 10 < 7 ? (r >> 8) % 3 * 1111 : r >> 16) & mask; break; }
        case 3: k[i] = (i * (mask + 1) / n) & mask; break;
        default: k[i] = ((n - 1 - i) * (mask + 1) / n) & mask; break;
      }
    }
    std::vector<uint32_t> want(n);
    std::iota(want.begin(), want.end(), 0u);
    std::stable_sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
    const raftqk::RadixPlan p = raftqk::radix_plan(n);
    uint64_t *kA, *kB;
    uint32_t *vA, *vB;
    void* scratch;
    CK(hipMalloc(&kA, n * 8));
    CK(hipMalloc(&kB, n * 8));
    CK(hipMalloc(&vA, n * 4));
    CK(hipMalloc(&vB, n * 4));
    CK(hipMalloc(&scratch, p.bytes));
    CK(hipMemcpy(kA, k.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(vA, v.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int in_b = 0;
    CK(hipEventRecord(e0, 0));
    CK(raftqk::radix_sort_pairs(0, scratch, kA, vA, kB, vB, n, c.bits, &in_b));
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> gk(n);
    std::vector<uint32_t> gv(n);
    CK(hipMemcpy(gk.data(), in_b ? kB : kA, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gv.data(), in_b ? vB : vA, n * 4, hipMemcpyDeviceToHost));
    uint64_t wrong = 0;
    for (uint64_t i = 0; i < n; ++i) wrong += gv[i] != want[i] || gk[i] != k[want[i]];
    for (int rep = 0; rep < 5; ++rep) {  // warm timing: the same launches again (the data is sorted by now; the work is the same)
      int dummy = 0;
      CK(hipEventRecord(e0, 0));
      CK(raftqk::radix_sort_pairs(0, scratch, kA, vA, kB, vB, n, c.bits, &dummy));
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float t = 0;
      CK(hipEventElapsedTime(&t, e0, e1));
      ms = std::min(ms, t);
    }
    std::printf("n %9llu bits %2d dist %d tiles %5u rounds %3u %s: %.1f us %s\n", (unsigned long long)n, c.bits, c.dist, p.nb, p.rounds,
                p.scanned ? "scanned" : "self   ", ms * 1e3, wrong ? "MISMATCH" : "ok");
    bad += wrong != 0;
    CK(hipFree(kA));
    CK(hipFree(kB));
    CK(hipFree(vA));
    CK(hipFree(vB));
    CK(hipFree(scratch));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
  }
  std::printf(bad ? "FAILED %d cases\n" : "ALL OK\n", bad);
  return bad ? 1 : 0;
}
