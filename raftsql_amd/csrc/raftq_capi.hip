// raftq_capi.hip -- implementation of include/raftq.h over the gfx950 kernels.
// Host side of the drop-in boundary: owns the padded SoA state in HBM, the
// stream, the double-buffered commit index and the per-wave tallies.  There is
// deliberately no CPU fallback: without a GPU every entry point that needs one
// returns RAFTQ_ENODEV / RAFTQ_EHIP.
#include "raftq.h"

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>
#include <cerrno>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <unordered_set>
#include <vector>

#include "raftq_internal.hpp"

#ifndef RAFTQ_GPL
#define RAFTQ_GPL 4 /* groups per lane; 4 measured best on MI355X (profiles/tune_r01.txt) */
#endif
#ifndef RAFTQ_AUTO_STREAM_BYTES
#define RAFTQ_AUTO_STREAM_BYTES (128ull << 20) /* footprint above which sweeps stream non-temporally */
#endif
#ifndef RAFTQ_LDS_GPL
#define RAFTQ_LDS_GPL 4
#endif

using namespace raftqk;

namespace {

constexpr int kGPL = RAFTQ_GPL;
constexpr int kLdsGPL = RAFTQ_LDS_GPL;
constexpr uint64_t kTile = (uint64_t)kBlock * kGPL;
constexpr uint64_t kRowStagger = 288;  // groups; must stay a multiple of 8 (vote rows are read 8 B at a time)
static_assert(kTileMax % (kBlock * RAFTQ_GPL) == 0, "ld granule must be a multiple of the tile");
static_assert(kTileMax % (kBlock * RAFTQ_LDS_GPL) == 0, "ld granule must be a multiple of the LDS tile");

thread_local std::string g_err;

}  // namespace

namespace raftq_detail {

int fail(raftq_t* h, int code, const std::string& msg) {
  g_err = msg;
  if (h) h->err = msg;
  return code;
}

}  // namespace raftq_detail
using raftq_detail::fail;

namespace {

// What a sweep launch covers: one handle (args by value), or a set of K handles through a device-resident
// table of their SweepArgs -- as one K-deep grid (blockIdx.y = member) or as a persistent grid that walks all
// K x tiles tiles.
struct SweepLaunch {
  const SweepArgs* one = nullptr;  // single handle
  const SweepArgs* tab = nullptr;  // set: device table [members]
  uint32_t members = 1;
  uint32_t want_bits = 0;
  uint32_t persist_wgs = 0;        // > 0: persistent walk with this many workgroups
  uint64_t gpad = 0;
};

// nt: 0 = cached loads and stores, 1 = non-temporal loads, normal stores, 3 = both non-temporal
// Tile size of the K-deep set grid, per peer count (profiles/r02/tune3_focus_*.jsonl, medians of 9 interleaved
// runs): 2048-group tiles (GPL 8) win for N <= 5 (10.92 vs 11.15 us per 1M x 5 batch), 512-group tiles (GPL 2)
// from N = 6 up, where the 8-group register set would not fit 5 waves per SIMD (N = 7: 29.0 vs 30.2 us).
constexpr int set_gpl(int n_peers) { return n_peers <= 5 ? 8 : 2; }

template <int N, bool COMMIT, bool GATED, bool VOTES, int POLICY>
hipError_t launch_pol(const SweepLaunch& L, hipStream_t s) {
  const unsigned tiles = (unsigned)(L.gpad / kTile);
  if (L.tab == nullptr) {
    hipLaunchKernelGGL((sweep_kernel<N, kGPL, COMMIT, GATED, VOTES, POLICY, true>), dim3(tiles), dim3(kBlock), 0, s, *L.one);
  } else if (L.persist_wgs == 0) {
    constexpr int G = set_gpl(N);
    hipLaunchKernelGGL((sweep_set_kernel<N, G, COMMIT, GATED, VOTES, POLICY, true>),
                       dim3((unsigned)(L.gpad / ((uint64_t)kBlock * G)), L.members), dim3(kBlock), 0, s, L.tab, L.want_bits);
  } else {
    const uint32_t total = tiles * L.members;
    hipLaunchKernelGGL((sweep_persist_kernel<N, kGPL, COMMIT, GATED, VOTES, POLICY, true>),
                       dim3(std::min<uint32_t>(L.persist_wgs, total)), dim3(kBlock), 0, s, L.tab, tiles, total, L.want_bits);
  }
  return hipGetLastError();
}

template <int N, bool COMMIT, bool GATED, bool VOTES>
hipError_t launch_reg(const SweepLaunch& L, int nt, hipStream_t s) {
  if (nt == 3) return launch_pol<N, COMMIT, GATED, VOTES, kLdNT | kStNT>(L, s);
  if (nt == 1) return launch_pol<N, COMMIT, GATED, VOTES, kLdNT>(L, s);
  return launch_pol<N, COMMIT, GATED, VOTES, 0>(L, s);
}

template <int N, bool GATED, bool VOTES>
hipError_t launch_lds(const SweepArgs& a, uint64_t gpad, hipStream_t s) {
  constexpr uint64_t tile = (uint64_t)kBlock * kLdsGPL;
  constexpr int rows = N + 1 + (GATED ? 1 : 0);
  constexpr size_t lds = (size_t)kWaves * (kLdsGPL / 2) * rows * 1024;
  const dim3 grid((unsigned)(gpad / tile));
  hipLaunchKernelGGL((sweep_lds_kernel<N, kLdsGPL, GATED, VOTES, true>), grid, dim3(kBlock), lds, s, a);
  return hipGetLastError();
}

template <int N>
hipError_t launch_n(const SweepLaunch& L, unsigned flags, int nt, hipStream_t s) {
  const bool commit = flags & (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED);
  const bool gated = flags & RAFTQ_SWEEP_GATED;
  const bool votes = flags & RAFTQ_SWEEP_VOTES;
  if ((flags & RAFTQ_SWEEP_LDS) && commit) {
    if (L.tab) return hipErrorInvalidValue;  // the LDS A/B variant sweeps one handle at a time
    if (gated) return votes ? launch_lds<N, true, true>(*L.one, L.gpad, s) : launch_lds<N, true, false>(*L.one, L.gpad, s);
    return votes ? launch_lds<N, false, true>(*L.one, L.gpad, s) : launch_lds<N, false, false>(*L.one, L.gpad, s);
  }
  if (commit && gated) return votes ? launch_reg<N, true, true, true>(L, nt, s) : launch_reg<N, true, true, false>(L, nt, s);
  if (commit) return votes ? launch_reg<N, true, false, true>(L, nt, s) : launch_reg<N, true, false, false>(L, nt, s);
  return launch_reg<N, false, false, true>(L, nt, s);
}

// the segmented turn's sweep (sweep_segments_kernel): commit-only sweeps of one handle
template <int N>
hipError_t launch_segments_n(const SweepArgs& a, uint32_t n_tiles, bool gated, int nt, Advance16* list, unsigned int* counts_host, unsigned int* counts_dev,
                             hipStream_t s, const Arrival& ar) {
  const dim3 grid(n_tiles), block(kBlock);
  if (gated) {
    if (nt == 3) hipLaunchKernelGGL((sweep_segments_kernel<N, kGPL, true, kLdNT | kStNT>), grid, block, 0, s, a, list, counts_host, counts_dev, ar);
    else if (nt == 1) hipLaunchKernelGGL((sweep_segments_kernel<N, kGPL, true, kLdNT>), grid, block, 0, s, a, list, counts_host, counts_dev, ar);
    else hipLaunchKernelGGL((sweep_segments_kernel<N, kGPL, true, 0>), grid, block, 0, s, a, list, counts_host, counts_dev, ar);
  } else {
    if (nt == 3) hipLaunchKernelGGL((sweep_segments_kernel<N, kGPL, false, kLdNT | kStNT>), grid, block, 0, s, a, list, counts_host, counts_dev, ar);
    else if (nt == 1) hipLaunchKernelGGL((sweep_segments_kernel<N, kGPL, false, kLdNT>), grid, block, 0, s, a, list, counts_host, counts_dev, ar);
    else hipLaunchKernelGGL((sweep_segments_kernel<N, kGPL, false, 0>), grid, block, 0, s, a, list, counts_host, counts_dev, ar);
  }
  return hipGetLastError();
}
hipError_t launch_segments(uint32_t N, const SweepArgs& a, uint32_t n_tiles, bool gated, int nt, Advance16* list, unsigned int* counts_host,
                           unsigned int* counts_dev, hipStream_t s, const Arrival& ar) {
  switch (N) {
    case 1: return launch_segments_n<1>(a, n_tiles, gated, nt, list, counts_host, counts_dev, s, ar);
    case 2: return launch_segments_n<2>(a, n_tiles, gated, nt, list, counts_host, counts_dev, s, ar);
    case 3: return launch_segments_n<3>(a, n_tiles, gated, nt, list, counts_host, counts_dev, s, ar);
    case 4: return launch_segments_n<4>(a, n_tiles, gated, nt, list, counts_host, counts_dev, s, ar);
    case 5: return launch_segments_n<5>(a, n_tiles, gated, nt, list, counts_host, counts_dev, s, ar);
    case 6: return launch_segments_n<6>(a, n_tiles, gated, nt, list, counts_host, counts_dev, s, ar);
    case 7: return launch_segments_n<7>(a, n_tiles, gated, nt, list, counts_host, counts_dev, s, ar);
    case 8: return launch_segments_n<8>(a, n_tiles, gated, nt, list, counts_host, counts_dev, s, ar);
    case 9: return launch_segments_n<9>(a, n_tiles, gated, nt, list, counts_host, counts_dev, s, ar);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_sweep(uint32_t N, const SweepLaunch& L, unsigned flags, int nt, hipStream_t s) {
  switch (N) {
    case 1: return launch_n<1>(L, flags, nt, s);
    case 2: return launch_n<2>(L, flags, nt, s);
    case 3: return launch_n<3>(L, flags, nt, s);
    case 4: return launch_n<4>(L, flags, nt, s);
    case 5: return launch_n<5>(L, flags, nt, s);
    case 6: return launch_n<6>(L, flags, nt, s);
    case 7: return launch_n<7>(L, flags, nt, s);
    case 8: return launch_n<8>(L, flags, nt, s);
    case 9: return launch_n<9>(L, flags, nt, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace
// something other than Step is about to change match / committed / first_idx / role on the device: the per-group records' copies
// of them go stale (raftq_step_kernels.hpp NodeRec)
static inline void dense_changed(raftq_t* h) { h->node_mirror_fresh = false; }

int raftq_detail::ensure_staging(raftq_t* h, size_t bytes) {
  if (bytes <= h->stage_bytes) return RAFTQ_OK;
  size_t want = std::max(bytes, h->stage_bytes * 2);
  want = std::max<size_t>(want, 1 << 20);
  if (h->stage_h) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipHostFree(h->stage_h));
    h->stage_h = h->stage_d = nullptr;
    h->stage_bytes = 0;
  }
  HIPCHK(h, hipHostMalloc(&h->stage_h, want, hipHostMallocMapped));
  HIPCHK(h, hipHostGetDevicePointer(&h->stage_d, h->stage_h, 0));
  h->stage_bytes = want;
  return RAFTQ_OK;
}

// Can this process really store into `p` (device memory behind the BAR)?  hipDeviceAttributeIsLargeBar says the aperture
// exists; whether the allocation is mapped writable into THIS process is read off /proc/self/maps -- no store is tried
// and no signal handler is touched.  (Round 2 probed with a guarded store and a temporary SIGSEGV / SIGBUS handler that
// siglongjmp'd out of the fault: undefined behaviour in a host with other threads that use those signals -- the Go
// runtime this library is meant to sit under requires SA_ONSTACK handlers, a JVM installs its own)
bool raftq_detail::host_can_write(void* p, size_t bytes) {
  const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
  std::FILE* f = std::fopen("/proc/self/maps", "r");
  if (!f) return false;  // cannot tell: stay on pinned host memory
  char line[512];
  uintptr_t covered = lo;  // [lo, covered) is mapped writable so far (maps is sorted by address)
  while (covered < hi && std::fgets(line, sizeof line, f)) {
    unsigned long long a = 0, b = 0;
    char perms[8] = {0};
    if (std::sscanf(line, "%llx-%llx %7s", &a, &b, perms) != 3) continue;
    if ((uintptr_t)b <= covered) continue;
    if ((uintptr_t)a > covered) break;           // a hole in front of the next mapping
    if (perms[0] != 'r' || perms[1] != 'w') break;  // mapped, but not for stores
    covered = (uintptr_t)b;
  }
  std::fclose(f);
  if (covered < hi) return false;
  // Mapped "rw" is not yet "a store works": in some containers / VMs the aperture is mapped and a store to it still faults
  // (ADVICE r03).  Try one without risking a signal: a one-byte pwrite through /proc/self/mem goes through the kernel's
  // access path and FAILS (EFAULT / EIO) where a store would have raised SIGBUS.  The buffer is fresh and about to be
  // overwritten by its producer, so the byte written does not matter.  Three outcomes: written -> device staging;
  // refused with EFAULT -> pinned host staging; the probe itself unavailable (no /proc/self/mem, EPERM, EIO: the mapping
  // has no access method) -> the maps answer stands.  RAFTQ_STAGE=host forces pinned staging whatever is found here.
  const int fd = ::open("/proc/self/mem", O_WRONLY | O_CLOEXEC);
  if (fd < 0) return true;
  const char zero = 0;
  bool ok = true;
  for (const uintptr_t at : {lo, hi - 1}) {
    if (::pwrite(fd, &zero, 1, (off_t)at) == 1) continue;
    if (errno == EFAULT) ok = false;
    break;
  }
  ::close(fd);
  return ok;
}

int raftq_detail::ensure_ingest(raftq_t* h, size_t bytes) {
  if (bytes <= h->ingest_bytes) return RAFTQ_OK;
  size_t want = std::max(bytes, h->ingest_bytes * 2);
  want = std::max<size_t>(want, 1 << 20);
  if (h->ingest_h) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->ingest_in_device) HIPCHK(h, hipFree(h->ingest_h));
    else HIPCHK(h, hipHostFree(h->ingest_h));
    h->ingest_h = h->ingest_d = nullptr;
    h->ingest_bytes = 0;
  }
  if (h->bar_staging) {
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, want, hipDeviceMallocFinegrained) == hipSuccess) {
      if (h->bar_probed || host_can_write(p, want)) {
        h->bar_probed = true;
        h->ingest_h = h->ingest_d = p;
        h->ingest_in_device = true;
        h->ingest_bytes = want;
        return RAFTQ_OK;
      }
      (void)hipFree(p);
    }
    (void)hipGetLastError();
    h->bar_staging = false;  // no host-writable device memory here: pinned host memory from now on
    if (std::getenv("RAFTQ_PROFILE"))
      std::fprintf(stderr, "[raftq] the ack / inbound staging stays in pinned host memory: device memory behind the BAR is not "
                           "host-writable here (RAFTQ_STAGE=host asks for this outright)\n");
  }
  HIPCHK(h, hipHostMalloc(&h->ingest_h, want, hipHostMallocMapped));
  HIPCHK(h, hipHostGetDevicePointer(&h->ingest_d, h->ingest_h, 0));
  h->ingest_in_device = false;
  h->ingest_bytes = want;
  return RAFTQ_OK;
}

namespace {
// the producer's stores into a device-resident ack buffer are write-combined: drain them before the doorbell
inline void publish_ingest(const raftq_t* h) {
#if defined(__x86_64__)
  if (h->ingest_in_device) __builtin_ia32_sfence();
#else
  (void)h;
#endif
}

}  // namespace
// The buffers a kernel writes for the host to read while the kernel may still be running (the advance list, the totals,
// the completion flag) are fine-grained: the device's stores go out over PCIe as they are issued instead of sitting in
// its L2 until the end-of-kernel write-back.  Stated explicitly -- without the flag the choice is the runtime's
// (HIP_HOST_COHERENT).  RAFTQ_HOST_COHERENT=0 leaves it to the runtime (A/B).
unsigned raftq_detail::host_coherence_flag() {
  static const unsigned f = [] {
    const char* e = std::getenv("RAFTQ_HOST_COHERENT");
    return e && e[0] == '0' ? 0u : (unsigned)hipHostMallocCoherent;
  }();
  return f;
}
using raftq_detail::host_coherence_flag;
namespace {

int ensure_adv(raftq_t* h, uint64_t entries) {
  if (entries <= h->adv_cap) return RAFTQ_OK;
  uint64_t want = std::max<uint64_t>(entries, h->adv_cap * 2);
  want = std::min<uint64_t>(std::max<uint64_t>(want, 4096), h->gpad);
  if (h->adv_h) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipHostFree(h->adv_h));
    h->adv_h = h->adv_d = nullptr;
    h->adv_cap = 0;
  }
  HIPCHK(h, hipHostMalloc((void**)&h->adv_h, want * sizeof(Advance), hipHostMallocMapped | host_coherence_flag()));
  HIPCHK(h, hipHostGetDevicePointer((void**)&h->adv_d, h->adv_h, 0));
  h->adv_cap = want;
  return RAFTQ_OK;
}

}  // namespace
int raftq_detail::use_device(raftq_t* h) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  HIPCHK(h, hipSetDevice(h->device));
  return RAFTQ_OK;
}
// Every call that reads or changes group state outside raftq_step_submit / _collect: a submitted Step batch
// is not necessarily applied until its collect (a batch with a long per-group run is replayed there), so
// state calls are refused while batches are in flight instead of silently running ahead of them.
int raftq_detail::use_device_idle(raftq_t* h, const char* who) {
  if (int rc = use_device(h)) return rc;
  if (h->step_collected != h->step_submitted)
    return fail(h, RAFTQ_ESTATE, std::string(who) + ": Step batches are in flight; collect them first");
  return RAFTQ_OK;
}
using raftq_detail::ensure_staging;
using raftq_detail::ensure_ingest;
using raftq_detail::use_device;
using raftq_detail::use_device_idle;
using raftq_detail::ensure_tick_state;

// host <-> device form of the RequestVote state.  The ABI speaks one byte per (peer, group); the device keeps one
// word per group, 2 bits per peer.  Bytes other than 1 / 2 are "no response" (include/raftq.h) and load as 00.
template <typename W>
static void pack_votes(const uint8_t* votes, uint64_t G, uint32_t N, W* out) {
  for (uint64_t g = 0; g < G; ++g) out[g] = 0;
  for (uint32_t p = 0; p < N; ++p) {
    const uint8_t* row = votes + (size_t)p * G;
    for (uint64_t g = 0; g < G; ++g) {
      const uint8_t v = row[g];
      out[g] = (W)(out[g] | (W)((v == 1 ? 1u : v == 2 ? 2u : 0u) << (2 * p)));
    }
  }
}
template <typename W>
static void unpack_votes(const W* in, uint64_t G, uint32_t N, uint8_t* votes) {
  for (uint32_t p = 0; p < N; ++p) {
    uint8_t* row = votes + (size_t)p * G;
    for (uint64_t g = 0; g < G; ++g) {
      const uint32_t f = ((uint32_t)in[g] >> (2 * p)) & 3u;
      row[g] = f == 3u ? 0 : (uint8_t)f;
    }
  }
}

extern "C" {

int raftq_abi_version(void) { return RAFTQ_ABI_VERSION; }

uint32_t raftq_quorum(uint32_t n_peers) { return n_peers / 2 + 1; }

int raftq_device_count(int* n) {
  if (!n) return fail(nullptr, RAFTQ_EINVAL, "raftq_device_count: null out");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    return fail(nullptr, RAFTQ_ENODEV, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *n = c;
  return RAFTQ_OK;
}

const char* raftq_last_error(const raftq_t* h) { return h ? h->err.c_str() : g_err.c_str(); }

uint64_t raftq_groups(const raftq_t* h) { return h ? h->G : 0; }
uint32_t raftq_peers(const raftq_t* h) { return h ? h->N : 0; }

int raftq_create(int device, uint64_t n_groups, uint32_t n_peers, raftq_t** out) {
  if (!out) return fail(nullptr, RAFTQ_EINVAL, "raftq_create: null out");
  *out = nullptr;
  if (n_groups == 0 || n_groups > RAFTQ_MAX_GROUPS)
    return fail(nullptr, RAFTQ_EINVAL, "raftq_create: n_groups out of range (1 .. RAFTQ_MAX_GROUPS = 2^30 per handle)");
  if (n_peers < 1 || n_peers > RAFTQ_MAX_PEERS) return fail(nullptr, RAFTQ_EINVAL, "raftq_create: n_peers must be 1..9");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, RAFTQ_ENODEV, "raftq_create: no HIP device (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(nullptr, RAFTQ_ENODEV, "raftq_create: device index out of range");
  raftq_t* h = new (std::nothrow) raftq();
  if (!h) return fail(nullptr, RAFTQ_ENOMEM, "raftq_create: host allocation failed");
  h->device = device;
  h->G = n_groups;
  h->N = n_peers;
  h->last_gpl = kGPL;
  h->gpad = (n_groups + kTileMax - 1) / kTileMax * kTileMax;
  // Row stride = padded groups + a stagger.  Rows exactly 2^k bytes apart put the N+1 row loads of
  // a wave on the same HBM channel at the same instant; 288 groups (2304 B of u64, 288 B of u8 --
  // not a multiple of 16 KiB) measured -2.3 % sweep time on MI355X (profiles/r01/tune_stagger.txt).
  h->ld = h->gpad + kRowStagger;
  const uint64_t ld = h->ld;
  // the finest tile any variant uses bounds the number of per-wave partials
  h->max_partials = h->gpad / (kBlock * 2) * kWaves;
  int rc = RAFTQ_OK;
  auto alloc = [&](void** p, size_t bytes) -> int {
    HIPCHK(h, hipMalloc(p, bytes));
    HIPCHK(h, hipMemsetAsync(*p, 0, bytes, h->stream));
    return RAFTQ_OK;
  };
  do {
    if ((rc = use_device(h))) break;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { rc = fail(h, RAFTQ_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); break; }
    h->own_stream = true;
    {
      int large_bar = 0;
      (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device);
      const char* st = std::getenv("RAFTQ_STAGE");
      h->bar_staging = large_bar == 1 && !(st && std::strcmp(st, "host") == 0);
    }
    if ((rc = alloc((void**)&h->match, (size_t)n_peers * ld * 8))) break;
    if ((rc = alloc((void**)&h->committed[0], ld * 8))) break;
    if ((rc = alloc((void**)&h->committed[1], ld * 8))) break;
    if ((rc = alloc((void**)&h->first_idx, ld * 8))) break;
    // RequestVote state: one packed word per group (2 bits per peer), outcomes 2 bits per group
    if ((rc = alloc((void**)&h->votes, (size_t)vote_word_bytes((int)n_peers) * ld))) break;
    if ((rc = alloc((void**)&h->outcome, ld / 4 + 64))) break;
    if ((rc = alloc((void**)&h->changed_bits, h->gpad / 8))) break;
    if ((rc = alloc((void**)&h->partials, h->max_partials * sizeof(uint4)))) break;
    if ((rc = alloc((void**)&h->offsets, (h->max_partials + 1) * 8))) break;
    if ((rc = alloc((void**)&h->compact_arrived, 64))) break;
    e = hipHostMalloc((void**)&h->h_partials, h->max_partials * sizeof(uint4), hipHostMallocDefault);
    if (e != hipSuccess) { rc = fail(h, RAFTQ_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); break; }
    e = hipHostMalloc((void**)&h->h_total, 64, hipHostMallocMapped | host_coherence_flag());
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&h->d_total, h->h_total, 0);
    if (e != hipSuccess) { rc = fail(h, RAFTQ_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e)); break; }
    e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) { rc = fail(h, RAFTQ_EHIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); break; }
    e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { rc = fail(h, RAFTQ_EHIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e)); break; }
  } while (0);
  if (rc != RAFTQ_OK) {
    std::string keep = h->err;
    raftq_destroy(h);
    g_err = keep;
    return rc;
  }
  *out = h;
  return RAFTQ_OK;
}

void raftq_destroy(raftq_t* h) {
  if (!h) return;
  if (h->in_set) {  // destroyed under its set: the set refuses further sweeps instead of touching freed memory
    raftq_set_t* s = h->in_set;
    s->broken = true;
    s->members.erase(std::remove(s->members.begin(), s->members.end(), h), s->members.end());
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(s->stream);
    h->stream = nullptr;
    h->own_stream = false;
  }
  if (h->prof_n && std::getenv("RAFTQ_PROFILE"))
    std::fprintf(stderr,
                 "[raftq] cycle phases, avg us over %llu calls: stage/validate %.1f | enqueue scatter %.1f | enqueue sweep "
                 "%.1f | enqueue collect %.1f | sync %.1f | copy-out %.1f | turns that ended in the blocking wait instead of the "
                 "polled flag: %llu\n",
                 (unsigned long long)h->prof_n, h->prof[0] / h->prof_n, h->prof[1] / h->prof_n, h->prof[2] / h->prof_n,
                 h->prof[3] / h->prof_n, h->prof[4] / h->prof_n, h->prof[5] / h->prof_n, (unsigned long long)h->flag_fallbacks);
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  (void)hipFree(h->match);
  (void)hipFree(h->committed[0]);
  (void)hipFree(h->committed[1]);
  (void)hipFree(h->first_idx);
  (void)hipFree(h->votes);
  (void)hipFree(h->outcome);
  (void)hipFree(h->changed_bits);
  (void)hipFree(h->partials);
  (void)hipFree(h->seg_d);
  if (h->seg_h) (void)hipHostFree(h->seg_h);
  (void)hipFree(h->offsets);
  (void)hipFree(h->compact_arrived);
  (void)hipFree(h->claim);
  (void)hipFree(h->delta_dev);
  (void)hipFree(h->delta_bad);
  (void)hipFree(h->role);
  (void)hipFree(h->elapsed);
  (void)hipFree(h->action);
  (void)hipFree(h->hup_bits);
  (void)hipFree(h->beat_bits);
  (void)hipFree(h->tick_partials);
  (void)hipFree(h->tick_offsets2);
  raftq_detail::free_node_state(h);
  raftq_detail::free_wire_state(h);
  if (h->stage_h) (void)hipHostFree(h->stage_h);
  if (h->ingest_h) (void)(h->ingest_in_device ? hipFree(h->ingest_h) : hipHostFree(h->ingest_h));
  if (h->adv_h) (void)hipHostFree(h->adv_h);
  if (h->tl_h) (void)hipHostFree(h->tl_h);
  if (h->h_partials) (void)hipHostFree(h->h_partials);
  if (h->h_total) (void)hipHostFree(h->h_total);
  if (h->arrive_count) (void)hipFree(h->arrive_count);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->many_aux) {
    (void)hipStreamSynchronize(h->many_aux);
    (void)hipEventDestroy(h->many_fork);
    (void)hipEventDestroy(h->many_join);
    (void)hipStreamDestroy(h->many_aux);
  }
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int raftq_set_stream(raftq_t* h, void* stream) {
  if (int rc = use_device_idle(h, "raftq_set_stream")) return rc;
  if (h->in_set) return fail(h, RAFTQ_ESTATE, "raftq_set_stream: the handle is a member of a sweep set, which owns its stream");
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->own_stream) HIPCHK(h, hipStreamDestroy(h->stream));
  h->stream = (hipStream_t)stream;
  h->own_stream = false;
  return RAFTQ_OK;
}

void* raftq_get_stream(const raftq_t* h) { return h ? (void*)h->stream : nullptr; }

int raftq_load_match(raftq_t* h, const uint64_t* match, const uint64_t* committed) {
  if (int rc = use_device_idle(h, "raftq_load_match")) return rc;
  if (!match && !committed) return fail(h, RAFTQ_EINVAL, "raftq_load_match: nothing to load");
  // one copy per peer row: a row of a large handle is wider than the 2D copy's pitch limit (2^31 - 1 bytes)
  if (match)
    for (uint32_t p = 0; p < h->N; ++p)
      HIPCHK(h, hipMemcpyAsync(h->match + (size_t)p * h->ld, match + (size_t)p * h->G, h->G * 8, hipMemcpyHostToDevice, h->stream));
  dense_changed(h);
  if (committed)
    HIPCHK(h, hipMemcpyAsync(h->committed[h->cur], committed, h->G * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RAFTQ_OK;
}

int raftq_load_terms(raftq_t* h, const uint64_t* cur_term, const uint64_t* first_idx_cur_term) {
  if (int rc = use_device_idle(h, "raftq_load_terms")) return rc;
  if (!cur_term || !first_idx_cur_term) return fail(h, RAFTQ_EINVAL, "raftq_load_terms: null argument");
  // A group whose term is 0 has never seen an election: it is not a leader and
  // commits nothing.  Fold that into the compact gate so the kernel reads one
  // array (DESIGN.md "term gate").
  std::vector<uint64_t> f;
  try {
    f.assign(first_idx_cur_term, first_idx_cur_term + h->G);
  } catch (...) {
    return fail(h, RAFTQ_ENOMEM, "raftq_load_terms: host allocation failed");
  }
  for (uint64_t g = 0; g < h->G; ++g)
    if (cur_term[g] == 0) f[g] = 0;
  HIPCHK(h, hipMemcpyAsync(h->first_idx, f.data(), h->G * 8, hipMemcpyHostToDevice, h->stream));
  dense_changed(h);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->have_terms = true;
  return RAFTQ_OK;
}

int raftq_load_votes(raftq_t* h, const uint8_t* votes) {
  if (int rc = use_device_idle(h, "raftq_load_votes")) return rc;
  if (!votes) return fail(h, RAFTQ_EINVAL, "raftq_load_votes: null argument");
  const size_t wb = (size_t)vote_word_bytes((int)h->N);
  std::vector<uint8_t> packed;
  try {
    packed.resize(h->G * wb);
  } catch (...) {
    return fail(h, RAFTQ_ENOMEM, "raftq_load_votes: host allocation failed");
  }
  if (wb == 2) pack_votes(votes, h->G, h->N, reinterpret_cast<uint16_t*>(packed.data()));
  else pack_votes(votes, h->G, h->N, reinterpret_cast<uint32_t*>(packed.data()));
  HIPCHK(h, hipMemcpyAsync(h->votes, packed.data(), packed.size(), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RAFTQ_OK;
}

// enqueue only (no sync): validate, copy into the pinned staging area at byte
// offset `off`, launch the scatter.  Shared by raftq_apply_* and raftq_cycle.
static int ensure_delta_dev(raftq_t* h, size_t bytes) {
  if (!h->delta_bad) {
    HIPCHK(h, hipMalloc((void**)&h->delta_bad, 64));
    HIPCHK(h, hipMemsetAsync(h->delta_bad, 0, 64, h->stream));
  }
  if (bytes <= h->delta_dev_bytes) return RAFTQ_OK;
  if (h->delta_dev) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipFree(h->delta_dev));
    h->delta_dev = nullptr;
    h->delta_dev_bytes = 0;
  }
  const size_t want = std::max(bytes * 2, (size_t)1 << 20);
  HIPCHK(h, hipMalloc(&h->delta_dev, want));
  h->delta_dev_bytes = want;
  return RAFTQ_OK;
}

// after the caller's sync: did the device find a bad record in the batch(es) enqueued since the last check?
static int check_deltas(raftq_t* h, bool trusted = false) {
  int rc = RAFTQ_OK;
  const char* what = trusted ? "; that record was dropped, every other record of the turn was applied (RAFTQ_CYCLE_TRUSTED)"
                             : "; nothing applied";
  if (h->delta_check[0] && h->h_total[1] == h->delta_check[0])
    rc = fail(h, RAFTQ_EINVAL, std::string("a match delta is out of range (group >= G or peer >= N)") + what);
  if (h->delta_check[1] && h->h_total[2] == h->delta_check[1])
    rc = fail(h, RAFTQ_EINVAL, std::string("a vote delta is invalid (group/peer out of range or vote not 1/2)") + what);
  h->delta_check[0] = h->delta_check[1] = 0;
  return rc;
}

}  // extern "C" (templates need C++ linkage)

// enqueue only (no sync): copy the batch(es) into the pinned staging area unless the caller filled it in place
// (raftq_stage), then device-side: bring the records into HBM validating them (both kinds first), then scatter.
// Every scatter kernel checks the verdict of BOTH kinds, so one bad record of either kind withholds the whole
// call (raftq_cycle: all or nothing).  The verdict is read by check_deltas() after the caller's sync.
// Shared by raftq_apply_* and raftq_cycle*; Rec is the match-delta layout (24-byte or packed 16-byte).
template <typename Rec, typename AbiRec>
static int enqueue_ingest(raftq_t* h, const AbiRec* d, uint64_t n, const raftq_vote_delta_t* vd, uint64_t nv, size_t off_votes,
                          bool trusted = false) {
  static_assert(sizeof(Rec) == sizeof(AbiRec), "ABI struct mismatch");
  static_assert(sizeof(VoteDeltaRec) == sizeof(raftq_vote_delta_t), "ABI struct mismatch");
  if (nv > 0xfffffffeull) return fail(h, RAFTQ_EINVAL, "vote delta batch too large");
  // A pointer INTO the ack buffer must be exactly where this call's counts and record layout put that array: records
  // staged with other counts (or through the other layout's raftq_stage*) would overlap their own destination, and
  // reading them back out of a device-resident buffer crosses the uncached BAR.
  auto inside = [&](const void* q) {
    return (const uint8_t*)q >= (const uint8_t*)h->ingest_h && (const uint8_t*)q < (const uint8_t*)h->ingest_h + h->ingest_bytes;
  };
  if (n) {
    AbiRec* dst = (AbiRec*)h->ingest_h;
    if ((const void*)d != (const void*)dst) {
      if (inside(d)) return fail(h, RAFTQ_EINVAL, "match deltas staged with other counts / another record layout than this call's");
      std::memcpy(dst, d, (size_t)n * sizeof(AbiRec));
    }
  }
  if (nv) {
    raftq_vote_delta_t* dst = (raftq_vote_delta_t*)((uint8_t*)h->ingest_h + off_votes);
    if ((const void*)vd != (const void*)dst) {
      if (inside(vd)) return fail(h, RAFTQ_EINVAL, "vote deltas staged with other counts / another record layout than this call's");
      std::memcpy(dst, vd, (size_t)nv * sizeof(raftq_vote_delta_t));
    }
    if (!h->claim) {
      const size_t bytes = (size_t)h->N * h->ld * sizeof(uint32_t);
      HIPCHK(h, hipMalloc((void**)&h->claim, bytes));
      HIPCHK(h, hipMemsetAsync(h->claim, 0xff, bytes, h->stream));
    }
  }
  // match and vote records share the device buffer: votes go behind the matches (same offsets as in staging)
  publish_ingest(h);
  if (int rc = ensure_delta_dev(h, h->ingest_bytes)) return rc;
  const unsigned long long em = n ? ++h->delta_epoch : kNoEpoch, ev = nv ? ++h->delta_epoch : kNoEpoch;
  h->delta_check[0] = n ? em : 0;
  h->delta_check[1] = nv ? ev : 0;
  Rec* dev_m = (Rec*)h->delta_dev;
  VoteDeltaRec* dev_v = (VoteDeltaRec*)((uint8_t*)h->delta_dev + off_votes);
  const unsigned long long* bad = h->delta_bad;
  const dim3 gm((unsigned)((n + kBlock - 1) / kBlock)), gv((unsigned)((nv + kBlock - 1) / kBlock));
  if (n) dense_changed(h);  // (match words move under Step's records)
  if (n && trusted)
    hipLaunchKernelGGL((deltas_in_apply_kernel<Rec>), gm, dim3(kBlock), 0, h->stream, (const Rec*)h->ingest_d, n, h->match, h->ld,
                       h->G, h->N, h->delta_bad, h->d_total + 1, em);
  else if (n)
    hipLaunchKernelGGL((deltas_in_kernel<Rec>), gm, dim3(kBlock), 0, h->stream, (const Rec*)h->ingest_d, dev_m, n, h->G, h->N,
                       h->delta_bad, h->d_total + 1, em);
  if (nv)
    hipLaunchKernelGGL(vote_deltas_in_kernel, gv, dim3(kBlock), 0, h->stream,
                       (const VoteDeltaRec*)((uint8_t*)h->ingest_d + off_votes), dev_v, nv, h->G, h->N, h->delta_bad + 1,
                       h->d_total + 2, ev);
  if (n && !trusted)
    hipLaunchKernelGGL((apply_deltas_kernel<Rec>), gm, dim3(kBlock), 0, h->stream, h->match, h->ld, (const Rec*)dev_m, n, bad, em,
                       ev);
  if (nv) {
    // trusted match deltas are dropped one by one, never as a batch: their verdict does not gate the votes
    const unsigned long long em_gate = trusted ? kNoEpoch : em;
    // and a trusted turn's vote records are dropped one by one as well: what a trusted turn applies never depends on
    // another record of the batch
    hipLaunchKernelGGL(vote_claim_kernel, gv, dim3(kBlock), 0, h->stream, h->claim, h->ld, (const VoteDeltaRec*)dev_v, nv, bad,
                       em_gate, ev, h->G, h->N, trusted ? 1 : 0);
    hipLaunchKernelGGL(vote_apply_kernel, gv, dim3(kBlock), 0, h->stream, h->votes, h->N > 8 ? 1 : 0, h->claim, h->ld,
                       (const VoteDeltaRec*)dev_v, nv, bad, em_gate, ev, h->G, h->N, trusted ? 1 : 0);
  }
  HIPCHK(h, hipGetLastError());
  return RAFTQ_OK;
}

extern "C" {

int raftq_apply_deltas(raftq_t* h, const raftq_delta_t* d, uint64_t n) {
  if (int rc = use_device_idle(h, "raftq_apply_deltas")) return rc;
  if (n == 0) return RAFTQ_OK;
  if (!d) return fail(h, RAFTQ_EINVAL, "raftq_apply_deltas: null argument");
  if (int rc = ensure_ingest(h, (size_t)n * sizeof(raftq_delta_t))) return rc;
  if (int rc = enqueue_ingest<DeltaRec>(h, d, n, nullptr, 0, 0)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));  // the staging area is reused by the next call
  return check_deltas(h);
}

int raftq_apply_term_deltas(raftq_t* h, const raftq_term_delta_t* d, uint64_t n) {
  static_assert(sizeof(TermDeltaRec) == sizeof(raftq_term_delta_t), "ABI struct mismatch");
  if (int rc = use_device_idle(h, "raftq_apply_term_deltas")) return rc;
  if (n == 0) return RAFTQ_OK;
  if (!d) return fail(h, RAFTQ_EINVAL, "raftq_apply_term_deltas: null argument");
  for (uint64_t i = 0; i < n; ++i)
    if (d[i].group >= h->G) return fail(h, RAFTQ_EINVAL, "a term delta is out of range; nothing applied");
  if (int rc = ensure_staging(h, (size_t)n * sizeof(raftq_term_delta_t))) return rc;
  // keep only the last record per group so the scatter has no write-write race
  raftq_term_delta_t* dst = (raftq_term_delta_t*)h->stage_h;
  uint64_t m = 0;
  {
    std::unordered_set<uint64_t> seen;
    seen.reserve((size_t)n * 2);
    std::vector<uint64_t> keep;
    keep.reserve(n);
    for (uint64_t i = n; i-- > 0;)
      if (seen.insert(d[i].group).second) keep.push_back(i);
    for (auto it = keep.rbegin(); it != keep.rend(); ++it) dst[m++] = d[*it];
  }
  const dim3 grid((unsigned)((m + kBlock - 1) / kBlock));
  dense_changed(h);
  hipLaunchKernelGGL(apply_term_deltas_kernel, grid, dim3(kBlock), 0, h->stream, h->first_idx,
                     (const TermDeltaRec*)h->stage_d, m);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->have_terms = true;
  return RAFTQ_OK;
}

int raftq_apply_vote_deltas(raftq_t* h, const raftq_vote_delta_t* d, uint64_t n) {
  if (int rc = use_device_idle(h, "raftq_apply_vote_deltas")) return rc;
  if (n == 0) return RAFTQ_OK;
  if (!d) return fail(h, RAFTQ_EINVAL, "raftq_apply_vote_deltas: null argument");
  if (int rc = ensure_ingest(h, (size_t)n * sizeof(raftq_vote_delta_t))) return rc;
  if (int rc = enqueue_ingest<DeltaRec>(h, (const raftq_delta_t*)nullptr, 0, d, n, 0)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return check_deltas(h);
}

// ---- one sweep = validate flags -> fill SweepArgs -> launch -> per-handle bookkeeping.  The three host-side
// pieces are shared by raftq_step_async (one handle per launch) and raftq_set_sweep_async (a set per launch).
static int sweep_check(raftq_t* h, unsigned flags, const char* who) {
  const unsigned known = RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED | RAFTQ_SWEEP_VOTES | RAFTQ_SWEEP_NO_ADOPT |
                         RAFTQ_SWEEP_LDS | RAFTQ_SWEEP_CHANGED | RAFTQ_SWEEP_STREAM | RAFTQ_SWEEP_CACHED;
  const std::string w(who);
  if (flags & ~known) return fail(h, RAFTQ_EINVAL, w + ": unknown flag");
  const bool commit = flags & (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED);
  const bool votes = flags & RAFTQ_SWEEP_VOTES;
  if (!commit && !votes) return fail(h, RAFTQ_EINVAL, w + ": nothing to sweep");
  if ((flags & RAFTQ_SWEEP_GATED) && !h->have_terms)
    return fail(h, RAFTQ_ESTATE, w + ": gated sweep before raftq_load_terms");
  if ((flags & RAFTQ_SWEEP_STREAM) && (flags & RAFTQ_SWEEP_CACHED))
    return fail(h, RAFTQ_EINVAL, w + ": RAFTQ_SWEEP_STREAM and RAFTQ_SWEEP_CACHED are exclusive");
  if ((flags & RAFTQ_SWEEP_CHANGED) && !commit)
    return fail(h, RAFTQ_EINVAL, w + ": RAFTQ_SWEEP_CHANGED needs a commit sweep");
  return RAFTQ_OK;
}

static SweepArgs sweep_args(const raftq_t* h, int cur, bool want_bits) {
  SweepArgs a;
  a.match = h->match;
  a.committed = h->committed[cur];
  a.committed_out = h->committed[cur ^ 1];
  a.first_idx = h->first_idx;
  a.votes = h->votes;
  a.outcome = h->outcome;
  a.changed_bits = want_bits ? h->changed_bits : nullptr;
  a.partials = h->partials;
  a.ld = h->ld;
  return a;
}

static uint64_t sweep_footprint(const raftq_t* h) { return h->ld * (8ull * h->N + 24 + vote_word_bytes((int)h->N) + 1); }

// Streaming policy (profiles/r01/tune_policy_ld_vs_ldst.txt): loads always non-temporal; stores too only
// once the state outgrows ~128 MiB (2M x 7: -1.7 % with NT stores; 1M x 3/5/9: +1-2 % without them,
// and +9 % when the state is cache-resident, because the written commit indices are re-read next sweep)
static int sweep_policy(unsigned flags, uint64_t footprint) {
  const bool stream = (flags & RAFTQ_SWEEP_STREAM) ? true
                      : (flags & RAFTQ_SWEEP_CACHED) ? false
                                                     : footprint >= RAFTQ_AUTO_STREAM_BYTES;
  return !stream ? 0 : footprint >= RAFTQ_AUTO_STREAM_BYTES ? 3 : 1;
}

static void sweep_done(raftq_t* h, unsigned flags, int gpl) {
  const bool commit = flags & (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED);
  h->n_partials = h->gpad / ((uint64_t)kBlock * gpl) * kWaves;
  h->last_flags = flags;
  h->last_gpl = gpl;
  if (commit) {
    h->last_old = h->committed[h->cur];
    h->last_new = h->committed[h->cur ^ 1];
    if (!(flags & RAFTQ_SWEEP_NO_ADOPT)) {
      h->cur ^= 1;
      dense_changed(h);  // the live commit indices are the sweep's now
    }
  }
}

// one sweep of `h`, enqueued on `s` (the handle's own stream, or the auxiliary stream of raftq_sweep_many_async)
static int sweep_on(raftq_t* h, unsigned flags, hipStream_t s, const char* who) {
  if (int rc = use_device_idle(h, who)) return rc;
  if (int rc = sweep_check(h, flags, who)) return rc;
  const bool commit = flags & (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED);
  const SweepArgs a = sweep_args(h, h->cur, flags & RAFTQ_SWEEP_CHANGED);
  const bool lds = (flags & RAFTQ_SWEEP_LDS) && commit;
  SweepLaunch L;
  L.one = &a;
  L.gpad = h->gpad;
  HIPCHK(h, launch_sweep(h->N, L, flags, sweep_policy(flags, sweep_footprint(h)), s));
  sweep_done(h, flags, lds ? kLdsGPL : kGPL);
  return RAFTQ_OK;
}

int raftq_step_async(raftq_t* h, unsigned flags) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  return sweep_on(h, flags, h->stream, "raftq_step_async");
}

// K launches of K handles.  Handles on streams of their own already overlap their launch boundaries (a launch's ramp
// with its predecessor's drain).  Handles that SHARE a stream -- the members of a sweep set -- serialise there: one
// 1M x 5 launch at a time reads at 0.55-0.58 of the HBM peak.  Those launches are alternated between the shared stream
// and one auxiliary stream (fork once, join once): 11.97 -> 10.58 us per launch, read 0.547 -> 0.619 of 8 TB/s
// (profiles/r04/single_launch_ab.jsonl; four streams, 512-group tiles and hipGraph replay are all within noise of, or
// slower than, the plain loop).  Only for pairwise distinct handles: the same handle twice is a dependency chain.
// RAFTQ_MANY_STREAMS=1 keeps everything on the one stream (A/B).
static bool many_alternates() {
  static const bool on = [] {
    const char* e = std::getenv("RAFTQ_MANY_STREAMS");
    return !(e && e[0] == '1' && e[1] == 0);
  }();
  return on;
}

int raftq_sweep_many_async(raftq_t* const* handles, uint32_t n, unsigned flags) {
  if (!handles && n) return fail(nullptr, RAFTQ_EINVAL, "raftq_sweep_many_async: null handle array");
  bool alternate = n >= 2 && many_alternates();
  for (uint32_t i = 0; alternate && i < n; ++i) {
    if (!handles[i] || handles[i]->stream != handles[0]->stream || handles[i]->device != handles[0]->device) alternate = false;
    for (uint32_t j = 0; alternate && j < i; ++j)
      if (handles[j] == handles[i]) alternate = false;
  }
  if (!alternate) {
    for (uint32_t i = 0; i < n; ++i)
      if (int rc = raftq_step_async(handles[i], flags)) return rc;
    return RAFTQ_OK;
  }
  raftq_t* h0 = handles[0];
  if (int rc = use_device(h0)) return rc;
  if (!h0->many_aux) {
    HIPCHK(h0, hipStreamCreateWithFlags(&h0->many_aux, hipStreamNonBlocking));
    HIPCHK(h0, hipEventCreateWithFlags(&h0->many_fork, hipEventDisableTiming));
    HIPCHK(h0, hipEventCreateWithFlags(&h0->many_join, hipEventDisableTiming));
  }
  HIPCHK(h0, hipEventRecord(h0->many_fork, h0->stream));
  HIPCHK(h0, hipStreamWaitEvent(h0->many_aux, h0->many_fork, 0));
  int rc = RAFTQ_OK;
  for (uint32_t i = 0; i < n && rc == RAFTQ_OK; ++i)
    rc = sweep_on(handles[i], flags, (i & 1) ? h0->many_aux : h0->stream, "raftq_sweep_many_async");
  // joined whatever happened: nothing of this call stays behind on the auxiliary stream
  HIPCHK(h0, hipEventRecord(h0->many_join, h0->many_aux));
  HIPCHK(h0, hipStreamWaitEvent(h0->stream, h0->many_join, 0));
  return rc;
}

int raftq_wait(raftq_t* h, raftq_counts_t* counts) {
  if (int rc = use_device_idle(h, "raftq_wait")) return rc;
  if (counts) {
    if (h->n_partials == 0) return fail(h, RAFTQ_ESTATE, "raftq_wait: no sweep to report on");
    HIPCHK(h, hipMemcpyAsync(h->h_partials, h->partials, h->n_partials * sizeof(uint4), hipMemcpyDeviceToHost, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (counts) {
    uint64_t c = 0, w = 0, l = 0;
    for (uint64_t i = 0; i < h->n_partials; ++i) {
      c += h->h_partials[i].x;
      w += h->h_partials[i].y;
      l += h->h_partials[i].z;
    }
    const bool commit = h->last_flags & (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED);
    const bool votes = h->last_flags & RAFTQ_SWEEP_VOTES;
    counts->n_changed = commit ? c : 0;
    counts->n_won = votes ? w : 0;
    counts->n_lost = votes ? l : 0;
  }
  return RAFTQ_OK;
}

int raftq_read_committed(raftq_t* h, uint64_t* out) {
  if (int rc = use_device_idle(h, "raftq_read_committed")) return rc;
  if (!out) return fail(h, RAFTQ_EINVAL, "raftq_read_committed: null argument");
  // after a NO_ADOPT sweep the caller wants the evaluated (shadow) values
  const uint64_t* src = (h->last_flags & RAFTQ_SWEEP_NO_ADOPT) && h->last_new ? h->last_new : h->committed[h->cur];
  HIPCHK(h, hipMemcpyAsync(out, src, h->G * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RAFTQ_OK;
}

int raftq_read_outcome(raftq_t* h, uint8_t* out) {
  if (int rc = use_device_idle(h, "raftq_read_outcome")) return rc;
  if (!out) return fail(h, RAFTQ_EINVAL, "raftq_read_outcome: null argument");
  std::vector<uint8_t> packed;
  try {
    packed.resize((h->G + 3) / 4);
  } catch (...) {
    return fail(h, RAFTQ_ENOMEM, "raftq_read_outcome: host allocation failed");
  }
  HIPCHK(h, hipMemcpyAsync(packed.data(), h->outcome, packed.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (uint64_t g = 0; g < h->G; ++g) out[g] = (uint8_t)((packed[g >> 2] >> (2 * (g & 3))) & 3u);  // 2 bits per group
  return RAFTQ_OK;
}

int raftq_read_match(raftq_t* h, uint64_t* out) {
  if (int rc = use_device_idle(h, "raftq_read_match")) return rc;
  if (!out) return fail(h, RAFTQ_EINVAL, "raftq_read_match: null argument");
  for (uint32_t p = 0; p < h->N; ++p)  // row by row, as raftq_load_match
    HIPCHK(h, hipMemcpyAsync(out + (size_t)p * h->G, h->match + (size_t)p * h->ld, h->G * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RAFTQ_OK;
}

int raftq_read_votes(raftq_t* h, uint8_t* out) {
  if (int rc = use_device_idle(h, "raftq_read_votes")) return rc;
  if (!out) return fail(h, RAFTQ_EINVAL, "raftq_read_votes: null argument");
  const size_t wb = (size_t)vote_word_bytes((int)h->N);
  std::vector<uint8_t> packed;
  try {
    packed.resize(h->G * wb);
  } catch (...) {
    return fail(h, RAFTQ_ENOMEM, "raftq_read_votes: host allocation failed");
  }
  HIPCHK(h, hipMemcpyAsync(packed.data(), h->votes, packed.size(), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (wb == 2) unpack_votes(reinterpret_cast<const uint16_t*>(packed.data()), h->G, h->N, out);
  else unpack_votes(reinterpret_cast<const uint32_t*>(packed.data()), h->G, h->N, out);
  return RAFTQ_OK;
}

int raftq_commit_advance(raftq_t* h, int gated, uint64_t* committed_out, uint64_t* n_changed) {
  if (int rc = raftq_step_async(h, gated ? (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED) : RAFTQ_SWEEP_COMMIT)) return rc;
  raftq_counts_t c;
  if (int rc = raftq_wait(h, &c)) return rc;
  if (n_changed) *n_changed = c.n_changed;
  if (committed_out) return raftq_read_committed(h, committed_out);
  return RAFTQ_OK;
}

int raftq_vote_tally(raftq_t* h, uint8_t* outcome_out, raftq_counts_t* counts) {
  if (int rc = raftq_step_async(h, RAFTQ_SWEEP_VOTES)) return rc;
  raftq_counts_t c;
  if (int rc = raftq_wait(h, &c)) return rc;
  if (counts) *counts = c;
  if (outcome_out) return raftq_read_outcome(h, outcome_out);
  return RAFTQ_OK;
}

}  // extern "C"

int raftq_detail::ensure_tick_state(raftq_t* h) {
  if (h->role) return RAFTQ_OK;
  auto alloc = [&](void** p, size_t bytes) -> int {
    HIPCHK(h, hipMalloc(p, bytes));
    HIPCHK(h, hipMemsetAsync(*p, 0, bytes, h->stream));
    return RAFTQ_OK;
  };
  if (int rc = alloc((void**)&h->role, h->ld)) return rc;
  if (int rc = alloc((void**)&h->elapsed, h->ld * 4)) return rc;
  if (int rc = alloc((void**)&h->action, h->ld)) return rc;
  if (int rc = alloc((void**)&h->hup_bits, h->gpad / 8)) return rc;
  if (int rc = alloc((void**)&h->beat_bits, h->gpad / 8)) return rc;
  if (int rc = alloc((void**)&h->tick_partials, h->gpad / 256 * sizeof(uint4))) return rc;
  return RAFTQ_OK;
}

static int ensure_tick_offsets2(raftq_t* h, uint64_t nw);
static int flag_mode();
static hipError_t wait_turn(raftq_t* h, uint64_t flag_epoch, int kind = 0);
static int flag_mode();
static int ensure_arrive(raftq_t* h);

extern "C" {

int raftq_set_timers(raftq_t* h, uint32_t election_tick, uint32_t heartbeat_tick, uint64_t seed) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (election_tick == 0 || heartbeat_tick == 0)
    return fail(h, RAFTQ_EINVAL, "raftq_set_timers: ticks must be >= 1");
  h->election_tick = election_tick;
  h->heartbeat_tick = heartbeat_tick;
  h->tick_seed = seed;
  return RAFTQ_OK;
}

int raftq_load_roles(raftq_t* h, const uint8_t* role, const uint32_t* elapsed) {
  if (int rc = use_device_idle(h, "raftq_load_roles")) return rc;
  if (!role) return fail(h, RAFTQ_EINVAL, "raftq_load_roles: null role array");
  for (uint64_t g = 0; g < h->G; ++g)
    if (role[g] > RAFTQ_ROLE_LEADER) return fail(h, RAFTQ_EINVAL, "raftq_load_roles: role must be 0, 1 or 2");
  if (int rc = ensure_tick_state(h)) return rc;
  HIPCHK(h, hipMemcpyAsync(h->role, role, h->G, hipMemcpyHostToDevice, h->stream));
  dense_changed(h);
  if (elapsed) HIPCHK(h, hipMemcpyAsync(h->elapsed, elapsed, h->G * 4, hipMemcpyHostToDevice, h->stream));
  else HIPCHK(h, hipMemsetAsync(h->elapsed, 0, h->ld * 4, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RAFTQ_OK;
}

static TickArgs tick_args(const raftq_t* h, uint64_t tick_no) {
  TickArgs a;
  a.role = h->role;
  a.elapsed = h->elapsed;
  a.action = h->action;
  a.hup_bits = h->hup_bits;
  a.beat_bits = h->beat_bits;
  a.partials = h->tick_partials;
  a.n_groups = h->G;
  a.seed = h->tick_seed;
  a.tick_no = tick_no;
  a.election_tick = h->election_tick;
  a.heartbeat_tick = h->heartbeat_tick;
  a.et_magic = h->election_tick == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / h->election_tick);
  return a;
}

int raftq_tick(raftq_t* h, raftq_tick_counts_t* counts) {
  if (int rc = use_device_idle(h, "raftq_tick")) return rc;
  if (int rc = ensure_tick_state(h)) return rc;
  h->tl_valid = false;  // (raftq_last_tick_lists must not hand out an older tick's lists as the last one's)
  hipLaunchKernelGGL(tick_kernel, dim3((unsigned)(h->gpad / 1024)), dim3(kBlock), 0, h->stream, tick_args(h, h->tick_no++));
  HIPCHK(h, hipGetLastError());
  h->ticked = true;
  if (counts) {
    const uint64_t nw = h->gpad / 256;
    HIPCHK(h, hipMemcpyAsync(h->h_partials, h->tick_partials, nw * sizeof(uint4), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    uint64_t hup = 0, beat = 0;
    for (uint64_t i = 0; i < nw; ++i) {
      hup += h->h_partials[i].x;
      beat += h->h_partials[i].y;
    }
    counts->n_hup = hup;
    counts->n_beat = beat;
  }
  return RAFTQ_OK;
}

// Tick + both lists: two launches and ONE wait (raftq_tick, raftq_collect_hups and raftq_collect_beats are five launches
// and two waits).  The lists land in the pinned advance buffer (8 bytes of its 24-byte entries each) and are copied out.
int raftq_tick_collect(raftq_t* h, uint64_t* hups, uint64_t hup_cap, uint64_t* n_hup, uint64_t* beats, uint64_t beat_cap, uint64_t* n_beat) {
  if (int rc = use_device_idle(h, "raftq_tick_collect")) return rc;
  if (!n_hup || !n_beat) return fail(h, RAFTQ_EINVAL, "raftq_tick_collect: null count");
  if ((hup_cap && !hups) || (beat_cap && !beats)) return fail(h, RAFTQ_EINVAL, "raftq_tick_collect: null list with cap > 0");
  if (int rc = ensure_tick_state(h)) return rc;
  const uint64_t cap_h = std::min<uint64_t>(hup_cap, h->G), cap_b = std::min<uint64_t>(beat_cap, h->G);
  if (int rc = ensure_adv(h, (cap_h + cap_b + 2) / 3 + 2)) return rc;
  const uint64_t nw = h->gpad / 256;
  if (nw > (1u << 14))
    if (int rc = ensure_tick_offsets2(h, nw)) return rc;  // (before the Tick: a call that fails has not ticked)
  h->tl_valid = false;  // raftq_last_tick_lists hands out the lists of the LAST tick, and this is a newer one
  hipLaunchKernelGGL(tick_kernel, dim3((unsigned)(h->gpad / 1024)), dim3(kBlock), 0, h->stream, tick_args(h, h->tick_no++));
  h->ticked = true;
  const uint64_t *off_h = nullptr, *off_b = nullptr;
  if (nw > (1u << 14)) {  // past 16K waves every workgroup summing its predecessors itself would show: scan first
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(1024), 0, h->stream, h->tick_partials, nw, h->offsets, h->d_total, 0);
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(1024), 0, h->stream, h->tick_partials, nw, h->tick_offsets2, h->d_total + 1, 1);
    off_h = h->offsets;
    off_b = h->tick_offsets2;
  }
  uint64_t* list_d = (uint64_t*)h->adv_d;
  hipLaunchKernelGGL(tick_lists_kernel, dim3((unsigned)(h->gpad / 1024)), dim3(kBlock), 0, h->stream, (const uint64_t*)h->hup_bits,
                     (const uint64_t*)h->beat_bits, (const uint4*)h->tick_partials, list_d, cap_h, list_d + cap_h, cap_b, h->d_total, off_h, off_b);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  *n_hup = h->h_total[0];
  *n_beat = h->h_total[1];
  const uint64_t* list_h = (const uint64_t*)h->adv_h;
  if (const uint64_t take = std::min(*n_hup, cap_h)) std::memcpy(hups, list_h, take * 8);
  if (const uint64_t take = std::min(*n_beat, cap_b)) std::memcpy(beats, list_h + cap_h, take * 8);
  h->adv_listed = 0;
  return RAFTQ_OK;
}

// Tick + both lists as they are meant to be read: 4-byte group ids LEFT IN PLACE in page-locked memory
// (raftq_last_tick_lists: no copy into caller arrays -- 110 of the 172 us raftq_tick_collect took at 1M groups / 400K entries),
// the MsgBeat groups as a group-order bitmap when the caller asks for it (RAFTQ_TICK_BEAT_BITMAP), and the wait on the turn's
// completion word instead of a stream synchronisation.  Three launches (tick, lists, flag), one wait.
int raftq_tick_collect_lists(raftq_t* h, unsigned flags, uint64_t hup_cap, uint64_t beat_cap, uint64_t* n_hup, uint64_t* n_beat) {
  if (int rc = use_device_idle(h, "raftq_tick_collect_lists")) return rc;
  if (!n_hup || !n_beat) return fail(h, RAFTQ_EINVAL, "raftq_tick_collect_lists: null count");
  if (flags & ~RAFTQ_TICK_BEAT_BITMAP) return fail(h, RAFTQ_EINVAL, "raftq_tick_collect_lists: unknown flag");
  if (int rc = ensure_tick_state(h)) return rc;
  const bool bitmap = flags & RAFTQ_TICK_BEAT_BITMAP;
  const uint64_t cap_h = std::min<uint64_t>(hup_cap, h->G), cap_b = bitmap ? 0 : std::min<uint64_t>(beat_cap, h->G);
  const uint64_t beat_at = (cap_h + 3) & ~3ull;  // the MsgBeat ids start on a 16-byte boundary (the lists leave in whole quads)
  const uint64_t map_off = ((beat_at + cap_b) * 4 + 255) / 256 * 256;
  const uint64_t need = map_off + h->gpad / 8 + 256;
  h->tl_valid = false;
  if (need > h->tl_bytes) {
    if (h->tl_h) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      HIPCHK(h, hipHostFree(h->tl_h));
      h->tl_h = h->tl_d = nullptr;
      h->tl_bytes = 0;
    }
    const uint64_t want = std::max<uint64_t>(need, 1 << 16);
    HIPCHK(h, hipHostMalloc((void**)&h->tl_h, want, hipHostMallocMapped | host_coherence_flag()));
    HIPCHK(h, hipHostGetDevicePointer((void**)&h->tl_d, h->tl_h, 0));
    h->tl_bytes = want;
  }
  // every allocation is made BEFORE the Tick is enqueued: a call that fails has not ticked (an allocation failure
  // behind the tick kernel left the timers advanced and the MsgHup / MsgBeat groups of that tick unreported)
  const uint64_t nw = h->gpad / 256;
  if (nw > (1u << 14))
    if (int rc = ensure_tick_offsets2(h, nw)) return rc;
  hipLaunchKernelGGL(tick_kernel, dim3((unsigned)(h->gpad / 1024)), dim3(kBlock), 0, h->stream, tick_args(h, h->tick_no++));
  h->ticked = true;
  const uint64_t *off_h = nullptr, *off_b = nullptr;
  if (nw > (1u << 14)) {  // past 16K waves every workgroup summing its predecessors itself would show: scan first
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(1024), 0, h->stream, h->tick_partials, nw, h->offsets, h->d_total, 0);
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(1024), 0, h->stream, h->tick_partials, nw, h->tick_offsets2, h->d_total + 1, 1);
    off_h = h->offsets;
    off_b = h->tick_offsets2;
  }
  uint32_t* const hup_d = h->tl_d;
  uint32_t* const beat_d = h->tl_d + beat_at;
  uint64_t* const map_d = (uint64_t*)((uint8_t*)h->tl_d + map_off);
  // blocks of 1,024 groups per workgroup of the lists kernel: 1.  Four (runs of ~1,600 ids instead of ~400 at the bench's density,
  // RAFTQ_TICK_LISTS_BPW=4, read per call) measured no better: 47.1 against 46.5 us a call -- 1.6 MB of ids take the link 34 us
  // however they are cut
  int bpw = 1;
  if (const char* e = std::getenv("RAFTQ_TICK_LISTS_BPW")) bpw = std::atoi(e) == 4 ? 4 : 1;
  const uint64_t per_wg = (uint64_t)kWaves * bpw;
  const dim3 grid((unsigned)((nw + per_wg - 1) / per_wg));
  // the completion word: a one-thread kernel behind the lists (only a kernel boundary orders eight XCDs' stores to host memory
  // before it -- raftq_cycle's finding), polled by the host; RAFTQ_CYCLE_FLAG=arrive: the lists kernel's last workgroup to arrive
  // raises it itself, =packet: the runtime's write-value packet; the blocking wait where the word cannot be had
  uint64_t epoch = 0;
  Arrival ar{nullptr, nullptr, 0};
  if (h->stream_write_ok) {
    epoch = ++h->compact_epoch;
    if ((uint32_t)epoch == 0) epoch = ++h->compact_epoch;
    h->flag_mask = ~0ull;
    if (flag_mode() == 3) {
      if (int rc = ensure_arrive(h)) return rc;
      epoch &= 0xffffffffull;  // (the arrival form's word carries 32 bits of epoch)
      ar = Arrival{(unsigned long long*)(h->d_total + 3), h->arrive_count, (uint32_t)epoch};
    }
  }
#define RAFTQ_TICK_LISTS_LAUNCH(BM, B)                                                                                                       \
  hipLaunchKernelGGL((tick_lists32_kernel<BM, B>), grid, dim3(kBlock), 0, h->stream, (const uint64_t*)h->hup_bits, (const uint64_t*)h->beat_bits, \
                     (const uint4*)h->tick_partials, nw, hup_d, cap_h, beat_d, cap_b, map_d, h->d_total, off_h, off_b, ar)
  if (bitmap) {
    if (bpw == 1) RAFTQ_TICK_LISTS_LAUNCH(true, 1);
    else RAFTQ_TICK_LISTS_LAUNCH(true, 4);
  } else {
    if (bpw == 1) RAFTQ_TICK_LISTS_LAUNCH(false, 1);
    else RAFTQ_TICK_LISTS_LAUNCH(false, 4);
  }
#undef RAFTQ_TICK_LISTS_LAUNCH
  HIPCHK(h, hipGetLastError());
  if (epoch && flag_mode() == 1) {
    hipLaunchKernelGGL(raise_flag_kernel, dim3(1), dim3(64), 0, h->stream, h->d_total + 3, epoch);
    HIPCHK(h, hipGetLastError());
  } else if (epoch && flag_mode() == 2 && hipStreamWriteValue64(h->stream, (void*)(h->d_total + 3), epoch, 0) != hipSuccess) {
    (void)hipGetLastError();
    epoch = 0;  // not supported here: the blocking wait
  }
  HIPCHK(h, wait_turn(h, epoch, 1));
  if (epoch) {
    // RAFTQ_CYCLE_CHECK=1 (tests, soaks): what the host reads when the word lands -- the two totals, the ids, the bitmap -- must be
    // what it reads after a full stream synchronisation
    static const bool check = [] { const char* e = std::getenv("RAFTQ_CYCLE_CHECK"); return e && e[0] == '1'; }();
    if (check) {
      auto digest = [&]() {
        const volatile uint64_t* tot = (const volatile uint64_t*)h->h_total;
        const uint64_t t0 = tot[0], t1 = tot[1];
        uint64_t d = (t0 * 0x9E3779B97F4A7C15ull) ^ t1;
        const volatile uint32_t* ids = (const volatile uint32_t*)h->tl_h;
        const uint64_t nh = std::min<uint64_t>(t0, cap_h), nb = std::min<uint64_t>(t1, cap_b);
        for (uint64_t i = 0; i < nh; ++i) d = (d ^ ids[i]) * 0xBF58476D1CE4E5B9ull;
        for (uint64_t i = 0; i < nb; ++i) d = (d ^ ids[beat_at + i]) * 0xBF58476D1CE4E5B9ull;
        if (bitmap) {
          const volatile uint64_t* mw = (const volatile uint64_t*)((const uint8_t*)h->tl_h + map_off);
          for (uint64_t i = 0; i < h->gpad / 64; ++i) d = (d ^ mw[i]) * 0xBF58476D1CE4E5B9ull;
        }
        return d;
      };
      const uint64_t at_flag = digest();
      HIPCHK(h, hipStreamSynchronize(h->stream));
      if (digest() != at_flag)
        return fail(h, RAFTQ_EHIP, "raftq_tick_collect_lists: the completion word arrived before the lists it announces (RAFTQ_CYCLE_CHECK)");
    }
  }
  *n_hup = h->tl_n_hup = h->h_total[0];
  *n_beat = h->tl_n_beat = h->h_total[1];
  h->tl_hup_cap = cap_h;
  h->tl_beat_cap = cap_b;
  h->tl_beat_at = beat_at;
  h->tl_map_off = map_off;
  h->tl_flags = flags;
  h->tl_valid = true;
  return RAFTQ_OK;
}

int raftq_last_tick_lists(raftq_t* h, const uint32_t** hups, uint64_t* n_hups, const uint32_t** beats, uint64_t* n_beats,
                          const uint64_t** beat_bitmap, uint64_t* bitmap_words) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!h->tl_valid) return fail(h, RAFTQ_ESTATE, "raftq_last_tick_lists: no raftq_tick_collect_lists before it");
  const bool bitmap = h->tl_flags & RAFTQ_TICK_BEAT_BITMAP;
  if (hups) *hups = h->tl_h;
  if (n_hups) *n_hups = std::min(h->tl_n_hup, h->tl_hup_cap);
  if (beats) *beats = bitmap ? nullptr : h->tl_h + h->tl_beat_at;
  if (n_beats) *n_beats = bitmap ? 0 : std::min(h->tl_n_beat, h->tl_beat_cap);
  if (beat_bitmap) *beat_bitmap = bitmap ? (const uint64_t*)((const uint8_t*)h->tl_h + h->tl_map_off) : nullptr;
  if (bitmap_words) *bitmap_words = bitmap ? (h->G + 63) / 64 : 0;
  return RAFTQ_OK;
}

}  // extern "C"
static int ensure_tick_offsets2(raftq_t* h, uint64_t nw) {
  if (h->tick_offsets2) return RAFTQ_OK;
  (void)nw;
  HIPCHK(h, hipMalloc((void**)&h->tick_offsets2, (h->gpad / 256 + 1) * 8));
  return RAFTQ_OK;
}
extern "C" {

int raftq_read_tick(raftq_t* h, uint8_t* action, uint32_t* elapsed, uint8_t* role) {
  if (int rc = use_device_idle(h, "raftq_read_tick")) return rc;
  if (!h->role) return fail(h, RAFTQ_ESTATE, "raftq_read_tick: no tick state (raftq_load_roles / raftq_tick first)");
  if (action) HIPCHK(h, hipMemcpyAsync(action, h->action, h->G, hipMemcpyDeviceToHost, h->stream));
  if (elapsed) HIPCHK(h, hipMemcpyAsync(elapsed, h->elapsed, h->G * 4, hipMemcpyDeviceToHost, h->stream));
  if (role) HIPCHK(h, hipMemcpyAsync(role, h->role, h->G, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RAFTQ_OK;
}

// ascending list of the groups the last raftq_tick flagged: field 0 = MsgHup, 1 = MsgBeat
static int collect_tick_list(raftq_t* h, const char* who, int field, uint64_t* groups, uint64_t cap, uint64_t* n) {
  if (int rc = use_device_idle(h, who)) return rc;
  if (!n) return fail(h, RAFTQ_EINVAL, std::string(who) + ": null count");
  if (!h->ticked) return fail(h, RAFTQ_ESTATE, std::string(who) + ": no raftq_tick yet");
  if (cap && !groups) return fail(h, RAFTQ_EINVAL, std::string(who) + ": null out with cap > 0");
  const uint64_t take_cap = std::min<uint64_t>(cap, h->G);
  // the advance buffer doubles as the (smaller) group-id list: 8 B of every 24
  if (take_cap)
    if (int rc = ensure_adv(h, (take_cap + 2) / 3 + 1)) return rc;
  const uint64_t nw = h->gpad / 256;
  hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(1024), 0, h->stream, h->tick_partials, nw, h->offsets,
                     h->d_total, field);
  if (take_cap)
    hipLaunchKernelGGL(compact_hups_kernel, dim3((unsigned)(h->gpad / 1024)), dim3(kBlock), 0, h->stream,
                       field ? h->beat_bits : h->hup_bits, h->offsets, (uint64_t*)h->adv_d, take_cap);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const uint64_t total = *h->h_total;
  *n = total;
  const uint64_t take = std::min(total, take_cap);
  if (take) std::memcpy(groups, h->adv_h, take * 8);
  h->adv_listed = 0;
  return RAFTQ_OK;
}

int raftq_collect_hups(raftq_t* h, uint64_t* groups, uint64_t cap, uint64_t* n) {
  return collect_tick_list(h, "raftq_collect_hups", 0, groups, cap, n);
}

int raftq_collect_beats(raftq_t* h, uint64_t* groups, uint64_t cap, uint64_t* n) {
  return collect_tick_list(h, "raftq_collect_beats", 1, groups, cap, n);
}

int raftq_campaign(raftq_t* h, const uint64_t* groups, uint64_t n, uint32_t self_peer) {
  if (int rc = use_device_idle(h, "raftq_campaign")) return rc;
  if (n == 0) return RAFTQ_OK;
  if (!groups) return fail(h, RAFTQ_EINVAL, "raftq_campaign: null argument");
  if (self_peer >= h->N) return fail(h, RAFTQ_EINVAL, "raftq_campaign: self_peer out of range");
  uint64_t bad = 0;
  for (uint64_t i = 0; i < n; ++i) bad |= (uint64_t)(groups[i] >= h->G);
  if (bad) return fail(h, RAFTQ_EINVAL, "raftq_campaign: a group is out of range; nothing applied");
  if (int rc = ensure_tick_state(h)) return rc;
  if (int rc = ensure_staging(h, (size_t)n * 8)) return rc;
  std::memcpy(h->stage_h, groups, n * 8);
  hipLaunchKernelGGL(campaign_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream,
                     h->role, h->elapsed, h->votes, h->N, self_peer, (const uint64_t*)h->stage_d, n);
  dense_changed(h);  // (role)
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return RAFTQ_OK;
}

// enqueue scan + compaction of the last RAFTQ_SWEEP_CHANGED sweep; the kernels
// write the total and up to `take_cap` entries straight into pinned host memory.
}  // extern "C" (templates need C++ linkage)

// scan-free compaction of the last RAFTQ_SWEEP_CHANGED sweep (compact_changed_kernel computes its own offsets):
// total and up to `take_cap` entries go straight into pinned host memory.  want_flag: also publish a completion
// flag the host can poll (its value is left in h->compact_epoch_armed): a one-thread kernel of ours behind the
// compaction (default), or the runtime's stream write-value packet (RAFTQ_CYCLE_FLAG=packet: a 3.4 us kernel that
// starts 5 us after the compaction ends, profiles/r03/cycle_kernel_trace_before.txt).
static int flag_mode() {  // 1 = raise_flag_kernel, 2 = write-value packet, 3 = the last workgroup to arrive raises it (where a kernel can: the segmented turn, the tick lists)
  static const int m = [] {
    const char* e = std::getenv("RAFTQ_CYCLE_FLAG");
    return e && std::strcmp(e, "packet") == 0 ? 2 : e && std::strcmp(e, "arrive") == 0 ? 3 : 1;
  }();
  return m;
}
// the arrival counter of mode 3: a device word of the handle, zero between kernels
static int ensure_arrive(raftq_t* h) {
  if (h->arrive_count) return RAFTQ_OK;
  HIPCHK(h, hipMalloc((void**)&h->arrive_count, 64));
  HIPCHK(h, hipMemsetAsync(h->arrive_count, 0, 64, h->stream));
  return RAFTQ_OK;
}

template <typename Adv>
static int enqueue_collect(raftq_t* h, uint64_t take_cap, bool want_flag) {
  static_assert(sizeof(Advance) == sizeof(raftq_advance_t) && sizeof(Advance16) == sizeof(raftq_advance16_t), "ABI struct mismatch");
  const int gpl = h->last_gpl;
  const dim3 grid((unsigned)(h->gpad / ((uint64_t)kBlock * gpl)));
  Adv* out = (Adv*)h->adv_d;  // the pinned list holds either layout (sized for the larger one)
  // Up to 16K waves (4M groups at 256 per wave) every compaction workgroup sums its predecessors' counts itself;
  // beyond that the quadratic total would show, so the one-workgroup scan runs first and hands in the offsets.
  const uint64_t* wave_offsets = nullptr;
  if (h->n_partials > (1u << 14)) {
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(1024), 0, h->stream, h->partials, h->n_partials, h->offsets,
                       h->d_total, 0);
    wave_offsets = h->offsets;
  }
  // the compaction mirrors the geometry of the sweep that wrote the bitmap and the per-wave counts
  switch (gpl) {
    case 2:
      hipLaunchKernelGGL((compact_changed_kernel<2, Adv>), grid, dim3(kBlock), 0, h->stream, h->changed_bits, h->partials,
                         h->last_old, h->last_new, out, take_cap, h->d_total, wave_offsets);
      break;
    case 8:
      hipLaunchKernelGGL((compact_changed_kernel<8, Adv>), grid, dim3(kBlock), 0, h->stream, h->changed_bits, h->partials,
                         h->last_old, h->last_new, out, take_cap, h->d_total, wave_offsets);
      break;
    default:
      static_assert((kGPL == 4 || kGPL == 8) && kLdsGPL == 4, "compaction instantiations cover GPL 2, 4, 8");
      hipLaunchKernelGGL((compact_changed_kernel<4, Adv>), grid, dim3(kBlock), 0, h->stream, h->changed_bits, h->partials,
                         h->last_old, h->last_new, out, take_cap, h->d_total, wave_offsets);
  }
  HIPCHK(h, hipGetLastError());
  h->compact_epoch_armed = 0;
  if (want_flag && h->stream_write_ok) {
    const uint64_t epoch = ++h->compact_epoch;
    if (flag_mode() == 1) {
      hipLaunchKernelGGL(raise_flag_kernel, dim3(1), dim3(64), 0, h->stream, h->d_total + 3, epoch);
      HIPCHK(h, hipGetLastError());
      h->compact_epoch_armed = epoch;
    } else if (hipStreamWriteValue64(h->stream, (void*)(h->d_total + 3), epoch, 0) == hipSuccess) {
      h->compact_epoch_armed = epoch;
    } else {
      h->stream_write_ok = false;  // not supported here: turns end in the blocking wait
    }
  }
  return RAFTQ_OK;
}

// RAFTQ_CYCLE_SEGMENTED: can this turn's sweep write the advance list itself (sweep_segments_kernel)?  A commit sweep (gated or
// not) with no vote tally and no A/B variant asked for, on a stack where the completion word reaches the host.
constexpr uint64_t kSegmentsMaxGroups = 1ull << 22;  // the pinned list has a slot per group: 16 B x 4M groups = 64 MB at most (larger handles: the contiguous list)
static bool segments_ok(const raftq_t* h, unsigned flags) {
  if (const char* e = std::getenv("RAFTQ_CYCLE_SEGMENTS"))  // =0: every turn produces the contiguous list (A/B; the tests' way to a consumer's fallback)
    if (e[0] == '0') return false;
  const unsigned allowed = RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED | RAFTQ_SWEEP_CHANGED | RAFTQ_SWEEP_NO_ADOPT | RAFTQ_SWEEP_STREAM | RAFTQ_SWEEP_CACHED;
  return (flags & (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED)) && !(flags & ~allowed) && h->gpad <= kSegmentsMaxGroups && h->stream_write_ok;
}
// sweep + list + flag, two kernels; leaves the handle as sweep_on + enqueue_collect would
static int enqueue_sweep_segments(raftq_t* h, unsigned flags) {
  const uint32_t n_tiles = (uint32_t)(h->gpad / kTile);
  if (int rc = ensure_adv(h, (h->gpad * sizeof(Advance16) + sizeof(Advance) - 1) / sizeof(Advance))) return rc;  // a segment per tile, a 16-byte slot per group
  if (h->seg_cap < n_tiles) {
    if (h->seg_h) {
      HIPCHK(h, hipStreamSynchronize(h->stream));
      HIPCHK(h, hipHostFree(h->seg_h));
      HIPCHK(h, hipFree(h->seg_d));
      h->seg_h = h->seg_hd = nullptr;
      h->seg_d = nullptr;
      h->seg_cap = 0;
    }
    HIPCHK(h, hipHostMalloc((void**)&h->seg_h, (size_t)n_tiles * 4, hipHostMallocMapped | host_coherence_flag()));
    HIPCHK(h, hipHostGetDevicePointer((void**)&h->seg_hd, h->seg_h, 0));
    HIPCHK(h, hipMalloc((void**)&h->seg_d, (size_t)n_tiles * 4));
    h->seg_cap = n_tiles;
  }
  const SweepArgs a = sweep_args(h, h->cur, true);
  uint64_t epoch = ++h->compact_epoch;
  if ((uint32_t)epoch == 0) epoch = ++h->compact_epoch;  // (the word's top half is never 0 for a turn in flight)
  Arrival ar{nullptr, nullptr, (uint32_t)epoch};
  if (flag_mode() == 3) {  // RAFTQ_CYCLE_FLAG=arrive: no flag kernel -- the sweep's last workgroup to arrive raises the word
    if (int rc = ensure_arrive(h)) return rc;
    ar.flag = (unsigned long long*)(h->d_total + 3);
    ar.count = h->arrive_count;
  }
  HIPCHK(h, launch_segments(h->N, a, n_tiles, (flags & RAFTQ_SWEEP_GATED) != 0, sweep_policy(flags, sweep_footprint(h)), (Advance16*)h->adv_d, h->seg_hd,
                            h->seg_d, h->stream, ar));
  sweep_done(h, flags, kGPL);
  if (ar.flag == nullptr) hipLaunchKernelGGL(raise_flag_segments_kernel, dim3(1), dim3(kBlock), 0, h->stream, h->d_total + 3, (uint32_t)epoch, h->seg_d, n_tiles);
  HIPCHK(h, hipGetLastError());
  h->compact_epoch_armed = (uint64_t)(uint32_t)epoch << 32;
  h->flag_mask = 0xffffffff00000000ull;
  h->seg_tiles = n_tiles;
  h->seg_stride = kTile;
  return RAFTQ_OK;
}

extern "C" {

int raftq_collect_changed(raftq_t* h, raftq_advance_t* out, uint64_t cap, uint64_t* n) {
  if (int rc = use_device_idle(h, "raftq_collect_changed")) return rc;
  if (!n) return fail(h, RAFTQ_EINVAL, "raftq_collect_changed: null count");
  if (!(h->last_flags & RAFTQ_SWEEP_CHANGED) || !h->last_old)
    return fail(h, RAFTQ_ESTATE, "raftq_collect_changed: last sweep did not set RAFTQ_SWEEP_CHANGED");
  if (cap && !out) return fail(h, RAFTQ_EINVAL, "raftq_collect_changed: null out with cap > 0");
  const uint64_t take_cap = std::min<uint64_t>(cap, h->G);
  if (take_cap)
    if (int rc = ensure_adv(h, take_cap)) return rc;
  if (int rc = enqueue_collect<Advance>(h, take_cap, false)) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  const uint64_t total = *h->h_total;
  *n = total;
  const uint64_t take = std::min(total, take_cap);
  h->adv_listed = take;
  h->adv_packed = false;
  h->adv_segmented = false;
  if (take) std::memcpy(out, h->adv_h, take * sizeof(Advance));
  return RAFTQ_OK;
}

static size_t vote_stage_offset(uint64_t n_deltas) {
  return ((size_t)n_deltas * sizeof(raftq_delta_t) + 255) / 256 * 256;
}

int raftq_stage(raftq_t* h, uint64_t n_deltas, uint64_t n_vote_deltas, raftq_delta_t** deltas,
                raftq_vote_delta_t** vote_deltas) {
  if (int rc = use_device_idle(h, "raftq_stage")) return rc;
  const size_t off_votes = vote_stage_offset(n_deltas);
  if (int rc = ensure_ingest(h, off_votes + (size_t)n_vote_deltas * sizeof(raftq_vote_delta_t) + 256)) return rc;
  if (deltas) *deltas = (raftq_delta_t*)h->ingest_h;
  if (vote_deltas) *vote_deltas = (raftq_vote_delta_t*)((uint8_t*)h->ingest_h + off_votes);
  return RAFTQ_OK;
}

int raftq_last_advances(raftq_t* h, const raftq_advance_t** list, uint64_t* n_listed) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!list || !n_listed) return fail(h, RAFTQ_EINVAL, "raftq_last_advances: null argument");
  if (h->adv_listed && h->adv_packed)
    return fail(h, RAFTQ_ESTATE, "raftq_last_advances: the last list was produced in the packed layout");
  *list = (const raftq_advance_t*)h->adv_h;
  *n_listed = h->adv_listed;
  return RAFTQ_OK;
}

}  // extern "C" (templates need C++ linkage)

// The end of a batching turn.  A turn is ~55 us of device work, so the wake-up of a blocking
// hipStreamSynchronize (interrupt + reschedule, ~8 us) is a visible slice of it.  When the turn ends in a
// compaction, a stream write-value packet behind that kernel stores a completion epoch into pinned host memory:
// the host polls the word -- bounded, then falls back to the blocking wait.
// (Polling hipStreamQuery instead was measured SLOWER than blocking: 88 vs 80 us per turn, profiles/r02.)
// RAFTQ_CYCLE_WAIT=block restores the plain blocking wait.
static hipError_t wait_turn(raftq_t* h, uint64_t flag_epoch, int kind) {
  static const bool poll = [] {
    const char* e = std::getenv("RAFTQ_CYCLE_WAIT");
    return !(e && std::strcmp(e, "block") == 0);
  }();
  // A miss (the flag did not land within 2 ms; a turn is ~50 us) is either a slow turn -- a huge handle, a contended
  // GPU, a profiler -- or a stack on which the flag never reaches the host.  One slow turn must not change the handle's
  // steady state, so the wake-up is only given up after kMisses misses IN A ROW, and then tried again every
  // kRetry turns; RAFTQ_PROFILE prints how many turns fell back to the blocking wait.
  constexpr uint32_t kMisses = 8, kRetry = 1024;
  const bool try_flag = poll && flag_epoch && (h->flag_misses[kind] < kMisses || (++h->flag_rested[kind] % kRetry) == 0);
  if (try_flag) {
    volatile uint64_t* flag = h->h_total + 3;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0;; ++i) {
      if ((*flag & h->flag_mask) == flag_epoch) {
        std::atomic_thread_fence(std::memory_order_acquire);
        h->flag_misses[kind] = 0;
        // the word says the stream's work is done; an asynchronous error of that work is only known to the runtime (ADVICE r05:
        // the poll returned success without asking, and the error surfaced in a later call).  Whatever this thread has on
        // record -- possibly a benign, older one -- is cleared and the stream itself is asked: its answer is the call's.
        if (hipPeekAtLastError() == hipSuccess) return hipSuccess;
        (void)hipGetLastError();
        return hipStreamSynchronize(h->stream);
      }
      if ((i & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) break;
    }
    ++h->flag_misses[kind];
  }
  if (flag_epoch) ++h->flag_fallbacks;
  return hipStreamSynchronize(h->stream);
}

hipError_t raftq_detail::wait_call(raftq_t* h) {
  static const bool block = [] {
    const char* e = std::getenv("RAFTQ_CALL_WAIT");
    return e && std::strcmp(e, "block") == 0;
  }();
  hipError_t e;
  if (block || !h->stream_write_ok || !h->d_total) {
    e = hipStreamSynchronize(h->stream);
  } else if (flag_mode() == 2) {  // RAFTQ_CYCLE_FLAG=packet (A/B): the runtime's write-value packet behind the call's kernels
    const uint64_t epoch = ++h->compact_epoch;
    h->flag_mask = ~0ull;
    e = hipStreamWriteValue64(h->stream, (void*)(h->d_total + 3), epoch, 0) != hipSuccess ? hipStreamSynchronize(h->stream) : wait_turn(h, epoch, 1);
  } else {
    const uint64_t epoch = ++h->compact_epoch;
    h->flag_mask = ~0ull;
    hipLaunchKernelGGL(raise_flag_kernel, dim3(1), dim3(64), 0, h->stream, h->d_total + 3, epoch);
    e = hipGetLastError() != hipSuccess ? hipStreamSynchronize(h->stream) : wait_turn(h, epoch, 1);
  }
  if (e == hipSuccess && h->wal_pending && !h->wal_pending_done) h->wal_pending_waited = true;  // the begun WAL encode was enqueued before this wait
  return e;
}

// One batching turn (raft.go:227-235 for every group at once): ingest -> sweep -> advance list, one wait.
// Rec / AbiRec: layout of the match deltas; Adv / AbiAdv: layout of the advance list.
template <typename Rec, typename Adv, typename AbiRec, typename AbiAdv>
static int cycle_impl(raftq_t* h, const char* who, const AbiRec* deltas, uint64_t n_deltas, const raftq_vote_delta_t* vote_deltas,
                      uint64_t n_vote_deltas, unsigned flags, AbiAdv* advances_out, uint64_t cap, uint64_t* n_advanced,
                      raftq_counts_t* counts) {
  if (int rc = use_device_idle(h, who)) return rc;
  if ((n_deltas && !deltas) || (n_vote_deltas && !vote_deltas))
    return fail(h, RAFTQ_EINVAL, std::string(who) + ": null array with non-zero length");
  const bool want_segments = (flags & RAFTQ_CYCLE_SEGMENTED) != 0;
  flags &= ~RAFTQ_CYCLE_SEGMENTED;
  if (int rc = sweep_check(h, flags & ~RAFTQ_CYCLE_TRUSTED, who)) return rc;  // before anything is enqueued: a refused turn applies nothing
  const bool commit = flags & (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED);
  const bool want_list = commit && (advances_out || n_advanced || cap);
  if (want_list) flags |= RAFTQ_SWEEP_CHANGED;
  const size_t off_votes = ((size_t)n_deltas * sizeof(AbiRec) + 255) / 256 * 256;
  {
    const size_t need = off_votes + (size_t)n_vote_deltas * sizeof(raftq_vote_delta_t) + 256;
    auto staged = [&](const void* q) {
      return q && h->ingest_h && (const uint8_t*)q >= (const uint8_t*)h->ingest_h &&
             (const uint8_t*)q < (const uint8_t*)h->ingest_h + h->ingest_bytes;
    };
    if (need > h->ingest_bytes && (staged(deltas) || staged(vote_deltas)))  // growing would free what the caller points into
      return fail(h, RAFTQ_EINVAL, std::string(who) + ": the staged arrays were sized for fewer records than this call names");
    if (int rc = ensure_ingest(h, need)) return rc;
  }
  const uint64_t take_cap = want_list ? std::min<uint64_t>(cap, h->G) : 0;
  if (take_cap)
    if (int rc = ensure_adv(h, take_cap)) return rc;
  // everything below is enqueued back to back on the handle's stream; one wait at the end
  using clk = std::chrono::steady_clock;
  auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  const auto t0 = clk::now();
  const bool trusted = flags & RAFTQ_CYCLE_TRUSTED;
  flags &= ~RAFTQ_CYCLE_TRUSTED;
  if (n_deltas || n_vote_deltas)
    if (int rc = enqueue_ingest<Rec>(h, deltas, n_deltas, vote_deltas, n_vote_deltas, off_votes, trusted)) return rc;
  const auto t1 = clk::now();
  const int cur_before = h->cur;
  const bool flag_wake = want_list && !counts;  // the compaction is the last thing on the stream: it can say "done"
  // RAFTQ_CYCLE_SEGMENTED: the sweep writes the list itself, a segment per tile (two kernels and the flag instead of four) --
  // whenever the turn can take that form; otherwise the contiguous list is presented as one segment
  const bool segmented = want_segments && flag_wake && !advances_out && sizeof(Adv) == sizeof(Advance16) && segments_ok(h, flags);
  h->flag_mask = ~0ull;
  h->adv_segmented = false;
  if (segmented) {
    if (int rc = enqueue_sweep_segments(h, flags | RAFTQ_SWEEP_CHANGED)) return rc;
  } else if (int rc = raftq_step_async(h, flags)) {
    return rc;
  }
  const auto t2 = clk::now();
  if (want_list && !segmented)
    if (int rc = enqueue_collect<Adv>(h, take_cap, flag_wake)) return rc;
  if (counts)
    HIPCHK(h, hipMemcpyAsync(h->h_partials, h->partials, h->n_partials * sizeof(uint4), hipMemcpyDeviceToHost,
                             h->stream));
  const auto t3 = clk::now();
  HIPCHK(h, wait_turn(h, flag_wake ? h->compact_epoch_armed : 0));
  if (flag_wake && h->compact_epoch_armed) {
    // RAFTQ_CYCLE_CHECK=1 (tests, soaks): the completion flag must never run ahead of the data it announces -- what
    // the host sees when the flag lands must be what it sees after a full stream synchronisation
    static const bool check = [] { const char* e = std::getenv("RAFTQ_CYCLE_CHECK"); return e && e[0] == '1'; }();
    if (check) {
      auto digest = [&]() {
        const volatile uint64_t* w = (const volatile uint64_t*)h->adv_h;
        if (segmented) {  // every segment's count and records
          uint64_t d = (h->h_total[3] & 0xffffffffull) * 0x9E3779B97F4A7C15ull;
          for (uint32_t t = 0; t < h->seg_tiles; ++t) {
            const uint32_t c = ((const volatile uint32_t*)h->seg_h)[t];
            d = (d ^ c) * 0xBF58476D1CE4E5B9ull;
            for (uint64_t i = 0; i < (uint64_t)c * 2; ++i) d = (d ^ w[(uint64_t)t * h->seg_stride * 2 + i]) * 0xBF58476D1CE4E5B9ull;
          }
          return d;
        }
        const uint64_t total = *h->h_total;
        const uint64_t words = std::min(total, take_cap) * (sizeof(Adv) / 8);
        uint64_t d = total * 0x9E3779B97F4A7C15ull;
        for (uint64_t i = 0; i < words; ++i) d = (d ^ w[i]) * 0xBF58476D1CE4E5B9ull;
        return d;
      };
      const uint64_t at_flag = digest();
      HIPCHK(h, hipStreamSynchronize(h->stream));
      if (digest() != at_flag)
        return fail(h, RAFTQ_EHIP, std::string(who) + ": the completion flag arrived before the advance list it announces "
                                                      "(RAFTQ_CYCLE_CHECK); set RAFTQ_CYCLE_WAIT=block on this stack");
    }
  }
  const auto t4 = clk::now();
  h->prof[1] += us(t0, t1);
  h->prof[2] += us(t1, t2);
  h->prof[3] += us(t2, t3);
  h->prof[4] += us(t3, t4);
  h->prof_n++;
  const int verdict = check_deltas(h, trusted);
  if (verdict != RAFTQ_OK && !trusted) {
    // the device found a record out of range: no record of either kind was scattered, and the sweep that ran on
    // the unchanged state is not adopted either -- the handle is exactly where it was before the call
    h->cur = cur_before;
    dense_changed(h);
    if (n_advanced) *n_advanced = 0;
    if (counts) *counts = raftq_counts_t{0, 0, 0};
    h->adv_listed = 0;
    return verdict;
  }
  if (counts) {
    uint64_t c = 0, w = 0, l = 0;
    for (uint64_t i = 0; i < h->n_partials; ++i) {
      c += h->h_partials[i].x;
      w += h->h_partials[i].y;
      l += h->h_partials[i].z;
    }
    counts->n_changed = commit ? c : 0;
    counts->n_won = (flags & RAFTQ_SWEEP_VOTES) ? w : 0;
    counts->n_lost = (flags & RAFTQ_SWEEP_VOTES) ? l : 0;
  }
  if (want_list && segmented) {
    const uint64_t total = h->h_total[3] & 0xffffffffull;  // the completion word: epoch << 32 | records in all segments
    if (n_advanced) *n_advanced = total;
    h->adv_listed = total;
    h->adv_packed = true;
    h->adv_segmented = true;
  } else if (want_list) {
    const uint64_t total = *h->h_total;
    if (n_advanced) *n_advanced = total;
    const uint64_t take = std::min(total, take_cap);
    h->adv_listed = take;
    h->adv_packed = sizeof(Adv) == sizeof(Advance16);
    // advances_out == NULL: the caller reads the pinned list in place (raftq_last_advances*)
    if (take && advances_out) std::memcpy(advances_out, h->adv_h, take * sizeof(Adv));
  }
  return verdict;  // RAFTQ_CYCLE_TRUSTED: a dropped record is still reported, with every output filled in
}

extern "C" {

int raftq_cycle(raftq_t* h, const raftq_delta_t* deltas, uint64_t n_deltas, const raftq_vote_delta_t* vote_deltas,
                uint64_t n_vote_deltas, unsigned flags, raftq_advance_t* advances_out, uint64_t cap,
                uint64_t* n_advanced, raftq_counts_t* counts) {
  return cycle_impl<DeltaRec, Advance>(h, "raftq_cycle", deltas, n_deltas, vote_deltas, n_vote_deltas, flags, advances_out, cap,
                                       n_advanced, counts);
}

int raftq_cycle_packed(raftq_t* h, const raftq_delta16_t* deltas, uint64_t n_deltas, const raftq_vote_delta_t* vote_deltas,
                       uint64_t n_vote_deltas, unsigned flags, raftq_advance16_t* advances_out, uint64_t cap,
                       uint64_t* n_advanced, raftq_counts_t* counts) {
  if (h && h->G > (1ull << 32))
    return fail(h, RAFTQ_EINVAL, "raftq_cycle_packed: the packed records carry 32-bit group ids (handle has more than 2^32 groups)");
  return cycle_impl<Delta16Rec, Advance16>(h, "raftq_cycle_packed", deltas, n_deltas, vote_deltas, n_vote_deltas, flags,
                                           advances_out, cap, n_advanced, counts);
}

int raftq_stage_packed(raftq_t* h, uint64_t n_deltas, uint64_t n_vote_deltas, raftq_delta16_t** deltas,
                       raftq_vote_delta_t** vote_deltas) {
  if (int rc = use_device_idle(h, "raftq_stage_packed")) return rc;
  const size_t off_votes = ((size_t)n_deltas * sizeof(raftq_delta16_t) + 255) / 256 * 256;
  if (int rc = ensure_ingest(h, off_votes + (size_t)n_vote_deltas * sizeof(raftq_vote_delta_t) + 256)) return rc;
  if (deltas) *deltas = (raftq_delta16_t*)h->ingest_h;
  if (vote_deltas) *vote_deltas = (raftq_vote_delta_t*)((uint8_t*)h->ingest_h + off_votes);
  return RAFTQ_OK;
}

int raftq_last_advances_packed(raftq_t* h, const raftq_advance16_t** list, uint64_t* n_listed) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!list || !n_listed) return fail(h, RAFTQ_EINVAL, "raftq_last_advances_packed: null argument");
  if (h->adv_listed && !h->adv_packed)
    return fail(h, RAFTQ_ESTATE, "raftq_last_advances_packed: the last list was produced in the 24-byte layout");
  if (h->adv_segmented)
    return fail(h, RAFTQ_ESTATE, "raftq_last_advances_packed: the last list lies in segments (raftq_last_advance_segments)");
  *list = (const raftq_advance16_t*)h->adv_h;
  *n_listed = h->adv_listed;
  return RAFTQ_OK;
}

int raftq_last_advance_segments(raftq_t* h, const raftq_advance16_t** recs, const uint32_t** counts, uint32_t* n_segments, uint64_t* stride) {
  if (!h) return fail(nullptr, RAFTQ_EINVAL, "null handle");
  if (!recs || !counts || !n_segments || !stride) return fail(h, RAFTQ_EINVAL, "raftq_last_advance_segments: null argument");
  if (h->adv_listed && !h->adv_packed)
    return fail(h, RAFTQ_ESTATE, "raftq_last_advance_segments: the last list was produced in the 24-byte layout");
  *recs = (const raftq_advance16_t*)h->adv_h;
  if (h->adv_segmented) {
    *counts = h->seg_h;
    *n_segments = h->seg_tiles;
    *stride = h->seg_stride;
  } else {  // a contiguous list is one segment
    h->seg_one = (uint32_t)h->adv_listed;
    *counts = &h->seg_one;
    *n_segments = 1;
    *stride = 0;
  }
  return RAFTQ_OK;
}

int raftq_timer_begin(raftq_t* h) {
  if (int rc = use_device(h)) return rc;
  HIPCHK(h, hipEventRecord(h->ev0, h->stream));
  return RAFTQ_OK;
}

int raftq_timer_end(raftq_t* h, float* elapsed_ms) {
  if (int rc = use_device(h)) return rc;
  if (!elapsed_ms) return fail(h, RAFTQ_EINVAL, "raftq_timer_end: null argument");
  HIPCHK(h, hipEventRecord(h->ev1, h->stream));
  HIPCHK(h, hipEventSynchronize(h->ev1));
  HIPCHK(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
  return RAFTQ_OK;
}

int raftq_host_alloc(void** p, uint64_t bytes) {
  if (!p || bytes == 0) return fail(nullptr, RAFTQ_EINVAL, "raftq_host_alloc: null pointer or zero size");
  *p = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(nullptr, RAFTQ_ENODEV, "raftq_host_alloc: no HIP device");
  const hipError_t e = hipHostMalloc(p, (size_t)bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    *p = nullptr;
    return fail(nullptr, e == hipErrorOutOfMemory ? RAFTQ_ENOMEM : RAFTQ_EHIP,
                std::string("raftq_host_alloc: ") + hipGetErrorString(e));
  }
  return RAFTQ_OK;
}

void raftq_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

}  // extern "C"

// ---- sweep sets (include/raftq.h "sweep sets") ---------------------------------------------
namespace {
thread_local std::string g_set_err;
int sfail(raftq_set_t* s, int code, const std::string& msg) {
  g_set_err = msg;
  if (s) s->err = msg;
  return fail(nullptr, code, msg);
}
#define SETCHK(s, expr)                                                                                   \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess)                                                                                 \
      return sfail((s), _e == hipErrorOutOfMemory ? RAFTQ_ENOMEM : RAFTQ_EHIP,                           \
                   std::string(#expr) + ": " + hipGetErrorString(_e));                                    \
  } while (0)

int set_ready(raftq_set_t* s, const char* who) {
  if (!s) return sfail(nullptr, RAFTQ_EINVAL, std::string(who) + ": null set");
  if (s->broken) return sfail(s, RAFTQ_ESTATE, std::string(who) + ": a member of the set was destroyed");
  SETCHK(s, hipSetDevice(s->device));
  return RAFTQ_OK;
}

void set_fill_table(const raftq_set_t* s, int cur, std::vector<SweepArgs>& t) {
  t.resize(s->members.size());
  for (size_t i = 0; i < s->members.size(); ++i)
    t[i] = sweep_args(s->members[i], cur < 0 ? s->members[i]->cur : cur, true);
}
}  // namespace

extern "C" {

int raftq_set_create(raftq_t* const* handles, uint32_t n, raftq_set_t** out) {
  if (!out) return sfail(nullptr, RAFTQ_EINVAL, "raftq_set_create: null out");
  *out = nullptr;
  if (!handles || n == 0) return sfail(nullptr, RAFTQ_EINVAL, "raftq_set_create: empty set");
  if (n > 65535) return sfail(nullptr, RAFTQ_EINVAL, "raftq_set_create: at most 65535 members");
  for (uint32_t i = 0; i < n; ++i) {
    raftq_t* h = handles[i];
    if (!h) return sfail(nullptr, RAFTQ_EINVAL, "raftq_set_create: null member");
    if (h->in_set) return sfail(nullptr, RAFTQ_ESTATE, "raftq_set_create: a handle is already a member of a set");
    if (h->device != handles[0]->device || h->N != handles[0]->N || h->gpad != handles[0]->gpad)
      return sfail(nullptr, RAFTQ_EINVAL,
                   "raftq_set_create: members must share the device, the peer count and the (padded) group count");
    if (h->step_collected != h->step_submitted)
      return sfail(nullptr, RAFTQ_ESTATE, "raftq_set_create: a member has Step batches in flight");
    for (uint32_t k = 0; k < i; ++k)
      if (handles[k] == h) return sfail(nullptr, RAFTQ_EINVAL, "raftq_set_create: duplicate member");
  }
  raftq_set_t* s = new (std::nothrow) raftq_set();
  if (!s) return sfail(nullptr, RAFTQ_ENOMEM, "raftq_set_create: host allocation failed");
  s->device = handles[0]->device;
  s->N = handles[0]->N;
  s->gpad = handles[0]->gpad;
  if (const char* e = std::getenv("RAFTQ_SET_MODE")) s->mode = std::atoi(e) == 1 ? 1 : 0;
  try {  // everything the set's calls will ever push into: no allocation (and no exception) after this point
    s->members.reserve(n);
    s->tab_host.reserve(n);
    s->np_host.reserve(n);
  } catch (...) {
    delete s;
    return sfail(nullptr, RAFTQ_ENOMEM, "raftq_set_create: host allocation failed");
  }
  auto bail = [&](int rc) {
    std::string keep = s->err;
    raftq_set_destroy(s);
    g_set_err = keep;
    return rc;
  };
  hipError_t e = hipSetDevice(s->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipMalloc((void**)&s->tab[k], (size_t)n * sizeof(SweepArgs));
  if (e == hipSuccess) e = hipMalloc((void**)&s->counts_d, (size_t)n * 32);
  if (e == hipSuccess) e = hipMalloc((void**)&s->np_d, (size_t)n * 8);
  if (e == hipSuccess) e = hipHostMalloc((void**)&s->counts_h, (size_t)n * 32, hipHostMallocDefault);
  if (e == hipSuccess) e = hipEventCreate(&s->ev0);
  if (e == hipSuccess) e = hipEventCreate(&s->ev1);
  if (e != hipSuccess)
    return bail(sfail(s, e == hipErrorOutOfMemory ? RAFTQ_ENOMEM : RAFTQ_EHIP, std::string("raftq_set_create: ") + hipGetErrorString(e)));
  // re-home every member onto the set's stream: loads, deltas and per-member read-backs stay ordered with the
  // set's sweeps without any event traffic
  for (uint32_t i = 0; i < n; ++i) {
    raftq_t* h = handles[i];
    e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && h->own_stream) e = hipStreamDestroy(h->stream);
    if (e != hipSuccess) return bail(sfail(s, RAFTQ_EHIP, std::string("raftq_set_create: ") + hipGetErrorString(e)));
    h->stream = s->stream;
    h->own_stream = false;
    h->in_set = s;
    s->members.push_back(h);
  }
  for (int cur = 0; cur < 2; ++cur) {
    set_fill_table(s, cur, s->tab_host);
    e = hipMemcpy(s->tab[cur], s->tab_host.data(), (size_t)n * sizeof(SweepArgs), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(sfail(s, RAFTQ_EHIP, std::string("raftq_set_create: ") + hipGetErrorString(e)));
  }
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device);
  s->persist_wgs = (uint32_t)std::max(cus, 1) * 3u;  // 3 resident 256-thread workgroups per CU (144 VGPRs: profiles/r03/isa_sweep.txt)
  if (const char* w = std::getenv("RAFTQ_SET_PERSIST_WGS")) s->persist_wgs = (uint32_t)std::max(1, std::atoi(w));
  *out = s;
  return RAFTQ_OK;
}

void raftq_set_destroy(raftq_set_t* s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (raftq_t* h : s->members) {  // members live on: each gets a stream of its own back
    h->in_set = nullptr;
    h->stream = nullptr;
    h->own_stream = false;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess) h->own_stream = true;
  }
  for (int k = 0; k < 3; ++k) (void)hipFree(s->tab[k]);
  (void)hipFree(s->counts_d);
  (void)hipFree(s->np_d);
  (void)hipFree(s->tick_tab);
  if (s->counts_h) (void)hipHostFree(s->counts_h);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

uint32_t raftq_set_size(const raftq_set_t* s) { return s ? (uint32_t)s->members.size() : 0; }
const char* raftq_set_last_error(const raftq_set_t* s) { return s ? s->err.c_str() : g_set_err.c_str(); }
void* raftq_set_get_stream(const raftq_set_t* s) { return s ? (void*)s->stream : nullptr; }

int raftq_set_mode(raftq_set_t* s, int mode, uint32_t persist_workgroups) {
  if (!s) return sfail(nullptr, RAFTQ_EINVAL, "raftq_set_mode: null set");
  if (mode != RAFTQ_SET_GRID && mode != RAFTQ_SET_PERSISTENT) return sfail(s, RAFTQ_EINVAL, "raftq_set_mode: unknown mode");
  s->mode = mode;
  if (persist_workgroups) s->persist_wgs = persist_workgroups;
  return RAFTQ_OK;
}

int raftq_set_sweep_async(raftq_set_t* s, unsigned flags) {
  if (int rc = set_ready(s, "raftq_set_sweep_async")) return rc;
  if (flags & RAFTQ_SWEEP_LDS) return sfail(s, RAFTQ_EINVAL, "raftq_set_sweep_async: RAFTQ_SWEEP_LDS sweeps one handle at a time");
  uint64_t footprint = 0;
  int cur = s->members[0]->cur;
  for (raftq_t* h : s->members) {
    if (h->step_collected != h->step_submitted)
      return sfail(s, RAFTQ_ESTATE, "raftq_set_sweep_async: a member has Step batches in flight; collect them first");
    if (int rc = sweep_check(h, flags, "raftq_set_sweep_async")) return sfail(s, rc, h->err);
    footprint += sweep_footprint(h);
    if (h->cur != cur) cur = -1;
  }
  const SweepArgs* tab = s->tab[cur < 0 ? 2 : cur];
  if (cur < 0) {  // members disagree on which commit buffer is current: build this launch's table
    set_fill_table(s, -1, s->tab_host);
    SETCHK(s, hipMemcpyAsync(s->tab[2], s->tab_host.data(), s->tab_host.size() * sizeof(SweepArgs), hipMemcpyHostToDevice,
                             s->stream));
  }
  SweepLaunch L;
  L.tab = tab;
  L.members = (uint32_t)s->members.size();
  L.want_bits = (flags & RAFTQ_SWEEP_CHANGED) ? 1u : 0u;
  L.persist_wgs = s->mode == RAFTQ_SET_PERSISTENT ? s->persist_wgs : 0u;
  L.gpad = s->gpad;
  // a set that streams, streams its stores too: with many members in one dispatch non-temporal stores are worth
  // 5-9 % (profiles/r02/tune3_focus_*.jsonl: 10.92 vs 11.78 us per batch), unlike one 1M-group launch at a time
  int nt = sweep_policy(flags, footprint);
  if (nt == 1) nt = 3;
  SETCHK(s, launch_sweep(s->N, L, flags, nt, s->stream));
  const int gpl = s->mode == RAFTQ_SET_PERSISTENT ? kGPL : set_gpl((int)s->N);
  for (raftq_t* h : s->members) sweep_done(h, flags, gpl);
  s->swept = true;
  s->last_flags = flags;
  return RAFTQ_OK;
}

// rc.node.Tick() for every group of every member: ONE dispatch (blockIdx.y = member), enqueued on the set's stream.  The
// members' per-handle results (action bytes, the MsgHup / MsgBeat bitmaps and counts behind raftq_collect_hups / _beats /
// raftq_read_tick) are what raftq_tick on each member would have left.
int raftq_set_tick(raftq_set_t* s) {
  if (int rc = set_ready(s, "raftq_set_tick")) return rc;
  const size_t K = s->members.size();
  bool stale = s->tick_tab == nullptr || s->tick_host.size() != K;
  for (size_t m = 0; m < K; ++m) {
    raftq_t* h = s->members[m];
    if (h->step_collected != h->step_submitted)
      return sfail(s, RAFTQ_ESTATE, "raftq_set_tick: a member has Step batches in flight; collect them first");
    const bool fresh = h->role == nullptr;
    if (int rc = ensure_tick_state(h)) return sfail(s, rc, h->err);
    if (stale || fresh) {
      stale = true;
      continue;
    }
    const TickArgs now = tick_args(h, h->tick_no), &was = s->tick_host[m];
    if (now.role != was.role || now.seed != was.seed || now.election_tick != was.election_tick || now.heartbeat_tick != was.heartbeat_tick ||
        now.tick_no != was.tick_no + s->tick_since)
      stale = true;  // a member was re-configured, or ticked on its own since the table was built
  }
  if (stale) {
    if (!s->tick_tab) SETCHK(s, hipMalloc((void**)&s->tick_tab, K * sizeof(TickArgs)));
    s->tick_host.resize(K);
    for (size_t m = 0; m < K; ++m) s->tick_host[m] = tick_args(s->members[m], s->members[m]->tick_no);
    SETCHK(s, hipMemcpyAsync(s->tick_tab, s->tick_host.data(), K * sizeof(TickArgs), hipMemcpyHostToDevice, s->stream));  // (pageable: staged before it returns)
    s->tick_since = 0;
  }
  const uint64_t n_blocks = s->gpad / 1024;
  // RAFTQ_TICK_SHAPE: wide1 (default: 16 groups per lane, one 1,024-group block per wave: 14.9 us per 8 x 1M groups = 0.70 of HBM
  // in steady state, profiles/r05), wide2 / wide4 (two / four blocks per wave: 16.1 / 17.2), narrow (round 4's 4 groups per lane,
  // four rounds per workgroup: 15.7).  Same results, byte for byte (tests/test_parity_gpu.py runs the set Tick in every shape).
  // (read at every call -- a getenv is nothing beside a launch -- so that a test or the bench can A/B the shapes in one process)
  int shape_now = 1;
  if (const char* e = std::getenv("RAFTQ_TICK_SHAPE"))
    shape_now = std::strcmp(e, "narrow") == 0 ? 0 : std::strcmp(e, "wide2") == 0 ? 2 : std::strcmp(e, "wide4") == 0 ? 4 : 1;
  auto wide_grid = [&](int r) { return dim3((unsigned)((n_blocks + (uint64_t)kWaves * r - 1) / ((uint64_t)kWaves * r)), (unsigned)K); };
  switch (shape_now) {
    case 0:
      hipLaunchKernelGGL(tick_set_kernel, dim3((unsigned)((n_blocks + kTickSetRounds - 1) / kTickSetRounds), (unsigned)K), dim3(kBlock), 0, s->stream,
                         (const TickArgs*)s->tick_tab, s->tick_since, n_blocks);
      break;
    case 2:
      hipLaunchKernelGGL((tick_set_wide_kernel<2>), wide_grid(2), dim3(kBlock), 0, s->stream, (const TickArgs*)s->tick_tab, s->tick_since, n_blocks);
      break;
    case 4:
      hipLaunchKernelGGL((tick_set_wide_kernel<4>), wide_grid(4), dim3(kBlock), 0, s->stream, (const TickArgs*)s->tick_tab, s->tick_since, n_blocks);
      break;
    default:
      hipLaunchKernelGGL((tick_set_wide_kernel<kTickWideBlocks>), wide_grid(kTickWideBlocks), dim3(kBlock), 0, s->stream, (const TickArgs*)s->tick_tab,
                         s->tick_since, n_blocks);
  }
  SETCHK(s, hipGetLastError());
  ++s->tick_since;
  for (raftq_t* h : s->members) {
    h->tl_valid = false;
    ++h->tick_no;
    h->ticked = true;
  }
  return RAFTQ_OK;
}

int raftq_set_wait(raftq_set_t* s, raftq_counts_t* per_member, raftq_counts_t* total) {
  if (int rc = set_ready(s, "raftq_set_wait")) return rc;
  const bool want = per_member || total;
  const size_t n = s->members.size();
  if (want) {
    if (!s->swept) return sfail(s, RAFTQ_ESTATE, "raftq_set_wait: no set sweep to report on");
    // the tallies of the LAST sweep of every member (a member swept on its own since then reports that sweep)
    set_fill_table(s, -1, s->tab_host);
    SETCHK(s, hipMemcpyAsync(s->tab[2], s->tab_host.data(), n * sizeof(SweepArgs), hipMemcpyHostToDevice, s->stream));
    // number of per-wave partials of every member's last sweep (pageable source: staged before the call returns)
    s->np_host.resize(n);
    for (size_t i = 0; i < n; ++i) s->np_host[i] = s->members[i]->n_partials;
    SETCHK(s, hipMemcpyAsync(s->np_d, s->np_host.data(), n * 8, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(set_counts_kernel, dim3((unsigned)n), dim3(kBlock), 0, s->stream, (const SweepArgs*)s->tab[2],
                       (const uint64_t*)s->np_d, s->counts_d);
    SETCHK(s, hipGetLastError());
    SETCHK(s, hipMemcpyAsync(s->counts_h, s->counts_d, n * 32, hipMemcpyDeviceToHost, s->stream));
  }
  SETCHK(s, hipStreamSynchronize(s->stream));
  if (want) {
    raftq_counts_t t{0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
      const unsigned f = s->members[i]->last_flags;
      const bool commit = f & (RAFTQ_SWEEP_COMMIT | RAFTQ_SWEEP_GATED);
      const bool votes = f & RAFTQ_SWEEP_VOTES;
      raftq_counts_t c{commit ? s->counts_h[4 * i] : 0, votes ? s->counts_h[4 * i + 1] : 0, votes ? s->counts_h[4 * i + 2] : 0};
      if (per_member) per_member[i] = c;
      t.n_changed += c.n_changed;
      t.n_won += c.n_won;
      t.n_lost += c.n_lost;
    }
    if (total) *total = t;
  }
  return RAFTQ_OK;
}

int raftq_set_timer_begin(raftq_set_t* s) {
  if (int rc = set_ready(s, "raftq_set_timer_begin")) return rc;
  SETCHK(s, hipEventRecord(s->ev0, s->stream));
  return RAFTQ_OK;
}

int raftq_set_timer_end(raftq_set_t* s, float* elapsed_ms) {
  if (int rc = set_ready(s, "raftq_set_timer_end")) return rc;
  if (!elapsed_ms) return sfail(s, RAFTQ_EINVAL, "raftq_set_timer_end: null argument");
  SETCHK(s, hipEventRecord(s->ev1, s->stream));
  SETCHK(s, hipEventSynchronize(s->ev1));
  SETCHK(s, hipEventElapsedTime(elapsed_ms, s->ev0, s->ev1));
  return RAFTQ_OK;
}

int raftq_clone_state(raftq_t* dst, raftq_t* src) {
  if (int rc = use_device_idle(dst, "raftq_clone_state")) return rc;
  if (!src) return fail(dst, RAFTQ_EINVAL, "raftq_clone_state: null source");
  if (src == dst) return RAFTQ_OK;
  if (src->device != dst->device || src->N != dst->N || src->G != dst->G)
    return fail(dst, RAFTQ_EINVAL, "raftq_clone_state: source and destination must have the same device, groups and peers");
  if (src->step_collected != src->step_submitted)
    return fail(dst, RAFTQ_ESTATE, "raftq_clone_state: the source has Step batches in flight");
  HIPCHK(dst, hipStreamSynchronize(src->stream));
  const uint64_t ld = dst->ld;
  HIPCHK(dst, hipMemcpyAsync(dst->match, src->match, (size_t)dst->N * ld * 8, hipMemcpyDeviceToDevice, dst->stream));
  HIPCHK(dst, hipMemcpyAsync(dst->committed[dst->cur], src->committed[src->cur], ld * 8, hipMemcpyDeviceToDevice, dst->stream));
  HIPCHK(dst, hipMemcpyAsync(dst->first_idx, src->first_idx, ld * 8, hipMemcpyDeviceToDevice, dst->stream));
  HIPCHK(dst, hipMemcpyAsync(dst->votes, src->votes, (size_t)vote_word_bytes((int)dst->N) * ld, hipMemcpyDeviceToDevice,
                             dst->stream));
  HIPCHK(dst, hipStreamSynchronize(dst->stream));
  dst->have_terms = src->have_terms;
  dense_changed(dst);
  return RAFTQ_OK;
}

}  // extern "C"
