// wire_parse_host.cpp -- raftsql_amd/csrc/raftq_wire_parse.hpp (the per-frame Unmarshal code the decode kernels run per
// lane) compiled FOR THE HOST, so the very source the GPU executes is checked against the codec oracle on the fuzz corpus
// without a GPU (tests/test_wire_parse_host.py).  TEST INFRASTRUCTURE: plain loops over the frames, the two passes of
// raftq_wire_decode (count, exclusive scan, entries) and raftq_wal_decode's parse step; nothing of the product links it.
#include <stdint.h>
#include <string.h>

#include "raftq_wire_parse.hpp"

using namespace raftqk;

extern "C" {

// -> number of malformed frames; *n_ents_out = total entries (all of them are written when cap allows)
uint64_t host_wire_decode(const uint8_t* stream, uint64_t nbytes, const uint64_t* off, uint64_t n, WireMsg* msgs, WireEnt* ents,
                          uint64_t ents_cap, uint64_t* n_ents_out) {
  uint64_t bad = 0, total = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t a = off[i], b = off[i + 1];
    WireMsg m;
    bool ok = frame_body(stream, nbytes, a, b, true);
    ArrayFile file;
    const ByteSrc src = {stream + a + 8, nbytes - a - 8};
    if (ok) ok = parse_msg<false>(src, b - a - 8, a + 8, file, m, nullptr, 0, 0, 0);
    if (!ok) {
      memset(&m, 0, sizeof m);
      m.flags = kWireMalformed;
      ++bad;
    }
    msgs[i] = m;
    total += m.n_ents;
  }
  uint64_t first = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t cnt = msgs[i].n_ents;
    if (cnt) {
      const uint64_t a = off[i], b = off[i + 1];
      WireMsg m;
      ArrayFile file;
      const ByteSrc src = {stream + a + 8, nbytes - a - 8};
      (void)parse_msg<true>(src, b - a - 8, a + 8, file, m, ents, first, ents_cap, cnt);
      msgs[i].ent_first = (uint32_t)first;
    }
    first += cnt;
  }
  *n_ents_out = total;
  return bad;
}

// the parse step of raftq_wal_decode: records + Record.data spans; flags = kWalMalformed where a frame does not parse
void host_wal_parse(const uint8_t* bytes, uint64_t nbytes, const uint64_t* off, uint64_t n, WalRec* recs, uint64_t* span_off,
                    uint64_t* span_len) {
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t a = off[i], b = off[i + 1];
    WalRec r;
    uint64_t d_off = 0, d_len = 0;
    bool ok = frame_body(bytes, nbytes, a, b, false);
    const ByteSrc src = {bytes + a + 8, nbytes - a - 8};
    if (ok) ok = parse_wal_rec(src, b - a - 8, a + 8, r, d_off, d_len);
    span_off[i] = span_len[i] = 0;
    if (!ok) {
      memset(&r, 0, sizeof r);
      r.flags = kWalMalformed;
    } else if (r.kind != kWalCrc) {
      span_off[i] = a + 8 + d_off;
      span_len[i] = d_len;
    }
    recs[i] = r;
  }
}

// CRC-32C (standard inversions, seed `crc`) of data[0, n) by the slicing-by-8 update the kernels use
uint32_t host_crc32c(uint32_t crc, const uint8_t* data, uint64_t n) {
  static uint32_t tab[kCrcTabs * 256];
  static bool built = false;
  if (!built) {
    crc_tables_build(tab);
    built = true;
  }
  const ByteSrc src = {data, n};
  return ~crc_span8(tab, ~crc, src, 0, n);
}

}  // extern "C"
