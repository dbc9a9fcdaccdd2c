// raftq_wire_parse.hpp -- the Unmarshal side of the wire / WAL codecs (include/raftq_wire.h), one frame per call.
//
// What runs per lane in wire_dec_kernel / wire_dec_ents_kernel / wal_dec_kernel (raftq_wire_kernels.hpp): the byte-level
// restatement of raftpb.Message.Unmarshal, raftpb.Entry.Unmarshal and walpb.Record.Unmarshal (2015-era generated
// code, the unmarshal behind rafthttp's message stream and w.ReadAll() -- reference call sites raft.go:268-270 and
// raft.go:124).  Plain functions over byte pointers, compiled for the device AND for the host: the very same source is
// built into tests/c/libwire_parse_host.so and checked against the codec oracle on the fuzz corpus without a GPU
// (tests/test_wire_parse_host.py), so a kernel change is validated before it is sent to the GPU box.
//
// Round 3: Message.Unmarshal is ONE flat loop, one field per iteration, no nested loops and no
// switch.  Round 2's form (key varint, value varint, switch on the field number, a nested Entry loop) ran at 4,400
// dynamic instructions and 44 dependent loads per wave of 64 frames -- 17 us for 64K frames with no entries, 45 us
// when 15 % of the frames carried entries (profiles/r03/wire_before_*): lanes that had met an Entry were one or more
// fields behind their neighbours ever after and the wave executed every arm of the switch on every iteration.  Now
//   * the key byte and the value varint of a field come out of ONE unaligned 8-byte load (keys of the fields both
//     messages know are one byte; values up to 2^49 fit the other seven) through one SWAR compress;
//   * the field is filed by predicated selects -- lanes at different fields cost nothing extra;
//   * an Entry is a scope (its end offset) of the same loop, not a nested loop: a lane inside an Entry and a lane at a
//     top-level field execute the same instructions;
//   * the empty Snapshot every stock encoder emits (8 fixed body bytes) is one compare; anything else goes through the
//     generic nested walk.
// Everything the fast form does not cover -- multi-byte keys, 9/10-byte varints, unknown fields, the last bytes of the
// buffer -- takes the byte-loop forms below, per field; results are identical by construction and by the fuzz tests.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RAFTQ_HD __host__ __device__
#else
#define RAFTQ_HD
#endif

namespace raftqk {

struct WireMsg {  // == raftq_wire_msg_t; the first 46 bytes are MsgRec's
  uint64_t group, term, log_term, index, commit, reject_hint;
  uint32_t from;
  uint8_t type, reject, to, flags;
  uint32_t ent_first, n_ents;
};
struct WireEnt {  // == raftq_wire_ent_t
  uint64_t term, index, data_off;
  uint32_t data_len, type;
};
struct WalRec {  // == raftq_wal_rec_t
  uint64_t group, term, index, data_off;
  uint32_t data_len, vote, crc;
  uint8_t kind, entry_type, flags, pad;
};
static_assert(sizeof(WireMsg) == 64 && sizeof(WireEnt) == 32 && sizeof(WalRec) == 48, "record layout");

constexpr uint8_t kWireMalformed = 1, kWireSnapshot = 2, kWireGroup = 4;
constexpr uint8_t kWalMetadata = 1, kWalEntry = 2, kWalState = 3, kWalCrc = 4, kWalSnapshot = 5;
constexpr uint8_t kWalMalformed = 1, kWalBadCrc = 2, kWalGroup = 4;

RAFTQ_HD inline uint64_t load_u64(const uint8_t* p) {
  uint64_t w;
  __builtin_memcpy(&w, p, 8);
  return w;
}
// 1-based index of the lowest set bit (x != 0)
RAFTQ_HD inline uint32_t ffs64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__ffsll((long long)x);
#else
  return (uint32_t)__builtin_ffsll((long long)x);
#endif
}
// the 7-bit groups of the varint in the low `len` bytes of w (len 1..8), packed
RAFTQ_HD inline uint64_t varint_compress(uint64_t w, uint32_t len) {
  if (len < 8) w &= (1ull << (8 * len)) - 1;
  w &= 0x7f7f7f7f7f7f7f7full;
  w = ((w & 0x7f007f007f007f00ull) >> 1) | (w & 0x007f007f007f007full);
  w = ((w & 0x3fff00003fff0000ull) >> 2) | (w & 0x00003fff00003fffull);
  w = ((w & 0x0fffffff00000000ull) >> 4) | (w & 0x000000000fffffffull);
  return w;
}

// the Unmarshal varint loop.  Returns bytes consumed, 0 = malformed (truncated or > 10 bytes).
RAFTQ_HD inline uint32_t get_varint(const uint8_t* p, uint64_t n, uint64_t* v) {
  if (n >= 8) {
    const uint64_t w = load_u64(p);
    const uint64_t stop = ~w & 0x8080808080808080ull;  // bit 7 of every byte that ends a varint
    if (stop) {
      const uint32_t len = ffs64(stop) >> 3;  // 1..8
      *v = varint_compress(w, len);
      return len;
    }
  }
  uint64_t r = 0;
  for (uint32_t i = 0; i < n && i < 10; ++i) {
    const uint8_t b = p[i];
    r |= (uint64_t)(b & 0x7f) << (7 * i);  // the 10th byte's high bits fall off, as in Go
    if (b < 0x80) {
      *v = r;
      return i + 1;
    }
  }
  return 0;
}

// skipRaft: bytes of one unknown field's value, 0 = malformed
RAFTQ_HD inline uint64_t skip_value(const uint8_t* p, uint64_t n, uint32_t wt) {
  uint64_t v;
  if (wt == 0) return get_varint(p, n, &v);
  if (wt == 1) return n >= 8 ? 8 : 0;
  if (wt == 5) return n >= 4 ? 4 : 0;
  if (wt == 2) {
    const uint32_t k = get_varint(p, n, &v);
    if (!k || v > n - k) return 0;
    return k + v;
  }
  return 0;  // groups and the two unassigned wire types
}

// one field key; false = malformed
struct Key {
  uint64_t fn;
  uint32_t wt;
};
RAFTQ_HD inline bool get_key(const uint8_t* p, uint64_t n, uint64_t& i, Key& k) {
  uint64_t key;
  const uint32_t used = get_varint(p + i, n - i, &key);
  if (!used) return false;
  i += used;
  k.wt = (uint32_t)(key & 7);
  k.fn = key >> 3;
  return k.fn != 0;  // "illegal tag 0"
}

// ---- raftpb.Entry -----------------------------------------------------------------------------------

// Entry.Unmarshal (the generic form: WAL records, where an Entry is the whole Record.data).  base = offset of p[0] in
// the enclosing buffer.  group == nullptr: field 5 is unknown.
RAFTQ_HD inline bool parse_entry(const uint8_t* p, uint64_t n, uint64_t base, WireEnt& e, uint64_t* group,
                                 bool* has_group) {
  e.term = e.index = e.data_off = 0;
  e.data_len = e.type = 0;
  uint64_t i = 0;
  while (i < n) {
    Key k;
    uint64_t v;
    if (!get_key(p, n, i, k)) return false;
    if (k.fn <= 3 || (k.fn == 5 && group)) {
      if (k.wt != 0) return false;
      const uint32_t used = get_varint(p + i, n - i, &v);
      if (!used) return false;
      i += used;
      if (k.fn == 1) e.type = (uint32_t)v;
      else if (k.fn == 2) e.term = v;
      else if (k.fn == 3) e.index = v;
      else {
        *group = v;
        *has_group = true;
      }
    } else if (k.fn == 4) {
      if (k.wt != 2) return false;
      const uint32_t used = get_varint(p + i, n - i, &v);
      if (!used || v > n - i - used || v > 0xffffffffull) return false;
      i += used;
      e.data_off = base + i;
      e.data_len = (uint32_t)v;
      i += v;
    } else {
      const uint64_t used = skip_value(p + i, n - i, k.wt);
      if (!used) return false;
      i += used;
    }
  }
  if (e.data_len == 0) e.data_off = 0;
  return true;
}

// ---- raftpb.Message ---------------------------------------------------------------------------------

// SnapshotMetadata{1 conf_state (message), 2 index, 3 term}: 1 = something set, 0 = all empty, -1 = malformed
RAFTQ_HD inline int snapshot_meta_nonempty(const uint8_t* p, uint64_t n) {
  uint64_t i = 0;
  int nonempty = 0;
  while (i < n) {
    Key k;
    uint64_t v;
    if (!get_key(p, n, i, k)) return -1;
    if (k.fn > 3) {
      const uint64_t used = skip_value(p + i, n - i, k.wt);
      if (!used) return -1;
      i += used;
      continue;
    }
    const bool is_len = k.fn == 1;
    if (k.wt != (is_len ? 2u : 0u)) return -1;
    const uint32_t used = get_varint(p + i, n - i, &v);
    if (!used) return -1;
    i += used;
    if (is_len) {
      if (v > n - i) return -1;
      i += v;
    }
    if (v) nonempty = 1;
  }
  return nonempty;
}
// Snapshot{1 data (bytes), 2 metadata (message)}
RAFTQ_HD inline int snapshot_nonempty(const uint8_t* p, uint64_t n) {
  uint64_t i = 0;
  int nonempty = 0;
  while (i < n) {
    Key k;
    uint64_t v;
    if (!get_key(p, n, i, k)) return -1;
    if (k.fn > 2) {
      const uint64_t used = skip_value(p + i, n - i, k.wt);
      if (!used) return -1;
      i += used;
      continue;
    }
    if (k.wt != 2) return -1;
    const uint32_t used = get_varint(p + i, n - i, &v);
    if (!used) return -1;
    i += used;
    if (v > n - i) return -1;
    if (k.fn == 2) {
      const int r = snapshot_meta_nonempty(p + i, v);
      if (r < 0) return -1;
      nonempty |= r;
    } else if (v) {
      nonempty = 1;
    }
    i += v;
  }
  return nonempty;
}

RAFTQ_HD inline uint32_t id_to_slot(uint64_t id, uint32_t none) {
  return id == 0 || id - 1 >= none ? none : (uint32_t)(id - 1);
}

// the body of the Snapshot a stock encoder writes for "no snapshot": 12 06 0a 00 10 00 18 00 (metadata{conf_state{},
// index 0, term 0}), as one little-endian word
constexpr uint64_t kEmptySnapshotBody = 0x00180010000a0612ull;

// Where a parse files the scalar fields it meets: slot fn for Message fields 1..12, slot 12 + fn for the fields 1..4 of
// the Entry being walked.  One store per field instead of a chain of selects over a dozen live registers; last write wins,
// as in the generated code.  On the device the slots are LDS words (LdsFile, raftq_wire_kernels.hpp: slot-major, one
// column per lane, so a wave's lanes never share a bank); on the host an array.
constexpr int kFileSlots = 17;
struct ArrayFile {
  uint64_t a[kFileSlots];
  RAFTQ_HD void put(uint32_t slot, uint64_t v) { a[slot] = v; }
  RAFTQ_HD uint64_t get(uint32_t slot) const { return a[slot]; }
};

// Where a parse reads its bytes: the frame in global memory, or -- when the kernel staged the wave's frames in LDS --
// dword-aligned LDS words (three aligned reads and two byte-aligns make one unaligned 8-byte window).
struct ByteSrc {
  const uint8_t* p;        // the message in its buffer (always valid: the byte-loop forms read here)
  uint64_t safe;           // 8-byte windows may start at offsets i with i + 8 <= safe
#if defined(__HIPCC__)
  const uint32_t* words;   // LDS copy, or nullptr
  uint32_t shift;          // message byte k is LDS byte k + shift
#endif
  RAFTQ_HD uint64_t ld8(uint64_t i) const {
#if defined(__HIP_DEVICE_COMPILE__)
    if (words) {
      const uint32_t o = (uint32_t)i + shift;
      const uint32_t* q = words + (o >> 2);
      const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
      const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, o & 3u), hi = __builtin_amdgcn_alignbyte(d2, d1, o & 3u);
      return ((uint64_t)hi << 32) | lo;
    }
#endif
    return load_u64(p + i);
  }
};

// the value of the varint in the low `len` bytes (1..7) of x
RAFTQ_HD inline uint64_t varint_value(uint64_t x, uint32_t len) {
  if (len <= 4) {  // up to 28 bits: the common case, in 32-bit arithmetic
    uint32_t y = (uint32_t)x;
    if (len < 4) y &= (1u << (8 * len)) - 1;
    y &= 0x7f7f7f7fu;
    y = ((y & 0x7f007f00u) >> 1) | (y & 0x007f007fu);
    y = ((y & 0x3fff0000u) >> 2) | (y & 0x00003fffu);
    return y;
  }
  return varint_compress(x, len);
}

// Message.Unmarshal over the n bytes of `src` as one flat loop (see the head of this file).
//   base      offset of the message's first byte in the enclosing buffer (entry payload offsets are relative to that buffer)
//   file      scratch for the scalar fields (see ArrayFile)
//   EMIT      entry k of this message goes to ents[ent_base + k] when that is below ents_cap, and the walk stops
//             after `stop_after` entries (the count a first pass found) -- what follows them was validated then
// false = malformed.  m.n_ents counts the entries either way; m.ent_first is the caller's.
template <bool EMIT, typename IX, typename File>
RAFTQ_HD inline bool parse_msg_ix(const ByteSrc& src, IX n, uint64_t base, File& file, WireMsg& m, WireEnt* ents,
                                  uint64_t ent_base, uint64_t ents_cap, uint32_t stop_after) {
  const uint8_t* p = src.p;
  for (uint32_t s = 1; s <= 12; ++s) file.put(s, 0);  // absent fields read back as zero: type 0, to / from "none", ...
  uint32_t flags = 0, n_ents = 0;
  uint64_t e_off = 0;        // payload offset of the Entry being walked
  IX i = 0, lim = n;         // lim: end of the current scope -- the message, or the Entry being walked.  IX: offsets inside
                             // the message -- 32 bits wide for every message shorter than 2 GiB (half the address arithmetic)
  bool in_ent = false;
  bool bad = false;          // malformed: the one way out of the loop besides its end (a lane-divergent `break` per check
                             // costs the wave a mask save / restore each; one sticky flag tested at the top costs one)
  // (A third form -- one straight line for "known one-byte-key varint field" and a single cold block with the byte-loop
  // forms for everything else, Entry and Snapshot fields included -- compiled to MORE instructions and ran mixed traffic at
  // 28 us instead of 21: every Entry field then pays the generic key + varint decode.  Not kept.)
  while (!bad) {
    if (i >= lim) {  // a scope ends exactly where its last field does (every step below is bounded by lim)
      if (!in_ent) break;
      if (EMIT) {
        const uint64_t slot = ent_base + n_ents;
        if (ents && slot < ents_cap) {
          WireEnt e;
          e.type = (uint32_t)file.get(13);
          e.term = file.get(14);
          e.index = file.get(15);
          e.data_len = (uint32_t)file.get(16);
          e.data_off = e.data_len ? e_off : 0;
          ents[slot] = e;
        }
      }
      ++n_ents;
      in_ent = false;
      lim = n;
      if (EMIT && n_ents >= stop_after) break;
      continue;
    }
    const IX rem = lim - i;
    const uint32_t known = in_ent ? 4u : 12u;
    // the one-load form: key byte + value varint out of one 8-byte window
    const bool have = (uint64_t)i + 8 <= src.safe;
    const uint64_t w = have ? src.ld8(i) : 0x80ull;  // (a window that cannot be read looks like a long key: byte-loop form)
    const uint64_t vstop = ~w & 0x8080808080808000ull;  // terminators among the seven bytes behind the key byte
    uint32_t fn = ((uint32_t)w >> 3) & 0x1fu;             // of a one-byte key (bit 7 clear: checked next)
    uint32_t wt = (uint32_t)w & 7u;
    const uint32_t used = vstop ? ffs64(vstop) >> 3 : 9u;  // key + varint bytes, 2..8
    uint64_t v = varint_value(w >> 8, (used - 1) & 7u);
    const bool fast = ((uint32_t)w & 0x80u) == 0 && vstop != 0 && fn != 0 && fn <= known;
    if (fast) {
      bad = used > rem;  // the varint runs over the end of its scope: io.ErrUnexpectedEOF
      i += (IX)used;
    } else {  // byte-loop forms, bounded by the scope: long keys, long varints, unknown fields, the buffer's tail
      Key k;
      uint64_t j = i;
      bool skip = false;
      if (!get_key(p, (uint64_t)lim, j, k)) {
        bad = true;
      } else if (k.fn > known) {
        const uint64_t u = skip_value(p + j, (uint64_t)lim - j, k.wt);
        bad = u == 0;
        i = (IX)(j + u);
        skip = true;
      } else {
        fn = (uint32_t)k.fn;
        wt = k.wt;
        const uint32_t u = get_varint(p + j, (uint64_t)lim - j, &v);
        bad = u == 0;
        i = (IX)(j + u);
      }
      if (bad || skip) continue;
    }
    // one known field (fn, wt, v); i is behind its varint
    const bool is_len = in_ent ? fn == 4 : (fn == 7 || fn == 9);
    bad |= wt != (is_len ? 2u : 0u);  // "wrong wireType"
    file.put(fn + (in_ent ? 12u : 0u), v);
    flags |= (!in_ent && fn == 12) ? kWireGroup : 0u;
    if (is_len && !bad) {
      if (v > (uint64_t)(lim - i) || (in_ent && v > 0xffffffffull)) {  // io.ErrUnexpectedEOF / a payload no Entry can hold
        bad = true;
      } else if (in_ent) {  // Entry.data
        e_off = base + i;
        i += (IX)v;
      } else if (fn == 7) {  // an Entry: a scope of this loop (an empty one is complete at once, on the next turn)
        in_ent = true;
        lim = i + (IX)v;
        file.put(13, 0);
        file.put(14, 0);
        file.put(15, 0);
        file.put(16, 0);
      } else {  // the Snapshot
        int r = 0;
        if (!(v == 8 && (uint64_t)i + 8 <= src.safe && src.ld8(i) == kEmptySnapshotBody)) r = snapshot_nonempty(p + i, v);
        bad = r < 0;
        flags |= r > 0 ? kWireSnapshot : 0u;
        i += (IX)v;
      }
    }
  }
  const uint64_t type = file.get(1);
  m.type = (uint32_t)type > 255 ? (uint8_t)255 : (uint8_t)type;
  m.to = (uint8_t)id_to_slot(file.get(2), 0xff);
  m.from = id_to_slot(file.get(3), 0xffffffffu);
  m.term = file.get(4);
  m.log_term = file.get(5);
  m.index = file.get(6);
  m.commit = file.get(8);
  m.reject = file.get(10) != 0 ? 1 : 0;
  m.reject_hint = file.get(11);
  m.group = file.get(12);
  m.flags = (uint8_t)flags;
  m.ent_first = 0;
  m.n_ents = n_ents;
  return !bad;
}

template <bool EMIT, typename File>
RAFTQ_HD inline bool parse_msg(const ByteSrc& src, uint64_t n, uint64_t base, File& file, WireMsg& m, WireEnt* ents,
                               uint64_t ent_base, uint64_t ents_cap, uint32_t stop_after) {
  if (n < (1ull << 31)) return parse_msg_ix<EMIT, uint32_t>(src, (uint32_t)n, base, file, m, ents, ent_base, ents_cap, stop_after);
  return parse_msg_ix<EMIT, uint64_t>(src, n, base, file, m, ents, ent_base, ents_cap, stop_after);
}

// ---- walpb.Record -----------------------------------------------------------------------------------

// One field of a message whose known field numbers are 1..known, at offset i of a scope that ends at lim: the key and --
// for a known field -- the varint behind it (its value, or the length of a length-delimited field), i moved past both.
// An unknown field is skipped whole (skipped = true).  The one-window form where it applies (one-byte key, value ending
// within seven bytes, window inside `safe`), the byte-loop forms otherwise.  false = malformed.
RAFTQ_HD inline bool next_field(const ByteSrc& src, uint64_t& i, uint64_t lim, uint32_t known, uint32_t& fn, uint32_t& wt, uint64_t& v,
                                bool& skipped) {
  skipped = false;
  const uint64_t rem = lim - i;
  if (i + 8 <= src.safe) {
    const uint64_t w = src.ld8(i);
    const uint64_t vstop = ~w & 0x8080808080808000ull;
    const uint32_t f = ((uint32_t)w >> 3) & 0x1fu;
    if (((uint32_t)w & 0x80u) == 0 && vstop != 0 && f != 0 && f <= known) {
      const uint32_t used = ffs64(vstop) >> 3;
      if (used > rem) return false;  // the varint runs over the end of its scope
      fn = f;
      wt = (uint32_t)w & 7u;
      v = varint_value(w >> 8, used - 1);
      i += used;
      return true;
    }
  }
  Key k;
  uint64_t j = i;
  if (!get_key(src.p, lim, j, k)) return false;
  if (k.fn > known) {
    const uint64_t u = skip_value(src.p + j, lim - j, k.wt);
    if (!u) return false;
    i = j + u;
    skipped = true;
    return true;
  }
  fn = (uint32_t)k.fn;
  wt = k.wt;
  const uint32_t u = get_varint(src.p + j, lim - j, &v);
  if (!u) return false;
  i = j + u;
  return true;
}

// Record.Unmarshal + the Data unmarshal ReadAll does per type, over the n bytes of `src`.  d_off / d_len: Record.data
// inside the record.  Round 3: the same one-window-per-field walk as parse_msg (the generic key / varint pair per field,
// twice nested, was round 2's form); Record.data is parsed after the record's own fields because its type may follow it.
RAFTQ_HD inline bool parse_wal_rec(const ByteSrc& src, uint64_t n, uint64_t base, WalRec& r, uint64_t& d_off, uint64_t& d_len) {
  r.group = r.term = r.index = r.data_off = 0;
  r.data_len = r.vote = r.crc = 0;
  r.kind = r.entry_type = r.flags = r.pad = 0;
  d_off = d_len = 0;
  uint64_t i = 0, type = 0, v;
  uint32_t fn, wt;
  bool skipped;
  while (i < n) {
    if (!next_field(src, i, n, 3, fn, wt, v, skipped)) return false;
    if (skipped) continue;
    if (fn <= 2) {
      if (wt != 0) return false;
      if (fn == 1) type = v;
      else r.crc = (uint32_t)v;
    } else {
      if (wt != 2 || v > n - i) return false;
      d_off = i;
      d_len = v;
      i += v;
    }
  }
  if (type < 1 || type > 5) return false;  // ReadAll: "unexpected block type"
  r.kind = (uint8_t)type;
  const uint64_t end = d_off + d_len;
  uint64_t j = d_off;
  if (r.kind == kWalEntry) {  // Entry{1 type, 2 term, 3 index, 4 data, 5 group}
    uint64_t e_off = 0;
    uint32_t e_len = 0;
    while (j < end) {
      if (!next_field(src, j, end, 5, fn, wt, v, skipped)) return false;
      if (skipped) continue;
      if (fn == 4) {
        if (wt != 2 || v > end - j || v > 0xffffffffull) return false;
        e_off = base + j;
        e_len = (uint32_t)v;
        j += v;
      } else {
        if (wt != 0) return false;
        if (fn == 1) r.entry_type = (uint8_t)(uint32_t)v;
        else if (fn == 2) r.term = v;
        else if (fn == 3) r.index = v;
        else {
          r.group = v;
          r.flags |= kWalGroup;
        }
      }
    }
    r.data_len = e_len;
    r.data_off = e_len ? e_off : 0;
  } else if (r.kind == kWalState || r.kind == kWalSnapshot) {
    const uint32_t known = r.kind == kWalState ? 4 : 2;
    while (j < end) {
      if (!next_field(src, j, end, known, fn, wt, v, skipped)) return false;
      if (skipped) continue;
      if (wt != 0) return false;
      if (r.kind == kWalState) {
        if (fn == 1) r.term = v;
        else if (fn == 2) r.vote = (uint32_t)v;
        else if (fn == 3) r.index = v;
        else {
          r.group = v;
          r.flags |= kWalGroup;
        }
      } else {
        if (fn == 1) r.index = v;
        else r.term = v;
      }
    }
  } else if (r.kind == kWalMetadata) {
    if (d_len > 0xffffffffull) return false;
    r.data_off = d_len ? base + d_off : 0;
    r.data_len = (uint32_t)d_len;
  }
  return true;
}

// ---- CRC-32C, eight bytes per step (slicing-by-8) ------------------------------------------------------------
// t[k][i] (k = 0..7, 256 words each): the raw register after byte i and k further zero bytes.  The byte-at-a-time update is
// a chain of dependent table reads -- on the device one LDS round trip per byte, 64 lanes hitting 64 random words of one
// 1 KB table: round 2's WAL decoder spent most of its 25.6 us there; eight independent reads per eight bytes put one round
// trip where there were eight.  Same arithmetic on the host (tests/test_wire_parse_host.py checks it against the oracle's CRC).
constexpr int kCrcTabs = 8;
RAFTQ_HD inline void crc_tables_entry(uint32_t i, uint32_t* col /*[kCrcTabs], stride 256*/) {
  uint32_t c = i;
  for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82f63b78u & (0u - (c & 1u)));
  col[0] = c;
}
RAFTQ_HD inline uint32_t crc_step1(const uint32_t* t, uint32_t raw, uint8_t b) { return t[(raw ^ b) & 0xffu] ^ (raw >> 8); }
RAFTQ_HD inline uint32_t crc_step8(const uint32_t* t, uint32_t raw, uint64_t w) {
  const uint32_t lo = (uint32_t)w ^ raw, hi = (uint32_t)(w >> 32);
  return t[7 * 256 + (lo & 0xffu)] ^ t[6 * 256 + ((lo >> 8) & 0xffu)] ^ t[5 * 256 + ((lo >> 16) & 0xffu)] ^ t[4 * 256 + (lo >> 24)] ^
         t[3 * 256 + (hi & 0xffu)] ^ t[2 * 256 + ((hi >> 8) & 0xffu)] ^ t[1 * 256 + ((hi >> 16) & 0xffu)] ^ t[hi >> 24];
}
// the raw register over n bytes at offset `off` of what `src` reads: whole 8-byte windows, then the tail out of one more
// window (a staged frame: bytes behind the data are inside the stage) or byte by byte (the buffer may end with the data)
RAFTQ_HD inline uint32_t crc_span8(const uint32_t* t, uint32_t raw, const ByteSrc& src, uint64_t off, uint64_t n) {
  uint64_t i = 0;
  for (; i + 8 <= n; i += 8) raw = crc_step8(t, raw, src.ld8(off + i));
  if (i < n) {
    bool window = false;
#if defined(__HIP_DEVICE_COMPILE__)
    window = src.words != nullptr;
#endif
    if (window) {
      uint64_t w = src.ld8(off + i);
      for (; i < n; ++i, w >>= 8) raw = crc_step1(t, raw, (uint8_t)w);
    } else {
      for (; i < n; ++i) raw = crc_step1(t, raw, src.p[off + i]);
    }
  }
  return raw;
}
// host-side table build (the device builds the same table in LDS, raftq_wire_kernels.hpp crc_table_init)
inline void crc_tables_build(uint32_t* t /*[kCrcTabs * 256]*/) {
  for (uint32_t i = 0; i < 256; ++i) crc_tables_entry(i, t + i);
  for (int k = 1; k < kCrcTabs; ++k)
    for (uint32_t i = 0; i < 256; ++i) t[k * 256 + i] = (t[(k - 1) * 256 + i] >> 8) ^ t[t[(k - 1) * 256 + i] & 0xffu];
}

// frame extent + length word; body = [a + 8, b)
RAFTQ_HD inline bool frame_body(const uint8_t* buf, uint64_t nbytes, uint64_t a, uint64_t b, bool big_endian) {
  if (!(a <= b && b <= nbytes && b - a >= 8)) return false;
  uint64_t w = load_u64(buf + a);
  if (big_endian) w = __builtin_bswap64(w);
  return w == b - a - 8;
}

}  // namespace raftqk
