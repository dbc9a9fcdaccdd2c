/*
 * raftq_wire_oracle.c -- CPU restatement of the wire / WAL codecs of include/raftq_wire.h.
 * TEST INFRASTRUCTURE ONLY (see raftq_oracle.h): never linked into libraftq.so; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg call it.
 *
 * The reference reaches these formats through
 *     rc.transport.Send(rd.Messages)           raft.go:230  (rafthttp: raftpb.Message.Marshal)
 *     rc.wal.Save(rd.HardState, rd.Entries)    raft.go:228  (wal.encoder.encode)
 *     w.ReadAll()                              raft.go:124  (wal.decoder.decode + Record.Validate)
 * and all of it lives in the un-vendored module github.com/coreos/etcd:
 *     raft/raftpb/raft.pb.go   Message / Entry / HardState / Snapshot  MarshalTo + Unmarshal
 *     wal/walpb/record.pb.go   Record / Snapshot
 *     wal/encoder.go, wal/decoder.go, wal/wal.go (Save, ReadAll, saveCrc)
 *     rafthttp/msg_codec.go    messageEncoder.encode / messageDecoder.decode
 *
 * PINNING.  The field numbers and the "write every non-nullable field" behaviour are recalled
 * from the 2015-era generated code (schema PARITY UNPINNED: no .proto on this machine).  What IS
 * pinned, by independent third-party code and published vectors:
 *   - every encoder output byte == google.protobuf (python runtime 7.x) serialising the same
 *     schema with all fields set (tests/golden/make_wire_golden.py -> tests/golden/wire_golden.json,
 *     and live in tests/test_wire_oracle.py);
 *   - every decoder result == the same runtime's ParseFromString, incl. shuffled field order,
 *     unknown fields, multi-byte varints;
 *   - CRC-32C == RFC 3720 appendix B.4 vectors + "123456789" -> e3069283, bitwise and
 *     table-driven implementations agreeing, and the combine identity
 *     crc(A||B) = crc(A) * x^(8|B|) + crc(B) that the GPU scan relies on.
 *
 * Shape: one message / record at a time, append-to-buffer, the shape of the Go original
 * (MarshalTo into a byte slice; Unmarshal's `for iNdEx < l` loop), not of the GPU kernels.
 */
#include <stdlib.h>
#include <string.h>

#include "raftq_oracle.h"
#include "raftq_wire.h"

/* ---- CRC-32C (Castagnoli), hash/crc32: reflected polynomial 0x82f63b78 -------------------- */

/* crc32.Update(crc, castagnoliTable, p) -- bit at a time, straight from the definition */
uint32_t rq_crc32c_update(uint32_t crc, const uint8_t* p, size_t n) {
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) {
    crc ^= p[i];
    for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0x82f63b78u & (0u - (crc & 1u)));
  }
  return ~crc;
}

/* second implementation: crc32.MakeTable + the simple table loop (crc32.update) */
uint32_t rq_crc32c_update_table(uint32_t crc, const uint8_t* p, size_t n) {
  static uint32_t tab[256];
  static int have;
  if (!have) {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ 0x82f63b78u : c >> 1;
      tab[i] = c;
    }
    have = 1;
  }
  crc = ~crc;
  for (size_t i = 0; i < n; ++i) crc = tab[(uint8_t)crc ^ p[i]] ^ (crc >> 8);
  return ~crc;
}

/* which of the two the WAL functions below use: 0 = bitwise (the definition; what the parity tests
 * run), 1 = table-driven (bench.py's cpu_baseline leg: Go's hash/crc32 is table / SSE4.2 driven, a
 * bit-at-a-time loop would be a strawman) */
static int g_crc_fast;
void rq_wire_set_fast_crc(int on) { g_crc_fast = on; }
static uint32_t crc_update(uint32_t crc, const uint8_t* p, size_t n) {
  return g_crc_fast ? rq_crc32c_update_table(crc, p, n) : rq_crc32c_update(crc, p, n);
}

/* a(x) * b(x) mod P in the reflected representation (bit 31 = x^0); zlib's multmodp */
uint32_t rq_crc32c_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (uint32_t m = 0x80000000u; m; m >>= 1) {
    if (a & m) p ^= b;
    b = (b & 1u) ? (b >> 1) ^ 0x82f63b78u : b >> 1;
  }
  return p;
}

/* x^(8 n) mod P */
uint32_t rq_crc32c_xpow8(uint64_t n) {
  uint32_t sq = 0x00800000u; /* x^8 */
  uint32_t r = 0x80000000u;  /* x^0 */
  for (; n; n >>= 1) {
    if (n & 1u) r = rq_crc32c_mulmod(r, sq);
    sq = rq_crc32c_mulmod(sq, sq);
  }
  return r;
}

/* crc(A || B) from crc(A), crc(B), |B| -- what chains Record.crc across records */
uint32_t rq_crc32c_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
  return rq_crc32c_mulmod(rq_crc32c_xpow8(len_b), crc_a) ^ crc_b;
}

/* ---- protobuf primitives (encodeVarintRaft / sovRaft / the Unmarshal varint loop) ---------- */

typedef struct {
  uint8_t* p;
  size_t n, cap;
  int overflow;
} buf_t;

static void put(buf_t* b, const void* src, size_t k) {
  if (b->n + k > b->cap) {
    b->overflow = 1;
    b->n += k; /* keep counting: the caller learns the size it needs */
    return;
  }
  if (k) memcpy(b->p + b->n, src, k); /* (k == 0 may come with src == NULL: an empty payload out of an empty pool) */
  b->n += k;
}
static void put_byte(buf_t* b, uint8_t v) { put(b, &v, 1); }
static void put_varint(buf_t* b, uint64_t v) {
  while (v >= 0x80) {
    put_byte(b, (uint8_t)(v | 0x80));
    v >>= 7;
  }
  put_byte(b, (uint8_t)v);
}
static size_t sov(uint64_t v) {
  size_t n = 1;
  while (v >= 0x80) {
    v >>= 7;
    ++n;
  }
  return n;
}
static void put_field(buf_t* b, uint8_t tag, uint64_t v) {
  put_byte(b, tag);
  put_varint(b, v);
}

/* returns bytes consumed, 0 = malformed (truncated, or > 10 bytes: gogoproto's ErrIntOverflow) */
static size_t get_varint(const uint8_t* p, size_t n, uint64_t* v) {
  uint64_t r = 0;
  for (size_t i = 0; i < n && i < 10; ++i) {
    r |= (uint64_t)(p[i] & 0x7f) << (7 * i); /* the 10th byte's high bits fall off, as in Go */
    if (p[i] < 0x80) {
      *v = r;
      return i + 1;
    }
  }
  return 0;
}

/* skipRaft: step over one unknown field's value.  0 = malformed. */
static size_t skip_value(const uint8_t* p, size_t n, unsigned wt) {
  uint64_t v;
  size_t k;
  switch (wt) {
    case 0: return get_varint(p, n, &v);
    case 1: return n >= 8 ? 8 : 0;
    case 2:
      k = get_varint(p, n, &v);
      if (!k || v > n - k) return 0;
      return k + (size_t)v;
    case 5: return n >= 4 ? 4 : 0;
    default: return 0; /* groups (3, 4) and 6, 7 */
  }
}

/* ---- raftpb.Entry / Message --------------------------------------------------------------- */

static size_t entry_size(const raftq_wire_ent_t* e, int with_group, uint64_t group) {
  size_t n = 1 + sov(e->type) + 1 + sov(e->term) + 1 + sov(e->index);
  if (e->data_len) n += 1 + sov(e->data_len) + e->data_len;
  if (with_group) n += 1 + sov(group);
  return n;
}
/* Entry.MarshalTo */
static void entry_marshal(buf_t* b, const raftq_wire_ent_t* e, const uint8_t* pool, int with_group, uint64_t group) {
  put_field(b, 0x08, e->type);
  put_field(b, 0x10, e->term);
  put_field(b, 0x18, e->index);
  if (e->data_len) { /* `if m.Data != nil` -- an empty payload is the nil of becomeLeader's entry */
    put_field(b, 0x22, e->data_len);
    put(b, pool + e->data_off, e->data_len);
  }
  if (with_group) put_field(b, 0x28, group);
}

static const uint8_t kEmptySnapshot[10] = {0x4a, 0x08, 0x12, 0x06, 0x0a, 0x00, 0x10, 0x00, 0x18, 0x00};

static size_t msg_body_size(const raftq_wire_msg_t* m, const raftq_wire_ent_t* ents) {
  size_t n = 1 + sov(m->type) + 1 + sov((uint64_t)m->to + 1) + 1 + sov((uint64_t)m->from + 1) + 1 + sov(m->term) + 1 +
             sov(m->log_term) + 1 + sov(m->index);
  for (uint32_t k = 0; k < m->n_ents; ++k) {
    const size_t es = entry_size(&ents[m->ent_first + k], 0, 0);
    n += 1 + sov(es) + es;
  }
  n += 1 + sov(m->commit) + sizeof kEmptySnapshot + 2 + 1 + sov(m->reject_hint) + 1 + sov(m->group);
  return n;
}
/* Message.MarshalTo, fields in number order, every non-nullable one written */
static void msg_marshal(buf_t* b, const raftq_wire_msg_t* m, const raftq_wire_ent_t* ents, const uint8_t* pool) {
  put_field(b, 0x08, m->type);
  put_field(b, 0x10, (uint64_t)m->to + 1);
  put_field(b, 0x18, (uint64_t)m->from + 1);
  put_field(b, 0x20, m->term);
  put_field(b, 0x28, m->log_term);
  put_field(b, 0x30, m->index);
  for (uint32_t k = 0; k < m->n_ents; ++k) {
    const raftq_wire_ent_t* e = &ents[m->ent_first + k];
    put_field(b, 0x3a, entry_size(e, 0, 0));
    entry_marshal(b, e, pool, 0, 0);
  }
  put_field(b, 0x40, m->commit);
  put(b, kEmptySnapshot, sizeof kEmptySnapshot);
  put_field(b, 0x50, m->reject ? 1 : 0);
  put_field(b, 0x58, m->reject_hint);
  put_field(b, 0x60, m->group);
}

/* messageEncoder.encode per message: binary.Write(w, binary.BigEndian, uint64(m.Size())); w.Write(m.Marshal()) */
uint64_t rq_wire_encode(const raftq_wire_msg_t* msgs, uint64_t n, const raftq_wire_ent_t* ents, const uint8_t* pool,
                        uint8_t* out, uint64_t cap, uint64_t* frame_off) {
  buf_t b = {out, 0, (size_t)cap, 0};
  for (uint64_t i = 0; i < n; ++i) {
    if (frame_off) frame_off[i] = b.n;
    const uint64_t sz = msg_body_size(&msgs[i], ents);
    uint8_t be[8];
    for (int k = 0; k < 8; ++k) be[k] = (uint8_t)(sz >> (56 - 8 * k));
    put(&b, be, 8);
    const size_t before = b.n;
    msg_marshal(&b, &msgs[i], ents, pool);
    if (b.n - before != sz) abort(); /* Size() and MarshalTo must agree */
  }
  if (frame_off) frame_off[n] = b.n;
  return b.n;
}

/* is this Snapshot message non-empty?  -1 = malformed.
 * depth 0: Snapshot{1 data (bytes), 2 metadata (message)}
 * depth 1: SnapshotMetadata{1 conf_state (message: non-empty iff it has bytes), 2 index, 3 term} */
static int snapshot_nonempty(const uint8_t* p, size_t n, int depth) {
  size_t i = 0;
  int nonempty = 0;
  while (i < n) {
    uint64_t key, v;
    size_t k = get_varint(p + i, n - i, &key);
    if (!k) return -1;
    i += k;
    const unsigned wt = (unsigned)(key & 7);
    const uint64_t fn = key >> 3;
    if (fn == 0) return -1;
    const int known = depth == 0 ? fn <= 2 : fn <= 3;
    if (!known) {
      k = skip_value(p + i, n - i, wt);
      if (!k) return -1;
      i += k;
      continue;
    }
    const int is_len = depth == 0 || fn == 1;
    if (wt != (is_len ? 2u : 0u)) return -1;
    k = get_varint(p + i, n - i, &v);
    if (!k) return -1;
    i += k;
    if (!is_len) {
      if (v) nonempty = 1;
      continue;
    }
    if (v > n - i) return -1;
    if (depth == 0 && fn == 2) {
      const int r = snapshot_nonempty(p + i, (size_t)v, 1);
      if (r < 0) return -1;
      nonempty |= r;
    } else if (v) {
      nonempty = 1; /* snapshot data, or a conf_state with members */
    }
    i += (size_t)v;
  }
  return nonempty;
}

/* Entry.Unmarshal; `base` = offset of p[0] in the enclosing buffer (data_off is relative to that buffer) */
static int entry_unmarshal(const uint8_t* p, size_t n, uint64_t base, raftq_wire_ent_t* e, uint64_t* group, int* has_group) {
  memset(e, 0, sizeof *e);
  size_t i = 0;
  while (i < n) {
    uint64_t key, v;
    size_t k = get_varint(p + i, n - i, &key);
    if (!k) return -1;
    i += k;
    const unsigned wt = (unsigned)(key & 7);
    const uint64_t fn = key >> 3;
    if (fn == 0) return -1;
    if (fn >= 1 && fn <= 3) {
      if (wt != 0) return -1;
      k = get_varint(p + i, n - i, &v);
      if (!k) return -1;
      i += k;
      if (fn == 1) e->type = (uint32_t)v;
      if (fn == 2) e->term = v;
      if (fn == 3) e->index = v;
    } else if (fn == 4) {
      if (wt != 2) return -1;
      k = get_varint(p + i, n - i, &v);
      if (!k || v > n - i - k || v > 0xffffffffu) return -1;
      i += k;
      e->data_off = base + i;
      e->data_len = (uint32_t)v;
      i += (size_t)v;
    } else if (fn == 5 && group) {
      if (wt != 0) return -1;
      k = get_varint(p + i, n - i, &v);
      if (!k) return -1;
      i += k;
      *group = v;
      *has_group = 1;
    } else {
      k = skip_value(p + i, n - i, wt);
      if (!k) return -1;
      i += k;
    }
  }
  if (e->data_len == 0) e->data_off = 0;
  return 0;
}

static uint32_t id_to_slot(uint64_t id, uint32_t none) { return id == 0 || id - 1 >= none ? none : (uint32_t)(id - 1); }

/* Message.Unmarshal.  ents == NULL: count entries only.  Returns 0, or -1 = malformed. */
static int msg_unmarshal(const uint8_t* p, size_t n, uint64_t base, raftq_wire_msg_t* m, raftq_wire_ent_t* ents,
                         uint64_t ents_cap, uint64_t* n_ents_total) {
  memset(m, 0, sizeof *m);
  m->to = 0xff;
  m->from = 0xffffffffu;
  m->ent_first = (uint32_t)*n_ents_total;
  size_t i = 0;
  while (i < n) {
    uint64_t key, v;
    size_t k = get_varint(p + i, n - i, &key);
    if (!k) return -1;
    i += k;
    const unsigned wt = (unsigned)(key & 7);
    const uint64_t fn = key >> 3;
    if (fn == 0) return -1; /* "illegal tag 0" */
    const int is_len = fn == 7 || fn == 9;
    if (fn >= 1 && fn <= 12) {
      if (wt != (is_len ? 2u : 0u)) return -1; /* "wrong wireType" */
      k = get_varint(p + i, n - i, &v);
      if (!k) return -1;
      i += k;
      if (is_len && v > n - i) return -1; /* io.ErrUnexpectedEOF */
      switch (fn) {
        case 1: m->type = (uint32_t)v > 255 ? 255 : (uint8_t)v; break; /* int32: low 32 bits */
        case 2: m->to = (uint8_t)id_to_slot(v, 0xff); break;
        case 3: m->from = id_to_slot(v, 0xffffffffu); break;
        case 4: m->term = v; break;
        case 5: m->log_term = v; break;
        case 6: m->index = v; break;
        case 7: {
          raftq_wire_ent_t e;
          if (entry_unmarshal(p + i, (size_t)v, base + i, &e, NULL, NULL)) return -1;
          if (ents && *n_ents_total < ents_cap) ents[*n_ents_total] = e;
          ++*n_ents_total;
          ++m->n_ents;
          i += (size_t)v;
          break;
        }
        case 8: m->commit = v; break;
        case 9: {
          const int r = snapshot_nonempty(p + i, (size_t)v, 0);
          if (r < 0) return -1;
          if (r) m->flags |= RAFTQ_WIRE_F_SNAPSHOT;
          i += (size_t)v;
          break;
        }
        case 10: m->reject = v != 0; break;
        case 11: m->reject_hint = v; break;
        case 12:
          m->group = v;
          m->flags |= RAFTQ_WIRE_F_GROUP;
          break;
      }
    } else {
      k = skip_value(p + i, n - i, wt);
      if (!k) return -1;
      i += k;
    }
  }
  if (m->n_ents == 0) m->ent_first = 0;
  return 0;
}

/* messageDecoder.decode per frame.  Returns 0; *n_ents / *n_bad totals.  ents may be NULL. */
int rq_wire_decode(const uint8_t* stream, uint64_t nbytes, const uint64_t* frame_off, uint64_t n, raftq_wire_msg_t* msgs,
                   raftq_wire_ent_t* ents, uint64_t ents_cap, uint64_t* n_ents, uint64_t* n_bad) {
  uint64_t total = 0, bad = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t a = frame_off[i], b = frame_off[i + 1];
    int ok = a <= b && b <= nbytes && b - a >= 8;
    if (ok) {
      uint64_t len = 0;
      for (int k = 0; k < 8; ++k) len = len << 8 | stream[a + k];
      ok = len == b - a - 8;
    }
    const uint64_t before = total;
    if (ok) ok = msg_unmarshal(stream + a + 8, (size_t)(b - a - 8), a + 8, &msgs[i], ents, ents_cap, &total) == 0;
    if (!ok) {
      memset(&msgs[i], 0, sizeof msgs[i]);
      msgs[i].flags = RAFTQ_WIRE_F_MALFORMED;
      total = before; /* a malformed frame contributes no entries */
      ++bad;
    }
  }
  *n_ents = total;
  *n_bad = bad;
  return 0;
}

/* the length-word walk both decoders start with */
int rq_wire_scan_frames(const uint8_t* buf, uint64_t nbytes, int big_endian, uint64_t* off, uint64_t cap,
                        uint64_t* n_frames, uint64_t* consumed) {
  uint64_t pos = 0, k = 0;
  while (k < cap && nbytes - pos >= 8) {
    uint64_t len = 0;
    for (int b = 0; b < 8; ++b) len |= (uint64_t)buf[pos + b] << (big_endian ? 56 - 8 * b : 8 * b);
    if (len > nbytes - pos - 8) break; /* torn tail (io.ErrUnexpectedEOF) */
    off[k++] = pos;
    pos += 8 + len;
  }
  off[k] = pos;
  *n_frames = k;
  *consumed = pos;
  return 0;
}

/* ---- WAL ----------------------------------------------------------------------------------- */

static size_t wal_data_size(const raftq_wal_rec_t* r) {
  raftq_wire_ent_t e;
  switch (r->kind) {
    case RAFTQ_WAL_ENTRY:
      e.type = r->entry_type, e.term = r->term, e.index = r->index, e.data_off = r->data_off, e.data_len = r->data_len;
      return entry_size(&e, 1, r->group);
    case RAFTQ_WAL_STATE: return 1 + sov(r->term) + 1 + sov(r->vote) + 1 + sov(r->index) + 1 + sov(r->group);
    case RAFTQ_WAL_SNAPSHOT: return 1 + sov(r->index) + 1 + sov(r->term);
    case RAFTQ_WAL_METADATA: return r->data_len;
    default: return 0;
  }
}
static void wal_data_marshal(buf_t* b, const raftq_wal_rec_t* r, const uint8_t* pool) {
  raftq_wire_ent_t e;
  switch (r->kind) {
    case RAFTQ_WAL_ENTRY: /* saveEntry: pbutil.MustMarshal(e) */
      e.type = r->entry_type, e.term = r->term, e.index = r->index, e.data_off = r->data_off, e.data_len = r->data_len;
      entry_marshal(b, &e, pool, 1, r->group);
      break;
    case RAFTQ_WAL_STATE: /* saveState: pbutil.MustMarshal(s) */
      put_field(b, 0x08, r->term);
      put_field(b, 0x10, r->vote);
      put_field(b, 0x18, r->index);
      put_field(b, 0x20, r->group);
      break;
    case RAFTQ_WAL_SNAPSHOT: /* SaveSnapshot: walpb.Snapshot{Index, Term} */
      put_field(b, 0x08, r->index);
      put_field(b, 0x10, r->term);
      break;
    case RAFTQ_WAL_METADATA: put(b, pool + r->data_off, r->data_len); break;
    default: break;
  }
}

/* wal.Save record by record through encoder.encode:
 *     e.crc.Write(rec.Data); rec.Crc = e.crc.Sum32(); data := rec.Marshal()
 *     writeInt64(e.bw, int64(len(data))); e.bw.Write(data) */
uint64_t rq_wal_encode(const raftq_wal_rec_t* recs, uint64_t n, const uint8_t* pool, uint32_t prev_crc, uint8_t* out,
                       uint64_t cap, uint64_t* frame_off, uint32_t* last_crc) {
  buf_t b = {out, 0, (size_t)cap, 0};
  uint32_t crc = prev_crc;
  size_t scratch_cap = 1 << 16;
  uint8_t* scratch = malloc(scratch_cap);
  for (uint64_t i = 0; i < n; ++i) {
    const raftq_wal_rec_t* r = &recs[i];
    if (frame_off) frame_off[i] = b.n;
    const size_t dsz = wal_data_size(r);
    if (dsz > scratch_cap) {
      scratch_cap = dsz * 2;
      scratch = realloc(scratch, scratch_cap);
    }
    buf_t d = {scratch, 0, scratch_cap, 0};
    wal_data_marshal(&d, r, pool);
    if (d.n != dsz) abort();
    crc = crc_update(crc, scratch, dsz);
    /* Record.MarshalTo: 08 type, 10 crc, `if m.Data != nil` 1a len data */
    const int has_data = r->kind != RAFTQ_WAL_CRC && !(r->kind == RAFTQ_WAL_METADATA && dsz == 0);
    const uint64_t rsz = 1 + sov(r->kind) + 1 + sov(crc) + (has_data ? 1 + sov(dsz) + dsz : 0);
    uint8_t le[8];
    for (int k = 0; k < 8; ++k) le[k] = (uint8_t)(rsz >> (8 * k));
    put(&b, le, 8);
    put_field(&b, 0x08, r->kind);
    put_field(&b, 0x10, crc);
    if (has_data) {
      put_field(&b, 0x1a, dsz);
      put(&b, scratch, dsz);
    }
  }
  free(scratch);
  if (frame_off) frame_off[n] = b.n;
  if (last_crc) *last_crc = crc;
  return b.n;
}

/* Record.Unmarshal + the per-type Data unmarshal */
static int wal_rec_unmarshal(const uint8_t* p, size_t n, uint64_t base, raftq_wal_rec_t* r, uint64_t* d_off,
                             uint64_t* d_len) {
  memset(r, 0, sizeof *r);
  *d_off = *d_len = 0;
  size_t i = 0;
  uint64_t type = 0;
  while (i < n) {
    uint64_t key, v;
    size_t k = get_varint(p + i, n - i, &key);
    if (!k) return -1;
    i += k;
    const unsigned wt = (unsigned)(key & 7);
    const uint64_t fn = key >> 3;
    if (fn == 0) return -1;
    if (fn == 1 || fn == 2) {
      if (wt != 0) return -1;
      k = get_varint(p + i, n - i, &v);
      if (!k) return -1;
      i += k;
      if (fn == 1) type = v;
      else r->crc = (uint32_t)v;
    } else if (fn == 3) {
      if (wt != 2) return -1;
      k = get_varint(p + i, n - i, &v);
      if (!k || v > n - i - k) return -1;
      i += k;
      *d_off = i;
      *d_len = v;
      i += (size_t)v;
    } else {
      k = skip_value(p + i, n - i, wt);
      if (!k) return -1;
      i += k;
    }
  }
  if (type < 1 || type > 5) return -1; /* ReadAll: "unexpected block type" */
  r->kind = (uint8_t)type;
  const uint8_t* d = p + *d_off;
  const size_t dn = (size_t)*d_len;
  if (r->kind == RAFTQ_WAL_ENTRY) {
    raftq_wire_ent_t e;
    int hg = 0;
    if (entry_unmarshal(d, dn, base + *d_off, &e, &r->group, &hg)) return -1;
    r->term = e.term, r->index = e.index, r->data_off = e.data_off, r->data_len = e.data_len;
    r->entry_type = (uint8_t)e.type;
    if (hg) r->flags |= RAFTQ_WAL_F_GROUP;
  } else if (r->kind == RAFTQ_WAL_STATE || r->kind == RAFTQ_WAL_SNAPSHOT) {
    size_t j = 0;
    while (j < dn) {
      uint64_t key, v;
      size_t k = get_varint(d + j, dn - j, &key);
      if (!k) return -1;
      j += k;
      const unsigned wt = (unsigned)(key & 7);
      const uint64_t fn = key >> 3;
      if (fn == 0) return -1;
      const uint64_t known = r->kind == RAFTQ_WAL_STATE ? 4 : 2;
      if (fn <= known) {
        if (wt != 0) return -1;
        k = get_varint(d + j, dn - j, &v);
        if (!k) return -1;
        j += k;
        if (r->kind == RAFTQ_WAL_STATE) {
          if (fn == 1) r->term = v;
          if (fn == 2) r->vote = (uint32_t)v;
          if (fn == 3) r->index = v;
          if (fn == 4) r->group = v, r->flags |= RAFTQ_WAL_F_GROUP;
        } else {
          if (fn == 1) r->index = v;
          if (fn == 2) r->term = v;
        }
      } else {
        k = skip_value(d + j, dn - j, wt);
        if (!k) return -1;
        j += k;
      }
    }
  } else if (r->kind == RAFTQ_WAL_METADATA) {
    if (dn > 0xffffffffu) return -1;
    r->data_off = dn ? base + *d_off : 0;
    r->data_len = (uint32_t)dn;
  }
  return 0;
}

/* ReadAll's loop: decode, then per type; crcType re-seeds the chain:
 *     crc := decoder.crc.Sum32()
 *     if crc != 0 && rec.Validate(crc) != nil { ErrCRCMismatch }
 *     decoder.updateCRC(rec.Crc)
 * every other record: decoder.crc.Write(rec.Data); rec.Validate(decoder.crc.Sum32()) */
int rq_wal_decode(const uint8_t* bytes, uint64_t nbytes, const uint64_t* frame_off, uint64_t n, uint32_t prev_crc,
                  raftq_wal_rec_t* recs, uint64_t* n_valid, uint32_t* last_crc) {
  uint32_t crc = prev_crc, crc_valid = prev_crc; /* crc_valid: the chain after the last record before the first bad one */
  uint64_t first_bad = n;
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t a = frame_off[i], b = frame_off[i + 1];
    int ok = a <= b && b <= nbytes && b - a >= 8;
    if (ok) {
      uint64_t len = 0;
      for (int k = 0; k < 8; ++k) len |= (uint64_t)bytes[a + k] << (8 * k);
      ok = len == b - a - 8;
    }
    uint64_t d_off = 0, d_len = 0;
    if (ok) ok = wal_rec_unmarshal(bytes + a + 8, (size_t)(b - a - 8), a + 8, &recs[i], &d_off, &d_len) == 0;
    if (!ok) {
      memset(&recs[i], 0, sizeof recs[i]);
      recs[i].flags = RAFTQ_WAL_F_MALFORMED;
      if (first_bad == n) first_bad = i;
      continue; /* the chain cannot be followed through a record that does not parse: it skips it */
    }
    if (recs[i].kind == RAFTQ_WAL_CRC) {
      if (crc != 0 && recs[i].crc != crc) recs[i].flags |= RAFTQ_WAL_F_BADCRC;
      crc = recs[i].crc;
    } else {
      crc = crc_update(crc, bytes + a + 8 + d_off, (size_t)d_len);
      if (recs[i].crc != crc) recs[i].flags |= RAFTQ_WAL_F_BADCRC;
    }
    if ((recs[i].flags & RAFTQ_WAL_F_BADCRC) && first_bad == n) first_bad = i;
    if (first_bad == n) crc_valid = crc;
  }
  if (last_crc) *last_crc = crc_valid;
  *n_valid = first_bad;
  return 0;
}
