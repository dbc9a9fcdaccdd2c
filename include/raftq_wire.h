/*
 * raftq_wire.h -- C-ABI of the batched wire / WAL codecs: the
 * data formats either side of the quorum path.
 *
 * The reference moves two byte formats around its one raft group, both owned by
 * the un-vendored etcd dependency:
 *     rc.transport.Send(rd.Messages)        raft.go:230   raftpb.Message over rafthttp
 *     rc.wal.Save(rd.HardState, rd.Entries) raft.go:228   walpb.Record frames, CRC-32C chained
 *     w.ReadAll() in replayWAL              raft.go:122-134
 * With G groups per process those are per-message / per-record marshal calls in
 * G goroutines.  The entry points below do a whole batch on the GPU: protobuf
 * field encoding / parsing one lane per message, CRC-32C one lane per record
 * with the chain resolved by a parallel scan over (crc, x^(8 len)) pairs.
 *
 * FORMATS (restated from the published 2015-era etcd sources, v2.2-v2.3 line;
 * PARITY UNPINNED for the schema -- the .proto files are not on this machine --
 * but every byte produced is checked against the protobuf runtime on that schema
 * and against RFC 3720's CRC-32C vectors, see oracle/raftq_wire_oracle.c):
 *
 *   raftpb.Message   1 type  2 to  3 from  4 term  5 logTerm  6 index  7 entries*
 *                    8 commit  9 snapshot  10 reject  11 rejectHint  [12 group]
 *   raftpb.Entry     1 Type  2 Term  3 Index  4 Data  [5 group, WAL only]
 *   raftpb.HardState 1 term  2 vote  3 commit  [4 group]
 *   walpb.Record     1 type  2 crc  3 data       walpb.Snapshot  1 index  2 term
 *
 *   [n group] is this library's one extension: etcd runs a single group per
 *   process, so its messages carry none.  It is an ordinary varint field with an
 *   unused number, inside the CRC-covered bytes; a stock decoder skips it.
 *
 *   Encoding is canonical gogoproto (`nullable=false`): every scalar field is
 *   written, zero or not, in field order; Message.snapshot is always written
 *   (empty: 4a 08 12 06 0a 00 10 00 18 00); Entry.Data is omitted when empty.
 *   Decoding is ordinary protobuf: any field order, last scalar wins, unknown
 *   fields skipped, wrong wire type on a known field = malformed.
 *
 *   stream frame (rafthttp messageEncoder):  u64 BIG-endian length | Message
 *   WAL frame (wal/encoder.go, pre-3.0: no padding):  i64 LITTLE-endian length | Record
 *   Record.crc = crc32.Update(previous record's crc, castagnoli, Record.data)
 *   -- the running CRC of all Data bytes since the segment's crcType record.
 *
 * raft IDs on the wire are 1-based (raft.go:148-151); the structs carry peer
 * SLOTS (ID - 1) like the rest of this ABI.  An absent / zero ID decodes to
 * slot 0xFF (to) / 0xFFFFFFFF (from), which raftq_step_* rejects.
 *
 * No CPU path: all entry points need the handle's GPU.
 *
 * How a call moves its bytes.  When every array of a call is page-locked (hipHostMalloc /
 * hipHostRegister, or the handle's own staging areas) and 16-byte aligned, the call is ONE
 * kernel that reads the inputs and writes the outputs where they lie, both directions of
 * the link busy at once ("streaming form").  Otherwise (pageable memory, odd alignment, or
 * RAFTQ_WIRE_STREAMING=0 in the environment) inputs are copied in, outputs copied out
 * ("copying form").  Same results, byte for byte.  One difference a caller can see: a
 * streaming ENCODE that is refused (RAFTQ_EINVAL: cap too small, a range out of bounds, a
 * bad slot) has already been writing, so out[0 .. cap) and frame_off are unspecified after
 * it -- nothing at or behind out[cap] is ever touched; the copying form leaves out alone.
 * A streaming DECODE that finds more entries than ents_cap has likewise filled msgs and the
 * first ents_cap entry headers before it says so.
 */
#ifndef RAFTQ_WIRE_H
#define RAFTQ_WIRE_H

#include "raftq_step.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RAFTQ_MSG_SNAP 7 /* decoded, never accepted by Step */
#define RAFTQ_MSG_PROP 2 /* decoded, never stepped: a follower forwarding a proposal to its leader -- the log owner's (raftq_step_frames holds it) */

/* raftq_wire_msg_t.flags */
#define RAFTQ_WIRE_F_MALFORMED 0x01u /* frame did not parse: every other field of the record is 0 */
#define RAFTQ_WIRE_F_SNAPSHOT 0x02u  /* a non-empty Message.snapshot was present (skipped) */
#define RAFTQ_WIRE_F_GROUP 0x04u     /* field 12 was present (absent = group 0, a stock etcd peer) */

/* One message header.  Same layout as raftq_msg_t (raftq_step.h) in the fields Step reads,
 * so a decoded array can be handed to raftq_step_submit unchanged. */
typedef struct raftq_wire_msg {
  uint64_t group;
  uint64_t term;
  uint64_t log_term;
  uint64_t index;
  uint64_t commit;
  uint64_t reject_hint;
  uint32_t from;      /* sender's peer slot */
  uint8_t type;       /* raftpb.MessageType; > 255 decodes to 255 */
  uint8_t reject;
  uint8_t to;         /* addressee's peer slot */
  uint8_t flags;      /* RAFTQ_WIRE_F_* (decode); raftq_step_frames adds RAFTQ_MSGF_* (0x10 .. 0x80) */
  uint32_t ent_first; /* this message's entries are ents[ent_first .. ent_first + n_ents) */
  uint32_t n_ents;
} raftq_wire_msg_t; /* 64 bytes */

/* One log entry.  data_off points into the payload pool (encode) or into the
 * decoded stream itself (decode: no payload byte is copied). */
typedef struct raftq_wire_ent {
  uint64_t term;
  uint64_t index;
  uint64_t data_off;
  uint32_t data_len;
  uint32_t type; /* raftpb.EntryType: 0 EntryNormal, 1 EntryConfChange */
} raftq_wire_ent_t; /* 32 bytes */

typedef struct raftq_wire_counts {
  uint64_t n_msgs;
  uint64_t n_ents;
  uint64_t n_malformed;
  uint64_t bytes; /* encoded / consumed */
} raftq_wire_counts_t;

/* Marshal n messages into rafthttp stream frames, in order.  frame_off (may be NULL) receives
 * n + 1 byte offsets into out; cap is out's size.  RAFTQ_EINVAL if cap is too small (counts->bytes
 * then says what is needed), if an entry range or payload range is out of bounds, or to/from >= 255. */
int raftq_wire_encode(raftq_t* h, const raftq_wire_msg_t* msgs, uint64_t n, const raftq_wire_ent_t* ents,
                      uint64_t n_ents, const void* pool, uint64_t pool_bytes, void* out, uint64_t cap,
                      uint64_t* frame_off /*[n+1]|NULL*/, raftq_wire_counts_t* counts /*|NULL*/);

/* Unmarshal n frames.  frame_off[i] .. frame_off[i+1] is frame i (its 8-byte length included) --
 * the receive loop knows the boundaries, it read the lengths to read the bodies.  A frame whose
 * length word disagrees with its extent, or whose body does not parse, is flagged MALFORMED and
 * counted; the call still succeeds.  ents receives the entry headers of all messages (message
 * order); RAFTQ_EINVAL if there are more than ents_cap (counts->n_ents says how many). */
int raftq_wire_decode(raftq_t* h, const void* stream, uint64_t nbytes, const uint64_t* frame_off /*[n+1]*/, uint64_t n,
                      raftq_wire_msg_t* msgs /*[n]*/, raftq_wire_ent_t* ents /*[ents_cap]|NULL*/, uint64_t ents_cap,
                      raftq_wire_counts_t* counts /*|NULL*/);

/* Walk the length words of a byte buffer (host side; pointer chasing, inherently serial):
 * off[0..n] for the n whole frames found.  big_endian = 1 for rafthttp streams, 0 for WAL files.
 * *n_frames = whole frames; *consumed = bytes they cover (a torn tail is left to the caller). */
int raftq_wire_scan_frames(const void* buf, uint64_t nbytes, int big_endian, uint64_t* off /*[cap+1]*/, uint64_t cap,
                           uint64_t* n_frames, uint64_t* consumed);

/* Step straight from the wire: decode on the device into the batch's message records (the 64-byte
 * records never cross PCIe), then exactly raftq_step_submit.  Frames that are malformed, or whose
 * type Step does not take, fail the batch at its collect like any malformed message.  Entry
 * headers of the batch (MsgApp) are available after the collect through raftq_step_wire_entries,
 * msgs[i].ent_first / n_ents through raftq_step_wire_msgs -- both in pinned memory, valid until
 * the next submit into that slot. */
int raftq_step_submit_wire(raftq_t* h, const void* stream, uint64_t nbytes, const uint64_t* frame_off, uint64_t n);
/* Zero-copy form: the arrays the NEXT raftq_step_submit_wire will take, with room for n_cap frames and nbytes_cap stream
 * bytes -- fine-grained device memory behind a large BAR (the receive path's stores, or a NIC's, land in HBM and the
 * frames are decoded where they lie: no DMA, no copy), pinned host memory otherwise; write-only for the host either
 * way.  Fill frame_off[0..n] and the stream, pass the SAME two pointers to raftq_step_submit_wire.  The arrays belong to
 * the library from that submit until the batch is collected; asking for a slot's arrays again ends the validity of
 * raftq_step_wire_msgs / _entries for the batch that was decoded in them. */
int raftq_step_stage_wire(raftq_t* h, uint64_t n_cap, uint64_t nbytes_cap, uint64_t** frame_off, void** stream);
int raftq_step_wire_msgs(raftq_t* h, const raftq_wire_msg_t** msgs, uint64_t* n);
int raftq_step_wire_entries(raftq_t* h, const raftq_wire_ent_t** ents, uint64_t* n_ents);

/* A node's inbound half of a turn -- rafthttp's decoder, the checks a node makes on what it received, rc.node.Step for every
 * message (raft.go:268-270) -- as ONE submission with one wait: raftq_wire_decode into msgs / ents, then raftq_step_batch over
 * ALL n frames in arrival order, the decoder saying in every record's flag byte what Step is to make of it (RAFTQ_MSGF_*,
 * raftq_step.h; the handle must have opted in, raftq_step_set_msg_flags):
 *   - a frame that did not parse, is of a kind a peer never sends (anything but MsgProp, MsgApp, MsgAppResp, MsgVote,
 *     MsgVoteResp, MsgHeartbeat, MsgHeartbeatResp), names a group >= G or a sender >= N, or is addressed to a slot other than
 *     the handle's own (raftq_set_self): RAFTQ_MSGF_SKIP -> RAFTQ_OUT_SKIPPED.  rafthttp would log and drop the stream; a node
 *     has to survive whatever bytes a peer throws at it.
 *   - MsgProp (appending is the log owner's): RAFTQ_MSGF_HOLD -> RAFTQ_OUT_HELD, the group's later frames RAFTQ_OUT_DEFERRED.
 *   - MsgApp: RAFTQ_MSGF_BARRIER, and with tail_appends != 0 RAFTQ_MSGF_ENTRIES (reject_hint, which a MsgApp does not use, is
 *     overwritten with the Term of its last entry) -- one that lands on the log's tail is RAFTQ_OUT_APPENDED.
 * Results: raftq_step_results (or _c), n records, out[i] answers frame i.  msgs / ents / counts as raftq_wire_decode, except
 * that more entries than ents_cap is NOT an error here -- the frames have been stepped; counts->n_ents says how many there
 * are, the first ents_cap are written, raftq_wire_decode fetches the rest.
 * Every array must be page-locked and 16-byte aligned (RAFTQ_EINVAL otherwise: make the two calls).  No batch may be in
 * flight.  raftq_node's turn is this call + one for what goes out. */
int raftq_step_frames(raftq_t* h, const void* stream, uint64_t nbytes, const uint64_t* frame_off /*[n+1]*/, uint64_t n, int tail_appends,
                      raftq_wire_msg_t* msgs /*[n]*/, raftq_wire_ent_t* ents /*[ents_cap]|NULL*/, uint64_t ents_cap,
                      raftq_wire_counts_t* counts /*|NULL*/);

/* A node's OUTBOUND half of a turn for what it was asked to propose (round 6; raft.go:211-215 -> :227-230: rc.node.Propose ->
 * appendEntry -> bcastAppend -> rc.transport.Send) as ONE submission with one wait: for every record of props[] the leader's
 * appendEntry (lastIndex += n_ents, lastTerm = Term, its own Progress.Match) on the device-resident state, and bcastAppend --
 * the N - 1 MsgApp{Term, LogTerm: old lastTerm, Index: old lastIndex, Commit, Entries} headers and their entry headers are
 * written INTO THE ENCODER'S INPUT IN HBM (they never exist in host memory); then raftq_wire_encode over msgs[] (what the
 * caller queued itself this turn: responses, resends, heartbeats) followed by those MsgApps.  What the host keeps of a
 * proposal is its payload bytes (in `pool`, where data_off points) and its own log.
 *
 * props[i]: a group THIS handle's node leads (role == leader; anything else fails the call), at most once per call, with
 * every follower's Progress.Next at the log's tail -- the state bcastAppend leaves behind, i.e. every group outside a
 * catch-up; the caller, who owns Progress.Next (raftq_step.h), sends the others itself -- and 1 <= n_ents entries
 * prop_ents[ent_first .. ent_first + n_ents) in log order (every record of prop_ents[] must name a payload inside the pool).  The new entries get Term = the group's Term, Index = old lastIndex
 * + 1 + k.  The handle needs more than one peer (with one the append commits: raftq_apply_log_deltas reports that).
 *
 * The stream: the frames of msgs[0 .. n_msgs), then for every peer slot p != self, ascending, the n_props MsgApps addressed to
 * p in props[] order -- frame_off (may be NULL) gets n_msgs + (N - 1) * n_props + 1 offsets; peer p's MsgApps are ONE slice.
 * Byte for byte what raftq_wire_encode makes of the same messages built on the host (tests/test_wire_gpu.py::
 * test_propose_frames_*).  Every array must be page-locked and 16-byte aligned (RAFTQ_EINVAL otherwise), no Step batch may be
 * in flight.  A call that fails has applied nothing (a validation kernel runs first); `out` is unspecified after a refusal, as
 * with the streaming raftq_wire_encode.  raftq_node's turn is raftq_step_frames + this. */
typedef struct raftq_prop {
  uint64_t group;
  uint32_t ent_first; /* into prop_ents[] */
  uint32_t n_ents;    /* >= 1 */
} raftq_prop_t; /* 16 bytes */
typedef struct raftq_prop_ent {
  uint64_t data_off; /* byte offset of Entry.Data in pool; ignored when data_len == 0 */
  uint32_t data_len;
  uint32_t type; /* raftpb.EntryType */
} raftq_prop_ent_t; /* 16 bytes */
int raftq_propose_frames(raftq_t* h, const raftq_prop_t* props, uint64_t n_props, const raftq_prop_ent_t* prop_ents, uint64_t n_prop_ents,
                         const raftq_wire_msg_t* msgs, uint64_t n_msgs, const raftq_wire_ent_t* ents, uint64_t n_ents, const void* pool,
                         uint64_t pool_bytes, void* out, uint64_t cap, uint64_t* frame_off /*[n_msgs + (N-1) n_props + 1]|NULL*/,
                         raftq_wire_counts_t* counts /*|NULL*/);

/* ---- WAL ------------------------------------------------------------------------------------ */

/* walpb record types (wal/wal.go) */
#define RAFTQ_WAL_METADATA 1
#define RAFTQ_WAL_ENTRY 2
#define RAFTQ_WAL_STATE 3
#define RAFTQ_WAL_CRC 4
#define RAFTQ_WAL_SNAPSHOT 5

#define RAFTQ_WAL_F_MALFORMED 0x01u /* record or its Data did not parse */
#define RAFTQ_WAL_F_BADCRC 0x02u    /* Record.crc is not the running CRC (wal.ErrCRCMismatch) */
#define RAFTQ_WAL_F_GROUP 0x04u     /* the group extension field was present */

/* One WAL record:
 *   ENTRY     Data = Entry{Type: entry_type, Term: term, Index: index, Data: pool[data_off, +data_len), group}
 *   STATE     Data = HardState{term, vote (raft ID, 0 = None), commit = index, group}
 *   SNAPSHOT  Data = walpb.Snapshot{index, term}
 *   METADATA  Data = pool[data_off, +data_len) verbatim
 *   CRC       no Data; Record.crc = the running CRC (wal.saveCrc at the head of a segment) */
typedef struct raftq_wal_rec {
  uint64_t group;
  uint64_t term;
  uint64_t index;
  uint64_t data_off;
  uint32_t data_len;
  uint32_t vote;
  uint32_t crc;       /* decode: Record.crc as stored */
  uint8_t kind;       /* RAFTQ_WAL_* */
  uint8_t entry_type;
  uint8_t flags;      /* RAFTQ_WAL_F_* (decode) */
  uint8_t _pad;
} raftq_wal_rec_t; /* 48 bytes */

typedef struct raftq_wal_counts {
  uint64_t n_recs;
  uint64_t n_valid;  /* decode: records before the first malformed / CRC-mismatching one */
  uint64_t bytes;
  uint32_t last_crc; /* running CRC after the last (valid) record: prev_crc of the next batch */
  uint32_t _pad;
} raftq_wal_counts_t;

/* wal.Save for a batch: n records -> WAL frames appended to a segment whose running CRC is
 * prev_crc (0 for a new file, whose first record must then be a CRC record).  Records are
 * written in order; every record's crc continues the chain. */
int raftq_wal_encode(raftq_t* h, const raftq_wal_rec_t* recs, uint64_t n, const void* pool, uint64_t pool_bytes,
                     uint32_t prev_crc, void* out, uint64_t cap, uint64_t* frame_off /*[n+1]|NULL*/,
                     raftq_wal_counts_t* counts /*|NULL*/);

/* raftq_wal_encode in two halves, so that a turn's two encodes are ONE submission: _begin enqueues the encode on the handle's
 * stream and returns without waiting (page-locked buffers only: RAFTQ_EINVAL otherwise); the wait of whatever is called next
 * on the handle -- raftq_wire_encode in a node's turn -- covers it; _end reports what raftq_wal_encode would have (and waits
 * itself if nothing has).  out / frame_off are not to be read, nor recs / pool reused, before _end.  One at a time. */
int raftq_wal_encode_begin(raftq_t* h, const raftq_wal_rec_t* recs, uint64_t n, const void* pool, uint64_t pool_bytes,
                           uint32_t prev_crc, void* out, uint64_t cap, uint64_t* frame_off /*[n+1]|NULL*/);
int raftq_wal_encode_end(raftq_t* h, raftq_wal_counts_t* counts /*|NULL*/);

/* w.ReadAll for a batch of frames (boundaries from raftq_wire_scan_frames, big_endian = 0):
 * parse every record, recompute the CRC chain from prev_crc and compare.  As ReadAll, a CRC
 * record re-seeds the chain (and must equal the running value unless that is 0).
 * counts->n_valid = index of the first bad record (n if none); records from there on still
 * carry their parsed fields and flags, the caller decides (the reference log.Fatalf's, raft.go:126). */
int raftq_wal_decode(raftq_t* h, const void* bytes, uint64_t nbytes, const uint64_t* frame_off /*[n+1]*/, uint64_t n,
                     uint32_t prev_crc, raftq_wal_rec_t* recs /*[n]*/, raftq_wal_counts_t* counts /*|NULL*/);

#ifdef __cplusplus
}
#endif
#endif /* RAFTQ_WIRE_H */
