"""The etcd raftpb / walpb schema (2015-era, recalled -- SURVEY.md F2: no .proto on this machine),
built at run time for the google.protobuf runtime.  TEST INFRASTRUCTURE: the independent
third-party encoder / decoder the wire oracle is pinned against.

proto2 with explicit presence: a field that is SET serialises even when zero, which is exactly
gogoproto's `nullable=false` "write every field" output when every field is set."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto
_cache = None


def classes():
    global _cache
    if _cache is not None:
        return _cache
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "raftq_test_raftpb.proto", "raftpb", "proto2"
    O, R = _F.LABEL_OPTIONAL, _F.LABEL_REPEATED
    U64, MSG, BYTES = _F.TYPE_UINT64, _F.TYPE_MESSAGE, _F.TYPE_BYTES

    def msg(name, fields):
        m = fd.message_type.add()
        m.name = name
        for fname, num, typ, label, tname in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, typ, label
            if tname:
                f.type_name = ".raftpb." + tname

    e = fd.enum_type.add()
    e.name = "EntryType"
    for n, v in (("EntryNormal", 0), ("EntryConfChange", 1)):
        x = e.value.add()
        x.name, x.number = n, v
    msg("Entry", [("Type", 1, _F.TYPE_ENUM, O, "EntryType"), ("Term", 2, U64, O, None), ("Index", 3, U64, O, None),
                  ("Data", 4, BYTES, O, None), ("group", 5, U64, O, None)])
    msg("ConfState", [("nodes", 1, U64, R, None)])
    msg("SnapshotMetadata", [("conf_state", 1, MSG, O, "ConfState"), ("index", 2, U64, O, None), ("term", 3, U64, O, None)])
    msg("Snapshot", [("data", 1, BYTES, O, None), ("metadata", 2, MSG, O, "SnapshotMetadata")])
    msg("Message", [("type", 1, _F.TYPE_INT32, O, None), ("to", 2, U64, O, None), ("from", 3, U64, O, None),
                    ("term", 4, U64, O, None), ("logTerm", 5, U64, O, None), ("index", 6, U64, O, None),
                    ("entries", 7, MSG, R, "Entry"), ("commit", 8, U64, O, None), ("snapshot", 9, MSG, O, "Snapshot"),
                    ("reject", 10, _F.TYPE_BOOL, O, None), ("rejectHint", 11, U64, O, None), ("group", 12, U64, O, None)])
    msg("HardState", [("term", 1, U64, O, None), ("vote", 2, U64, O, None), ("commit", 3, U64, O, None),
                      ("group", 4, U64, O, None)])
    msg("Record", [("type", 1, _F.TYPE_INT64, O, None), ("crc", 2, _F.TYPE_UINT32, O, None), ("data", 3, BYTES, O, None)])
    msg("WalSnapshot", [("index", 1, U64, O, None), ("term", 2, U64, O, None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    _cache = {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("raftpb." + n))
              for n in ("Entry", "Snapshot", "SnapshotMetadata", "ConfState", "Message", "HardState", "Record", "WalSnapshot")}
    return _cache


def entry_pb(type_, term, index, data: bytes, group=None):
    e = classes()["Entry"]()
    e.Type, e.Term, e.Index = int(type_), int(term), int(index)
    if data:
        e.Data = bytes(data)
    if group is not None:
        e.group = int(group)
    return e


def message_bytes(m, ents, pool: bytes) -> bytes:
    """Canonical (gogoproto-shaped) marshal of one raftq_wire_msg_t record by the protobuf runtime."""
    pm = classes()["Message"]()
    pm.type, pm.to = int(m["type"]), int(m["to"]) + 1
    setattr(pm, "from", int(m["from"]) + 1)
    pm.term, pm.logTerm, pm.index = int(m["term"]), int(m["log_term"]), int(m["index"])
    for k in range(int(m["n_ents"])):
        e = ents[int(m["ent_first"]) + k]
        d = pool[int(e["data_off"]): int(e["data_off"]) + int(e["data_len"])]
        pm.entries.append(entry_pb(e["type"], e["term"], e["index"], d))
    pm.commit = int(m["commit"])
    pm.snapshot.metadata.conf_state.SetInParent()
    pm.snapshot.metadata.index = 0
    pm.snapshot.metadata.term = 0
    pm.reject, pm.rejectHint, pm.group = bool(m["reject"]), int(m["reject_hint"]), int(m["group"])
    return pm.SerializeToString()


def frame_be(body: bytes) -> bytes:
    return len(body).to_bytes(8, "big") + body


def frame_le(body: bytes) -> bytes:
    return len(body).to_bytes(8, "little") + body
