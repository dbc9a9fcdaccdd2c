"""CPU: a static check of the Go binding (go/raftq/*.go) against the C headers (include/*.h), since `go vet` cannot run here
(no Go toolchain in the image).  What a compiler would have caught and this catches:
  * a `C.raftq_*` call whose argument count differs from the prototype, or that passes a pointer where the prototype takes
    an integer (or the other way round) -- for the arguments whose kind can be read off the source;
  * a Go struct documented as "layout-identical" to a C struct whose fields do not sit at the same offsets with the same
    sizes (natural alignment on both sides), or whose field names disagree;
  * an export of the headers that no Go file binds (the host drivers raftq_pipe_* / raftq_node_* / raftq_crank_* / raftq_shards_* are C++
    stand-ins for code that IS Go in a Go deployment -- go/raftq/node.go, batcher.go -- and are exempt, by name);
  * a `C.raftq_*` name the headers do not declare.
It reads source text only: nothing here proves the binding runs."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = sorted(glob.glob(os.path.join(ROOT, "go", "raftq", "*.go")))
HEADERS = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))

# C++ host drivers that stand in for Go host code: not bound on purpose
EXEMPT_PREFIXES = ("raftq_pipe_", "raftq_node_", "raftq_crank_", "raftq_shards_")

SCALARS = {"uint64_t": 8, "int64_t": 8, "uint32_t": 4, "int32_t": 4, "uint16_t": 2, "uint8_t": 1, "int": 4, "unsigned": 4, "float": 4,
           "size_t": 8, "char": 1}
GO_SIZES = {"uint64": 8, "int64": 8, "uint32": 4, "int32": 4, "uint16": 2, "uint8": 1, "byte": 1, "float32": 4}

# Go struct -> C struct it documents itself as layout-identical to
LAYOUTS = {
    "Delta": "raftq_delta", "VoteDelta": "raftq_vote_delta", "Advance": "raftq_advance", "Counts": "raftq_counts",
    "Msg": "raftq_msg", "StepOut": "raftq_step_out", "StepOutC": "raftq_step_out_c", "LogDelta": "raftq_log_delta", "Msg40": "raftq_msg40",
    "WireMsg": "raftq_wire_msg", "WireEnt": "raftq_wire_ent", "WalRec": "raftq_wal_rec",
    "TermDelta": "raftq_term_delta", "TickCounts": "raftq_tick_counts", "Delta16": "raftq_delta16", "Advance16": "raftq_advance16",
    "Prop": "raftq_prop", "PropEnt": "raftq_prop_ent", "StepOutS": "raftq_step_out_s",
}
# Go field name -> C field name where the binding renames on purpose
ALIASES = {("Msg", "wireto"): "_pad", ("Msg", "flags"): "_pad", ("Msg", "resv"): "_resv"}


def strip_c_comments(t):
    return re.sub(r"/\*.*?\*/", " ", t, flags=re.S)


def split_args(s):
    """top-level comma split of an argument list (parentheses, brackets and braces nest)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def balanced(text, open_at):
    """text[open_at] == '(' -> the text between it and its closing parenthesis"""
    depth = 0
    for i in range(open_at, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return text[open_at + 1:i]
    raise AssertionError("unbalanced call at %d" % open_at)


def c_prototypes():
    protos = {}
    for h in HEADERS:
        t = strip_c_comments(open(h).read())
        for m in re.finditer(r"^[ \t]*((?:const\s+)?[A-Za-z_]\w*(?:\s*\*+)?)\s*\b(raftq_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", t, flags=re.M):
            params = [p for p in split_args(" ".join(m.group(3).split())) if p and p != "void"]
            protos[m.group(2)] = ["ptr" if "*" in p else "int" for p in params]
    return protos


def c_structs():
    structs = {}
    for h in HEADERS:
        t = strip_c_comments(open(h).read())
        for m in re.finditer(r"typedef\s+struct\s+(raftq_\w+)\s*\{(.*?)\}\s*\w+\s*;", t, flags=re.S):
            fields = []
            for decl in m.group(2).split(";"):
                decl = " ".join(decl.split())
                if not decl:
                    continue
                tm = re.match(r"(\w+)\s+(.*)$", decl)
                assert tm and tm.group(1) in SCALARS, (m.group(1), decl)
                for one in tm.group(2).split(","):  # `uint64_t a, b[2]`
                    fm = re.match(r"\s*(\w+)(?:\[(\d+)\])?\s*$", one)
                    assert fm, (m.group(1), decl)
                    fields.append((fm.group(1), SCALARS[tm.group(1)], int(fm.group(2) or 1)))
            structs[m.group(1)] = fields
    return structs


def layout(fields):
    """[(name, elem size, count)] -> ([(name, offset, total bytes)], struct size), natural alignment"""
    off, out, align = 0, [], 1
    for name, size, count in fields:
        off = (off + size - 1) // size * size
        out.append((name, off, size * count))
        off += size * count
        align = max(align, size)
    return out, (off + align - 1) // align * align


def go_structs():
    structs, documented = {}, set()
    for g in GO:
        t = open(g).read()
        for m in re.finditer(r"type\s+(\w+)\s+struct\s*\{(.*?)\}", t, flags=re.S):
            fields = []
            for line in m.group(2).replace(";", "\n").splitlines():
                line = line.split("//")[0].strip()
                if not line:
                    continue
                fm = re.match(r"([\w_, ]+?)\s+(\[(\d+)\])?(\w+)$", line)
                if not fm or fm.group(4) not in GO_SIZES:
                    fields = None  # not a plain-data struct (pointers, C types, slices): not a layout candidate
                    break
                for name in [x.strip() for x in fm.group(1).split(",")]:
                    fields.append((name, GO_SIZES[fm.group(4)], int(fm.group(3) or 1)))
            if fields:
                structs[m.group(1)] = fields
            head = t[max(0, m.start() - 400):m.start()]
            last_comment = head[head.rfind("\n\n") + 1:] if "\n\n" in head else head
            if "layout-identical" in last_comment:
                documented.add(m.group(1))
    return structs, documented


def test_struct_layouts_match_the_headers():
    cs, (gs, documented) = c_structs(), go_structs()
    assert documented <= set(LAYOUTS), "a struct says layout-identical but is not in this test's table: %s" % (documented - set(LAYOUTS))
    for gname, cname in LAYOUTS.items():
        assert gname in gs, "Go struct %s is gone (or no longer plain data)" % gname
        assert cname in cs, cname
        (gl, gsize), (cl, csize) = layout(gs[gname]), layout(cs[cname])
        assert gsize == csize, (gname, gsize, cname, csize)
        # byte coverage must agree field by field where the C side has scalars; a C array may be split into Go fields
        # same partition of the bytes, allowing one side to split an array the other keeps whole
        g_cuts = {o for _, o, _ in gl} | {gsize}
        c_cuts = {o for _, o, _ in cl} | {csize}
        assert g_cuts >= c_cuts or c_cuts >= g_cuts, (gname, sorted(g_cuts), cname, sorted(c_cuts))
        for name, off, size in gl:
            if name == "_":
                continue
            owner = [c for c in cl if c[1] <= off < c[1] + c[2]]
            assert owner, (gname, name)
            cfield = owner[0][0]
            want = ALIASES.get((gname, name.lower()), None)
            norm = lambda s: s.replace("_", "").lower()  # noqa: E731
            ok = want == cfield if want else (norm(name) in norm(cfield) or norm(cfield) in norm(name))  # Old ~ old_commit, Hup ~ n_hup
            assert ok, "%s.%s sits on %s.%s" % (gname, name, cname, cfield)


def go_calls():
    """-> [(file, function name, [arg text], enclosing Go function text)] for every C.raftq_* call and every raftq_* call in a preamble"""
    calls = []
    for g in GO:
        t = open(g).read()
        for m in re.finditer(r"/\*(.*?)\*/\s*import\s+\"C\"", t, flags=re.S):  # the cgo preamble is C
            pre = re.sub(r"//[^\n]*", "", m.group(1))
            for c in re.finditer(r"(?<![\w.])(raftq_[a-z_0-9]+)\s*\(", pre):
                before = pre[:c.start()].rstrip()
                if before.endswith(("int", "void", "static", "*")):  # a definition of a helper, not a call
                    continue
                calls.append((g, c.group(1), split_args(balanced(pre, c.end() - 1)), ""))
        body = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
        body = re.sub(r"//[^\n]*", "", body)
        funcs = [(f.start(), f.group(0)) for f in re.finditer(r"^func .*?^}", body, flags=re.S | re.M)]
        for c in re.finditer(r"C\.(raftq_[a-z_0-9]+)\s*\(", body):
            enclosing = next((src for start, src in funcs if start <= c.start() < start + len(src)), "")
            calls.append((g, c.group(1), split_args(balanced(body, c.end() - 1)), enclosing))
    return calls


def arg_kind(arg, func_src):
    a = arg.strip()
    if a == "nil" or a.startswith("&") or re.match(r"\(\*+C\.", a) or a in ("e.h", "s.s", "src.h", "n.h", "p.h") or a.startswith("unsafe.Pointer("):
        return "ptr"
    if re.match(r"C\.(int|uint|uint64_t|uint32_t|uint8_t|size_t|float|int64_t|int32_t)\(", a) or re.fullmatch(r"\d+", a):
        return "int"
    if re.fullmatch(r"[A-Za-z_]\w*", a) and func_src:  # a local: look for its declaration
        if re.search(r"\bvar\s+[\w, ]*\b%s\b[\w, ]*\s+\*+C\." % a, func_src) or re.search(r"\b%s\s*:?=\s*\(\*+C\." % a, func_src) or \
                re.search(r"\b%s\s+unsafe\.Pointer\b" % a, func_src) or re.search(r"\b%s\s*:?=\s*(?:unsafe\.Pointer|C\.malloc)\(" % a, func_src):
            return "ptr"
        if re.search(r"\b%s\s*:=\s*C\.(int|uint|uint64_t|uint32_t|uint8_t)\(" % a, func_src) or re.search(r"\bvar\s+[\w, ]*\b%s\b[\w, ]*\s+C\.(int|uint|uint64_t|uint32_t)" % a, func_src):
            return "int"
    return None  # not readable off the source: not judged


def test_calls_match_the_prototypes():
    protos = c_prototypes()
    assert len(protos) > 100
    helpers = {"raftq_create_msg", "raftq_device_count_msg", "raftq_set_create_msg"}  # static functions of raftq.go's preamble
    judged = total = 0
    for g, name, args, src in go_calls():
        if name in helpers:
            continue
        assert name in protos, "%s calls %s, which no header declares" % (os.path.basename(g), name)
        want = protos[name]
        assert len(args) == len(want), "%s: %s takes %d arguments, called with %d: %s" % (os.path.basename(g), name, len(want), len(args), args)
        for i, (a, k) in enumerate(zip(args, want)):
            total += 1
            got = arg_kind(a, src) if src else ("ptr" if a in ("NULL",) or "*" in a or a.startswith("&") else None)
            if got is not None:
                judged += 1
                assert got == k, "%s: argument %d of %s is %s in the header, `%s` at the call" % (os.path.basename(g), i, name, k, a)
    assert total > 300 and judged >= 0.85 * total, (judged, total)  # the check has teeth: most arguments are classifiable


def test_every_export_is_bound_or_exempt():
    protos = c_prototypes()
    used = set()
    for g in GO:
        used |= set(re.findall(r"\b(raftq_[a-z_0-9]+)\s*\(", open(g).read()))
    unbound = sorted(n for n in protos if n not in used and not n.startswith(EXEMPT_PREFIXES))
    assert not unbound, "exports without a Go binding: %s" % unbound


def test_the_check_catches_a_renamed_parameter():
    """the self-test VERDICT asks for: drop an argument from one call, flip a pointer for an integer in another"""
    protos = c_prototypes()
    assert protos["raftq_set_timers"] == ["ptr", "int", "int", "int"] and protos["raftq_tick"] == ["ptr", "ptr"]
    assert arg_kind("C.uint64_t(len(d))", "") == "int" and arg_kind("(*C.raftq_delta_t)(unsafe.Pointer(&d[0]))", "") == "ptr"
    assert arg_kind("dp", "func f() {\n\tvar dp *C.raftq_delta_t\n}") == "ptr"
    assert len(split_args("e.h, (*C.uint64_t)(unsafe.Pointer(&x[0])), C.uint64_t(len(x))")) == 3
    (gl, size), _ = layout([("a", 8, 1), ("b", 4, 1), ("c", 1, 1), ("d", 1, 2)]), None
    assert size == 16 and gl[2] == ("c", 12, 1) and gl[3] == ("d", 13, 2)
