#!/usr/bin/env python3
"""Regenerate the machine-written blocks of DESIGN.md from tracked evidence, so that no figure in them can drift from
the file it was read off (DESIGN said 118 / 161 VGPRs where profiles/r02/isa_sweep.txt said
110 / 144).

  <!-- isa:begin --> ... <!-- isa:end -->     from profiles/r06/isa_sweep.txt (tools/isa_report.py)

usage: tools/design_facts.py [--check]      (--check: exit 1 if DESIGN.md is not what would be generated;
                                             tests/test_docs.py runs it)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ISA = os.path.join(ROOT, "profiles", "r06", "isa_sweep.txt")
DESIGN = os.path.join(ROOT, "DESIGN.md")

HEADLINE = [
    ("`sweep_kernel<5,4,commit,votes,stream>` (one handle per launch)", "raftqk::sweep_kernel<5, 4, true, false, true, 3, true, 256>"),
    ("`sweep_set_kernel<5,8,…>` (the bench's dispatch)", "raftqk::sweep_set_kernel<5, 8, true, false, true, 3, true, 256>"),
    ("`sweep_persist_kernel<5,4,…>`", "raftqk::sweep_persist_kernel<5, 4, true, false, true, 3, true, 1>"),
    ("`sweep_set_kernel<7,2,…>` (config 4's dispatch)", "raftqk::sweep_set_kernel<7, 2, true, false, true, 3, true, 256>"),
    ("`sweep_set_kernel<5,8,gated,…>` (config 5's dispatch)", "raftqk::sweep_set_kernel<5, 8, true, true, false, 3, true, 256>"),
]


def isa_block() -> str:
    rows, tail = {}, ""
    for ln in open(ISA):
        if ln.startswith("# kernels listed"):
            tail = ln[2:].strip()
        m = re.match(r"\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(raftqk::.*)$", ln)
        if m:
            rows[m.group(7).split("(")[0]] = m.groups()
    out = ["| kernel | VGPRs | SGPRs | scratch B | LDS B | waves / SIMD |", "|---|---|---|---|---|---|"]
    for label, name in HEADLINE:
        v = rows[name]
        out.append("| %s | %s | %s | %s | %s | %s |" % (label, v[0], v[2], v[3], v[4], v[5]))
    out.append("")
    out.append("(`profiles/r06/isa_sweep.txt`, written by `tools/isa_report.py` from `-Rpass-analysis=kernel-resource-usage`; %s; "
               "this table is regenerated from that file by `tools/design_facts.py` and `tests/test_docs.py` fails when the two disagree.)" % tail)
    return "\n".join(out)


def main():
    s = open(DESIGN).read()
    a, b = s.index("<!-- isa:begin -->"), s.index("<!-- isa:end -->")
    new = s[:a] + "<!-- isa:begin -->\n" + isa_block() + "\n" + s[b:]
    if "--check" in sys.argv:
        if new != s:
            sys.exit("DESIGN.md's ISA table is not what profiles/r06/isa_sweep.txt says: run tools/design_facts.py")
        return
    open(DESIGN, "w").write(new)


if __name__ == "__main__":
    main()
