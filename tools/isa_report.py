#!/usr/bin/env python3
"""Per-kernel resource usage of the sweep kernels as compiled for gfx950 (VGPR / SGPR / scratch / LDS / occupancy),
with the toolchain and the exact command line -- the tracked evidence behind DESIGN.md's ISA claims.

usage: tools/isa_report.py [out.txt]      (default profiles/r02/isa_sweep.txt)
Compiles raftsql_amd/csrc/raftq_capi.hip (device side) with -Rpass-analysis=kernel-resource-usage; no GPU needed.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raftsql_amd import build as b  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02", "isa_sweep.txt")
    src = os.path.join(b.CSRC, "raftq_capi.hip")
    cmd = [b._hipcc()] + b._flags() + ["--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        sys.exit(p.stderr[-3000:])
    kernels, cur = [], None
    for ln in p.stderr.splitlines():
        m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[[^\]]+\])?: (.*?) \[-Rpass", ln)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
            kernels.append(cur)
        elif cur is not None:
            cur[k] = v
    dem = subprocess.run(["c++filt"] + [k["name"] for k in kernels], capture_output=True, text=True)
    names = dem.stdout.splitlines() if dem.returncode == 0 else [k["name"] for k in kernels]
    tc = b.toolchain()
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "raftsql_amd/csrc"], capture_output=True, text=True).stdout.strip()
    lines = ["# gfx950 resource usage of the sweep kernels (raftq_capi.hip), from -Rpass-analysis=kernel-resource-usage",
             "# tree: %s%s" % (head, " + uncommitted changes under raftsql_amd/csrc" if dirty else ""),
             "# " + " | ".join(tc["version"][:2]), "# " + " ".join(cmd), "#",
             "# %-6s %-6s %-6s %-8s %-5s %-4s kernel" % ("VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ")]
    want = re.compile(r"raftqk::(sweep_kernel|sweep_set_kernel|sweep_persist_kernel|sweep_lds_kernel|set_counts_kernel|sweep_segments_kernel|tick_set_wide_kernel|tick_set_kernel|tick_lists32_kernel|"
                      r"deltas_in|apply_deltas|scan_partials|compact_changed|tick_kernel)")
    rows = []
    for k, n in zip(kernels, names):
        if not want.search(n):
            continue
        n = re.sub(r"^void ", "", n)
        rows.append((n, "  %-6s %-6s %-6s %-8s %-5s %-4s %s" % (
            k.get("VGPRs", "?"), k.get("AGPRs", "?"), k.get("TotalSGPRs", k.get("SGPRs", "?")), k.get("ScratchSize", "?"),
            k.get("LDS Size", "?"), k.get("Occupancy", "?"), n)))
    rows.sort()
    lines += [r for _, r in rows]
    worst = max((int(k.get("ScratchSize", "0")) for k in kernels), default=0)
    lines.append("# kernels listed: %d of %d in the unit; largest scratch in the unit: %d bytes/lane" % (len(rows), len(kernels), worst))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(l for l in lines if "<5, " in l or l.startswith("#")))


if __name__ == "__main__":
    main()
