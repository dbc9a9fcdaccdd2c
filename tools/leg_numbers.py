#!/usr/bin/env python3
"""Print the timing / rate / fraction leaves of a leg's output (tools/profile_*.py writes one dict per run, JSON or repr)."""
import ast, json, sys

def walk(d, p=""):
    for k, v in d.items():
        if isinstance(v, dict):
            walk(v, p + k + ".")
        elif isinstance(v, (int, float)) and not isinstance(v, bool) and ("us" in k or "per_s" in k or k == "frac"):
            print(f"  {p}{k} = {v:.4g}")

for f in sys.argv[1:]:
    print("==", f)
    for line in open(f).read().splitlines():
        if line.startswith("{"):
            try:
                d = json.loads(line)
            except Exception:
                d = ast.literal_eval(line)
            walk(d)
