"""aggregate tools/probe/pcsample.c output by function (addr2line) -- python tools/probe/pcsample_report.py /tmp/pcsample.txt [lib-substring]"""
import collections
import subprocess
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "libraftq"
by_mod = collections.defaultdict(list)
for line in open(path):
    mod, off, sym = line.split()[:3]
    by_mod[mod].append((off, sym))
total = sum(len(v) for v in by_mod.values())
fn = collections.Counter()
ln = collections.Counter()
for mod, lst in by_mod.items():
    if want in mod:
        out = subprocess.run(["addr2line", "-f", "-C", "-i", "-e", mod] + ["0x" + o for o, _ in lst], capture_output=True, text=True).stdout.split("\n")
        # -i prints inlined frames too: take the outermost non-inlined pair per address is hard; use the first (innermost) pair
        res = subprocess.run(["addr2line", "-f", "-C", "-e", mod] + ["0x" + o for o, _ in lst], capture_output=True, text=True).stdout.split("\n")
        for i in range(0, len(res) - 1, 2):
            fn[res[i][:90]] += 1
            ln[res[i + 1].split("/")[-1]] += 1
    else:
        for _, sym in lst:
            fn[mod.split("/")[-1] + ":" + sym] += 1
print("samples", total)
for k, v in fn.most_common(25):
    print("%5.1f%%  %s" % (100.0 * v / total, k))
print("-- lines")
for k, v in ln.most_common(30):
    print("%5.1f%%  %s" % (100.0 * v / total, k))
