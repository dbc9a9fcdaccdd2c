import os, sys, subprocess
def cpus(node):
    s=open(f"/sys/devices/system/node/node{node}/cpulist").read().strip()
    out=set()
    for part in s.split(","):
        a,_,b=part.partition("-")
        out.update(range(int(a), int(b or a)+1))
    return out
print("node0", open("/sys/devices/system/node/node0/cpulist").read().strip(), "node1", open("/sys/devices/system/node/node1/cpulist").read().strip())
print("allowed", len(os.sched_getaffinity(0)))
for tag, aff in (("none", None), ("node0", cpus(0)), ("node1", cpus(1))):
    env=dict(os.environ, NODE_THREADS="1")
    env.pop("RAFTQ_PROFILE", None)
    code = "import os\n" + (f"os.sched_setaffinity(0, {sorted(aff)!r})\n" if aff else "") + "import runpy; runpy.run_path('tools/node_profile.py', run_name='__main__')\n"
    p=subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=200)
    lines=[l for l in (p.stdout+p.stderr).splitlines() if l.startswith(("election:","waves:"))]
    print(tag, " | ".join(l.split("{")[0] for l in lines))
