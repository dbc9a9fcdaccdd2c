"""Build tests/c's oracle-backed engine (tests/test_hostsim.py) WITHOUT sanitizers at -O2, for timing the host phases of
raftq_node.cpp on a box with no GPU:  python tools/hostsim_opt.py && RAFTQ_TEST_ENGINE_DOUBLE=tests/c/libraftq_hostsim_opt.so
RAFTQ_PROFILE=1 python -c 'import tests.conftest, runpy; runpy.run_path("tools/node_profile.py")' (the swap lives in tests/conftest.py).  Test infrastructure only: the device phases (decode, step, deltas, encode) are the
oracle's speed here and mean nothing."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hostsim as th  # noqa: E402

print(th._build(san=["-O2", "-g"], lib=os.path.join(th.CDIR, "libraftq_hostsim_opt.so"), bdir_name="build_hostsim_opt"))
