"""CPU: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the
reference has no race/sanitizer tooling; the oracle is the checker, so it gets checked).  The
sanitized build (oracle/Makefile `asan`) is loaded in a subprocess with libasan preloaded and driven
through the sweep oracles, the tick, and a few thousand random Step messages."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ROOT)
from oracle import pyoracle
pyoracle._SO = os.path.join(ROOT, "oracle", "libraftq_oracle_asan.so")
pyoracle._lib = None
from raftsql_amd import synth
from tests import _stepgen
for n in (1, 2, 3, 4, 5, 7, 9):
    st = synth.concat(synth.make_groups(3000, n, seed=n, with_terms=True), synth.adversarial_block(n))
    a, _ = pyoracle.commit_advance(st.match, st.committed)
    b, _ = pyoracle.commit_advance(st.match, st.committed, True, st.first_idx_cur_term)
    assert np.all(b <= a)
    pyoracle.vote_tally(st.votes)
    rng = np.random.default_rng(n)
    s = _stepgen.random_state(rng, 200, n, n // 2)
    for _ in range(5):
        m = _stepgen.random_batch(rng, s, 1500)
        s.step_batch(m)
        g = rng.integers(0, 200, 50)
        s.apply_log_deltas(g, s.last_index[g] + 1, s.last_term[g], s.committed[g])
role = (np.arange(5000) % 3).astype(np.uint8); el = np.zeros(5000, np.uint32); act = np.zeros(5000, np.uint8)
for t in range(30):
    pyoracle.lib().rq_oracle_tick(role, el, 5000, 10, 1, 7, t, act, None, None)
print("SANITIZED-OK")
"""


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", "ROOT=%r\n" % ROOT + DRIVER], capture_output=True, text=True, env=env,
                       timeout=600)
    assert p.returncode == 0 and "SANITIZED-OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
    assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-3000:]
