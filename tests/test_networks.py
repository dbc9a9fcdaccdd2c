"""CPU: the comparator lists compiled into the HIP kernel are sorting networks
(0-1 principle), checked from the kernel source itself."""
import itertools
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "raftsql_amd", "csrc", "raftq_kernels.hpp")).read()


def _networks():
    body = SRC[SRC.index("select_quorum_network"):SRC.index("#undef CE")]
    nets = {}
    parts = re.split(r"if constexpr \(N == (\d)\)", body)
    for k in range(1, len(parts), 2):
        nets[int(parts[k])] = [(int(a), int(b)) for a, b in re.findall(r"CE\((\d), (\d)\)", parts[k + 1])]
    return nets


def test_all_sizes_present():
    nets = _networks()
    assert sorted(nets) == list(range(2, 10))
    # optimal comparator counts for n = 2..9
    assert [len(nets[n]) for n in range(2, 10)] == [1, 3, 5, 9, 12, 16, 19, 25]


def test_zero_one_principle_descending():
    for n, net in _networks().items():
        for bits in itertools.product((0, 1), repeat=n):
            v = list(bits)
            for a, b in net:
                assert a < b < n
                if v[a] < v[b]:
                    v[a], v[b] = v[b], v[a]
            assert all(v[i] >= v[i + 1] for i in range(n - 1)), (n, bits)
