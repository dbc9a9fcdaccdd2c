"""raftsql_amd -- MI355X-native batched multi-raft quorum engine.

One hot path of chzchzchz/raftsql, rebuilt for gfx950: per raft group, the
commit-index advance (q-th largest matchIndex) and the RequestVote majority
tally that the reference reaches through raft.go:214/224/269 into etcd/raft.
The product is libraftq.so (hand-written HIP behind the C-ABI of
include/raftq.h); this package is the Python host mirror used by the tests
and the bench.  See DESIGN.md and INTEGRATION.md.
"""
from . import synth  # noqa: F401
from .synth import quorum, shard_range  # noqa: F401

__all__ = ["synth", "quorum", "shard_range", "QuorumEngine"]


def __getattr__(name):
    # engine needs the native library; keep `import raftsql_amd` (synth, build)
    # usable before it is built, but never substitute anything for it.
    if name in ("QuorumEngine", "engine"):
        import importlib
        engine = importlib.import_module(__name__ + ".engine")
        return engine if name == "engine" else engine.QuorumEngine
    raise AttributeError(name)
