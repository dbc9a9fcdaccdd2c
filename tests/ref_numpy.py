"""Third, independent restatement of the quorum path (numpy, sort-free where
it can be): used to cross-check the C oracle and to make the golden fixtures.
TEST INFRASTRUCTURE."""
import numpy as np


def quorum(n):
    return n // 2 + 1


def mci(match):
    """q-th largest per column = ascending position N-q (np.sort, not the oracle's insertion sort)."""
    match = np.asarray(match, dtype=np.uint64)
    n = match.shape[0]
    return np.sort(match, axis=0)[n - quorum(n)]


def mci_bruteforce(match):
    """max{ i in match values : #{p : match[p] >= i} >= q } per column."""
    match = np.asarray(match, dtype=np.uint64)
    n = match.shape[0]
    q = quorum(n)
    best = np.zeros(match.shape[1], dtype=np.uint64)
    for c in range(n):
        ge = (match >= match[c][None, :]).sum(axis=0)
        ok = ge >= q
        best = np.where(ok & (match[c] > best), match[c], best)
    return best


def commit_advance(match, committed, gated=False, first_idx=None):
    m = mci(match)
    committed = np.asarray(committed, dtype=np.uint64)
    adv = m > committed
    if gated:
        f = np.asarray(first_idx, dtype=np.uint64)
        adv &= (f != 0) & (m >= f)
    out = np.where(adv, m, committed).astype(np.uint64)
    return out, int(adv.sum())


def vote_tally(votes):
    votes = np.asarray(votes, dtype=np.uint8)
    n = votes.shape[0]
    q = quorum(n)
    g = (votes == 1).sum(axis=0)
    r = (votes == 2).sum(axis=0)
    out = np.where(g >= q, 1, np.where(r >= q, 2, 0)).astype(np.uint8)
    return out, int((out == 1).sum()), int((out == 2).sum())


# ---- the election-timeout draw (oracle/raftq_oracle.c rq_oracle_tick_key / rq_oracle_tick_rand), vectorised ----------
_M64 = (1 << 64) - 1


def tick_key(seed, tick_no):
    """splitmix64's finaliser over (seed, tick number): ONE 64-bit key per tick (python ints: no silent wraparound)."""
    z = (seed ^ ((tick_no * 0xD1B54A32D192ED03) & _M64)) & _M64
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def tick_rand(seed, tick_no, groups):
    """murmur3's 32-bit finaliser of (group ^ key.lo), xored with key.hi -> u32 [len(groups)]"""
    key = tick_key(seed, tick_no)
    g = np.asarray(groups, dtype=np.uint64)
    x = (g.astype(np.uint32) ^ (g >> np.uint64(32)).astype(np.uint32)) ^ np.uint32(key & 0xFFFFFFFF)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x85EBCA6B)
    x ^= x >> np.uint32(13)
    x *= np.uint32(0xC2B2AE35)
    x ^= x >> np.uint32(16)
    return x ^ np.uint32(key >> 32)
