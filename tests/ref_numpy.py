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
