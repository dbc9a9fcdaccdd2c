"""Adds more of etcd's own tables, AS RECALLED, to kat.json -> "upstream_step_tables_recalled" (round 2).

raft_paper_test.go (2015-era): TestLeaderAcknowledgeCommit, TestLeaderOnlyCommitsLogFromCurrentTerm, TestVoter, TestFollowerVote,
TestLeaderElectionInOneRoundRPC; log_test.go: TestIsUpToDate.  The module is absent from this machine: these rows are what the builder remembers of
upstream's expectations -- alignment evidence, not a pin.  Where upstream reaches the state under test through calls that
are the host's here (becomeLeader's and MsgProp's appendEntry), the row starts from that state: the comments say which.
Run once; idempotent (rows of these tables are replaced).  Member ids 1..n map to slots 0..n-1."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "kat.json")
FOLLOWER, CANDIDATE, LEADER = 0, 1, 2
HUP, APP_RESP, VOTE, VOTE_RESP = 0, 4, 5, 6
OUT_VOTE_RESP = 1

cases = []

# TestLeaderAcknowledgeCommit: a leader at term 1 whose empty entry (index 1) is committed everywhere proposes once
# (index 2: its own match is 2, committed = li = 1); the acceptors acknowledge index 2; `committed > li` iff wack.
# (Upstream's first row, a single-member group, commits inside appendEntry -- the host's log-tail report here -- and is
# covered by tests/test_node_scenarios_gpu.py::test_single_node_candidate_and_commit.)
rows = [(3, [], False), (3, [2], True), (3, [2, 3], True), (5, [], False), (5, [2], False), (5, [2, 3], True),
        (5, [2, 3, 4], True), (5, [2, 3, 4, 5], True)]
for i, (size, acc, wack) in enumerate(rows):
    cases.append({"table": "TestLeaderAcknowledgeCommit", "row": i + 1, "n": size, "self": 0,
                  "init": {"role": LEADER, "term": 1, "lead": 1, "vote": 1, "last_index": 2, "last_term": 1, "first_idx": 1,
                           "committed": 1, "match": [2] + [1] * (size - 1)},
                  "msgs": [{"type": APP_RESP, "term": 1, "from": a - 1, "index": 2} for a in acc],
                  "want_out": [], "want_state": {"committed": 2 if wack else 1, "role": LEADER}})

# TestLeaderOnlyCommitsLogFromCurrentTerm: log (term 1, index 1), (2, 2); leader of {1, 2} at term 3 after becomeLeader's
# empty entry (index 3) and one proposal (index 4); peer 2 acknowledges `index`: nothing of an earlier term commits by
# counting (raft 5.4.2)
for i, (index, wcommit) in enumerate([(1, 0), (2, 0), (3, 3)]):
    cases.append({"table": "TestLeaderOnlyCommitsLogFromCurrentTerm", "row": i, "n": 2, "self": 0,
                  "init": {"role": LEADER, "term": 3, "lead": 1, "vote": 1, "last_index": 4, "last_term": 3, "first_idx": 3,
                           "committed": 0, "match": [4, 0]},
                  "msgs": [{"type": APP_RESP, "term": 3, "from": 1, "index": index}],
                  "want_out": [], "want_state": {"committed": wcommit}})

# TestVoter: a member of {1, 2} holding `ents` receives MsgVote{Term: 3, LogTerm, Index} from 2
rows = [((1, 1), 1, 1, False), ((1, 1), 1, 2, False), ((2, 1), 1, 1, True),          # same logterm   (last_index, last_term)
        ((1, 1), 2, 1, False), ((1, 1), 2, 2, False), ((2, 1), 2, 1, False),         # candidate higher logterm
        ((1, 2), 1, 1, True), ((1, 2), 1, 2, True), ((2, 1), 1, 1, True)]            # voter higher logterm
for i, ((li, lt), logterm, index, wreject) in enumerate(rows):
    cases.append({"table": "TestVoter", "row": i, "n": 2, "self": 0,
                  "init": {"role": FOLLOWER, "term": 0, "vote": 0, "last_index": li, "last_term": lt},
                  "msgs": [{"type": VOTE, "term": 3, "from": 1, "index": index, "log_term": logterm}],
                  "want_out": [{"type": OUT_VOTE_RESP, "reject": int(wreject), "to": 1}], "want_state": {"term": 3}})

# TestFollowerVote: HardState{Term: 1, Vote: vote}, empty log; MsgVote{Term: 1} from nvote
for i, (vote, nvote, wreject) in enumerate([(0, 1, False), (0, 2, False), (1, 1, False), (2, 2, False), (1, 2, True), (2, 1, True)]):
    cases.append({"table": "TestFollowerVote", "row": i, "n": 3, "self": 2,  # the voter is member 3 here: upstream lets
                  # member 1 receive a request "from 1"; a batch that names this node as a peer sender is malformed
                  "init": {"role": FOLLOWER, "term": 1, "vote": vote},
                  "msgs": [{"type": VOTE, "term": 1, "from": nvote - 1, "index": 0, "log_term": 0}],
                  "want_out": [{"type": OUT_VOTE_RESP, "reject": int(wreject), "to": nvote - 1}],
                  "want_state": {"term": 1, "vote": vote if wreject or vote else nvote}})

# TestLeaderElectionInOneRoundRPC: MsgHup, then the listed MsgVoteResp (true = granted)
rows = [(3, {2: True, 3: True}, LEADER), (3, {2: True}, LEADER), (5, {2: True, 3: True, 4: True, 5: True}, LEADER),
        (5, {2: True, 3: True, 4: True}, LEADER), (5, {2: True, 3: True}, LEADER),
        (3, {2: False, 3: False}, FOLLOWER), (5, {2: False, 3: False, 4: False, 5: False}, FOLLOWER),
        (5, {2: True, 3: False, 4: False, 5: False}, FOLLOWER),
        (3, {}, CANDIDATE), (5, {2: True}, CANDIDATE), (5, {2: False, 3: False}, CANDIDATE), (5, {}, CANDIDATE)]
for i, (size, votes, state) in enumerate(rows):
    cases.append({"table": "TestLeaderElectionInOneRoundRPC", "row": i + 1, "n": size, "self": 0,
                  "init": {"role": FOLLOWER, "term": 0},
                  "msgs": [{"type": HUP, "term": 0}] +
                          [{"type": VOTE_RESP, "term": 1, "from": k - 1, "reject": int(not v)} for k, v in votes.items()],
                  "want_out": [], "want_state": {"role": state, "term": 1}})

# TestIsUpToDate (log_test.go): a log ending (index 3, term 3); a candidate's (lastIndex, term) is up to date iff its term is
# greater, or equal with lastIndex >= 3.  Reached through Step as a MsgVote of a higher term: granted iff up to date.
rows = [(2, 4, True), (3, 4, True), (4, 4, True), (2, 2, False), (3, 2, False), (4, 2, False), (2, 3, False), (3, 3, True), (4, 3, True)]
for i, (last_index, term, up_to_date) in enumerate(rows):
    cases.append({"table": "TestIsUpToDate", "row": i, "n": 3, "self": 0,
                  "init": {"role": FOLLOWER, "term": 3, "vote": 0, "last_index": 3, "last_term": 3},
                  "msgs": [{"type": VOTE, "term": 5, "from": 1, "index": last_index, "log_term": term}],
                  "want_out": [{"type": OUT_VOTE_RESP, "reject": int(not up_to_date), "to": 1}],
                  "want_state": {"term": 5, "vote": 2 if up_to_date else 0}})

kat = json.load(open(PATH))
rec = kat["upstream_step_tables_recalled"]
mine = {c["table"] for c in cases}
rec["cases"] = [c for c in rec["cases"] if c["table"] not in mine] + cases
note = (" Round 2 (tests/golden/make_recalled_tables_r02.py) adds raft_paper_test.go's TestLeaderAcknowledgeCommit, "
        "TestLeaderOnlyCommitsLogFromCurrentTerm, TestVoter, TestFollowerVote, TestLeaderElectionInOneRoundRPC and log_test.go's "
        "TestIsUpToDate; rows whose "
        "upstream setup appends to the log (the host's job here) start from the state that setup produces.")
rec["_note"] = rec["_note"].split(" Round 2 (tests/golden/make_recalled_tables_r02.py)")[0] + note
json.dump(kat, open(PATH, "w"), indent=1)
print(len(cases), "rows written;", len(rec["cases"]), "recalled rows in all")
