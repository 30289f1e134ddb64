"""TEST INFRASTRUCTURE ONLY -- the oracle for the IGGT forward path.

Nothing in the product package (`iggt_official_amd`, `iggt`) may import from here.
Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.

Pinning status: the reference ships no golden vectors / tests / checkpoint
(SURVEY.md section 8c).  The oracle is pinned against *outputs of the reference's own
modules run in the build container* (oracle/make_golden.py imports /root/reference
through oracle/ref_shim.py and writes tests/golden/*.pt); oracle/restate.py is then
checked against those fixtures by tests/test_oracle_golden.py.
"""
