"""CPU oracle for the dusty_v2 G+D training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain fp32 CPU restatement (torch
CPU tensors + numpy) of the reference algorithm for the path named in
BASELINE.json `north_star`.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it, and there only as the checker
or as the timed CPU baseline -- never as the thing shipped.  The product path
(`dusty-gan-v2_amd/`) must not import anything from here and has no CPU fallback.

Parity pinning: every function here is checked against golden vectors produced
by importing the reference itself on CPU in the build container
(`tests/golden/make_golden.py`, fixtures committed under `tests/golden/`), see
`tests/test_oracle_golden.py`.  The reference has no tests or golden vectors of
its own, so those import-generated fixtures are the pin.

Each function cites the reference file:line (relative to the reference tree)
whose behaviour it restates.  Formulations are deliberately different from the
reference (direct gather / polyphase forms instead of zero-insert + conv2d,
functional state-dict access instead of nn.Module trees) so that agreement is
evidence, not tautology.
"""
