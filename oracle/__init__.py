"""CPU oracle for the RoHM denoising hot path -- TEST INFRASTRUCTURE ONLY.

This package restates, in plain PyTorch-CPU arithmetic (fp32 by default, fp64 on
request), the algorithm of the reference's hot path: PoseNet /
TrajNet / ControlNet forwards, the DDPM ancestral sampling loops, the 6-D-rotation /
quaternion / axis-angle helpers, SMPL-X (smplx==0.1.28) linear blend skinning and
the two test-time guidance gradients; and, for the rows either side of the samplers,
the between-stage trajectory re-derivation (`rederive.py`), the drivers' iteration loops
(`scheme.py`), the evaluation metrics (`metrics.py`) and the DDIM update.  Every
function cites the reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it, and only as the checker -- the shipped path (`rohm_amd`) never
routes through it and raises if the HIP library is missing.

Parity pinning: the reference has no tests or golden vectors.  The
networks / diffusion / guidance restatements are pinned against the reference's own
modules imported from /root/reference (see `oracle/refload.py`,
`oracle/make_golden.py`; outputs committed under `tests/golden/`).  The SMPL-X
arithmetic lives in the third-party package smplx==0.1.28, which is not vendored in
the reference nor installed here: that part is a restatement of its published
algorithm and is **parity unpinned** (self-consistency checks only).
"""
