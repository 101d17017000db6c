"""CPU oracle for the PyramidBox + IoU-tracker hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy for the box/index arithmetic, PyTorch's
CPU backend for the convolution arithmetic the reference itself delegates to
ATen) of the reference algorithm on the path named by BASELINE.json.  Each
function cites the reference file:line it follows.

Who may import it: `tests/` (incl. the diagnostic `tests/stage_diff.py`),
`__graft_entry__.smoke()` and the `cpu_baseline` legs of `bench.py` -- as the
checker / the timed CPU baseline, never as the thing shipped.  Nothing under
`face-detection-and-tracking_amd/` or `tools/` imports it
(tests/test_cabi_and_host.py enforces both); the product path raises when the HIP
library is missing.

Pinning: every function here is checked in `tests/test_oracle_golden.py` against
fixtures in `tests/golden/` that were produced by importing the reference itself
in the build container (`tests/golden/make_golden.py`, committed).
"""
