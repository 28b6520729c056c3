"""oracle/ — TEST INFRASTRUCTURE, not product code.

A CPU restatement (plain fp32 PyTorch, functional, driven by a state_dict with the reference's key names) of the
algorithm on MuDG's denoising path: schedule math, the 3D-UNet, the DDIM update and the AutoencoderKL decoder.
It exists to check the HIP path (`mudg_amd`, `lvdm`) and to be timed as the CPU baseline in bench.py.

Rules (enforced by tests/test_repo_rules.py):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
  * nothing under mudg_amd/ or lvdm/ imports it, and the product path has no CPU fallback at all.

Pinning: every function here is checked against golden vectors captured by importing the reference itself in the
build container (tests/golden/make_golden.py -> tests/golden/*.pt, tests/test_oracle_golden.py).  The reference has
no tests or fixtures of its own, so those captured vectors are the pin.
"""
