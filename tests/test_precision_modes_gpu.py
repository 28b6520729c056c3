"""The other operand-type builds of the library through the parity suites, each in a child process (the operand type is
fixed per process when the library loads): fp16 (libmudg_hip_fp16.so) and the split-operand precision modes bf16x3 /
bf16x6 (libmudg_hip_x3.so / _x6.so), which are the modes that must meet north_star's 1e-3 decoded-frame tolerance — the
pipeline tests assert it with the literal constant there.  The children's measured errors are echoed."""
import os
import subprocess
import sys

import pytest

from mudg_amd import hip

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUITES = {
    "fp16": ["tests/test_kernels_gpu.py", "tests/test_operand_modes_gpu.py", "tests/test_unet_gpu.py",
             "tests/test_pipeline_gpu.py", "tests/test_sampler_options_gpu.py", "tests/test_fullsize_gpu.py"],
    # test_kernels_gpu.py builds its inputs as plain 16-bit tensors; the split modes run the mode-agnostic kernel suite
    "bf16x3": ["tests/test_operand_modes_gpu.py", "tests/test_unet_gpu.py", "tests/test_pipeline_gpu.py",
               "tests/test_sampler_options_gpu.py"],
    "bf16x6": ["tests/test_operand_modes_gpu.py", "tests/test_unet_gpu.py", "tests/test_pipeline_gpu.py"],
}


@pytest.mark.parametrize("mode", list(SUITES))
def test_parity_suites_in_operand_mode(cuda, mode):
    if hip.operand_name() == mode:
        pytest.skip("already running in this mode")
    env = dict(os.environ, MUDG_OPERAND=mode, MUDG_SKIP_FULLSIZE_ORACLE="1")     # the minute-long CPU oracle forward runs once, in the default mode
    r = subprocess.run([sys.executable, "-m", "pytest", *SUITES[mode], "-m", "gpu", "-q", "-s", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = "\n".join(l for l in r.stdout.splitlines() if "rel-L2" in l or "passed" in l or "failed" in l)
    print(tail)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
