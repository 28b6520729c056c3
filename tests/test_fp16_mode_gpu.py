"""The fp16-operand build (libmudg_hip_fp16.so, MUDG_OPERAND=fp16) through the same parity suite, in a child process
(the operand type is fixed per process when the library loads).  Tolerances tighten 6-8x there; see the test files."""
import os
import subprocess
import sys

import pytest

from mudg_amd import hip

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(hip.operand_name() == "fp16", reason="already running in fp16 mode")
def test_parity_suite_in_fp16_operand_mode(cuda):
    env = dict(os.environ, MUDG_OPERAND="fp16")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_kernels_gpu.py", "tests/test_unet_gpu.py",
                        "tests/test_pipeline_gpu.py", "tests/test_fullsize_gpu.py", "-m", "gpu", "-q", "-s", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join(l for l in r.stdout.splitlines() if "rel-L2" in l or "passed" in l or "failed" in l)
    print(tail)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
