"""Every contraction-kernel variant through the same kernel parity tests, in child processes on the debug-variants build
(libmudg_hip_dbg.so, MUDG_DEBUG_VARIANTS=1: the only library that reads the MUDG_<switch> variables, once per process):
the generic 64-bit-address path of all kernels with the buffer-descriptor (FAST) path disabled, the single-buffer
short-K kernel forced on / off for every FAST problem, the phase-scheduled large-tile kernel forced on / off, the persistent kernel
off / forced at 4, 3 and 2 workgroups per CU for every problem it accepts (its residual-seed / GroupNorm-partial variants included), both
flash-attention kernels (32 / 64 queries per wave) forced, the LDS-table GroupNorm apply kernel and the VALU temporal-attention kernel
(the defaults are the register-table kernel and the MFMA kernel), the 3x3 convs without the shared activation stage of their dx taps, the 288 x 320 tile forced for every problem it can run (and, in
that child, its bit-identity with the 128 x 128 kernels: test_wide288_is_bit_identical_to_the_one_tile_kernels), the two-workgroup
144 x 256 GEGLU kernel of round 6 forced for every GEGLU problem (both K-loop forms) and switched off, the 160 x 320 tile of round 6 forced
for every problem it can run (test_wide160_is_bit_identical_to_the_one_tile_kernels runs in the children)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECT = "gemm or conv or tconv or resblock or transformer or geglu or fused or attention or attn or wide or persistent or norm or temporal"


VARIANTS = [
    {"MUDG_GEMM_FAST": "0"},
    {"MUDG_GEMM_SB": "2"},
    {"MUDG_GEMM_SB": "0"},
    {"MUDG_GEMM_WIDE": "1"},
    {"MUDG_GEMM_WIDE": "0"},
    {"MUDG_GEMM_PERSIST": "0"},
    {"MUDG_GEMM_PERSIST": "4"},
    {"MUDG_GEMM_PERSIST": "3"},
    {"MUDG_GEMM_PERSIST": "2"},
    {"MUDG_ATTN_Q": "32"},
    {"MUDG_ATTN_Q": "64"},
    {"MUDG_GN_REG": "0", "MUDG_TATTN_MFMA": "0"},
    {"MUDG_CONV_XSHARE": "0"},
    {"MUDG_GEMM_W288": "2"},
    {"MUDG_GEMM_W288": "2", "MUDG_GEMM_W288P": "2"},
    {"MUDG_GEMM_H144": "2"},                                  # the two-workgroup 144 x 256 GEGLU kernel for every GEGLU problem it can run
    {"MUDG_GEMM_H144": "2", "MUDG_GEMM_H144PF": "0"},         # ... its plain (non-prefetching) K loop
    {"MUDG_GEMM_H144": "0"},                                  # ... and never (round 5's selection)
    {"MUDG_GEMM_W160": "2"},                                  # the 160 x 320 tile (round 6) for every problem it can run that the 288-row tile's rule does not take
]
# The bf16x3 build's own debug-variants library (libmudg_hip_x3_dbg.so): its 288 x 320 kernel forced for every problem it can run
# (short K, GEGLU on the 288 x 256 tile, ragged M — the rule sends none of those to it), and switched off (every conv / temporal conv
# on the fused-piece 128 x 128 kernels), under the mode-agnostic kernel suite and the UNet parity tests.
VARIANTS_X3 = [{"MUDG_GEMM_W288": "2"}, {"MUDG_GEMM_W288": "0"}]
_name = lambda e: ",".join(f"{k[5:]}={v}" for k, v in e.items())


@pytest.fixture(scope="module")
def children():
    """All variant children of this module, four at a time (helpers.ChildRuns)."""
    from helpers import ChildRuns
    runs = ChildRuns(workers=4)
    for i, env in enumerate(VARIANTS):
        runs.submit(_name(env), [sys.executable, "-m", "pytest", "tests/test_kernels_gpu.py", "tests/test_operand_modes_gpu.py", "tests/test_unet_gpu.py",
                                 "-m", "gpu", "-q", "-k", SELECT, "-p", "no:cacheprovider"],
                    ROOT, dict(os.environ, MUDG_DEBUG_VARIANTS="1", **env), 900, alone=(i == 0))
    for env in VARIANTS_X3:
        runs.submit("x3:" + _name(env), [sys.executable, "-m", "pytest", "tests/test_operand_modes_gpu.py", "tests/test_unet_gpu.py", "-m", "gpu", "-q",
                                         "-p", "no:cacheprovider"],
                    ROOT, dict(os.environ, MUDG_DEBUG_VARIANTS="1", MUDG_OPERAND="bf16x3", MUDG_PARITY_CHILD="1", MUDG_SKIP_FULLSIZE_ORACLE="1",
                               MUDG_SKIP_CONFIG0_CUT="1", **env), 900)
    yield runs
    runs.shutdown()


@pytest.mark.parametrize("env", VARIANTS, ids=_name)
def test_kernel_parity_under_variant(cuda, env, request):
    if os.environ.get("MUDG_DEBUG_VARIANTS") == "1":
        pytest.skip("already running under a variant switch")
    rc, stdout = request.getfixturevalue("children").result(_name(env))
    print("\n".join(l for l in stdout.splitlines() if "passed" in l or "failed" in l or l.startswith("[child")))
    assert rc == 0, stdout[-5000:]


@pytest.mark.parametrize("env", VARIANTS_X3, ids=_name)
def test_kernel_parity_under_variant_in_the_bf16x3_build(cuda, env, request):
    if os.environ.get("MUDG_DEBUG_VARIANTS") == "1" or os.environ.get("MUDG_PARITY_CHILD") == "1":
        pytest.skip("already running under a variant switch / inside a mode child")
    rc, stdout = request.getfixturevalue("children").result("x3:" + _name(env))
    print("\n".join(l for l in stdout.splitlines() if "passed" in l or "failed" in l or l.startswith("[child")))
    assert rc == 0, stdout[-5000:]
