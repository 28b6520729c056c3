"""The C-ABI without Python in the loop: tests/cabi/smoke.c is plain C99 against include/mudg_hip.h and libmudg_hip.so.
CPU: the header compiles as C and every entry point links.  GPU: the same binary runs a GEMM through the ABI, checks it
against host arithmetic and checks that a bad descriptor is refused with a message."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROCM = "/opt/rocm"


def _build(tmp_path):
    from mudg_amd import build as mbuild
    mbuild.build(verbose=False)
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.isdir(ROCM):
        pytest.skip("gcc / ROCm not available")
    exe = str(tmp_path / "cabi_smoke")
    lib = os.path.join(ROOT, "mudg_amd")
    cmd = [gcc, "-std=c99", "-Wall", "-Werror=implicit-function-declaration", os.path.join(ROOT, "tests", "cabi", "smoke.c"),
           "-I" + os.path.join(ROOT, "include"), "-I" + ROCM + "/include", "-D__HIP_PLATFORM_AMD__", "-L" + lib, "-lmudg_hip",
           "-L" + ROCM + "/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + lib, "-Wl,-rpath," + ROCM + "/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_header_is_c_and_every_entry_point_links(tmp_path):
    r = subprocess.run([_build(tmp_path), "--link-only"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "symbols linked" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_gemm_through_the_c_abi_without_python(cuda, tmp_path):
    r = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and "gemm rel-L2" in r.stdout and "refused as expected" in r.stdout, r.stdout + r.stderr
