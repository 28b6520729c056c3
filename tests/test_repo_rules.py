"""Structural rules of the scope contract: the oracle is test infrastructure, the product has no fallback."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(*dirs):
    for d in dirs:
        for base, _, files in os.walk(os.path.join(ROOT, d)):
            if "__pycache__" in base:
                continue
            for f in files:
                if f.endswith(".py"):
                    yield os.path.join(base, f)


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    offenders = [p for p in _py_files("mudg_amd", "lvdm", "utils") if pat.search(open(p).read())]
    assert not offenders, offenders


def test_oracle_headers_say_test_infrastructure():
    for p in _py_files("oracle"):
        assert "TEST INFRASTRUCTURE" in open(p).read(), p


def test_no_compat_layers_in_tree():
    bad = re.compile(r"__HIP_PLATFORM_AMD__|hipify|import triton|from triton")
    csrc = os.path.join(ROOT, "mudg_amd", "csrc")
    for p in list(_py_files("mudg_amd", "lvdm")) + [os.path.join(csrc, f) for f in os.listdir(csrc)]:
        assert not bad.search(open(p).read()), p


def test_gpu_side_code_does_not_read_the_reference_tree():
    for p in _py_files("tests", "mudg_amd", "lvdm", "oracle"):
        if p.endswith("make_golden.py") or p.endswith("test_repo_rules.py"):
            continue
        assert "/root/reference" not in open(p).read(), p
    for name in ("bench.py", "__graft_entry__.py"):
        assert "/root/reference" not in open(os.path.join(ROOT, name)).read()
