"""Properties of the generated gfx950 code that the source relies on and the compiler does not know about (CPU: hipcc cross-compiles).

wq_kernel<..., RS = 2> (csrc/wgemm.hip, the deferred residual of the 160 x 320 tile) requests a lane's 15 residual pieces with inline-assembly
loads: the compiler believes their destination registers are valid at once, the kernel guarantees their arrival by a counted s_waitcnt in
front of the epilogue.  Correct only if NOTHING reads or moves those registers in between — a live-range split, a spill, a copy of a
pending register would carry garbage.  This test compiles the file to assembly and checks exactly that, for every instantiation."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regs(line):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", line):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", line):
        out.add(int(a))
    return out


@pytest.fixture(scope="module")
def wgemm_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from mudg_amd import build
    out = tmp_path_factory.mktemp("isa") / "wgemm.s"
    cmd = [hipcc, *build.FLAGS, "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-S", os.path.join(ROOT, "mudg_amd", "csrc", "wgemm.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    return out.read_text()


def test_deferred_residual_registers_are_untouched_until_the_counted_wait(wgemm_asm):
    s = wgemm_asm
    kernels = sorted(set(re.findall(r"^(_ZN[^:\s]*9wq_kernelILi\dELi5ELb0ELi2ELi5EEE[^:\s]*):", s, re.M)))
    assert len(kernels) == 3, kernels                                      # plain GEMM, 3x3 conv, temporal conv
    for name in kernels:
        i = s.index(name + ":")
        body = [l.strip() for l in s[i:s.index(".end_amdhsa_kernel", i)].split("\n")]
        pending, last = set(), None
        for k, l in enumerate(body):
            m = re.match(r"global_load_dwordx[24] v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\]", l)
            if m:
                pending |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                last = k
        assert len(pending) == 50, (name, len(pending))                    # 5 rows x (2 x 4 + 2) registers
        mfma = [k for k, l in enumerate(body) if l.startswith("v_mfma")]
        assert mfma and last < mfma[0]
        wait = [k for k, l in enumerate(body) if l.startswith("s_waitcnt vmcnt(0)") and k > mfma[-1]]
        assert wait, name
        for k in range(last + 1, wait[0]):
            l = body[k]
            if not l or l.startswith((";", ".")) or l.endswith(":"):
                continue
            assert not (_regs(l) & pending), f"{name}: `{l}` touches a pending residual register"
            assert not l.startswith("scratch_"), f"{name}: scratch traffic inside the counted region: `{l}`"
        # every lane of every wave issues exactly 15 of them, unconditionally (the counted waits add 15)
        loads = [k for k, l in enumerate(body) if re.match(r"global_load_dwordx[24] v\[\d+:\d+\], v\d+, s\[", l)]
        assert len(loads) == 15
        assert not any(l.startswith(("s_cbranch", "s_branch")) or l.endswith(":") for l in body[loads[0]:loads[-1]] if l and not l.startswith(";")), name


def test_tile_kernels_of_wgemm_do_not_spill(wgemm_asm):
    """A scratch reload inside a K loop is a vector-memory operation the hand-counted vmcnt waits of these kernels do not know (and it
    carries its own vmcnt(0): the DMA ring drains).  Every kernel of the file — the 288-row tile in its one-tile and persistent forms, the
    half-height GEGLU kernel, the 160-row tile — is held to zero scratch in the release build."""
    md = wgemm_asm[wgemm_asm.index("amdhsa.kernels"):]
    seen = {}
    for m in re.finditer(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)", md, re.S):
        name, scratch = m.group(1), int(m.group(2))
        for family in ("wq_kernel", "wgemm_kernel", "wgemm_pkernel", "hgeglu_kernel"):
            if family in name:
                seen[family] = seen.get(family, 0) + 1
                assert scratch == 0, (name, scratch)
    assert seen.get("wq_kernel", 0) >= 10 and seen.get("wgemm_kernel", 0) >= 4 and seen.get("wgemm_pkernel", 0) >= 2 and seen.get("hgeglu_kernel", 0) >= 2, seen
