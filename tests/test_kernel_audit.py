"""The update kernels keep data in the accumulation registers a[0:63] BY NAME (inline assembly: the accumulators of
k_update, the parked tiles of the run launch's panel solve).  The compiler knows them only as clobbers, so nothing but
an audit of the generated code shows that it never uses one for a value of its own (cdna_hip_programming.md 5.7 item 4):
no `v_accvgpr_*` / AGPR operand outside an asm block wherever named registers hold data (the update and panel-solve paths),
MFMA builtins in VGPR form, no spills in the level-by-level kernels."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pastix_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC) and not shutil.which("hipcc"), reason="no hipcc")
def test_no_compiler_use_of_the_named_accumulators(tmp_path):
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^FLAGS_kernels_update\s*=\s*(.*)$", mk, re.M).group(1).split()
    out = str(tmp_path / "ku.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", *flags, "-S", "--cuda-device-only",
                           os.path.join(CSRC, "kernels_update.hip"), "-o", out], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    kernels = [i for i, l in enumerate(lines) if re.match(r"^_ZN10pastix_amd(8k_update|12k_run_update)", l) and l.rstrip().endswith(":") is False and ":" in l]
    assert len(kernels) >= 3
    # Where the compiler MAY use accumulation registers for values of its own: inside a diagonal-blok ticket of the run launch
    # (between the DIAG_TICKET markers of k_run_update<0 / 1, true>: real LLt / LDLt bloks live in LDS, the update path's
    # accumulators are dead there and nothing else is parked) -- everywhere else every AGPR belongs to the asm statements.
    for start in kernels:
        name = lines[start].split(":")[0]
        onek = re.search(r"k_run_updateILi(\d)ELb1", name)
        inasm = diag = False
        nasm = ndiag = nown = 0
        for l in lines[start:]:
            if ".Lfunc_end" in l:
                break
            if "PASTIX_AMD_DIAG_TICKET_BEGIN" in l:
                assert not diag, name
                diag = True
                ndiag += 1
            elif "PASTIX_AMD_DIAG_TICKET_END" in l:
                assert diag, name
                diag = False
            if "ASMSTART" in l:
                inasm = True
                nasm += 1
            elif "ASMEND" in l:
                inasm = False
            elif not inasm:
                code = l.split(";")[0]
                if "accvgpr" in code or re.search(r"\ba\[?\d", code):
                    assert diag, (name, l)
                    nown += 1
        assert not diag, name
        assert nasm > 100, name
        if onek:
            assert ndiag == 1 and int(onek.group(1)) <= 1, name
        else:
            assert ndiag == 0 and nown == 0, name
    txt = "\n".join(lines)
    # per kernel metadata: 64 accumulation registers, no scratch in the level-by-level kernels
    meta = txt[txt.index("amdhsa.kernels:"):]
    blocks = [b for b in meta.split("\n  - ") if ".name:" in b]
    seen = 0
    for b in blocks:
        nm = re.search(r"\.name:\s+(\S+)", b).group(1)
        if "k_update" in nm or "k_run_update" in nm:
            seen += 1
            assert re.search(r"\.agpr_count:\s+64", b), nm
            if "k_run_update" not in nm:
                assert re.search(r"\.vgpr_spill_count:\s+0", b), nm
    assert seen >= 3
