"""The update kernels keep data in the accumulation registers a[0:63] BY NAME (inline assembly: the accumulators of
k_update, the parked tiles of the run launch's panel solve).  The compiler does not know they are live, so it must never
allocate an AGPR itself.  Round 5's build says so to the register allocator (csrc/Makefile: the kernels' IR attribute
groups get "amdgpu-agpr-alloc"="0" -- every AGPR reserved -- and a 64-VGPR budget; before that only THIS audit stood between
a change of the kernel and an allocator that reloaded a spilled tuple into a[0:31]).  The audit stays, on the assembly of the
same code generation the library is linked from (make build/ku.s): no `v_accvgpr_*` / AGPR operand outside an asm block
ANYWHERE in the update kernels, 64 + 64 registers per wave, MFMA builtins in VGPR form, no spills in the level-by-level
kernels (cdna_hip_programming.md 5.7 item 4)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pastix_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC) and not shutil.which("hipcc"), reason="no hipcc")
def test_no_compiler_use_of_the_named_accumulators():
    subprocess.check_call(["make", "-C", CSRC, "build/ku.s"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ll = open(os.path.join(CSRC, "build", "ku_r.ll")).read()
    groups = [l for l in ll.split("\n") if l.startswith("attributes #") and '"amdgpu-flat-work-group-size"' in l]
    assert groups and all('"amdgpu-agpr-alloc"="0"' in g and '"amdgpu-num-vgpr"="32"' in g for g in groups)
    lines = open(os.path.join(CSRC, "build", "ku.s")).read().split("\n")
    kernels = [i for i, l in enumerate(lines) if re.match(r"^_ZN10pastix_amd(8k_update|12k_run_update)\S*:", l)]
    assert len(kernels) >= 9
    for start in kernels:
        name = lines[start].split(":")[0]
        onek = re.search(r"k_run_updateILi(\d)ELb1", name)
        inasm = False
        nasm = ndiag = 0
        for l in lines[start:]:
            if ".Lfunc_end" in l:
                break
            if "PASTIX_AMD_DIAG_TICKET_BEGIN" in l:
                ndiag += 1
            if "ASMSTART" in l:
                inasm = True
                nasm += 1
            elif "ASMEND" in l:
                inasm = False
            elif not inasm:
                code = l.split(";")[0]
                assert not ("accvgpr" in code or re.search(r"\ba\[?\d", code)), (name, l)
        assert nasm > 100, name
        # the diagonal-blok tickets exist in the one-kernel instances of real LLt / LDLt only
        assert ndiag == (1 if onek else 0), name
        if onek:
            assert int(onek.group(1)) <= 1, name
    txt = "\n".join(lines)
    # per kernel metadata: 64 VGPRs + 64 accumulation registers, no scratch in the level-by-level kernels
    meta = txt[txt.index("amdhsa.kernels:"):]
    blocks = [b for b in meta.split("\n  - ") if ".name:" in b]
    seen = 0
    for b in blocks:
        nm = re.search(r"\.name:\s+(\S+)", b).group(1)
        if "k_update" in nm or "k_run_update" in nm:
            seen += 1
            assert re.search(r"\.agpr_count:\s+64", b), nm
            assert int(re.search(r"\.vgpr_count:\s+(\d+)", b).group(1)) <= 128, nm
            if "k_run_update" not in nm:
                assert re.search(r"\.vgpr_spill_count:\s+0", b), nm
    assert seen >= 9
    # the compiler's own MFMAs (panel-solve tickets) are in VGPR form: no AGPR destination outside the asm blocks is
    # already covered above; the flag that asks for it must be on the code generation line
    mk = open(os.path.join(CSRC, "Makefile")).read()
    cg = mk[mk.index("KU_CODEGEN ="):]
    assert "-amdgpu-mfma-vgpr-form" in cg[:cg.index("\n\n")]
