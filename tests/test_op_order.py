"""CPU: the per-Gaussian projection of the HIP preprocess kernels is the SAME tree of IEEE operations as the reference build's
preprocessCUDA (tests/golden/preprocess_op_order.txt, derived from the reference build's assembly by tools/check_op_order.py
--make-fixture).  Equal trees <=> bit-identical projected centres and conics on every input, which is what every
alpha >= 1/255 decision of the blends is a function of (gaustar_amd/csrc/gsr_ref_order.h; profiles/r06_parity_report.txt: 94 % of
the threshold flips against the reference build came from last-bit differences of exactly these values).  Compiles
gsr_preprocess.hip to assembly (hipcc, no GPU needed) and executes it symbolically (tools/symfp.py)."""
import os
import shutil
import sys

import pytest

from conftest import ROOT


def test_preprocess_kernels_compute_centre_and_conic_in_the_reference_builds_operation_order():
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("needs hipcc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_op_order
    assert check_op_order.check() == 0


def test_symbolic_executor_on_a_known_snippet(tmp_path):
    """tools/symfp.py itself: packed f32 with op_sel / neg modifiers, fmac, the IEEE division sequence, 2 x == x + x."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import symfp
    asm = """k:
	s_load_dwordx2 s[4:5], s[0:1], 0x10
	s_waitcnt lgkmcnt(0)
	global_load_dwordx2 v[2:3], v1, s[4:5]
	global_load_dwordx2 v[4:5], v1, s[4:5] offset:8
	v_pk_mul_f32 v[6:7], v[2:3], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]
	v_pk_fma_f32 v[8:9], v[2:3], v[4:5], v[6:7] neg_lo:[0,0,1] neg_hi:[0,0,1]
	v_fmac_f32_e32 v8, v2, v3
	v_mul_f32_e32 v10, 2.0, v8
	v_div_scale_f32 v11, s[2:3], v9, v9, v10
	v_rcp_f32_e32 v12, v11
	v_div_fmas_f32 v11, v11, v12, v12
	v_div_fixup_f32 v13, v11, v9, v10
	global_store_dword v1, v13, s[4:5]
	s_endpgm
"""
    f = tmp_path / "k.s"
    f.write_text(asm)
    M = symfp.run(symfp.extract(str(f), "k"), {0x10: "a"})
    (_ln, _t, vals), = M.stores
    prog, res = symfp.ssa([("out", vals[0])], {"a[0]": "A", "a[1]": "B", "a[2]": "C", "a[3]": "D"})
    text = "\n".join(prog)
    # lo lane: A*C - (B*C) then + A*B, doubled as a sum; hi lane: B*D - (A*D); quotient = a true division
    assert "mul(B, C)" in text and "mul(A, D)" in text and "fma(A, C, -t0)" in text.replace("t1", "t0") or "fma(A, C" in text
    assert any(l.endswith("= add(%s, %s)" % (l.split(" = ")[1][4:].split(",")[0], l.split(" = ")[1][4:].split(",")[0])) for l in prog if "= add(" in l)
    assert prog[-1].startswith(res[0][1] + " = div(")
