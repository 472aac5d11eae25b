"""tools/check_vmem_hazards.py: the static check that no instruction touches a VGPR an outstanding vector-memory load still has
to write (in-order retirement, `s_waitcnt vmcnt(N)` = at most N outstanding).  It guards the hand-counted forms of the one-launch
decode member (csrc/wqaa_gemm_kernel.h, DESIGN 3.2a''): their loads are inline assembly, invisible to the compiler's own wait
insertion.  First the checker itself on hand-made instruction streams, then the built library."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import check_vmem_hazards as chk  # noqa: E402


def _asm(*instrs):
    """fake llvm-objdump lines: 4-byte instructions from address 0x1000 (an 8-byte one is written with a trailing '+')"""
    lines, addr = [], 0x1000
    for t in instrs:
        wide = t.endswith("+")
        t = t.rstrip("+")
        lines.append(f"\t{t:<60}// {addr:012X}: 00000000")
        addr += 8 if wide else 4
    return lines


def test_read_before_the_wait_is_found_and_the_wait_clears_it():
    bad = _asm("global_load_dwordx4 v[4:7], v[0:1], off+", "v_add_u32_e32 v8, v4, v5", "s_endpgm")
    assert [f[2] for f in chk.scan("k", bad)] == [[4, 5]]
    good = _asm("global_load_dwordx4 v[4:7], v[0:1], off+", "s_waitcnt vmcnt(0)", "v_add_u32_e32 v8, v4, v5", "s_endpgm")
    assert chk.scan("k", good) == []


def test_counted_waits_follow_the_order_of_issue():
    prog = ["global_load_dwordx4 v[4:7], v[0:1], off+", "global_load_dwordx4 v[8:11], v[0:1], off+", "global_load_lds_dwordx4 v[0:1], off+",
            "global_store_dword v[2:3], v20, off+"]
    # two operations younger than the second load: vmcnt(2) hands over both loads, vmcnt(3) only the first
    assert chk.scan("k", _asm(*prog, "s_waitcnt vmcnt(2)", "v_mov_b32_e32 v30, v9", "s_endpgm")) == []
    assert chk.scan("k", _asm(*prog, "s_waitcnt vmcnt(3)", "v_mov_b32_e32 v30, v5", "s_endpgm")) == []
    assert len(chk.scan("k", _asm(*prog, "s_waitcnt vmcnt(3)", "v_mov_b32_e32 v30, v9", "s_endpgm"))) == 1
    # overwriting a destination in flight (a spill reload, a copy) is a finding too
    assert len(chk.scan("k", _asm(*prog, "v_mov_b32_e32 v10, 0", "s_waitcnt vmcnt(0)", "s_endpgm"))) == 1


def test_paths_are_merged_by_age_and_exclusive_arms_do_not_alias():
    # if / else: the load is in one arm, the constant in the other (what a linear scan would flag)
    prog = _asm("s_cbranch_vccz 3",                                   # -> else arm
                "global_load_dwordx2 v[96:97], v[0:1], off+",
                "s_branch 1",                                         # -> join
                "v_mov_b32_e32 v96, 0",
                "s_waitcnt vmcnt(0)",
                "v_add_u32_e32 v1, v96, v97",
                "s_endpgm")
    assert chk.scan("k", prog) == []
    # a load issued before a loop and read inside it without a wait is found on the path around the back edge as well
    loop = _asm("global_load_dword v5, v[0:1], off+",
                "v_add_u32_e32 v6, 1, v6",
                "s_cbranch_scc1 65534",                               # back to the add
                "v_mov_b32_e32 v7, v5",
                "s_endpgm")
    assert [f[2] for f in chk.scan("k", loop)] == [[5]]


def test_the_hand_counted_decode_kernels_of_the_built_library_are_clean():
    from bitblas_amd import lib as wlib
    if not (os.path.exists(chk.OBJDUMP) and shutil.which("objcopy")):
        pytest.skip("no llvm-objdump / objcopy on this box")
    import io
    from contextlib import redirect_stdout
    buf = io.StringIO()
    old = sys.argv
    sys.argv = ["check_vmem_hazards.py", "--lib", wlib.LIB_PATH]
    try:
        with redirect_stdout(buf):
            rc = chk.main()
    finally:
        sys.argv = old
    out = buf.getvalue()
    assert rc == 0, out[-3000:]
    assert "kernels checked, 0 with findings" in out and int(out.strip().split("\n")[-1].split()[0]) >= 8, out[-500:]


def test_store_data_overwritten_within_two_wait_states_is_found():
    """round 5: a vector-memory store of more than 64 bits reads its data registers for two more issue cycles; an inline-assembly
    store gets no hazard padding from the compiler (tools/store_hazard_lab.hip: lanes 8-15 / 12-15 of every 16 store the overwriting
    value at gap 0 / 1, nothing from gap 2 on).  The mid-M member's first two-launch build had exactly this stream."""
    st = "global_store_dwordx4 v[34:35], v[2:5], off sc0 sc1+"
    assert [f[2] for f in chk.scan("k", _asm(st, "v_lshl_add_u64 v[2:3], v[0:1], 0, s[2:3]+", "s_endpgm"))] == [[2, 3]]
    assert len(chk.scan("k", _asm(st, "s_lshl_b64 s[2:3], s[2:3], 10", "v_mov_b32_e32 v5, 0", "s_endpgm"))) == 1        # one wait state: still inside
    assert chk.scan("k", _asm(st, "s_nop 1", "v_mov_b32_e32 v2, 0", "s_endpgm")) == []
    assert chk.scan("k", _asm(st, "s_mov_b32 s2, 0", "s_mov_b32 s3, 0", "v_mov_b32_e32 v2, 0", "s_endpgm")) == []          # two instructions in between
    assert chk.scan("k", _asm(st, "v_mov_b32_e32 v34, 0", "v_mov_b32_e32 v6, 0", "s_endpgm")) == []                         # the ADDRESS may be re-used
    assert chk.scan("k", _asm("global_store_dwordx2 v[34:35], v[2:3], off+", "v_mov_b32_e32 v2, 0", "s_endpgm")) == []     # 64 bits: no hazard
    # across a branch: the window follows both arms
    prog = _asm(st, "s_cbranch_scc1 1", "s_nop 0", "v_mov_b32_e32 v4, 0", "s_endpgm")
    assert len(chk.scan("k", prog)) == 1
    # a LOAD into the data registers is no overwrite inside the window (its data is a memory latency away; the compiler's spill code
    # has `scratch_store v[a:d]; global_load v[a:d]` back to back)
    assert chk.scan("k", _asm(st, "global_load_dwordx4 v[2:5], v[0:1], off+", "s_waitcnt vmcnt(0)", "s_endpgm")) == []


def test_the_compilers_long_branch_is_followed_and_branches_on_constants_are_pruned():
    """round 5: with the K-sliced form in it a decode kernel grew past the 16-bit branch range and the compiler wrote
    `s_getpc_b64; s_add_u32; s_addc_u32; s_setpc_b64` - without following it the walk fell through into the code behind the jump with the
    jumping path's loads in flight (866 findings per kernel, none real)."""
    load = "global_load_dwordx4 v[4:7], v[0:1], off+"
    # 0: load (8 B)  8: getpc (4)  12: add (8)  20: addc (8)  28: setpc (4)  32: v_add (touches the load: NOT reachable)  36: waitcnt  40: endpgm
    prog = _asm(load, "s_getpc_b64 s[98:99]", "s_add_u32 s98, s98, 0x18+", "s_addc_u32 s99, s99, 0+", "s_setpc_b64 s[98:99]",
                "v_add_u32_e32 v8, v4, v5", "s_waitcnt vmcnt(0)", "s_endpgm")
    assert chk.scan("k", prog) == []
    # the same with the jump landing ON the instruction that touches the load: found
    bad = _asm(load, "s_getpc_b64 s[98:99]", "s_add_u32 s98, s98, 0x14+", "s_addc_u32 s99, s99, 0+", "s_setpc_b64 s[98:99]",
               "v_add_u32_e32 v8, v4, v5", "s_waitcnt vmcnt(0)", "s_endpgm")
    assert len(chk.scan("k", bad)) == 1
    # a branch on a constant: `s_mov_b64 sx, 0; s_and_b64 vcc, exec, sx; s_cbranch_vccnz` is never taken
    never = _asm(load, "s_mov_b64 s[4:5], 0", "s_and_b64 vcc, exec, s[4:5]", "s_cbranch_vccnz 1", "s_branch 1", "v_add_u32_e32 v8, v4, v5",
                 "s_waitcnt vmcnt(0)", "s_endpgm")
    assert chk.scan("k", never) == []
    with pytest.raises(RuntimeError):
        chk.scan("k", _asm(load, "s_setpc_b64 s[30:31]", "s_endpgm"))


def test_every_gemm_kernel_of_the_built_library_is_clean_of_both_hazards():
    """the whole GEMM family (the split-K and ping-pong members write their partial sums / output tiles with inline-assembly
    write-through stores too)"""
    from bitblas_amd import lib as wlib
    if not (os.path.exists(chk.OBJDUMP) and shutil.which("objcopy")):
        pytest.skip("no llvm-objdump / objcopy on this box")
    import io
    from contextlib import redirect_stdout
    buf = io.StringIO()
    old = sys.argv
    sys.argv = ["check_vmem_hazards.py", "--lib", wlib.LIB_PATH, "--match", "wq_gemm_(pp|mid)_kernel|wq_gemm_kernelINS_10GemmPolicyILi0ELi1ELi0ELi2"]
    try:
        with redirect_stdout(buf):
            rc = chk.main()
    finally:
        sys.argv = old
    out = buf.getvalue()
    assert rc == 0, out[-3000:]
    assert "kernels checked, 0 with findings" in out and int(out.strip().split("\n")[-1].split()[0]) >= 80, out[-500:]


def test_the_mid_m_kernels_of_the_built_library_are_clean():
    """round 5: wq_gemm_mid_kernel (csrc/wqaa_gemm_mid_kernel.h) issues every register load as inline assembly next to its LDS-DMA
    and waits once - the same contract, checked the same way (its first build had a run-time branch that joined two register
    assignments of in-flight loads: 30 of 70 instantiations had findings)"""
    from bitblas_amd import lib as wlib
    if not (os.path.exists(chk.OBJDUMP) and shutil.which("objcopy")):
        pytest.skip("no llvm-objdump / objcopy on this box")
    import io
    from contextlib import redirect_stdout
    buf = io.StringIO()
    old = sys.argv
    sys.argv = ["check_vmem_hazards.py", "--lib", wlib.LIB_PATH, "--match", "wq_gemm_mid_kernel"]
    try:
        with redirect_stdout(buf):
            rc = chk.main()
    finally:
        sys.argv = old
    out = buf.getvalue()
    assert rc == 0, out[-3000:]
    assert "kernels checked, 0 with findings" in out and int(out.strip().split("\n")[-1].split()[0]) >= 50, out[-500:]
