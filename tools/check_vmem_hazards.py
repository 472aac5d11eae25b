#!/usr/bin/env python
"""Static check of the built library: no instruction touches a VGPR that an OUTSTANDING vector-memory load still has to write.

    python tools/check_vmem_hazards.py [--lib bitblas_amd/libwqaa_hip.so] [--match REGEX] [-v]

Why: the hand-counted forms of `wq_gemm_decode_lds_kernel` (csrc/wqaa_gemm_kernel.h, DESIGN 3.2a'') issue their loads as inline
assembly and count the `s_waitcnt vmcnt(N)` by hand - the compiler does not know the destination registers are in flight, so a
spill, a copy or a re-use between a load and its wait would go unnoticed (it did once: every output NaN).  The model is the
hardware's: vector-memory operations of a wave retire in order, `vmcnt(N)` waits until at most N are outstanding; loads AND
stores count (gfx9 family).  Input: llvm-objdump of the code objects inside the library's .hip_fatbin.
The analysis is a forward data flow over each kernel's control-flow graph (branch targets from the disassembly's addresses): the
abstract state is the queue of outstanding operations' destination registers, youngest first; at a join the queues are merged
position by position.  Compiler-tracked code passes by construction (the compiler's own waits satisfy the same model), so a
finding is a place where hand-written loads and waits - or a future compiler - broke the contract."""
from __future__ import annotations

import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
DEFAULT_MATCH = r"wq_gemm_decode_lds_kernelINS_10GemmPolicyILi[04]ELi\dELi0ELi[123]E"   # int4 / lut4, 16-bit activations, Scale (+ Zeros)


BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def _compressed_bundles(blob):
    """the library is built with --offload-compress (bitblas_amd/build.py): its .hip_fatbin holds one "CCOB" container per
    translation unit (header: magic, u16 version, u16 method, then the container's size - u32 in version 2, u64 in version 3);
    clang-offload-bundler unbundles (and decompresses) a container given as a file"""
    import tempfile
    pos = 0
    while True:
        i = blob.find(b"CCOB", pos)
        if i < 0:
            return
        ver, = struct.unpack_from("<H", blob, i + 4)
        size = struct.unpack_from("<Q", blob, i + 8)[0] if ver >= 3 else struct.unpack_from("<I", blob, i + 8)[0]
        if size <= 24 or i + size > len(blob):
            pos = i + 4
            continue
        with tempfile.TemporaryDirectory() as tmp:
            src, dst = os.path.join(tmp, "bundle.bin"), os.path.join(tmp, "dev.o")
            with open(src, "wb") as f:
                f.write(blob[i: i + size])
            res = subprocess.run([BUNDLER, "--unbundle", "--type=o", f"--input={src}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                  f"--output={dst}"], capture_output=True)
            if res.returncode == 0 and os.path.exists(dst):
                with open(dst, "rb") as f:
                    yield f.read()
        pos = i + size


def code_objects(path):
    blob = subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, "/dev/stdout"], capture_output=True).stdout
    if b"CCOB" in blob[:64] or (b"__CLANG_OFFLOAD_BUNDLE__" not in blob and b"CCOB" in blob):
        yield from _compressed_bundles(blob)
        return
    magic, pos = b"__CLANG_OFFLOAD_BUNDLE__", 0
    while True:
        i = blob.find(magic, pos)
        if i < 0:
            return
        n, = struct.unpack_from("<Q", blob, i + len(magic))
        p = i + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p: p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                yield blob[i + off: i + off + size]
        pos = i + len(magic)


REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def first_operand_regs(ops):
    return regs_of(ops.split(",")[0])


VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")
ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")
CAP = 64                                                        # vmcnt counts to 63
CHECK_LDS = False                                               # --lds-dma: also no DS instruction while an LDS-DMA operation is outstanding


def parse(lines):
    """-> list of (address, opcode, operands, text)"""
    out = []
    for line in lines:
        m = ADDR.search(line)
        code = line.split("//")[0].strip()
        if not m or not code or code.endswith(":"):
            continue
        parts = code.split(None, 1)
        out.append((int(m.group(1), 16), parts[0], parts[1] if len(parts) > 1 else "", code))
    return out


LDS_DMA = frozenset({-1})                                       # marker: an operation that writes LDS (global_load_lds_*, buffer_load ... lds)


def dest_of(op, ops):
    if re.search(r"\blds\b", ops) or "_load_lds_" in op:      # the data goes to LDS (with `buffer_load ... lds` the first operand is an address)
        return LDS_DMA
    if "_load" in op and "lds" not in op:
        return frozenset(first_operand_regs(ops))
    if "_atomic" in op and ("sc0" in ops or "glc" in ops):
        return frozenset(first_operand_regs(ops))
    return frozenset()


def step(state, ins, report=None):
    """one instruction on the abstract queue: a tuple of register sets, YOUNGEST first (position = number of younger operations)"""
    addr, op, ops, code = ins
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", ops)
        return state[: int(m.group(1))] if m else state
    if report is not None and state and CHECK_LDS and op.startswith("ds_") and any(-1 in regs for regs in state):
        report.append((addr, code, ["LDS"], max(age for age, regs in enumerate(state) if -1 in regs)))
    if report is not None and state:
        touched = regs_of(ops)
        if VMEM.match(op) and dest_of(op, ops) and dest_of(op, ops) != LDS_DMA:
            # a load into a register an older load still writes is in order (loads retire in order): only its address operands count
            touched = regs_of(ops.split(",", 1)[1]) if "," in ops else set()
        if touched:
            for age, regs in enumerate(state):
                hit = touched & regs
                if hit:
                    report.append((addr, code, sorted(hit), age))
                    break
    if VMEM.match(op):
        state = ((dest_of(op, ops),) + state)[:CAP]
    return state


def merge(a, b):
    if a is None:
        return b
    if len(a) < len(b):
        a, b = b, a
    return tuple((a[i] | b[i]) if i < len(b) else a[i] for i in range(len(a)))


WIDE_STORE = re.compile(r"^(global|buffer|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)$")
STORE_DATA_WAIT_STATES = 2          # gfx950, measured (tools/store_hazard_lab.hip -> profiles/r05_lab_store_hazard.txt); the compiler's own: s_nop 1


def store_data_hazards(ins, index_of_addr):
    """Round 5.  A vector-memory store of MORE than 64 bits reads its data registers over the two issue cycles after its own: an
    instruction that writes one of them within STORE_DATA_WAIT_STATES wait states stores its own result in lanes 8-15 (no gap) /
    12-15 (one wait state) of every 16.  The compiler's hazard recognizer pads stores it knows (`s_nop 1`); an inline-assembly store
    is opaque to it - csrc/wqaa_gemm_mid_kernel.h's first two-launch build re-used a store's data registers for the next address
    and every output of two rows per fragment was wrong.  -> [(addr, code, registers, -1)]"""
    out = []

    def walk(k, data, left, seen, origin):
        while left > 0 and k < len(ins):
            if (k, left) in seen:
                return
            seen.add((k, left))
            addr, op, ops, code = ins[k]
            if op == "s_endpgm":
                return
            if op == "s_nop":
                left -= int(ops.split()[0], 0) + 1
                k += 1
                continue
            written = set()
            if op.startswith("v_") and not op.startswith(("v_cmp", "v_nop", "v_readfirstlane", "v_readlane")):
                written = first_operand_regs(ops)
            # (a load that names the data registers as its destination is no writer here: its data comes back a memory latency later,
            # and the compiler itself issues `scratch_store v[a:d]; global_load v[a:d]` back to back in its spill code)
            hit = written & data
            if hit:
                out.append((origin[0], origin[1] + "   <-   " + code, sorted(hit), -1))
                return
            left -= 1
            if op == "s_branch" or op.startswith("s_cbranch"):
                off = int(ops.split()[0], 0)
                off = off - 65536 if off >= 32768 else off
                t = index_of_addr.get(addr + 4 + 4 * off)
                if t is not None:
                    walk(t, data, left, seen, origin)
                if op == "s_branch":
                    return
            k += 1

    for k, (addr, op, ops, code) in enumerate(ins):
        if not WIDE_STORE.match(op):
            continue
        parts = [x.strip() for x in ops.split(",")]
        data = regs_of(parts[0]) if op.startswith("buffer") else (regs_of(parts[1]) if len(parts) > 1 else set())
        if data:
            walk(k + 1, data, STORE_DATA_WAIT_STATES, set(), (addr, code))
    return out


def long_branch_target(ins, k):
    """target of the s_setpc_b64 at index k when it ends the compiler's long-branch sequence, else None"""
    if k < 3:
        return None
    (a0, o0, p0, _), (_, o1, p1, _), (_, o2, p2, _), (_, _, p3, _) = ins[k - 3], ins[k - 2], ins[k - 1], ins[k]
    m0 = re.fullmatch(r"s\[(\d+):(\d+)\]", p0.strip())
    if o0 != "s_getpc_b64" or o1 != "s_add_u32" or o2 != "s_addc_u32" or not m0 or p3.strip() != p0.strip():
        return None
    lo, hi = int(m0.group(1)), int(m0.group(2))
    f1 = [x.strip() for x in p1.split(",")]
    f2 = [x.strip() for x in p2.split(",")]
    if f1[:2] != [f"s{lo}", f"s{lo}"] or f2[:2] != [f"s{hi}", f"s{hi}"] or len(f1) != 3 or len(f2) != 3:
        return None
    imm = int(f1[2], 0) & 0xFFFFFFFF
    carry_hi = int(f2[2], 0)
    if carry_hi not in (0, -1, 0xFFFFFFFF):
        return None
    if imm >= 0x80000000:
        imm -= 1 << 32
    return a0 + 4 + imm


def scan(name, lines, verbose=False):
    """forward data flow over the kernel's control-flow graph: at a join the queues are merged position by position, counted from
    the youngest operation (sound for in-order retirement: whatever a path has outstanding is in the merged queue at the same age)"""
    ins = parse(lines)
    if not ins:
        return []
    index = {a: k for k, (a, _, _, _) in enumerate(ins)}
    target = {}
    leaders = {0}
    for k, (addr, op, ops, _) in enumerate(ins):
        if op == "s_setpc_b64":
            # the compiler's long branch (a kernel beyond the 16-bit branch range): s_getpc_b64 s[a:a+1]; s_add_u32 sa, sa, imm;
            # s_addc_u32 sa+1, sa+1, 0|-1; s_setpc_b64 s[a:a+1] - the target is (address after s_getpc) + imm.  Anything else: refuse
            t = long_branch_target(ins, k)
            if t is None or index.get(t) is None:
                raise RuntimeError(f"{name}: s_setpc_b64 at {addr:#x} is not a long branch this tool can follow")
            if k + 1 < len(ins):
                leaders.add(k + 1)
            target[k] = index[t]
            leaders.add(index[t])
            continue
        if op == "s_endpgm" or op == "s_branch" or op.startswith("s_cbranch"):
            if k + 1 < len(ins):
                leaders.add(k + 1)
            if op != "s_endpgm":
                off = int(ops.split()[0], 0)
                off = off - 65536 if off >= 32768 else off
                t = index.get(addr + 4 + 4 * off)
                if t is None:
                    raise RuntimeError(f"{name}: branch at {addr:#x} to an unknown address")
                target[k] = t
                leaders.add(t)
    starts = sorted(leaders)
    block_of = {}
    blocks = []
    for bi, st in enumerate(starts):
        en = starts[bi + 1] if bi + 1 < len(starts) else len(ins)
        blocks.append((st, en))
        block_of[st] = bi
    succ = []
    for (st, en) in blocks:
        last = ins[en - 1]
        sx = []
        if last[1] == "s_endpgm":
            pass
        elif last[1] == "s_branch" or last[1] == "s_setpc_b64":
            sx.append(block_of[target[en - 1]])
        elif last[1].startswith("s_cbranch"):
            # `s_mov_b64 s[x:y], 0 / -1; s_and_b64 vcc, exec, s[x:y]; s_cbranch_vccnz / vccz` inside one block: a branch on a constant (how
            # the compiler writes "fall into the long branch that follows"): only the edge that can be taken
            const = None
            if en - 3 >= st and last[1] in ("s_cbranch_vccnz", "s_cbranch_vccz"):
                mv, an = ins[en - 3], ins[en - 2]
                mm = re.fullmatch(r"(s\[\d+:\d+\]), (0|-1)", mv[2].strip())
                if mv[1] == "s_mov_b64" and mm and an[1] == "s_and_b64" and an[2].replace(" ", "") == f"vcc,exec,{mm.group(1)}":
                    const = mm.group(2) == "-1"                 # vcc != 0 (for the lanes that run)
            taken = None if const is None else (const if last[1] == "s_cbranch_vccnz" else not const)
            if taken is not False:
                sx.append(block_of[target[en - 1]])
            if en < len(ins) and taken is not True:
                sx.append(block_of[en])
        elif en < len(ins):
            sx.append(block_of[en])
        succ.append(sx)
    state_in = [None] * len(blocks)
    state_in[0] = ()
    work = [0]
    while work:
        b = work.pop()
        st = state_in[b]
        for k in range(blocks[b][0], blocks[b][1]):
            st = step(st, ins[k])
        for sb in succ[b]:
            m = merge(state_in[sb], st)
            if m != state_in[sb]:
                state_in[sb] = m
                work.append(sb)
    findings = []
    for b, (s0, e0) in enumerate(blocks):
        st = state_in[b]
        if st is None:
            continue
        for k in range(s0, e0):
            st = step(st, ins[k], findings)
    findings += store_data_hazards(ins, index)
    return findings


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "bitblas_amd", "libwqaa_hip.so"))
    ap.add_argument("--match", default=DEFAULT_MATCH)
    ap.add_argument("-v", action="store_true")
    ap.add_argument("--lds-dma", action="store_true",
                    help="also: no ds_* instruction while a global_load_lds / buffer_load ... lds is outstanding.  A DIAGNOSTIC, not a "
                         "gate: it has no addresses, and on the one-launch decode member every finding it has is a path of the graph "
                         "that cannot run (the compiler merges the fragment-count arms behind flag registers: from the tile's DMA "
                         "straight to the meeting); the register rule has no such findings")
    args = ap.parse_args()
    global CHECK_LDS
    CHECK_LDS = args.lds_dma
    rx = re.compile(args.match)
    total, bad = 0, 0
    with tempfile.TemporaryDirectory() as td:
        for k, co in enumerate(code_objects(args.lib)):
            if not rx.search(co.decode("latin1")):
                continue
            f = os.path.join(td, f"co{k}.elf")
            open(f, "wb").write(co)
            dis = subprocess.run([OBJDUMP, "-d", f], capture_output=True, text=True).stdout.split("\n")
            cur, body = None, []
            kernels = []
            for line in dis:
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    if cur:
                        kernels.append((cur, body))
                    cur, body = m.group(1), []
                elif cur:
                    body.append(line)
            if cur:
                kernels.append((cur, body))
            for name, body in kernels:
                if not rx.search(name):
                    continue
                total += 1
                fnd = scan(name, body, args.v)
                loads = sum(1 for b in body if re.match(r"\s*global_load_dwordx4", b))
                print(f"{'HAZARD' if fnd else 'ok    '} {name[:110]}  ({len(body)} instructions, {loads} 16-byte loads, {len(fnd)} findings)")
                if fnd:
                    bad += 1
                    for addr, code, hit, age in fnd[:8 if not args.v else 1000]:
                        if age < 0:
                            print(f"      {addr:#x}: {code}   overwrites store data v{hit} within {STORE_DATA_WAIT_STATES} wait states of the store")
                        else:
                            print(f"      {addr:#x}: {code}   touches v{hit}: possibly in flight, {age} younger operation(s)")
    print(f"{total} kernels checked, {bad} with findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
