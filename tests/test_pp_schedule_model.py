"""A model of the ping-pong member's LDS-DMA schedule (csrc/wqaa_gemm_pp_kernel.h, `wq_gemm_pp_kernel`): does every counted wait cover what is
read after it?

The kernel keeps several k-tiles of activations and a chunk of packed weights in flight by LDS-DMA and never drains the queue inside its loop:
`s_waitcnt vmcnt(N)` returns when at most N of the wave's vector-memory operations are outstanding, and they retire in issue order - so a wait
is a statement about ISSUE ORDER: "everything older than the N youngest has landed".  The counts are written for the loop's steady state.  Round 6
found (by repeating first launches on fresh operands, profiles/r06_repetition.txt) that the 128-row tile's `vmcnt(10)` did not hold behind its
prologue: there only six pieces follow k-tiles 1 and 2, the wait returned at once and a wave could read tiles that had not landed - about once in
2000 first launches.  This file replays the issue order of both tile heights (prologue + loop, as the header issues them) against the waits
AS WRITTEN IN THE HEADER (the template arguments are read from its text) and checks every read; with the prologue wait of before the fix the model
reports exactly the tiles the hardware showed.  A CPU test: the order below is a restatement - keep it next to the kernel when either changes."""
import os
import re

import pytest

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bitblas_amd", "csrc", "wqaa_gemm_pp_kernel.h")


def _waits_from_header():
    """the three counted waits of wq_gemm_pp_kernel as Python callables of (HALF, D, NPH, tq)"""
    src = open(HEADER).read()
    start = src.index(" wq_gemm_pp_kernel(const GemmArgs a) {")
    body = src[start:src.index("// ---- epilogue:", start)]
    exprs = re.findall(r"pp_wait_vmcnt<([^;]+)>\(\);", body)
    def py(e):
        """a C integer expression with ?: && || -> Python"""
        e = e.strip()
        depth, q = 0, -1
        for i, ch in enumerate(e):                      # the first ?: at parenthesis depth 0 (lowest precedence, right associative)
            depth += ch == "("
            depth -= ch == ")"
            if ch == "?" and depth == 0:
                q = i
                break
        if q >= 0:
            depth = nest = 0
            for j in range(q + 1, len(e)):
                ch = e[j]
                depth += ch == "("
                depth -= ch == ")"
                if depth == 0 and ch == "?":
                    nest += 1
                if depth == 0 and ch == ":":
                    if nest == 0:
                        return f"(({py(e[q + 1:j])}) if ({py(e[:q])}) else ({py(e[j + 1:])}))"
                    nest -= 1
            raise ValueError(e)
        out, i = "", 0
        while i < len(e):                               # no ternary at this level: translate the parenthesised parts
            if e[i] == "(":
                depth, j = 1, i + 1
                while depth:
                    depth += e[j] == "("
                    depth -= e[j] == ")"
                    j += 1
                out += "(" + py(e[i + 1:j - 1]) + ")"
                i = j
            else:
                out += e[i]
                i += 1
        return out.replace("&&", " and ").replace("||", " or ")

    prologue = next(e for e in exprs if "NPH" in e)
    loop256 = next(e for e in exprs if "tq == 2" in e)
    loop128 = next(e for e in exprs if e.strip() == "10")
    chunk = next(e for e in exprs if e.strip() == "2")
    return {k: (lambda HALF, D, NPH, tq, _e=py(v): int(eval(_e, {}, dict(HALF=HALF, D=D, NPH=NPH, tq=tq))))
            for k, v in dict(prologue=prologue, loop256=loop256, loop128=loop128, chunk=chunk).items()}


class Wave:
    """one wave's vector-memory queue: operations retire in issue order"""

    def __init__(self):
        self.issued = []          # tags, in issue order
        self.done = 0             # how many of them are known to have completed
        self.violations = []

    def dma(self, tag):
        self.issued.append(tag)

    def wait(self, n):
        self.done = max(self.done, len(self.issued) - n)

    def read(self, tag, where):
        pending = [i for i, t in enumerate(self.issued) if t == tag and i >= self.done]
        if tag not in self.issued:
            self.violations.append(f"{where}: {tag} was never asked for")
        elif pending:
            self.violations.append(f"{where}: {tag} read with {len(pending)} of its pieces possibly in flight")


def run_model(half, ntiles, has_meta, waits, prologue_wait=None):
    D, NPH = (4, 2) if half else (2, 4)
    w = Wave()
    wtiles = 16
    # ---- prologue (the header: dma_meta(0); WBUFS chunks; D k-tiles; the counted wait; barrier) ----
    if has_meta:
        w.dma(("meta", 0))
    for c in range(2 if half else 1):
        for p in range(4):
            w.dma(("W", c))
    for tt in range(D):
        for j in range(NPH):
            w.dma(("A", tt))
    w.wait(waits["prologue"](half, D, NPH, 0) if prologue_wait is None else prologue_wait)
    w.read(("W", 0), "prologue decode")
    # ---- main loop ----
    for t in range(ntiles):
        tq = t & 3
        if tq == 0 and has_meta and (t & (wtiles - 1)) == 0:
            w.dma(("meta", (t >> 4) + 1))
        for p in range(NPH):
            w.read(("A", t), f"tile {t} load segment {p}")
            w.dma(("A", t + D))                                     # (clamped beyond the last tile: still an operation of the queue)
            if not half and tq == 2:
                w.dma(("W", (t >> 2) + 1))
            if half and tq == 2 and p == 1:
                for _ in range(4):
                    w.dma(("W", (t >> 2) + 2))
            if not half and p == 1:
                if tq == 3:
                    w.wait(waits["chunk"](half, D, NPH, tq))
                    w.read(("W", (t >> 2) + 1), f"tile {t}: next chunk's words")
                if tq == 1:
                    w.read(("W", t >> 2), f"tile {t}: second half of the chunk")
            if not half and p == 2:
                w.wait(waits["loop256"](half, D, NPH, tq))
            if half and p == 0:
                if tq == 3:
                    w.read(("W", (t >> 2) + 1), f"tile {t}: next chunk's words")
                if tq == 1:
                    w.read(("W", t >> 2), f"tile {t}: second half of the chunk")
            if half and p == 1:
                w.wait(waits["loop128"](half, D, NPH, tq))
    return w.violations


@pytest.mark.parametrize("has_meta", [False, True])
@pytest.mark.parametrize("ntiles", [4, 8, 16, 64])
@pytest.mark.parametrize("half", [False, True], ids=["256-row", "128-row"])
def test_every_read_is_covered_by_the_waits_in_the_header(half, ntiles, has_meta):
    assert run_model(half, ntiles, has_meta, _waits_from_header()) == []


def test_the_model_sees_the_race_of_round_6():
    """the 128-row tile with the prologue wait it had - (D - 1) * NPH = 6, tiles 1 .. 3 in flight: tiles 1 and 2 are read uncovered"""
    bad = run_model(True, 16, False, _waits_from_header(), prologue_wait=6)
    assert bad and all(("('A', 1)" in v or "('A', 2)" in v) for v in bad), bad
    assert any("('A', 1)" in v for v in bad) and any("('A', 2)" in v for v in bad)
    # ... and the 256-row tile never had it: its prologue ends with the activation tiles, the loop's count is exact from tile 0 on
    assert run_model(False, 16, True, _waits_from_header(), prologue_wait=4) == []


# ---- the dense members (wq_gemm_pp8_kernel both tile heights, pp8w, pp8s): their prologues keep "the order the loop keeps" -------------
def _dense_model(kind, ntiles):
    w = Wave()

    def tile_ops(tt, order):                       # order: a string of 'A' / 'W' pieces of k-tile tt
        for ch in order:
            w.dma((ch, tt))

    if kind == "pp8_128":                          # [A p0, A p1, W x 4] per tile, three tiles ahead; vmcnt(12) = the two younger tiles
        for tt in range(3):
            tile_ops(tt, "AAWWWW")
        w.wait(12)
        for t in range(ntiles):
            w.read(("W", t), f"tile {t}")
            w.read(("A", t), f"tile {t} segment 0")
            w.dma(("A", t + 3))
            w.read(("A", t), f"tile {t} segment 1")
            w.dma(("A", t + 3))
            tile_ops(t + 3, "WWWW")
            w.wait(12)
    elif kind == "pp8_256":                        # W(t + 1) piece 3 with segment 0, W(t + 2) pieces 0..2 with segments 1..3; vmcnt(5) in segment 2
        tile_ops(0, "WWWW")
        tile_ops(0, "AAAA")
        tile_ops(1, "WWW")
        tile_ops(1, "AAAA")
        w.wait(7)
        for t in range(ntiles):
            for p in range(4):
                if p == 0:
                    w.read(("W", t), f"tile {t}")
                w.read(("A", t), f"tile {t} segment {p}")
                w.dma(("W", t + 1) if p == 0 else ("W", t + 2))
                w.dma(("A", t + 2))
                if p == 2:
                    w.wait(5)
    elif kind == "pp8w":                           # [A p0, p1] with segment 0, [A p2, p3, W x 4] with segment 1; vmcnt(8) = tile t + 2's pieces
        for tt in range(2):
            tile_ops(tt, "AAAAWWWW")
        w.wait(8)
        for t in range(ntiles):
            for p in range(2):
                w.read(("A", t), f"tile {t} segment {p}")
                w.read(("W", t), f"tile {t} segment {p}")
                tile_ops(t + 2, "AA" if p == 0 else "AAWWWW")
                if p == 1:
                    w.wait(8)
    elif kind == "pp8s":                           # one segment per tile: [A x 2, W x 4] three tiles ahead; vmcnt(12)
        for tt in range(3):
            tile_ops(tt, "AAWWWW")
        w.wait(12)
        for t in range(ntiles):
            w.read(("W", t), f"tile {t}")
            w.read(("A", t), f"tile {t}")
            tile_ops(t + 3, "AAWWWW")
            w.wait(12)
    return w.violations


@pytest.mark.parametrize("ntiles", [2, 8, 32])
@pytest.mark.parametrize("kind", ["pp8_128", "pp8_256", "pp8w", "pp8s"])
def test_dense_members_keep_the_order_their_waits_count(kind, ntiles):
    assert _dense_model(kind, ntiles) == []


def test_dense_members_waits_are_the_ones_in_the_header():
    """the constants of the model above, where the header has them (a changed count must come back to this file)"""
    src = open(HEADER).read()
    pp8 = src[src.index(" wq_gemm_pp8_kernel(const GemmArgs a) {"):src.index(" wq_gemm_pp8w_kernel(const GemmArgs a) {")]
    pp8w = src[src.index(" wq_gemm_pp8w_kernel(const GemmArgs a) {"):src.index(" wq_gemm_pp8s_kernel(const GemmArgs a) {")]
    pp8s = src[src.index(" wq_gemm_pp8s_kernel(const GemmArgs a) {"):]
    counts = lambda body: [int(x) for x in re.findall(r"pp_wait_vmcnt<(\d+)>\(\);", body)]      # noqa: E731
    assert counts(pp8) == [12, 7, 12, 5, 0], counts(pp8)
    assert counts(pp8w) == [8, 8, 0], counts(pp8w)
    assert counts(pp8s) == [12, 12, 0], counts(pp8s)
