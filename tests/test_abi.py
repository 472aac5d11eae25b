"""The C ABI: the library loads without a GPU and exports every symbol include/wqaa.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import set_knobs

from bitblas_amd import lib as wlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "wqaa.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:void|int|int64_t|uint64_t|char\s*\*|const char\s*\*)\s*\*?\s*(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_and_binding_agree():
    names = declared_functions()
    assert "init" in names and "wqaa_matmul" in names
    assert sorted(wlib.EXPORTED_SYMBOLS) == names


def test_library_exports_every_declared_symbol():
    cdll = ctypes.CDLL(wlib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(cdll, name), f"{name} missing from libwqaa_hip.so"


def test_struct_sizes_match_header():
    assert ctypes.sizeof(wlib.MatmulDesc) == 16 * 4
    assert ctypes.sizeof(wlib.Plan) == 11 * 4 + 96


def test_peer_exchange_descriptor_and_argument_checks():
    """struct wqaa_peer_exchange_desc as the binding lays it out; a malformed exchange is refused before any launch"""
    assert ctypes.sizeof(wlib.PeerExchangeDesc) == 8 + 8 + 4 * 4 + 2 * wlib.PEER_MAX * 8 + 8 + 8
    L = wlib.load_library()
    assert L.wqaa_peer_exchange(None, None) == wlib.ERR_BAD_DESC
    d = wlib.PeerExchangeDesc()
    d.world, d.rank, d.bytes, d.src, d.flags, d.status = 1, 0, 64, 256, 256, 256       # a world of one has nothing to exchange
    assert L.wqaa_peer_exchange(ctypes.byref(d), None) == wlib.ERR_BAD_DESC
    d.world, d.bytes = 2, 24                                                              # not whole 16-byte pieces
    assert L.wqaa_peer_exchange(ctypes.byref(d), None) == wlib.ERR_BAD_DESC
    d.bytes = 64                                                                          # peer 1 has no destination
    assert L.wqaa_peer_exchange(ctypes.byref(d), None) == wlib.ERR_BAD_DESC
    assert b"peer 1" in L.wqaa_last_error_string()
    assert L.wqaa_peer_export(None, None) == wlib.ERR_BAD_DESC and L.wqaa_peer_open(None, None) == wlib.ERR_BAD_DESC


def test_init_is_idempotent_and_error_channel_works():
    L = wlib.load_library()
    L.init()
    L.init()
    assert L.wqaa_abi_version() == 4
    d = wlib.make_desc(N=64, K=64, a_dtype=wlib.F16, w_format=wlib.W_UINT, w_bits=4, out_dtype=wlib.F16)
    d.struct_size = 4  # wrong ABI size must be refused
    assert L.wqaa_matmul(ctypes.byref(d), 1, 1, None, None, None, None, 1, 1, None) == wlib.ERR_BAD_DESC
    assert L.wqaa_last_error() == wlib.ERR_BAD_DESC
    assert b"ABI" in L.wqaa_last_error_string()


def test_m_zero_returns_ok_without_touching_anything():
    L = wlib.load_library()
    d = wlib.make_desc(N=64, K=64, a_dtype=wlib.F16, w_format=wlib.W_UINT, w_bits=4, out_dtype=wlib.F16)
    assert L.wqaa_matmul(ctypes.byref(d), None, None, None, None, None, None, None, 0, None) == wlib.OK


def test_selector_answers_without_gpu():
    p1 = wlib.select(wlib.make_desc(N=4096, K=4096, a_dtype=wlib.F16, w_format=wlib.W_INT, w_bits=4,
                                    out_dtype=wlib.F16, group_size=128, with_scaling=True,
                                    w_layout=wlib.LAYOUT_LOP3), 1)
    assert p1["kernel_family"] == 1 and p1["grid"] >= 256 and "gemv" in p1["name"]
    with pytest.raises(wlib.WqaaError) as e:
        wlib.select(wlib.make_desc(N=64, K=40, a_dtype=wlib.F16, w_format=wlib.W_INT, w_bits=4,
                                   out_dtype=wlib.F16), 1)
    assert e.value.code == wlib.ERR_UNSUPPORTED


def test_unsupported_is_loud_not_silent():
    with pytest.raises(wlib.WqaaError):
        wlib.select(wlib.make_desc(N=64, K=64, a_dtype=wlib.F32, w_format=wlib.W_INT, w_bits=4,
                                   out_dtype=wlib.F16), 1)


def test_c_packer_against_oracle_and_golden(golden):
    import wqaa_oracle as oracle
    rng = np.random.default_rng(0)
    for bits in (1, 2, 4):
        for code, tgt in ((wlib.F16, "float16"), (wlib.I8, "int8")):
            codes = rng.integers(0, 1 << bits, size=(9, 128), dtype=np.int8)
            plain = wlib.pack_weight(codes, bits, wlib.LAYOUT_PLAIN, code)
            assert np.array_equal(plain, oracle.general_compress(codes, bits))
            lop3 = wlib.pack_weight(codes, bits, wlib.LAYOUT_LOP3, code)
            assert np.array_equal(lop3, oracle.interleave_weight(plain, bits, tgt))
            assert np.array_equal(wlib.unpack_weight(lop3, 128, bits, wlib.LAYOUT_LOP3, code), codes)
    # int4 activations: 2-bit weights interleaved for a 4-bit target (lop3_permutate_impl.py:27-35, 131-134)
    codes = rng.integers(0, 4, size=(9, 128), dtype=np.int8)
    plain = wlib.pack_weight(codes, 2, wlib.LAYOUT_PLAIN, wlib.I4)
    lop3 = wlib.pack_weight(codes, 2, wlib.LAYOUT_LOP3, wlib.I4)
    assert np.array_equal(lop3, oracle.interleave_weight(plain, 2, "int4"))
    assert np.array_equal(wlib.unpack_weight(lop3, 128, 2, wlib.LAYOUT_LOP3, wlib.I4), codes)
    for key in golden.files:
        if key.startswith("interleave_"):
            _, tgt, tag = key.split("_", 2)
            bits = int(tag[1])
            code = wlib.F16 if tgt == "float16" else wlib.I8
            assert np.array_equal(wlib.pack_weight(golden["codes_" + tag], bits, wlib.LAYOUT_LOP3, code), golden[key]), key
        if key.startswith("compress_"):
            tag = key[len("compress_"):]
            bits = int(tag[1])
            assert np.array_equal(wlib.pack_weight(golden["codes_" + tag], bits, wlib.LAYOUT_PLAIN, wlib.F16), golden[key])


def test_relayout_and_threaded_packer(monkeypatch):
    """wqaa_relayout_weight (the LOP3Permutate stage on packed bytes, both directions) against the oracle's restatement of
    the reference's interleave, and the row-parallel path of all three packer entries (taken from 2 M fields on; forced
    to 5 threads here, ragged against the row count) against the serial one - bit identical."""
    import wqaa_oracle as oracle
    rng = np.random.default_rng(11)
    for bits in (1, 2, 4):
        for code, tgt in ((wlib.F16, "float16"), (wlib.I8, "int8")) + (((wlib.I4, "int4"),) if bits == 2 else ()):
            codes = rng.integers(0, 1 << bits, size=(13, 256), dtype=np.int8)
            plain = wlib.pack_weight(codes, bits, wlib.LAYOUT_PLAIN, code)
            lop3 = wlib.relayout_weight(plain, bits, wlib.LAYOUT_PLAIN, wlib.LAYOUT_LOP3, code)
            assert np.array_equal(lop3, oracle.interleave_weight(plain, bits, tgt)), (bits, tgt)
            assert np.array_equal(lop3, wlib.pack_weight(codes, bits, wlib.LAYOUT_LOP3, code))
            assert np.array_equal(wlib.relayout_weight(lop3, bits, wlib.LAYOUT_LOP3, wlib.LAYOUT_PLAIN, code), plain)
    assert np.array_equal(wlib.relayout_weight(plain, 8, wlib.LAYOUT_PLAIN, wlib.LAYOUT_LOP3, wlib.F16), plain)   # 8 bit: nothing moves
    with pytest.raises(wlib.WqaaError):                     # a row that is not a whole number of 32-bit words
        wlib.relayout_weight(np.zeros((4, 6), dtype=np.int8), 4, wlib.LAYOUT_PLAIN, wlib.LAYOUT_LOP3, wlib.F16)
    rows, cols = 1027, 2048                                 # 2.1 M fields: the threaded path
    for bits in (4, 2):
        codes = rng.integers(0, 1 << bits, size=(rows, cols), dtype=np.int8)
        monkeypatch.setenv("WQAA_PACK_THREADS", "1")
        serial = [wlib.pack_weight(codes, bits, lay, wlib.F16) for lay in (wlib.LAYOUT_PLAIN, wlib.LAYOUT_LOP3)]
        monkeypatch.setenv("WQAA_PACK_THREADS", "5")
        threaded = [wlib.pack_weight(codes, bits, lay, wlib.F16) for lay in (wlib.LAYOUT_PLAIN, wlib.LAYOUT_LOP3)]
        assert np.array_equal(serial[0], threaded[0]) and np.array_equal(serial[1], threaded[1])
        assert np.array_equal(serial[0], oracle.general_compress(codes, bits))
        assert np.array_equal(wlib.unpack_weight(threaded[1], cols, bits, wlib.LAYOUT_LOP3, wlib.F16), codes)
        assert np.array_equal(wlib.relayout_weight(threaded[0], bits, wlib.LAYOUT_PLAIN, wlib.LAYOUT_LOP3, wlib.F16), serial[1])
    monkeypatch.delenv("WQAA_PACK_THREADS")


def test_packer_fuzz_ragged_rows_and_tails():
    """random (rows, cols) incl. rows that are not a whole number of 32-bit words (the byte-at-a-time tail), every bit width,
    both layouts and interleave targets, against the oracle's restatement of general_compress / interleave_weight"""
    import wqaa_oracle as oracle
    rng = np.random.default_rng(0)
    for bits in (1, 2, 4):
        epb = 8 // bits
        for _ in range(60):
            rows, cols = int(rng.integers(1, 9)), int(rng.integers(1, 40)) * epb
            codes = rng.integers(0, 1 << bits, size=(rows, cols), dtype=np.int8)
            plain = wlib.pack_weight(codes, bits, wlib.LAYOUT_PLAIN, wlib.F16)
            assert np.array_equal(plain, oracle.general_compress(codes, bits)), (bits, rows, cols)
            assert np.array_equal(wlib.unpack_weight(plain, cols, bits, wlib.LAYOUT_PLAIN, wlib.F16), codes)
            if (cols * bits // 8) % 4 == 0:
                for code, tgt in ((wlib.F16, "float16"), (wlib.I8, "int8")):
                    lop3 = wlib.pack_weight(codes, bits, wlib.LAYOUT_LOP3, code)
                    assert np.array_equal(lop3, oracle.interleave_weight(plain, bits, tgt)), (bits, rows, cols, tgt)
                    assert np.array_equal(wlib.unpack_weight(lop3, cols, bits, wlib.LAYOUT_LOP3, code), codes)
                    assert np.array_equal(wlib.relayout_weight(plain, bits, wlib.LAYOUT_PLAIN, wlib.LAYOUT_LOP3, code), lop3)
            else:
                with pytest.raises(wlib.WqaaError):
                    wlib.pack_weight(codes, bits, wlib.LAYOUT_LOP3, wlib.F16)
    c8 = rng.integers(-128, 128, size=(3, 16), dtype=np.int8)
    assert np.array_equal(wlib.pack_weight(c8, 8, wlib.LAYOUT_PLAIN, wlib.F16), c8)
    assert wlib.pack_weight(np.zeros((0, 64), dtype=np.int8), 4, wlib.LAYOUT_PLAIN, wlib.F16).shape == (0, 32)


def test_selector_invariants_over_a_random_configuration_sweep():
    """Host logic only (wqaa_select needs no GPU): whatever the selector answers must be launchable - workgroup
    size a multiple of 64 and <= 1024, LDS within the 160 KB of a CU, non-empty grid, a name in the reference's
    kernel-name style (general_matmul/__init__.py:240-318) - and a refusal must carry a message."""
    import re
    rng = np.random.default_rng(7)
    name_re = re.compile(r"^matmul_m\d+n\d+k\d+_[a-z0-9]+x[a-z0-9_]+_(gemv_b\d+r\d+d\d+(k\d+)?(_areg)?|gemvx_b\d+r\d+d\d+k\d+(_areg)?|(dq_)?tcx\d+x\d+x\d+(xr)?(pp(t\d+)?|xs|xdl[ptkw]?|xd|xw|xmk)?)$")
    pairs = [(wlib.F16, wlib.W_UINT, b) for b in (1, 2, 4, 8)] + [(wlib.F16, wlib.W_INT, b) for b in (1, 2, 4, 8)] + \
            [(wlib.F16, wlib.W_NF, 4), (wlib.F16, wlib.W_FP4, 4), (wlib.F16, wlib.W_E4M3, 8),
             (wlib.BF16, wlib.W_UINT, 4), (wlib.BF16, wlib.W_NF, 4),
             (wlib.I8, wlib.W_INT, 2), (wlib.I8, wlib.W_INT, 4), (wlib.I8, wlib.W_NATIVE, 8),
             (wlib.I4, wlib.W_NATIVE, 4), (wlib.I4, wlib.W_INT, 2),
             (wlib.E4M3, wlib.W_E4M3, 8), (wlib.E5M2, wlib.W_E5M2, 8), (wlib.F16, wlib.W_NATIVE, 16)]
    ok = refused = 0
    for _ in range(3000):
        a, wf, bits = pairs[rng.integers(len(pairs))]
        N = int(rng.choice([16, 48, 200, 256, 1000, 1024, 4096, 5120, 11008, 28672]))
        K = int(rng.choice([256, 512, 768, 1024, 2048, 4096, 8192, 11008, 28672]))
        M = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 33, 64, 65, 100, 128, 256, 777, 1024, 4096]))
        fp_act = a in (wlib.F16, wlib.BF16)
        quant = wf in (wlib.W_UINT, wlib.W_INT, wlib.W_NF, wlib.W_FP4, wlib.W_E4M3) and fp_act
        g = int(rng.choice([-1, 32, 64, 128, 256])) if quant else -1
        scaling = bool(rng.integers(2)) if quant else False
        zmode = int(rng.choice([wlib.Z_NONE, wlib.Z_ORIGINAL, wlib.Z_RESCALE, wlib.Z_QUANTIZED])) \
            if (scaling and wf in (wlib.W_UINT, wlib.W_INT)) else wlib.Z_NONE
        out = {wlib.I8: wlib.I32, wlib.I4: wlib.I32, wlib.BF16: wlib.F32, wlib.E4M3: wlib.F32, wlib.E5M2: wlib.F32}.get(a, wlib.F16)
        layout = int(rng.integers(2)) if (wf in (wlib.W_UINT, wlib.W_INT) and bits < 8 and a != wlib.BF16) else wlib.LAYOUT_PLAIN
        desc = wlib.make_desc(N=N, K=K, a_dtype=a, w_format=wf, w_bits=bits, out_dtype=out, group_size=g,
                              with_scaling=scaling, zeros_mode=zmode, with_bias=bool(rng.integers(2)), w_layout=layout)
        try:
            p = wlib.select(desc, M)
        except wlib.WqaaError as e:
            assert e.code in (wlib.ERR_UNSUPPORTED, wlib.ERR_BAD_DESC) and str(e), (N, K, M, a, wf, bits)
            refused += 1
            continue
        ok += 1
        ctx = (N, K, M, a, wf, bits, g, scaling, zmode, layout, p)
        assert p["kernel_family"] in (1, 2, 4), ctx           # (4: B_decode + this library's dense member, round 4's automatic two-pass form)
        assert p["threads"] % 64 == 0 and 64 <= p["threads"] <= 1024, ctx
        assert 0 <= p["lds_bytes"] <= 160 * 1024, ctx
        assert p["grid"] >= 1, ctx
        assert name_re.match(p["name"]), ctx
        assert f"m{M}n{N}k{K}_" in p["name"], ctx
        if M <= 2 and p["kernel_family"] == 2:
            # only when the GEMV family has no member for the configuration (groups below its lane chunk)
            assert g != -1 and g < 128, ctx
    assert ok > 1500 and refused > 0


def test_c_packer_reproduces_the_reference_int4_tests_operands():
    """tests/golden/int4_golden.npz holds the weight operand exactly as the reference's int4 test packs it by hand
    (test_general_matmul_ops_int4.py:49-57, produced by running that test): the C packer must emit the same bytes
    from the unpacked fields."""
    import os
    import wqaa_oracle as oracle
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "int4_golden.npz"))
    for i in range(2):
        bits = {"int4": 4, "int2": 2}[str(g[f"c{i}_W_dtype"])]
        packed = g[f"c{i}_B"]
        codes = oracle.general_decompress(packed, bits).astype(np.int8)
        assert np.array_equal(wlib.pack_weight(codes, bits, wlib.LAYOUT_PLAIN, wlib.I4), packed)
        assert np.array_equal(wlib.unpack_weight(packed, codes.shape[1], bits, wlib.LAYOUT_PLAIN, wlib.I4), codes)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/wqaa.h is the drop-in boundary: it must compile as C (no C++, no HIP, no torch types) and a C program that
    sees nothing but the header must be able to plan an operator and a group through the shared library."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    src = tmp_path / "use_wqaa.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "wqaa.h"
int main(void) {
  wqaa_matmul_desc d, k;
  wqaa_plan p;
  const wqaa_matmul_desc* grp[3];
  int launches = 0;
  memset(&d, 0, sizeof d);
  d.struct_size = (int32_t)sizeof d; d.N = 4096; d.K = 4096; d.a_dtype = WQAA_F16; d.w_format = WQAA_W_INT; d.w_bits = 4;
  d.out_dtype = WQAA_F16; d.group_size = 128; d.with_scaling = 1; d.w_layout = WQAA_LAYOUT_LOP3;
  init();
  if (wqaa_abi_version() != WQAA_ABI_VERSION) return 2;
  if (wqaa_select(&d, 1, &p) != WQAA_OK || p.kernel_family != 1) return 3;
  printf("%s\n", p.name);
  k = d; k.N = 1024;
  grp[0] = &d; grp[1] = &k; grp[2] = &k;
  if (wqaa_group_plan(grp, 3, 1, &launches, &p) != WQAA_OK || launches != 1) return 4;
  printf("%s\n", p.name);
  if (wqaa_matmul(&d, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) != WQAA_OK) return 5;   /* m == 0: returns at once */
  if (wqaa_matmul_group(NULL, 0, 1, NULL) != WQAA_OK) return 6;
  d.struct_size = 4;
  if (wqaa_select(&d, 1, &p) != WQAA_ERR_BAD_DESC || wqaa_last_error() != WQAA_ERR_BAD_DESC) return 7;
  return 0;
}
''')
    exe = tmp_path / "use_wqaa"
    libdir = os.path.dirname(wlib.LIB_PATH)
    subprocess.run([cc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-lwqaa_hip", f"-Wl,-rpath,{libdir}"], check=True, capture_output=True, text=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, (res.returncode, res.stdout, res.stderr)
    lines = res.stdout.split()
    assert "gemv" in lines[0] and lines[1].endswith("_x3")


def test_tune_and_dequantize_entries_without_a_device():
    """wqaa_tune is a no-op where there is nothing to measure on; wqaa_dequantize checks its operands before it needs a device"""
    L = wlib.load_library()
    d = wlib.make_desc(N=4096, K=4096, a_dtype=wlib.F16, w_format=wlib.W_UINT, w_bits=4, out_dtype=wlib.F16, group_size=128,
                       with_scaling=True, zeros_mode=wlib.Z_ORIGINAL)
    if L.wqaa_device_count() == 0:
        assert L.wqaa_tune(ctypes.byref(d), 4096, None) == wlib.OK
    assert L.wqaa_dequantize(ctypes.byref(d), None, None, None, None, None, None) == wlib.ERR_BAD_DESC
    # the tuned threshold is part of the descriptor: planning honours it where a two-pass member exists - since round 4 that is
    # B_decode + this library's own dense member (no device, no vendor library needed); below the threshold: the fused member
    d.two_pass_min_m = 1024
    p = wlib.select(d, 4096)
    assert p["kernel_family"] == 4 and "_dq_" in p["name"], p
    assert wlib.select(d, 512)["kernel_family"] == 2
    d.two_pass_min_m = 0
    for m in (4096, 2048, 1024):                               # a format with a fused ping-pong member never takes it unasked - also where
        assert wlib.select(d, m)["kernel_family"] == 2, m      # the round estimate prefers the lockstep member (1024 x 4096^2)
    d8 = wlib.make_desc(N=4096, K=4096, a_dtype=wlib.F16, w_format=wlib.W_INT, w_bits=8, out_dtype=wlib.F16)
    assert wlib.select(d8, 4096)["kernel_family"] == 4         # float16 x int8 has none: B_decode + the dense member
    assert wlib.select(d8, 512)["kernel_family"] == 2


def test_vendor_library_is_not_a_link_dependency():
    """hipBLASLt is an opt-in yardstick: libwqaa_hip.so must load on a box without it (dlopen on first use, never DT_NEEDED)"""
    import subprocess
    out = subprocess.run(["readelf", "-d", wlib.LIB_PATH], capture_output=True, text=True).stdout
    needed = [ln for ln in out.splitlines() if "NEEDED" in ln]
    assert needed, "readelf gave no dynamic section"
    assert not any("hipblaslt" in ln.lower() for ln in needed), needed


def _gfx950_code_objects(path):
    """the device ELFs inside the library's .hip_fatbin section (one - compressed - clang offload bundle per translation unit;
    tools/check_vmem_hazards.py knows both the plain and the --offload-compress container)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_vmem_hazards", os.path.join(ROOT, "tools", "check_vmem_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.code_objects(path)


def test_counted_decode_members_keep_their_loads_in_registers(tmp_path):
    """The persistent decode member of the 4-bit Scale / Zeros formats issues its weight loads as inline assembly and counts the
    waits by hand (csrc/wqaa_gemm_kernel.h, `issue` / `landed`): a register spill or an out-of-line call between a load and its
    wait would copy registers the data has not reached yet - it did once (zeros-rescale, every output NaN).  The compiler's own
    metadata of the BUILT library says whether that can happen: no scratch, no dynamic stack in any such instantiation."""
    import re
    import shutil
    import subprocess
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(readelf) and shutil.which("objcopy")):
        pytest.skip("no llvm-readelf / objcopy on this box")
    seen = {}
    for k, co in enumerate(_gfx950_code_objects(wlib.LIB_PATH)):
        f = tmp_path / f"co{k}.elf"
        f.write_bytes(co)
        notes = subprocess.run([readelf, "--notes", str(f)], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes):
            m = re.search(r"\.name:\s+(_ZN4wqaa25wq_gemm_decode_lds_kernelINS_10GemmPolicyILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E\S*)", blk)
            if not m:
                continue
            kind, _layout, at, mode = (int(m.group(j)) for j in (2, 3, 4, 5))
            if kind in (0, 4) and at == 0 and mode in (1, 2, 3):                 # DK_INT4 / DK_LUT4, 16-bit float activations, MD_S / ZO / ZR
                scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
                dyn = re.search(r"\.uses_dynamic_stack:\s+(\w+)", blk)
                seen[m.group(1)] = (scratch, dyn.group(1) if dyn else "false")
    assert len(seen) >= 8, sorted(seen)                                          # {int4, lut4} x layouts x modes x {f16, bf16}, as instantiated
    bad = {k: v for k, v in seen.items() if v != (0, "false")}
    assert not bad, bad


def test_mid_m_members_keep_their_loads_in_registers(tmp_path):
    """the same for the mid-M member (csrc/wqaa_gemm_mid_kernel.h, round 5): every register load an inline-assembly instruction,
    one wait - no scratch, no dynamic stack in any of its instantiations"""
    import re
    import shutil
    import subprocess
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(readelf) and shutil.which("objcopy")):
        pytest.skip("no llvm-readelf / objcopy on this box")
    seen = {}
    for k, co in enumerate(_gfx950_code_objects(wlib.LIB_PATH)):
        f = tmp_path / f"co{k}.elf"
        f.write_bytes(co)
        notes = subprocess.run([readelf, "--notes", str(f)], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", notes):
            m = re.search(r"\.name:\s+(_ZN4wqaa18wq_gemm_mid_kernelINS_9MidPolicy\S*)", blk)
            if not m:
                continue
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
            dyn = re.search(r"\.uses_dynamic_stack:\s+(\w+)", blk)
            seen[m.group(1)] = (scratch, dyn.group(1) if dyn else "false")
    assert len(seen) >= 50, len(seen)                                            # 2 layouts x 5 modes x 5 (rows, k-steps) shapes
    bad = {k: v for k, v in seen.items() if v != (0, "false")}
    assert not bad, bad


def test_decode_batch_forms_are_chosen_where_they_were_measured(monkeypatch):
    """one-launch decode member, rounds 4 and 5 (no device needed): persistent on wide outputs at K <= 4096 (`xdlp`, up to six rounds of
    fragments for the hand-counted formats), whole tile on long K where M-sized slots fit (`xdlt`: M <= 8 at K <= 8192, M <= 4 at
    K <= 12288, more than one fragment per workgroup), the block-by-block form otherwise"""
    monkeypatch.delenv("WQAA_GEMM_TUNE", raising=False)

    def name(m, N, K):
        d = wlib.make_desc(N=N, K=K, a_dtype=wlib.F16, w_format=wlib.W_UINT, w_bits=4, out_dtype=wlib.F16, group_size=128, with_scaling=True,
                           zeros_mode=wlib.Z_ORIGINAL, w_layout=wlib.LAYOUT_LOP3)
        return wlib.select(d, m)["name"]

    for m, N, K, suffix in ((8, 11008, 4096, "xdlp"), (16, 22016, 4096, "xdlp"), (3, 12288, 4096, "xdlp"), (8, 11008, 3840, "xdlp"),
                            (8, 8192, 8192, "xdlt"), (3, 12288, 8192, "xdlt"), (4, 8192, 11008, "xdlt"), (4, 8176, 12288, "xdlt"),
                            (8, 8192, 8192, "xdlt"), (5, 8192, 11008, "xdl"), (4, 4096, 11008, "xdl"), (16, 4096, 4096, "xdl"), (4, 8192, 28672, "xdl"),
                            # round 5, the K-sliced form (`xdlk`): two rounds of fragments or more, M >= 13 at K >= 8192, M >= 5 at K >= 24576
                            (9, 8192, 8192, "xdl"), (13, 8192, 8192, "xdlk"), (16, 12288, 8192, "xdlk"), (16, 11008, 8192, "xdlk"), (8, 8192, 28672, "xdlk"), (16, 8192, 28672, "xdlk"),
                            (16, 4096, 11008, "xdl"), (8, 12288, 8192, "xdlt"),
                            # round 5, a wave per fragment (`xdlw`): outputs wider than the persistent / whole-tile forms reach, where the tile fits LDS
                            (8, 32000, 4096, "xdlw"), (16, 28672, 4096, "xdlw"), (3, 128256, 4096, "xdlw"), (8, 16384, 8192, "xdlw"), (8, 24576, 4096, "xdlp")):
        assert name(m, N, K).endswith("_f16xu4_tcx16x16x128" + suffix), (m, N, K, name(m, N, K))
    set_knobs(monkeypatch, "gemm", decode_long="0")
    assert not name(8, 8192, 8192).endswith("xdlt") and not name(16, 8192, 28672).endswith("xdlk")
    set_knobs(monkeypatch, "gemm", decode_long="2")          # the round-4 selector: whole tile, never K-sliced
    assert name(8, 8192, 8192).endswith("xdlt") and not name(16, 8192, 28672).endswith("xdlk")
    set_knobs(monkeypatch, "gemm", decode_long="3")          # K-sliced wherever the shape fits (the parity tests)
    assert name(8, 4096, 11008).endswith("xdlk") and name(3, 512, 8192).endswith("xdlk")
    set_knobs(monkeypatch, "gemm", decode_long="4")          # a wave per fragment wherever the tile fits
    assert name(8, 4096, 4096).endswith("xdlw") and name(5, 1000, 8192).endswith("xdlw") and not name(16, 512, 8192).endswith("xdlw")
    set_knobs(monkeypatch, "gemm", decode_long="2")
    assert not name(8, 32000, 4096).endswith("xdlw")
