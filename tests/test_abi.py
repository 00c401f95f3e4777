"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/dir_hip.h declares, the host-side entry points compute, and the product never touches oracle/."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "dir_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dir_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    names = _header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(built_lib, n), "libdir_hip.so does not export %s" % n


def test_binding_table_matches_header(built_lib):
    from dir_amd import _lib
    assert sorted(_lib.SIGNATURES) == _header_functions()
    assert built_lib.dir_version() == 202


def test_host_hash_matches_published_kats(built_lib):
    from tests.test_oracle_kat import FARM_KATS, _s64
    from dir_amd import ops
    from oracle import np_ref as R
    for s, exp in FARM_KATS:
        assert _s64(ops.fingerprint64(s)) == exp
    assert ops.hash_bucket_strings(["Hello", "TensorFlow", "2.x"], 3).tolist() == [0, 2, 2]
    # every length branch (<=16, 17-32, 33-64, >64): C++ product vs the pure-Python restatement
    rng = np.random.default_rng(3)
    for n in list(range(0, 70)) + [64, 65, 127, 128, 129, 200, 1000]:
        s = bytes(rng.integers(0, 256, size=n, dtype=np.uint8).tolist())
        assert ops.fingerprint64(s) == R.fingerprint64(s), n


def test_host_shard_owner(built_lib):
    from oracle import np_ref as R
    o, l = ctypes.c_int(), ctypes.c_int64()
    for V, P in [(10, 4), (1000000, 8), (7, 8), (100000000, 8)]:
        ids = np.unique(np.concatenate([np.arange(min(V, 40)), np.random.default_rng(5).integers(0, V, 100), [V - 1]]))
        ro, rl = R.shard_div_owner(ids, V, P)
        for i, ri, rli in zip(ids, ro, rl):
            built_lib.dir_shard_div_owner(int(i), V, P, ctypes.byref(o), ctypes.byref(l))
            assert (o.value, l.value) == (ri, rli)


def test_argument_errors_without_gpu(built_lib):
    # argument validation happens before any HIP call, so it is testable here
    rc = built_lib.dir_embedding_bag_f32(None, 1, 1, None, None, None, 0, 0, 0, 0, 1, None, 1, None)
    assert rc == -1 and b"null pointer" in built_lib.dir_last_error()
    rc = built_lib.dir_cin_layer_f32(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 26, 26, 128, 10, 4,
                                     ctypes.c_void_p(16), None, 0, None)
    assert rc == -4 and b"D=10" in built_lib.dir_last_error()


def test_ops_refuse_cpu_tensors(built_lib):
    import torch
    from dir_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fm_logit(torch.zeros(4, 8), 2, 4)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "details-in-recommendation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "libdir_oracle" not in txt, f


def test_no_packed_fp32_beside_wide_mfma_in_the_built_objects(built_lib):
    """dir_amd/isa_check.py over every object of the in-tree build: no kernel that issues 16x16x32-class MFMAs may hold a packed fp32 VALU
    instruction whose op_sel sends a source's high register to the low result (the pair loses the low result in lanes 48..63 on gfx950:
    tools/pk_mfma_probe.hip), and no other kernel holds the form either (a bf16-MFMA kernel of another stream could corrupt it)."""
    import glob
    import os
    import dir_amd
    from dir_amd import isa_check
    objs = sorted(glob.glob(os.path.join(os.path.dirname(dir_amd.library_path()), "csrc", "_build", "*.o")))
    assert len(objs) >= 18
    errors, exposed = isa_check.check(objs)
    assert errors == [], errors
    assert exposed == [], exposed
    # the scanner itself: the failing pair of the round-2 DIN build is recognised, harmless forms are not
    text = """0000000000001000 <victim>:
\tv_pk_add_f32 v[138:139], v[138:139], v[252:253] op_sel:[0,1]   // 000000001000: D3B24086 1803F98A
\tv_mfma_f32_16x16x32_bf16 v[146:149], v[208:211], v[114:117], v[146:149] // 000000001008: D3B58092 0A4AE5D0
0000000000002000 <fine>:
\tv_pk_add_f32 v[10:11], v[10:11], v[12:13] op_sel_hi:[1,0]        // 000000002000: D3B2400A 0802190A
\tv_pk_add_f32 v[10:11], v[10:11], v[12:13] neg_lo:[0,1] neg_hi:[0,1] // 000000002008: D3B2400A 0802190A
\tv_mfma_f32_16x16x32_bf16 v[146:149], v[208:211], v[114:117], v[146:149] // 000000002010: D3B58092 0A4AE5D0
"""
    per = isa_check.scan(text)
    assert per["victim"][0] == 1 and len(per["victim"][1]) == 1
    assert per["fine"][0] == 1 and per["fine"][1] == []
