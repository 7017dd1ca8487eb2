"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads without a GPU and exports
every symbol include/mi355pose.h declares; the ctypes table covers the header one to one; the product
path refuses to run without the HIP library / on CPU tensors (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, PKG


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'mi355pose.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(mi355_\w+)\s*\(', txt)))


@pytest.fixture(scope='module')
def lib_path():
    import importlib.util
    spec = importlib.util.spec_from_file_location('mi355_build', os.path.join(PKG, 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(verbose=False)


def test_library_exports_every_header_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), 'missing export ' + s
    lib.mi355_version.restype = ctypes.c_int
    assert lib.mi355_version() >= 100


def test_isa_gate_on_the_shipped_library(lib_path):
    """build.py's gate: the linked library holds no packed-fp32 arithmetic (any VOP3P v_pk_*_f32 opcode) that takes an operand
    half from the other register of its pair (the instruction form behind the round-2 dropped-addend fault, DESIGN.md
    section 7) -- and the gate itself recognises such instructions when it sees them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('mi355_build', os.path.join(PKG, 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.isa_gate(lib_path) > 100000                   # every device instruction of the library was inspected
    bad = ['v_pk_add_f32 v[4:5], v[4:5], v[22:23] op_sel:[0,1] op_sel_hi:[1,0]',      # the round-2 fault
           'v_pk_mul_f32 v[2:3], v[2:3], v[8:9] op_sel_hi:[0,1]', 'v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,0,0]',
           'v_pk_max_f32 v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0]']                          # any VOP3P fp32-pair opcode
    ok = ['v_pk_add_f32 v[4:5], v[4:5], v[22:23]', 'v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel_hi:[1,1,1]',
          'v_pk_mov_b32 v[22:23], v[22:23], v[22:23] op_sel:[1,0]', 'v_pk_add_f16 v1, v2, v3 op_sel:[1,0]']
    assert all(mod.swaps_halves(i) for i in bad) and not any(mod.swaps_halves(i) for i in ok)
    assert '-fno-slp-vectorize' in mod.FLAGS


def test_ctypes_table_matches_header(lib_path):
    import mi355
    assert sorted(mi355.SIGNATURES) == _header_symbols()
    mi355.load()


def test_no_cpu_fallback(lib_path):
    import mi355
    from mi355 import ops
    with pytest.raises(mi355.Mi355Error):
        ops.to_nhwc(torch.zeros(1, 3, 8, 8))
    with pytest.raises(mi355.Mi355Error):
        ops.argmax2d(torch.zeros(1, 21, 8, 8))
    with pytest.raises(mi355.Mi355Error):
        mi355.load.__wrapped__ if hasattr(mi355.load, '__wrapped__') else None
        saved, mi355._lib = mi355._lib, None
        try:
            mi355.load('/nonexistent/libmi355pose.so')
        finally:
            mi355._lib = saved


def test_argument_validation_without_gpu(lib_path):
    """Host-side validation paths return error codes before any launch (safe on a CPU-only box)."""
    import mi355
    lib = mi355.load()
    bad = mi355.ConvDesc(1, 8, 8, 24, 8, 8, 64, 3, 3, 3, 1, mi355.BF16)   # stride 3 unsupported
    assert lib.mi355_conv_fwd(ctypes.byref(bad), 0, 0, 0, 0, 0, 0) == -1
    assert b'stride' in lib.mi355_last_error()
    assert lib.mi355_sgd_nesterov(0, 0, 0, 0, 0, 0.9, 0.0, 1, 0, 0) == -1
    assert lib.mi355_bn_workspace(4096, 256) > 0
