"""The C-ABI library loads on a GPU-less host and exports every symbol that
include/fcn8s_hip.h declares; the GPU-free layout queries agree with the oracle's
variable table.  No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "fcn8s_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fcn8s_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    from fcn8s_tensorflow_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 50
    raw = C.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), "libfcn8s_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "ctypes binding lacks %s" % s
    assert sorted(_lib.SIGNATURES) == syms


def _cfg(C_=20, widths=None):
    from fcn8s_tensorflow_amd import _lib
    cfg = _lib.Config(); cfg.num_classes = C_
    for i in range(7):
        cfg.widths[i] = widths[i] if widths else 0
    return cfg


def test_layout_matches_reference_variable_table():
    from fcn8s_tensorflow_amd import _lib
    from oracle import fcn8s_oracle as orc
    for widths in (None, (8, 16, 32, 64, 64, 128, 128)):
        cfg = _cfg(20, widths)
        specs = orc.param_specs(20, widths or orc.DEFAULT_WIDTHS)
        n = _lib.lib.fcn8s_layout_num_params(C.byref(cfg))
        assert n == len(specs) == 42
        total = _lib.lib.fcn8s_param_floats(C.byref(cfg))
        prev_end = 0
        names = []
        for i in range(n):
            name = C.create_string_buffer(64); nd = C.c_int32(); shp = (C.c_int64 * 4)(); off = C.c_int64()
            assert _lib.lib.fcn8s_layout_param(C.byref(cfg), i, name, C.byref(nd), C.byref(shp), C.byref(off)) == 0
            nm = name.value.decode(); names.append(nm)
            assert tuple(shp[k] for k in range(nd.value)) == specs[nm]
            assert off.value % 64 == 0 and off.value >= prev_end        # 256-byte aligned, non-overlapping
            prev_end = off.value + int(np.prod(specs[nm]))
        assert names == list(specs) and prev_end <= total
        # gradient buckets: contiguous partition of the flat buffer, produced in backward order
        nb = _lib.lib.fcn8s_layout_num_buckets(C.byref(cfg))
        assert 3 <= nb <= _lib.MAX_BUCKETS
        rng = []
        for b in range(nb):
            o = C.c_size_t(); m = C.c_size_t()
            assert _lib.lib.fcn8s_layout_bucket(C.byref(cfg), b, C.byref(o), C.byref(m)) == 0
            rng.append((o.value, m.value))
        assert _lib.lib.fcn8s_layout_bucket(C.byref(cfg), nb, C.byref(o), C.byref(m)) == _lib.ERR_BAD_ARG
        # ... the last-produced bucket starts at 0, each earlier one follows it, the first-produced one ends the buffer
        assert rng[-1][0] == 0 and rng[0][0] + rng[0][1] == total
        for b in range(nb - 1, 0, -1):
            assert rng[b][0] + rng[b][1] == rng[b - 1][0]
        # fc6's kernel has a bucket of its own (its 411 MB must not hold back the rest of the head, nor wait for it)
        offs = {nm: None for nm in names}
        for i in range(n):
            name = C.create_string_buffer(64); nd = C.c_int32(); shp = (C.c_int64 * 4)(); off = C.c_int64()
            _lib.lib.fcn8s_layout_param(C.byref(cfg), i, name, C.byref(nd), C.byref(shp), C.byref(off))
            offs[name.value.decode()] = off.value
        inb = lambda nm: [b for b, (o_, n_) in enumerate(rng) if o_ <= offs[nm] < o_ + n_][0]
        assert inb("fc7/weights") == inb("pool3_1x1/kernel") == inb("fc7_pool4_pool3_conv2d_trans/bias") == 0
        assert inb("fc6/weights") == inb("fc6/biases") == 1 and inb("conv5_3/filter") == inb("conv4_1/filter") == 2 and inb("conv1_1/filter") == inb("conv3_3/biases") == 3


def test_bad_config_is_rejected_without_gpu():
    from fcn8s_tensorflow_amd import _lib
    cfg = _cfg(0)
    assert _lib.lib.fcn8s_param_floats(C.byref(cfg)) == 0
    h = C.c_void_p()
    cfg = _cfg(3)                                 # not a multiple of 4
    assert _lib.lib.fcn8s_create(C.byref(cfg), C.byref(h)) == _lib.ERR_BAD_ARG
    assert b"num_classes" in _lib.lib.fcn8s_last_error(None)


def test_create_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    from fcn8s_tensorflow_amd import _lib
    h = C.c_void_p()
    rc = _lib.lib.fcn8s_create(C.byref(_cfg(20, (8, 16, 32, 64, 64, 128, 128))), C.byref(h))
    assert rc != 0 and b"no CPU fallback" in _lib.lib.fcn8s_last_error(None)
