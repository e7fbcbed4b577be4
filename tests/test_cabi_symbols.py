"""CPU-side checks of the boundary: the library loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from tacotronv2_wavernn_chinese_amd import _cabi
    hdr = open(os.path.join(ROOT, 'include', 'wavernn_amd.h')).read()
    declared = sorted(set(re.findall(r'\b(wrnn_[a-z_]+)\s*\(', hdr)))
    assert declared == sorted(_cabi.EXPORTED_SYMBOLS)
    lib = _cabi.load_library()
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.wrnn_abi_version() == 3


def test_create_rejects_bad_config_without_gpu():
    from tacotronv2_wavernn_chinese_amd import _cabi
    lib = _cabi.load_library()
    cfg = _cabi.Config()
    h = ctypes.c_void_p()
    cfg.n_upsample = 0
    assert lib.wrnn_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    with pytest.raises(RuntimeError):
        _cabi.NativeVocoder(rnn_dims=512, fc_dims=512, bits=10, pad=2, upsample_factors=(5, 5, 11), feat_dims=80,
                            compute_dims=128, res_out_dims=128, res_blocks=10, hop_length=275, sample_rate=22050,
                            mode='XYZ', device=0)
    with pytest.raises(_cabi.WrnnError):  # unsupported dims are refused loudly, not silently mis-computed
        _cabi.NativeVocoder(rnn_dims=256, fc_dims=512, bits=10, pad=2, upsample_factors=(5, 5, 11), feat_dims=80,
                            compute_dims=128, res_out_dims=128, res_blocks=10, hop_length=275, sample_rate=22050,
                            mode='RAW', device=0)
