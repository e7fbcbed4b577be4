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
    hdr_abi = int(re.search(r"#define WRNN_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.wrnn_abi_version() == hdr_abi == _cabi.ABI_VERSION


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


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The ctypes mirrors in _cabi.py against the C header itself: a C program that includes include/wavernn_amd.h prints
    sizeof / offsetof of every field; a field added on one side only (or reordered) fails here, not as silent garbage in a
    device call."""
    import ctypes as C
    import os
    import subprocess
    from tacotronv2_wavernn_chinese_amd import _cabi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {'wrnn_config': _cabi.Config, 'wrnn_tensor_desc': _cabi.TensorDesc, 'wrnn_sample_opts': _cabi.SampleOpts,
               'wrnn_timing': _cabi.Timing}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{root}/include/wavernn_amd.h"', 'int main(void) {']
    for cname, st in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c11', '-o', str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True).split('\n')
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out if l.strip()}
    for cname, st in structs.items():
        assert got[(cname, 'size')] == C.sizeof(st), cname
        for fname, _ in st._fields_:
            assert got[(cname, fname)] == getattr(st, fname).offset, (cname, fname)


def test_stale_library_is_refused(monkeypatch):
    """load_library() compares wrnn_abi_version() with the ABI the binding was written against: a stale .so (a git-ignored
    build artefact) must fail at load, not read the structs at shifted offsets."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    real = _cabi.ABI_VERSION
    monkeypatch.setattr(_cabi, '_lib', None)
    monkeypatch.setattr(_cabi, 'ABI_VERSION', real - 1)
    with pytest.raises(RuntimeError, match='ABI'):
        _cabi.load_library()
    monkeypatch.setattr(_cabi, 'ABI_VERSION', real)
    assert _cabi.load_library().wrnn_abi_version() == real


def test_generate_refuses_an_unknown_opts_struct_size():
    """wrnn_sample_opts.struct_size (ABI 4): a caller built against another revision of the header is WRNN_ERR_INVALID.
    The check precedes every device call, so it runs without a GPU (wrnn_create itself needs a HIP device: skipped where it
    fails for that reason)."""
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS
    try:
        nat = _cabi.NativeVocoder(device=0, mode='RAW', **DEFAULT_DIMS)
    except _cabi.WrnnError as e:
        if e.code == -2:
            pytest.skip('no HIP device: wrnn_create cannot make a handle here')
        raise
    o = _cabi.SampleOpts()
    o.struct_size = ctypes.sizeof(_cabi.SampleOpts) - 8
    rc = nat.lib.wrnn_generate(nat._h, 1, 1, 1, 0, 0, 0, ctypes.byref(o), None, 1, None)
    assert rc == -1 and b'struct_size' in nat.lib.wrnn_last_error(nat._h)
