"""INTEGRATION.md option B (the ctypes stub a maintainer pastes into the reference class) is the (b) row's evidence: it is pinned here.

CPU: the stub's three ctypes.Structure definitions are extracted from the markdown, executed, and compared field by field with the
gcc-compiled header (the same check tests/test_cabi_symbols.py applies to the package's own binding), and the ABI number the stub asserts is
the header's.  GPU: the stub's `_generate_device` is bound onto a reference-shaped module (same attribute names and state_dict as
fatchord_version.WaveRNN, :93-129) and its output for a T = 21 clip is checked against the oracle.
"""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_namespace():
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = md[md.index('## Option B'):]
    code = re.search(r'```python\n(.*?)```', sec, re.S).group(1)
    ns = {}
    exec(compile(code, 'INTEGRATION.md:option-B', 'exec'), ns)
    return ns, code


def test_option_b_structs_match_the_header(tmp_path):
    ns, code = _stub_namespace()
    hdr = open(os.path.join(ROOT, 'include', 'wavernn_amd.h')).read()
    hdr_abi = int(re.search(r'#define WRNN_ABI_VERSION (\d+)', hdr).group(1))
    assert int(re.search(r'wrnn_abi_version\(\) == (\d+)', code).group(1)) == hdr_abi
    structs = {'wrnn_config': ns['_Cfg'], 'wrnn_tensor_desc': ns['_Desc'], 'wrnn_sample_opts': ns['_Opts']}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/wavernn_amd.h"', 'int main(void) {']
    for cname, st in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c11', '-o', str(exe), str(src)])
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in subprocess.check_output([str(exe)], text=True).split('\n') if l.strip()}
    for cname, st in structs.items():
        assert got[(cname, 'size')] == C.sizeof(st), cname
        for fname, _ in st._fields_:
            assert got[(cname, fname)] == getattr(st, fname).offset, (cname, fname)
    # the entry points the stub calls exist in the header
    for sym in set(re.findall(r'lib\.(wrnn_[a-z_]+)', code)):
        assert re.search(r'\b' + sym + r'\s*\(', hdr), sym


@pytest.mark.parametrize('mode,bits', [('RAW', 10), ('MOL', 9)])
def test_option_b_config_and_descriptors_from_the_real_reference_class(mode, bits):
    """The stub reads attributes of the class it is pasted into (self.rnn_dims, self.fc1.out_features, self.pad, self.upsample..., :93-129).
    The GPU test binds it onto this package's own WaveRNN (the GPU box has no reference tree); HERE, where /root/reference exists, the stub's
    ``_cfg_of`` / ``_descs_of`` are run on the REAL ``wavernn.models.fatchord_version.WaveRNN`` and must give the wrnn_config and the
    tensor table the package's binding builds for the same weights."""
    if not os.path.isdir('/root/reference/wavernn'):
        pytest.skip('needs the reference tree (build container only)')
    import torch
    from oracle import ref_harness as rh
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    ns, _ = _stub_namespace()
    sd = make_state_dict(0, mode=mode, variant='default', bits=bits)
    ref_model = rh.build_reference_model(sd, mode=mode, bits=bits)
    assert type(ref_model).__module__ == 'wavernn.models.fatchord_version'
    cfg = ns['_cfg_of'](ref_model, 3)
    dims = dict(DEFAULT_DIMS, bits=bits)
    want = dict(rnn_dims=dims['rnn_dims'], fc_dims=dims['fc_dims'], bits=bits, pad=dims['pad'], n_upsample=len(dims['upsample_factors']),
                feat_dims=dims['feat_dims'], compute_dims=dims['compute_dims'], res_out_dims=dims['res_out_dims'], res_blocks=dims['res_blocks'],
                hop_length=dims['hop_length'], sample_rate=dims['sample_rate'], mode=_cabi.MODE_RAW if mode == 'RAW' else _cabi.MODE_MOL, device=3)
    for k, v in want.items():
        assert getattr(cfg, k) == v, k
    assert list(cfg.upsample_factors)[:cfg.n_upsample] == list(dims['upsample_factors'])
    keep, descs = ns['_descs_of'](ref_model)
    ours = WaveRNN(**dims, mode=mode)
    ours.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    ours_sd = {k: v.numpy() for k, v in ours.state_dict().items() if v.dtype == torch.float32}
    got = {d.name.decode(): (d.dtype, tuple(d.shape[:d.ndim])) for d in descs}
    assert got == {k: (_cabi.DTYPE_F32, a.shape) for k, a in ours_sd.items()}
    for d in descs:   # the data pointers carry the reference module's own weights
        a = np.ctypeslib.as_array(C.cast(d.data, C.POINTER(C.c_float)), shape=(int(np.prod(d.shape[:d.ndim])),))
        np.testing.assert_array_equal(a, ours_sd[d.name.decode()].reshape(-1))


@pytest.mark.gpu
def test_option_b_stub_generates_what_the_oracle_generates(monkeypatch):
    import torch
    from oracle import oracle as orc
    from tacotronv2_wavernn_chinese_amd import _cabi
    from tacotronv2_wavernn_chinese_amd.synth import DEFAULT_DIMS, make_mels, make_state_dict
    from tacotronv2_wavernn_chinese_amd.vocoder import WaveRNN
    from tests.parity_util import NEAR_TIE
    from tests.philox_ref import philox_uniform_raw_torch
    monkeypatch.setenv('WAVERNN_AMD_LIB', _cabi.LIB_PATH)
    ns, _ = _stub_namespace()
    sd = make_state_dict(0, variant='peaky')
    # the parameter container with the reference's attribute names and state_dict keys (fatchord_version.py:93-129); only the stub touches the library
    m = WaveRNN(**DEFAULT_DIMS, mode='RAW')
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    T = 21
    mels = make_mels(5, 1, T)
    seeds = []
    real_randint = torch.randint
    monkeypatch.setattr(torch, 'randint', lambda *a, **k: (lambda r: (seeds.append(int(r)), r)[1])(real_randint(*a, **k)))
    out = ns['_generate_device'](m, mels, False, 11000, 550)
    assert out.shape == (1, T * 275) and out.dtype == np.float64
    lab = np.rint((out[0] + 1.0) * 1023.0 / 2.0).astype(np.int64)
    np.testing.assert_allclose(2.0 * lab / 1023.0 - 1.0, out[0], atol=1e-6)
    # the oracle on the device's Philox draws (replayed on the host), driven along the stub's own trajectory
    L = T * 275
    u = philox_uniform_raw_torch(seeds[-1], 0, L, [0], device='cuda')
    q = (-torch.log(u.to(torch.float64))).to(torch.float32).cpu().numpy()
    om = orc.OracleModel(sd, fast=True)
    cm, ca = om.conditioning(mels)
    ref = om.loop(cm, ca, orc.NOISE_EXPO, q, x_forced=np.ascontiguousarray(out.T.astype(np.float32)))
    bad = np.flatnonzero(lab != ref['labels'][:, 0])
    for t in bad:
        assert ref['margin'][t, 0] < NEAR_TIE and lab[t] == ref['runner'][t, 0], f'step {t}'
    assert bad.size <= 1
