"""ctypes binding of libwavernn_amd.so (C-ABI: include/wavernn_amd.h).

The library is the product's only compute path: importing this module without
a built ``csrc/libwavernn_amd.so`` raises -- there is no CPU or PyTorch
fallback (the CPU restatement under ``oracle/`` is test infrastructure and is
never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libwavernn_amd.so')

MODE_RAW, MODE_MOL = 0, 1
NOISE_PHILOX, NOISE_INJECTED, NOISE_ARGMAX = 0, 1, 2
KERNEL_AUTO, KERNEL_SIMPLE, KERNEL_TEAM2, KERNEL_BATCH, KERNEL_BATCH_CS = 0, 1, 3, 4, 5
KERNEL_NAMES = {KERNEL_AUTO: 'auto', KERNEL_SIMPLE: 'simple', KERNEL_TEAM2: 'team2', KERNEL_BATCH: 'batch', KERNEL_BATCH_CS: 'batch_cs'}
KERNEL_IDS = {v: k for k, v in KERNEL_NAMES.items()}
DTYPE_F32, DTYPE_I64 = 0, 1
ERR_NAMES = {0: 'WRNN_OK', -1: 'WRNN_ERR_INVALID', -2: 'WRNN_ERR_HIP', -3: 'WRNN_ERR_STATE',
             -4: 'WRNN_ERR_MISSING_KEY', -5: 'WRNN_ERR_TIMEOUT', -6: 'WRNN_ERR_BUSY'}
ERR_INVALID, ERR_TIMEOUT, ERR_BUSY = -1, -5, -6
ABI_VERSION = 6   # WRNN_ABI_VERSION of the include/wavernn_amd.h this binding was written against

# every symbol include/wavernn_amd.h declares (checked by tests/test_cabi_symbols.py)
EXPORTED_SYMBOLS = ('wrnn_create', 'wrnn_load_weights', 'wrnn_conditioning', 'wrnn_plan', 'wrnn_generate',
                    'wrnn_last_timing', 'wrnn_n_classes', 'wrnn_loop_weight_bytes', 'wrnn_last_error',
                    'wrnn_abi_version', 'wrnn_destroy', 'wrnn_epilogue', 'wrnn_epilogue_rows', 'wrnn_epilogue_tables', 'wrnn_loss',
                    'wrnn_phase_profile', 'wrnn_phase_cycles', 'wrnn_train_step', 'wrnn_train_forward', 'wrnn_train_backward', 'wrnn_sync_status', 'wrnn_train_force_step_kernels',
                    'wrnn_dm_create', 'wrnn_dm_load_weights', 'wrnn_dm_generate', 'wrnn_dm_last_error', 'wrnn_dm_destroy',
                    'wrnn_dm_set_kernel', 'wrnn_dm_sync_status', 'wrnn_team_info', 'wrnn_debug_force_no_teams')


def epilogue_tables(n_classes: int, overlap: int, hop: int):
    """Host-only: (dec, fade_in, fade_out, tail) float64 arrays as ``wrnn_epilogue`` uses them."""
    lib = load_library()
    dec = np.empty(n_classes, np.float64)
    fin, fout = np.empty(max(overlap, 0), np.float64), np.empty(max(overlap, 0), np.float64)
    tail = np.empty(20 * hop, np.float64)
    rc = lib.wrnn_epilogue_tables(n_classes, overlap, hop, dec.ctypes.data, fin.ctypes.data if overlap else None,
                                  fout.ctypes.data if overlap else None, tail.ctypes.data)
    if rc != 0:
        raise WrnnError(rc, 'wrnn_epilogue_tables: invalid arguments')
    return dec, fin, fout, tail


class WrnnError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f'{ERR_NAMES.get(code, code)}: {msg}')
        self.code = code


class Config(C.Structure):
    _fields_ = [('rnn_dims', C.c_int32), ('fc_dims', C.c_int32), ('bits', C.c_int32), ('pad', C.c_int32),
                ('n_upsample', C.c_int32), ('upsample_factors', C.c_int32 * 4), ('feat_dims', C.c_int32),
                ('compute_dims', C.c_int32), ('res_out_dims', C.c_int32), ('res_blocks', C.c_int32),
                ('hop_length', C.c_int32), ('sample_rate', C.c_int32), ('mode', C.c_int32),
                ('device', C.c_int32)]


class TensorDesc(C.Structure):
    _fields_ = [('name', C.c_char_p), ('dtype', C.c_int32), ('ndim', C.c_int32), ('shape', C.c_int64 * 4),
                ('data', C.c_void_p)]


class SampleOpts(C.Structure):
    _fields_ = [('struct_size', C.c_uint32), ('noise_mode', C.c_int32), ('kernel', C.c_int32), ('mels_padded', C.c_int32),
                ('seed', C.c_uint64), ('noise1_dev', C.c_void_p), ('noise2_dev', C.c_void_p), ('x_forced_dev', C.c_void_p),
                ('logits_out_dev', C.c_void_p), ('x_init_dev', C.c_void_p), ('frames_dev', C.c_void_p),
                ('batch_rows', C.c_int32), ('team2_segment', C.c_int32)]


LOOP_PARAM_FIELDS = ('I_w', 'I_b', 'rnn1_w_ih', 'rnn1_w_hh', 'rnn1_b_ih', 'rnn1_b_hh', 'rnn2_w_ih', 'rnn2_w_hh', 'rnn2_b_ih',
                     'rnn2_b_hh', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'fc3_w', 'fc3_b')
# wrnn_loop_params field -> state_dict key of the reference module (fatchord_version.py:115-123)
LOOP_PARAM_KEYS = ('I.weight', 'I.bias', 'rnn1.weight_ih_l0', 'rnn1.weight_hh_l0', 'rnn1.bias_ih_l0', 'rnn1.bias_hh_l0',
                   'rnn2.weight_ih_l0', 'rnn2.weight_hh_l0', 'rnn2.bias_ih_l0', 'rnn2.bias_hh_l0', 'fc1.weight', 'fc1.bias',
                   'fc2.weight', 'fc2.bias', 'fc3.weight', 'fc3.bias')


class LoopParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in LOOP_PARAM_FIELDS]


class Timing(C.Structure):
    _fields_ = [('prologue_ms', C.c_float), ('loop_ms', C.c_float), ('kernel', C.c_int32), ('rows', C.c_int32),
                ('steps', C.c_int64), ('launches', C.c_int32), ('reserved_', C.c_int32)]


_lib: Optional[C.CDLL] = None


def _preload_hip_runtime() -> str:
    """One HIP runtime per process.  libwavernn_amd.so is linked with -no-hip-rt (no DT_NEEDED on a
    specific libamdhip64): PyTorch-ROCm wheels bundle their own runtime, and mixing it with /opt/rocm's in
    one process gives two HSA instances that do not share streams or events.  So: use torch's copy when
    torch is importable (device pointers, streams and events then belong to the same runtime), else ROCm's."""
    cands = []
    try:
        import torch
        cands.append(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
    except Exception:  # pragma: no cover - torch is a hard dependency of the host side
        pass
    cands += ['/opt/rocm/lib/libamdhip64.so', 'libamdhip64.so']
    for c in cands:
        try:
            C.CDLL(c, mode=C.RTLD_GLOBAL)
            return c
        except OSError:
            continue
    raise RuntimeError('no HIP runtime (libamdhip64.so) found')


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    _preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  There is no fallback path.')
    lib = C.CDLL(LIB_PATH)
    lib.wrnn_abi_version.argtypes = []
    lib.wrnn_abi_version.restype = C.c_int32
    got = int(lib.wrnn_abi_version())
    if got != ABI_VERSION:   # the .so is a git-ignored build artefact: a stale one would read the structs at shifted offsets
        raise RuntimeError(f'{LIB_PATH} implements ABI {got}, this binding needs ABI {ABI_VERSION}: rebuild it '
                           '(`python -c "import __graft_entry__ as g; g.build()"`)')
    vp = C.c_void_p
    lib.wrnn_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.wrnn_create.restype = C.c_int
    lib.wrnn_load_weights.argtypes = [vp, C.POINTER(TensorDesc), C.c_int32]
    lib.wrnn_load_weights.restype = C.c_int
    lib.wrnn_conditioning.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]
    lib.wrnn_conditioning.restype = C.c_int
    lib.wrnn_plan.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32),
                              C.POINTER(C.c_int64)]
    lib.wrnn_plan.restype = C.c_int
    lib.wrnn_generate.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  C.POINTER(SampleOpts), vp, vp, vp]
    lib.wrnn_generate.restype = C.c_int
    lib.wrnn_epilogue.argtypes = [vp, vp, vp, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                  vp, vp]
    lib.wrnn_epilogue.restype = C.c_int
    lib.wrnn_epilogue_rows.argtypes = [vp, vp, vp, C.c_int32, C.c_int64, C.c_int32, C.c_int64, vp, vp, C.c_int64, vp]
    lib.wrnn_epilogue_rows.restype = C.c_int
    lib.wrnn_phase_profile.argtypes = [vp, C.c_int32]
    lib.wrnn_phase_profile.restype = C.c_int
    lib.wrnn_phase_cycles.argtypes = [vp, C.POINTER(C.c_double)]
    lib.wrnn_phase_cycles.restype = C.c_int
    lib.wrnn_train_step.argtypes = [vp, C.POINTER(LoopParams), C.POINTER(LoopParams), vp, vp, vp, vp, C.c_int32, C.c_int64, vp, vp,
                                    vp, vp, vp]
    lib.wrnn_train_step.restype = C.c_int
    lib.wrnn_train_forward.argtypes = [vp, C.POINTER(LoopParams), vp, vp, vp, C.c_int32, C.c_int64, vp, vp]
    lib.wrnn_train_forward.restype = C.c_int
    lib.wrnn_train_backward.argtypes = [vp, C.POINTER(LoopParams), C.POINTER(LoopParams), vp, vp, vp, vp, C.c_int32, C.c_int64, vp, vp, vp]
    lib.wrnn_train_backward.restype = C.c_int
    lib.wrnn_sync_status.argtypes = [vp, vp]
    lib.wrnn_sync_status.restype = C.c_int
    lib.wrnn_train_force_step_kernels.argtypes = [vp, C.c_int32]
    lib.wrnn_train_force_step_kernels.restype = C.c_int
    lib.wrnn_epilogue_tables.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp]
    lib.wrnn_epilogue_tables.restype = C.c_int
    lib.wrnn_loss.argtypes = [vp, vp, vp, C.c_int64, vp, vp]
    lib.wrnn_loss.restype = C.c_int
    lib.wrnn_last_timing.argtypes = [vp, C.POINTER(Timing)]
    lib.wrnn_last_timing.restype = C.c_int
    lib.wrnn_n_classes.argtypes = [vp]
    lib.wrnn_n_classes.restype = C.c_int32
    lib.wrnn_loop_weight_bytes.argtypes = [vp]
    lib.wrnn_loop_weight_bytes.restype = C.c_int64
    lib.wrnn_last_error.argtypes = [vp]
    lib.wrnn_last_error.restype = C.c_char_p
    lib.wrnn_destroy.argtypes = [vp]
    lib.wrnn_destroy.restype = None
    lib.wrnn_team_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_char_p)]
    lib.wrnn_team_info.restype = C.c_int32
    lib.wrnn_debug_force_no_teams.argtypes = [vp, C.c_int32]
    lib.wrnn_debug_force_no_teams.restype = C.c_int
    lib.wrnn_dm_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.wrnn_dm_create.restype = C.c_int
    lib.wrnn_dm_load_weights.argtypes = [vp, C.POINTER(TensorDesc), C.c_int32]
    lib.wrnn_dm_load_weights.restype = C.c_int
    lib.wrnn_dm_generate.argtypes = [vp, C.c_int64, C.c_int32, C.c_uint64, vp, vp, vp, vp]
    lib.wrnn_dm_generate.restype = C.c_int
    lib.wrnn_dm_set_kernel.argtypes = [vp, C.c_int32]
    lib.wrnn_dm_set_kernel.restype = C.c_int
    lib.wrnn_dm_sync_status.argtypes = [vp, vp]
    lib.wrnn_dm_sync_status.restype = C.c_int
    lib.wrnn_dm_last_error.argtypes = [vp]
    lib.wrnn_dm_last_error.restype = C.c_char_p
    lib.wrnn_dm_destroy.argtypes = [vp]
    lib.wrnn_dm_destroy.restype = None
    _lib = lib
    return lib


class NativeVocoder:
    """Thin owner of one ``wrnn_handle`` (one per device)."""

    def __init__(self, *, rnn_dims, fc_dims, bits, pad, upsample_factors, feat_dims, compute_dims, res_out_dims,
                 res_blocks, hop_length, sample_rate, mode, device: int):
        self.lib = load_library()
        cfg = Config()
        cfg.rnn_dims, cfg.fc_dims, cfg.bits, cfg.pad = rnn_dims, fc_dims, bits, pad
        cfg.n_upsample = len(upsample_factors)
        for i, s in enumerate(upsample_factors):
            cfg.upsample_factors[i] = int(s)
        cfg.feat_dims, cfg.compute_dims, cfg.res_out_dims = feat_dims, compute_dims, res_out_dims
        cfg.res_blocks, cfg.hop_length, cfg.sample_rate = res_blocks, hop_length, sample_rate
        if mode not in ('RAW', 'MOL'):
            raise RuntimeError("Unknown model mode value - ", mode)
        cfg.mode = MODE_RAW if mode == 'RAW' else MODE_MOL
        cfg.device = int(device)
        self.device = int(device)
        self.hop = int(hop_length)
        self._h = C.c_void_p()
        rc = self.lib.wrnn_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            msg = self.lib.wrnn_last_error(self._h).decode() if self._h else 'wrnn_create failed'
            if self._h:
                self.lib.wrnn_destroy(self._h)
                self._h = C.c_void_p()
            raise WrnnError(rc, msg)
        self.n_classes = int(self.lib.wrnn_n_classes(self._h))

    def _check(self, rc: int):
        if rc != 0:
            raise WrnnError(rc, self.lib.wrnn_last_error(self._h).decode())

    def close(self):
        if getattr(self, '_h', None):
            self.lib.wrnn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, state_dict: Dict[str, np.ndarray]):
        """state_dict: name -> contiguous numpy array (float32 / int64), reference key names.  Every parameter the
        path reads must be present with the reference's shape (WrnnError otherwise); other keys are ignored."""
        keep, descs = [], []
        for name, arr in state_dict.items():
            arr = np.ascontiguousarray(arr)
            if arr.dtype == np.float32:
                dt = DTYPE_F32
            elif arr.dtype == np.int64:
                dt = DTYPE_I64
            else:
                arr = arr.astype(np.float32)
                dt = DTYPE_F32
            if arr.ndim > 4:
                continue
            d = TensorDesc()
            d.name = name.encode()
            d.dtype, d.ndim = dt, arr.ndim
            for i, s in enumerate(arr.shape):
                d.shape[i] = s
            d.data = arr.ctypes.data
            keep.append(arr)
            descs.append(d)
        arr_t = (TensorDesc * len(descs))(*descs)
        self._check(self.lib.wrnn_load_weights(self._h, arr_t, len(descs)))
        self.loop_weight_bytes = int(self.lib.wrnn_loop_weight_bytes(self._h))

    def plan(self, B: int, T: int, batched: bool, target: int, overlap: int) -> Tuple[int, int]:
        rows, steps = C.c_int32(), C.c_int64()
        self._check(self.lib.wrnn_plan(self._h, B, T, int(bool(batched)), int(target), int(overlap),
                                       C.byref(rows), C.byref(steps)))
        return rows.value, steps.value

    def conditioning(self, mels_ptr: int, B: int, T: int, up_ptr: int, aux_ptr: int, stream: int, mels_padded: bool = False):
        self._check(self.lib.wrnn_conditioning(self._h, mels_ptr, B, T, int(bool(mels_padded)), up_ptr or None, aux_ptr or None,
                                               stream or None))

    def generate(self, mels_ptr: int, B: int, T: int, batched: bool, target: int, overlap: int, *,
                 labels_ptr: int, samples_ptr: int, stream: int, noise_mode: int = NOISE_PHILOX, seed: int = 0,
                 noise1_ptr: int = 0, noise2_ptr: int = 0, x_forced_ptr: int = 0, logits_ptr: int = 0,
                 kernel: int = KERNEL_AUTO, x_init_ptr: int = 0, mels_padded: bool = False, frames_ptr: int = 0,
                 batch_rows: int = 0, team2_segment: int = 0):
        o = SampleOpts()
        o.struct_size = C.sizeof(SampleOpts)
        o.frames_dev = frames_ptr or None
        o.batch_rows, o.team2_segment = int(batch_rows), int(team2_segment)
        o.noise_mode, o.kernel, o.seed = noise_mode, kernel, seed & 0xFFFFFFFFFFFFFFFF
        o.noise1_dev, o.noise2_dev = noise1_ptr or None, noise2_ptr or None
        o.x_forced_dev, o.logits_out_dev = x_forced_ptr or None, logits_ptr or None
        o.x_init_dev = x_init_ptr or None
        o.mels_padded = int(bool(mels_padded))
        self._check(self.lib.wrnn_generate(self._h, mels_ptr, B, T, int(bool(batched)), int(target), int(overlap),
                                           C.byref(o), labels_ptr or None, samples_ptr, stream or None))

    def epilogue(self, samples_ptr: int, labels_ptr: int, rows: int, steps: int, batched: bool, target: int,
                 overlap: int, mu_law: bool, wave_len: int, out_ptr: int, stream: int):
        self._check(self.lib.wrnn_epilogue(self._h, samples_ptr, labels_ptr or None, rows, steps, int(bool(batched)),
                                           int(target), int(overlap), int(bool(mu_law)), int(wave_len), out_ptr,
                                           stream or None))

    def epilogue_rows(self, samples_ptr: int, labels_ptr: int, rows: int, steps: int, mu_law: bool, wave_len: int,
                      frames_ptr: int, out_ptr: int, out_stride: int, stream: int):
        """Every row finished as an independent unbatched utterance, one launch (``wrnn_epilogue_rows``)."""
        self._check(self.lib.wrnn_epilogue_rows(self._h, samples_ptr, labels_ptr or None, rows, steps, int(bool(mu_law)),
                                                int(wave_len), frames_ptr or None, out_ptr, int(out_stride), stream or None))

    def phase_profile(self, enable: bool = True):
        """Developer instrumentation: the following TEAM2 / BATCH calls run the instrumented loop kernel."""
        self._check(self.lib.wrnn_phase_profile(self._h, int(bool(enable))))

    def phase_cycles(self) -> np.ndarray:
        """(8 waves, 32 markers) cycles per step of workgroup 0 of team 0 for the last instrumented call."""
        out = np.zeros(8 * 32, np.float64)
        self._check(self.lib.wrnn_phase_cycles(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out.reshape(8, 32)

    def train_step(self, w_ptrs, g_ptrs, x_ptr: int, mels_up_ptr: int, aux_ptr: int, y_ptr: int, B: int, L: int, loss_ptr: int,
                   logits_ptr: int, d_mels_up_ptr: int, d_aux_ptr: int, stream: int):
        """``wrnn_train_step``: w_ptrs / g_ptrs = sequences of 16 device pointers in ``LOOP_PARAM_FIELDS`` order (g_ptrs may be
        None: forward + loss only)."""
        w = LoopParams(*[int(p) for p in w_ptrs])
        g = LoopParams(*[int(p) for p in g_ptrs]) if g_ptrs is not None else None
        self._check(self.lib.wrnn_train_step(self._h, C.byref(w), C.byref(g) if g is not None else None, x_ptr, mels_up_ptr, aux_ptr,
                                             y_ptr or None, int(B), int(L), loss_ptr or None, logits_ptr or None,
                                             d_mels_up_ptr or None, d_aux_ptr or None, stream or None))

    def train_forward(self, w_ptrs, x_ptr: int, mels_up_ptr: int, aux_ptr: int, B: int, L: int, logits_ptr: int, stream: int):
        w = LoopParams(*[int(p) for p in w_ptrs])
        self._check(self.lib.wrnn_train_forward(self._h, C.byref(w), x_ptr, mels_up_ptr, aux_ptr, int(B), int(L), logits_ptr, stream or None))

    def train_backward(self, w_ptrs, g_ptrs, d_logits_ptr: int, x_ptr: int, mels_up_ptr: int, aux_ptr: int, B: int, L: int,
                       d_mels_up_ptr: int, d_aux_ptr: int, stream: int):
        w = LoopParams(*[int(p) for p in w_ptrs])
        g = LoopParams(*[int(p) for p in g_ptrs])
        self._check(self.lib.wrnn_train_backward(self._h, C.byref(w), C.byref(g), d_logits_ptr, x_ptr, mels_up_ptr, aux_ptr, int(B), int(L),
                                                 d_mels_up_ptr or None, d_aux_ptr or None, stream or None))

    def sync_status(self, stream: int):
        """Waits for the stream; raises WrnnError for a device-side team-kernel error of ``train_step`` (busy GPU / timeout)."""
        self._check(self.lib.wrnn_sync_status(self._h, stream or None))

    def train_force_step_kernels(self, on: bool):
        self._check(self.lib.wrnn_train_force_step_kernels(self._h, int(bool(on))))

    def loss(self, y_hat_ptr: int, y_ptr: int, n_rows: int, out_ptr: int, stream: int):
        self._check(self.lib.wrnn_loss(self._h, y_hat_ptr, y_ptr, int(n_rows), out_ptr, stream or None))

    def team_info(self) -> Tuple[bool, int, str]:
        """(the XCD-team kernels can run for this handle, number of 32-CU teams, reason when they cannot) -- ``wrnn_team_info``."""
        n, why = C.c_int32(), C.c_char_p()
        ok = self.lib.wrnn_team_info(self._h, C.byref(n), C.byref(why))
        return bool(ok), int(n.value), (why.value or b'').decode()

    def debug_force_no_teams(self, on: bool):
        """Test hook: AUTO behaves as if the team kernels' residency check had failed."""
        self._check(self.lib.wrnn_debug_force_no_teams(self._h, int(bool(on))))

    def last_timing(self) -> dict:
        t = Timing()
        self._check(self.lib.wrnn_last_timing(self._h, C.byref(t)))
        return dict(prologue_ms=t.prologue_ms, loop_ms=t.loop_ms, kernel=t.kernel, rows=t.rows, steps=t.steps,
                    launches=t.launches)


def _tensor_descs(state_dict: Dict[str, np.ndarray]):
    keep, descs = [], []
    for name, arr in state_dict.items():
        arr = np.ascontiguousarray(arr)
        if arr.dtype != np.float32 or arr.ndim > 4:
            continue
        d = TensorDesc()
        d.name = name.encode()
        d.dtype, d.ndim = DTYPE_F32, arr.ndim
        for i, sdim in enumerate(arr.shape):
            d.shape[i] = sdim
        d.data = arr.ctypes.data
        keep.append(arr)
        descs.append(d)
    return keep, (TensorDesc * len(descs))(*descs), len(descs)


class NativeDeepmind:
    """Owner of one ``wrnn_dm_handle`` (the secondary dual-softmax model)."""

    def __init__(self, hidden_size: int, quantisation: int, device: int):
        self.lib = load_library()
        self.device = int(device)
        self._h = C.c_void_p()
        rc = self.lib.wrnn_dm_create(hidden_size, quantisation, device, C.byref(self._h))
        if rc != 0:
            msg = self.lib.wrnn_dm_last_error(self._h).decode() if self._h else 'wrnn_dm_create failed'
            self.close()
            raise WrnnError(rc, msg)

    def _check(self, rc: int):
        if rc != 0:
            raise WrnnError(rc, self.lib.wrnn_dm_last_error(self._h).decode())

    def close(self):
        if getattr(self, '_h', None):
            self.lib.wrnn_dm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weights(self, state_dict: Dict[str, np.ndarray]):
        keep, arr, n = _tensor_descs(state_dict)
        self._check(self.lib.wrnn_dm_load_weights(self._h, arr, n))

    def generate(self, seq_len: int, coarse_ptr: int, fine_ptr: int, stream: int, noise_mode: int = NOISE_PHILOX,
                 seed: int = 0, noise_ptr: int = 0):
        self._check(self.lib.wrnn_dm_generate(self._h, int(seq_len), noise_mode, seed & 0xFFFFFFFFFFFFFFFF,
                                              noise_ptr or None, coarse_ptr, fine_ptr, stream or None))
        self._check(self.lib.wrnn_dm_sync_status(self._h, stream or None))   # waits; surfaces device-side errors

    def set_kernel(self, kernel: int):
        self._check(self.lib.wrnn_dm_set_kernel(self._h, int(kernel)))
