"""Mirror of the reference's vocoder CLI ``wavernn_gen.py``.

``gen_from_file(model, load_path, save_path, batched, target, overlap)`` follows
``wavernn_gen.py:13-43``: a ``.npy`` mel shaped ``(T, n_mels)`` in ``[0, 1]`` (the format
``tacotron_synthesize.py:114-116`` writes) is transposed, validated, wrapped to ``(1, n_mels, T)`` and
handed to ``model.generate``; the output file name pattern is the reference's (:35-39).

Deliberate differences: the reference hard-overrides ``batched = False`` and
``device = cpu`` after parsing its flags (:76-77, :93); here ``--batched`` is honoured and the model
runs on the MI355X.  Extensions: ``--target auto|per_xcd``, ``--noise reference [--seed N]``
(the reference's own noise stream: ``vocoder.reference_noise``).  The reference's ``.wav`` input branch is broken (undefined ``file_name``, :18-20)
and needs librosa feature extraction, which is out of scope: it raises ``ValueError`` here.
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch

from .hparams import hparams as hp
from .vocoder import WaveRNN


def gen_from_file(model: WaveRNN, load_path, save_path, batched, target, overlap, **generate_opts):
    k = model.get_step() // 1000
    load_path = str(load_path)
    if ".npy" in load_path:
        mel = np.load(load_path).T
        if mel.ndim != 2 or mel.shape[0] != hp.num_mels:
            raise ValueError(f'Expected a numpy array shaped (n_mels, n_hops), but got {mel.shape}!')
        _max, _min = np.max(mel), np.min(mel)
        if _max >= 1.01 or _min <= -0.01:
            raise ValueError(f'Expected spectrogram range in [0,1] but was instead [{_min}, {_max}]')
    elif ".wav" in load_path:
        raise ValueError('wav -> mel feature extraction is outside the mel->wav path; pass a (T, n_mels) .npy mel')
    else:
        raise ValueError(f"Expected an extension of .wav or .npy, but got {os.path.splitext(load_path)[1]}!")

    mel = torch.tensor(mel).unsqueeze(0)
    batch_str = f'gen_batched_target{target}_overlap{overlap}' if batched else 'gen_NOT_BATCHED'
    idx = load_path.split('/')[-1].strip().split('.')[0]
    save_str = os.path.join(str(save_path), idx + '_' + batch_str + '_' + 'step={}k'.format(k) + '.wav')
    _ = model.generate(mel, save_str, batched, target, overlap, hp.mu_law, **generate_opts)
    print('\n\nstep = {}'.format(k * 1000))
    return save_str


def default_weights_path(base: str = '.') -> str:
    """``Paths(hp.voc_model_id).voc_latest_weights`` of the reference (``wavernn/utils/paths.py:8-12``)."""
    return os.path.join(os.path.abspath(base), 'logs_wavernn', 'checkpoints', 'latest_weights.pyt')


def build_model_from_hparams() -> WaveRNN:
    """``wavernn_gen.py:99-110``."""
    return WaveRNN(rnn_dims=hp.voc_rnn_dims, fc_dims=hp.voc_fc_dims, bits=hp.bits, pad=hp.voc_pad,
                   upsample_factors=hp.voc_upsample_factors, feat_dims=hp.num_mels,
                   compute_dims=hp.voc_compute_dims, res_out_dims=hp.voc_res_out_dims,
                   res_blocks=hp.voc_res_blocks, hop_length=hp.hop_length, sample_rate=hp.sample_rate,
                   mode=hp.voc_mode)


def main(argv=None):
    from .hparams import DEFAULT_HPARAMS
    parser = argparse.ArgumentParser(description='Generate WaveRNN Samples')
    parser.add_argument('--batched', '-b', dest='batched', action='store_true', help='Fast Batched Generation')
    parser.add_argument('--unbatched', '-u', dest='batched', action='store_false', help='Slow Unbatched Generation')
    parser.add_argument('--samples', '-s', type=int, help='[int] number of utterances to generate')
    parser.add_argument('--target', '-t', type=lambda s: s if s in ('auto', 'per_xcd') else int(s),
                        help="[int] number of samples in each batch index ('auto': the fold length with the lowest predicted latency on this GPU; "
                             "'per_xcd': one fold per XCD)")
    parser.add_argument('--overlap', '-o', type=int, help='[int] number of crossover samples')
    parser.add_argument('--file', '-f', type=str, help='[string/path] (T, n_mels) .npy mel to vocode')
    parser.add_argument('--voc_weights', '-w', type=str, help='[string/path] Load in different WaveRNN weights')
    parser.add_argument('--gta', '-g', dest='gta', action='store_true', help='Generate from GTA testset')
    parser.add_argument('--force_cpu', '-c', action='store_true',
                        help='accepted for compatibility; this package has no CPU path and will raise')
    parser.add_argument('--noise', choices=['philox', 'reference'], default='philox',
                        help="extension: 'reference' replays the reference's own draws from the torch CPU generator (with --seed: the wav the "
                             "reference script produces after torch.manual_seed(seed)); 'philox' (default) = the device counter RNG")
    parser.add_argument('--seed', type=int, default=None, help='extension: torch.manual_seed(SEED) before generating')
    parser.add_argument('--hp_file', metavar='FILE', default=DEFAULT_HPARAMS,
                        help='The file to use for the hyperparameters')
    parser.set_defaults(batched=None)
    args = parser.parse_args(argv)

    hp.configure(args.hp_file)
    if args.target is None:
        args.target = hp.voc_target
    if args.overlap is None:
        args.overlap = hp.voc_overlap
    if args.batched is None:
        args.batched = hp.voc_gen_batched
    if args.samples is None:
        args.samples = hp.voc_gen_at_checkpoint
    if args.force_cpu or not torch.cuda.is_available():
        raise RuntimeError('this vocoder runs on an MI355X only (no CPU path)')

    device = torch.device('cuda')
    print('Using device:', device)
    print('\nInitialising Model...\n')
    model = build_model_from_hparams().to(device)
    # wavernn_gen.py:112-117: `--voc_weights`, else the latest checkpoint of the training run
    # (Paths.voc_latest_weights = <base>/logs_wavernn/checkpoints/latest_weights.pyt, wavernn/utils/paths.py:11-12; <base> is
    # the directory the script is started from here).  The reference then dies in torch.load when that file is absent; no
    # checkpoint ships with the repository (.MISSING_LARGE_BLOBS), so an absent default falls back to the seeded random
    # initialisation, loudly.
    voc_weights = args.voc_weights if args.voc_weights else default_weights_path()
    print(voc_weights)
    if args.voc_weights or os.path.exists(voc_weights):
        model.load(voc_weights)
    else:
        print(f'{voc_weights} does not exist and no --voc_weights given: using the randomly initialised model')
    if args.file:
        out_dir = './wavernn_inference_output'
        os.makedirs(out_dir, exist_ok=True)
        if args.seed is not None:
            torch.manual_seed(args.seed)
        gen_from_file(model, args.file, out_dir, args.batched, args.target, args.overlap, noise_mode=args.noise)
    print('\n\nExiting...\n')


if __name__ == "__main__":
    main()
