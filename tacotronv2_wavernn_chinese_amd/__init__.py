"""MI355X-native WaveRNN vocoder: drop-in for the mel->wav path of
lturing/tacotronv2_wavernn_chinese (``wavernn_gen.py`` / ``WaveRNN.generate``)."""
from .hparams import hparams  # noqa: F401

__all__ = ['hparams']
