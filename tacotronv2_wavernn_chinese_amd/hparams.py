"""Run-time configured hyper-parameter bag.

Mirror of the ``hparams`` singleton of ``wavernn/utils/__init__.py:40-104``:
attribute access raises until ``configure(path)`` has imported a python file;
``configure`` may be called once.
"""
from __future__ import annotations

import os
import re
from importlib.util import module_from_spec, spec_from_file_location
from pathlib import Path
from typing import Union

DEFAULT_HPARAMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'default_hparams.py')


class _HParams:
    def __init__(self):
        self._configured = False

    def __getattr__(self, item):
        if item.startswith('_'):
            raise AttributeError(item)
        if not self.__dict__.get('_configured', False):
            raise AttributeError("HParams not configured yet. Call self.configure()")
        raise AttributeError(item)

    def is_configured(self):
        return self._configured

    def configure(self, path: Union[str, Path] = DEFAULT_HPARAMS):
        if self.is_configured():
            raise RuntimeError("Cannot reconfigure hparams!")
        path = Path(path).expanduser()
        if not path.exists():
            raise FileNotFoundError(f"Could not find hparams file {path}")
        if path.suffix != ".py":
            raise ValueError("`path` must be a python file")
        spec = spec_from_file_location("hparams", path)
        mod = module_from_spec(spec)
        spec.loader.exec_module(mod)
        magic = re.compile(r"^__.+__$")
        for name, value in mod.__dict__.items():
            if magic.match(name):
                continue
            if name in self.__dict__:
                raise AttributeError(f"module at `path` cannot contain attribute {name}")
            setattr(self, name, value)
        self._configured = True


hparams = _HParams()
