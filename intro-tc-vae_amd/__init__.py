"""intro-tc-vae_amd: MI355X-native hot path of meffmadd/intro-tc-vae.

This directory is a DROP-IN ROOT: put it on ``sys.path`` ahead of the reference checkout and the
reference's ``main.py`` / ``train.py`` pick up ``models``, ``ops``, ``solvers`` and ``utils`` from
here unchanged.  It can also be imported as a package (``importlib.import_module("intro-tc-vae_amd")``),
which only registers the directory on ``sys.path`` and loads the HIP extension.
"""
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from hipvae import abi  # noqa: E402,F401  (raises if libitcv_hip.so is missing: no CPU fallback)
