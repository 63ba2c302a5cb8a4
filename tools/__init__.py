"""Test / benchmark scaffolding that is NOT part of the product package: synthetic clips and the calibrated synthetic
checkpoint (``synth.py``).  Nothing under ``v-floodnet_amd/`` imports from here."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
import vfloodnet_amd  # noqa: E402,F401  (the import shim for the hyphenated package directory)
