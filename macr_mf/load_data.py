"""`from load_data import Data` -- the MF loader/sampler interface of the reference
(macr_mf/load_data.py: class Data :24, ctor :504, sample :543), implemented in macr_amd.data."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from macr_amd.data import MFData as Data  # noqa: E402,F401
