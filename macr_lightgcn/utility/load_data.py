"""`from utility.load_data import *` -- the LightGCN loader interface of the reference
(macr_lightgcn/utility/load_data.py: class Data :14), implemented in macr_amd.data."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from macr_amd.data import LGCNData as Data  # noqa: E402,F401
