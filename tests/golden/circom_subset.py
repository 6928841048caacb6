"""The Circom-subset front-end lives in the package (circom-2-arithc_amd/circom_frontend.py: it is the host-side
producer of the flat gate list); the fixture generator and the tests reach it through this name."""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_m = importlib.import_module("circom-2-arithc_amd.circom_frontend")
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
