"""Runs a script of this repository against ANOTHER build of the library (same-box A/B measurements, tools/ab.sh):
    python tools/ab_run.py <other libzksaas_hip.so> <script.py> [args...]
The package loader has no environment override (a deployment cannot inherit one): the library path is set here, in process,
before anything is loaded; the loader still refuses a build that lacks a symbol of include/zksaas.h."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib, script = os.path.abspath(sys.argv[1]), sys.argv[2]
import zksaas_amd                                  # noqa: E402  (registers the hyphenated package directory)
from zksaas_amd import _lib                        # noqa: E402
assert _lib._lib is None, "the library was loaded before the override"
_lib.LIB_PATH = lib
sys.argv = [script] + sys.argv[3:]
runpy.run_path(script, run_name="__main__")
