"""Run a tool against an A/B build of the library:  python tools/with_lib.py <libempanada_hip_xxx.so> <script.py> [args...]
(the diagnostic builds under empanada-napari_amd/lib/diag/ are made by hand with an extra -D macro; the product loads
lib/libempanada_hip.so only)"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from empanada_napari_amd import build as _b  # noqa: E402

lib = sys.argv[1]
_b.LIB = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
