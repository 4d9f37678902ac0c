"""Entry script of one spawned multi-GPU rank process (multigpu.MultiGPUEngine3d._start).

Run through ``runpy.run_path`` with the globals PKG_INIT (path of this package's __init__.py) and ARGS (the arguments
of ``multigpu._rank_main``): the package lives in a directory whose name is not an identifier, so a spawned interpreter
cannot ``import`` it by name before this loader has registered it."""
import importlib.util
import os
import sys

NAME = 'empanada_napari_amd'
if NAME not in sys.modules:
    pkg_dir = os.path.dirname(PKG_INIT)      # noqa: F821  (injected by runpy)
    spec = importlib.util.spec_from_file_location(NAME, PKG_INIT, submodule_search_locations=[pkg_dir])   # noqa: F821
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)

from empanada_napari_amd import multigpu   # noqa: E402

multigpu._rank_main(*ARGS)                  # noqa: F821
