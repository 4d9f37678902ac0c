"""MI355X-native drop-in for the empanada panoptic-inference hot path.

The directory is named ``empanada-napari_amd`` (reference repo name + ``_amd``);
it is imported under the module name ``empanada_napari_amd`` (see
``__graft_entry__.load_package`` / ``setup.py``'s ``package_dir``).

Sub-modules
  weights   -- parameter spec, seeded init, BatchNorm folding (host, numpy)
  synth     -- seeded synthetic EM tiles / head tensors
  _abi      -- ctypes binding of the C-ABI library ``libempanada_hip.so``
  build     -- hipcc build recipe for csrc/ (gfx950 only)
  engines   -- PanopticDeepLabRenderEngine[3d] mirror (empanada/inference/engines.py)
  inference -- Engine2d / Engine3d mirror (empanada_napari/inference.py)
"""
__version__ = '0.1.0'
