"""CPU oracle for the empanada panoptic-inference hot path.

TEST INFRASTRUCTURE ONLY.  Everything under ``oracle/`` is a plain CPU
restatement (numpy for integer/byte work, torch-CPU fp32 for the floating
point network) of the reference algorithm, each function citing the reference
file:line it follows.  It exists so that the HIP path can be checked against
it; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it.  The product package (``empanada-napari_amd``)
never imports, calls or falls back to anything in here.

Parity pinning: the oracle itself is pinned against
  * the reference's own unit tests for this path (tests/test_array_utils.py,
    tests/test_zarr_utils.py -- restated in tests/test_oracle_ranges.py), and
  * golden vectors generated in the build container by importing the reference
    (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
Pieces whose arithmetic lives in un-vendored third-party code (scikit-image
``label``/``regionprops``, cztile) say "parity unpinned" in their docstring.
"""
