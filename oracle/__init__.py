"""
oracle — CPU restatement of the reference's hot-path algorithms.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mdhelper_amd/`` imports this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import, call, link or execute it,
and there only as the checker (or the timed CPU baseline), never as the
thing shipped.  The product path (``mdhelper_amd``) raises if the HIP
library is missing instead of falling back to anything in here.

Pinning status:

* ``oracle.correlation``  — pinned against the reference's own
  ``src/mdhelper/algorithm/correlation.py`` run in the build container
  (fixtures ``tests/golden/correlation_*.npz`` made by
  ``scripts/make_golden.py``) and against the closed-form answers of
  ``tests/test_algorithm_correlation.py:438-561``.
* ``oracle.polymer``      — ``EndToEndVector`` frame loop + ``unwrap_edge`` for linear chains
  restated; its ACF goes through ``oracle.correlation`` (pinned); the reference holds no test
  or fixture for the class itself, so the class-level restatement is **parity unpinned** and
  is checked against closed forms (rigid rotors, rotational diffusion).
* ``oracle.fourier``      — pinned against the reference's
  ``src/mdhelper/algorithm/accelerated.py`` loop bodies run as plain Python
  (fixtures ``tests/golden/fourier_*.npz``).
* ``oracle.rdf``          — binning is the reference's actual call
  (``numpy.histogram``, ``analysis/structure.py:104``); the pair-distance
  arithmetic underneath lives in MDAnalysis (not vendored, not installed,
  version unpinned: ``mdanalysis >= 2.2.0``), so **the distance arithmetic is
  "parity unpinned" at the ULP level**: it follows the published
  brute-force/orthorhombic algorithm of ``MDAnalysis/lib/src/calc_distances.h``
  as stated in SURVEY.md §8(a-1) and is anchored on the reference's own
  call site (``structure.py:92-104``) and its ``radial_histogram`` test
  geometry (``tests/test_analysis_structure.py:21-40``).  The same holds for
  triclinic cells (``triclinic_vectors`` / ``triclinic_wrap`` /
  ``pair_distances_triclinic``: MDAnalysis' ``_triclinic_pbc`` +
  ``minimum_image_triclinic`` restated, **parity unpinned**; checked against a wide
  lattice search for the true minimum image in ``tests/test_oracle_rdf.py``).
"""
