"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's learned-mask -> differentiable-ICP hot path
(SURVEY.md §8a).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package; the product
(``mm_masking_amd/``) never does and fails loudly when its HIP library is
missing.

Pinning status
--------------
* ``radar_ref`` / ``unet_ref`` / ``train_ref`` — pinned by golden vectors
  generated from the importable reference modules
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
* ``dicp_ref`` + ``nn_search.c`` — **parity unpinned**: the algorithm lives in
  the third-party dependency ``lisusdaniil/dICP`` (git submodule
  ``external/dICP``, no pinned commit; /root/reference/.gitmodules:4-6,
  requirements.txt:11) whose source is absent from /root/reference and the
  reference ships no tests or vectors for it.  The restatement follows the
  published algorithm (weighted, trimmed, robust Gauss-Newton ICP) and is
  anchored on the reference's call sites (icp_weight_policy.py:54-55,281-288;
  icp_weight_dataset.py:59-61,395) and on known-answer geometry.
"""
