"""CPU oracle for the pssr2_amd hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain CPU restatement (numpy + torch-CPU fp32) of the arithmetic on the
PSSR2 hot path that ``pssr2_amd`` implements in HIP.  It is the *checker*:

* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
  import it;
* nothing under ``pssr2_amd/`` imports it, and the product path raises when the HIP
  extension is missing rather than falling back to anything here.

Pinning status (see DESIGN.md "Oracle"):

* ``model_ref``   pinned  — checked against outputs of the genuine reference ``ResUNet`` /
  ``ResBlock`` / ``Reconstruction`` captured by ``oracle/gen_golden.py`` (tests/golden/*.npz).
* ``pairs_ref``   pinned  — Pillow-exact bilinear reduction, ``_gen_pair``, round/clip,
  ``_pred_array``, ``_patch_images``, tiling/val-split index maths, all against fixtures
  produced by the genuine reference functions.
* ``loss_ref``    Gaussian-L1 term pinned against the reference ``SSIMLoss`` (mix=0);
  the SSIM / MS-SSIM term restates pytorch_msssim 1.0.0 (``pyproject.toml:33``), which is
  absent from the reference tree and this image: **parity unpinned** for that term; it is
  cross-checked against an independent scipy.ndimage formulation in tests/test_oracle_loss.py.
* ``Blur``        restates skimage.filters.gaussian -> scipy.ndimage.gaussian_filter
  (mode="nearest", truncate=4.0); skimage absent: **parity unpinned**, cross-checked vs scipy.
"""
