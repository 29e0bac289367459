"""oracle/ -- CPU restatement of the RAM-DSIR training hot path.  TEST INFRASTRUCTURE ONLY.

Everything in this package is a checker: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product path (``ram-dsir_amd/``) never
imports, links or executes anything from here and fails loudly when the HIP library is missing.

Each function cites the reference file:line (relative to the reference checkout, ``code/...``) it
restates.  The restatement is pinned against the reference itself: ``tests/golden/make_golden.py``
imports the reference's Python in the build container and writes small input/output fixtures under
``tests/golden/``; ``tests/test_oracle_*.py`` checks this package against them (no GPU needed).

Modules
  ram      numpy  -- Random Amplitude Mixup (fft2 / amplitude window lerp / ifft2), dataset/fundus.py:13-61
  unet     torch  -- functional Encoder / Decoder / Rec_Decoder on a reference-keyed state dict, networks/unet.py
  losses   torch  -- dice_loss, dice_loss_multi, KD, utils/losses.py:8-33, train.py:85-88
  step     torch  -- one full ``--ram --rec --consistency`` training step incl. Adam + poly LR, train.py:225-296
  masks    numpy  -- Fundus gray-mask -> 2-channel multilabel encoding, dataset/fundus.py:227-239
"""
