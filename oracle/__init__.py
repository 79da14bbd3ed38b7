"""CPU oracle for the studiosr hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-fp32 / numpy *restatement* of the reference's
``model.forward()`` / ``model.inference()`` algorithms (SwinIR, HAT, EDSR, RCAN,
PixelShuffle upsampler), written as pure functions over a ``state_dict``.  It is
the checker for the HIP kernels in ``studiosr_amd``; it is never the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  ``studiosr_amd`` never imports it and has no CPU
fallback: its ops raise when the HIP library is missing.

Parity pin: every function here is checked (``tests/test_oracle_golden.py``)
against golden vectors in ``tests/golden/*.npz`` that were produced by importing
the reference itself from ``/root/reference`` with ``tests/golden/generate.py``
(the reference's own tests pin shapes only -- SURVEY.md section 8c).
"""
from . import functional, models, metrics  # noqa: F401
