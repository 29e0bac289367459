"""The oracle is the checker of every GPU parity test, and on the GPU box it runs on a different host (CPU model, thread
count, BLAS threading) than the container its fixtures were generated in.  This module re-collects the CPU tests that
pin `oracle/` to the reference's fixtures (tests/test_oracle_ram.py, test_oracle_unet.py, test_oracle_step.py) under the `gpu`
marker, so that `pytest -m gpu` on the GPU box also holds THAT instance of the oracle to the reference before it judges the
HIP kernels.  No GPU is touched here."""
import pytest

from test_oracle_ram import *        # noqa: F401,F403  (tests and their module-scoped fixtures)
from test_oracle_unet import *       # noqa: F401,F403
from test_oracle_step import *       # noqa: F401,F403

pytestmark = pytest.mark.gpu
