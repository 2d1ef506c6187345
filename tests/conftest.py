import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


# PyTorch bundles its own HIP runtime (libamdhip64 of its ROCm build); liblocohip.so links the system one. Whichever is
# loaded first serves both, and torch only finds its GPUs with its own: import torch before the library whenever a test
# session uses both (bench.py does the same for its multi-GPU reduction).
try:
    import torch  # noqa: F401
except Exception:          # pragma: no cover
    pass
