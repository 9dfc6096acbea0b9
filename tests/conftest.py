import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

torch.set_float32_matmul_precision("highest")

# CPU thread pools (LAPACK references, the oracle) sized by what the cgroup grants, not by what the host shows
from cpu_quota import usable_cpus  # noqa: E402  (oracle/ is on the path above)

torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus())))
try:
    import threadpoolctl

    _BLAS_LIMIT = threadpoolctl.threadpool_limits(limits=usable_cpus())   # numpy / scipy BLAS and OpenMP pools
except Exception:  # pragma: no cover
    _BLAS_LIMIT = None


# the filtered eigensolver remembers late declines per (device, shape) and skips the route for the next requests of
# that shape: tests that provoke a decline must not change the route other tests see (one test switches it back on)
os.environ.setdefault("PTD_EIGH_FILTER_BACKOFF", "0")
# the second forward of a metric step reuses the products ahead of the analysed layer (_engine.PrefixMemo): under test
# every reused output is recomputed and compared bit for bit, in every model the suite runs
os.environ.setdefault("PTD_PREFIX_MEMO_CHECK", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with -m gpu; skip them cleanly if someone
    # runs the whole suite on a box without a device.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
