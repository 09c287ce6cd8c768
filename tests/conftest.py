"""pytest configuration: registers the ``gpu`` marker and puts the repo root and
the flat drop-in module directory (``multimodal-isic_amd/``) on ``sys.path``."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-isic_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


def pytest_sessionstart(session):
    """The oracle legs run on torch CPU ops: keep them to the CPUs the container's quota allows (bench.host_threads --
    the GPU box reports 256 CPUs and a 16-CPU quota, and 128 default threads are throttled to a crawl)."""
    try:
        import torch
        from bench import host_threads
        torch.set_num_threads(min(torch.get_num_threads(), host_threads()))
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
