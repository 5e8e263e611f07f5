import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests never run silently on CPU: without a device they are skipped only
    when not explicitly selected with ``-m gpu``."""
    import importlib
    have_gpu = None
    for item in items:
        if "gpu" in item.keywords:
            if have_gpu is None:
                try:
                    torch = importlib.import_module("torch")
                    have_gpu = bool(torch.cuda.device_count() > 0)
                except Exception:
                    have_gpu = False
            if not have_gpu and "gpu" not in (config.getoption("-m") or ""):
                item.add_marker(pytest.mark.skip(reason="no GPU in this container"))


def pytest_sessionstart(session):
    """Build the native libraries (HIP engine, BAM reader, C oracle) when they are missing or older
    than their sources -- the same thing ``__graft_entry__.build()`` does; a no-op otherwise.
    Building is not counting: the product still refuses to run without its HIP library."""
    try:
        from plastid_amd import build
        build.build_library()
        build.build_bam_library()
    except Exception as e:  # no hipcc here: the tests that need the library say so themselves
        sys.stderr.write("conftest: native build skipped (%s)\n" % e)
    try:
        from oracle import oracle
        oracle.build()
    except Exception as e:
        sys.stderr.write("conftest: oracle build skipped (%s)\n" % e)
