import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A hung kernel or oracle loop must fail one test, not eat the GPU box's whole time limit."""
    import pytest
    if config.pluginmanager.hasplugin("timeout"):
        for it in items:
            if it.get_closest_marker("timeout") is None:
                it.add_marker(pytest.mark.timeout(240))
