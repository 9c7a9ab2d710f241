import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'gaussianprocess-mcmc_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


PER_TEST_TIMEOUT_S = 300      # every test here finishes in seconds; a hang must not hold a GPU box


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # pytest-timeout, when installed: a per-test limit unless the test sets its own
    if not config.pluginmanager.hasplugin('timeout'):
        return
    for item in items:
        if item.get_closest_marker('timeout') is None:
            item.add_marker(pytest.mark.timeout(PER_TEST_TIMEOUT_S))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
