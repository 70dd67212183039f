import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no GPU is visible — the product has no CPU fallback")
    from vqengine_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


@pytest.fixture
def set_opt(ctx):
    """set_opt(key, value): vqhip_set_option on the session's context for the duration of ONE test (the defaults come back afterwards)"""
    used = []

    def f(key, value):
        ctx.set_option(key, value)
        used.append(key)
    yield f
    for k in used:
        ctx.set_option(k, None)
