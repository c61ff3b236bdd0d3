import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    so = os.path.join(ROOT, "csv-simd_amd", "csrc", "libcsvsimd_hip.so")
    if not os.path.exists(so):
        graft.build()
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle():
    o = graft.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(GOLDEN, "expected.json")) as f:
        exp = json.load(f)
    out = {}
    for name, e in exp.items():
        with open(os.path.join(GOLDEN, name), "rb") as f:
            out[name] = (f.read(), e)
    return out


@pytest.fixture(scope="session")
def ctx(pkg):
    """GPU context: only valid in -m gpu tests. Fails loudly (no fallback) without a device."""
    import torch
    assert torch.cuda.is_available(), "gpu test on a box without a GPU"
    c = pkg.Context(0)
    yield c
    c.close()


ALPHABET = np.frombuffer(b',"\n\ra \\\x00\xff', dtype=np.uint8)


def random_csvish(rng, n, p_quote=None):
    """Random bytes over the alphabet SURVEY.md §8c names: , " LF CR a space backslash 0x00 0xff"""
    w = np.ones(ALPHABET.size)
    if p_quote is not None:
        w[1] = p_quote * ALPHABET.size
    w = w / w.sum()
    return ALPHABET[rng.choice(ALPHABET.size, size=n, p=w)].astype(np.uint8)
