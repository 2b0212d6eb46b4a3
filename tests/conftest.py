import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    # gpu tests never silently pass on a CPU box: they are skipped (and the -m filter is
    # what the drivers use); on a GPU box a missing extension must FAIL, not skip.
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason='no GPU in this container (gpu-marked test)')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session', autouse=True)
def _built_library():
    """The .so is git-ignored: on a fresh checkout build it once (hipcc cross-compiles gfx950 without a GPU, ~2 min) so
    that the C-ABI surface tests run; without hipcc those tests fail loudly, as the product does."""
    from easy_vitpose_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        try:
            from easy_vitpose_amd.build import build_library
            build_library(force=False)
        except Exception as e:   # leave the failure to the tests that need the library
            print(f'[conftest] could not build {_capi.LIB_PATH}: {e}', file=sys.stderr)
    yield


@pytest.fixture
def one_launch_family(monkeypatch):
    """Bit identity ACROSS batch sizes (crop i of a batch == crop i alone) is a property of kernels that share one accumulation order: the one-launch GEMM
    family.  Since round 6 a call of one or two crops runs mlp.fc2 as four k ranges + a fixed-order reduction (tile_rules.hip pick_splitk) -- run-to-run
    bit-identical, inside the north_star tolerances against the reference (tests/test_gpu_api.py::test_split_k_*, tests/test_gpu_production_sizes.py::
    test_content_*), but NOT bit-identical to the batched kernels: the parity bar is +-0.5 px / 1e-3 against the oracle, not equal bits at every batch size
    (VERDICT r5 item 1).  Tests that use cross-batch bit identity as their race screen pin the handles they create to the one-launch family (VP_SPLITK=0)."""
    monkeypatch.setenv('VP_SPLITK', '0')
