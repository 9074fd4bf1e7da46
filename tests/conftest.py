import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library sends a FEW ciphertexts (<= 16 at N = 8192, <= 4 at N = 16384) through the whole-polynomial pipelines and larger
# batches through the head / middle / tail pipelines (Evaluator::few_for_split_*: latency vs throughput, same bits).  The
# environment of the suite is the PRODUCT DEFAULT (nothing set here).  The parity tests use two or three ciphertexts per call,
# which alone would never reach the split kernels, so every GPU test runs with the selection pinned to "by parameters only"
# (`pipeline_selection` = "split": HIPBFV_NO_SMALL_BATCH=1, read by the library when an evaluator is created), and the
# representative modules below run a SECOND time under the product default ("product_default": the variable unset), inside
# the same `pytest -m gpu` run.
BOTH_SELECTIONS = ("test_gpu_parity.py", "test_gpu_baseline_configs.py", "test_gpu_program.py", "test_gpu_per_key.py")


def _is_gpu_test(definition) -> bool:
    return definition.get_closest_marker("gpu") is not None


def pytest_generate_tests(metafunc):
    if "pipeline_selection" in metafunc.fixturenames and _is_gpu_test(metafunc.definition):
        both = os.path.basename(str(metafunc.module.__file__)) in BOTH_SELECTIONS
        metafunc.parametrize("pipeline_selection", ["split", "product_default"] if both else ["split"], indirect=True)


@pytest.fixture(autouse=True)
def pipeline_selection(request, monkeypatch):
    sel = getattr(request, "param", None)
    if sel == "split":
        monkeypatch.setenv("HIPBFV_NO_SMALL_BATCH", "1")
    elif sel == "product_default":
        monkeypatch.delenv("HIPBFV_NO_SMALL_BATCH", raising=False)
    return sel


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with `-m gpu`; never run them implicitly on a box without a GPU.
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
