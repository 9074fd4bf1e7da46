import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The library sends a FEW ciphertexts (<= 16 at N = 8192, <= 4 at N = 16384) through the whole-polynomial pipelines and larger
# batches through the head / middle / tail pipelines (Evaluator::few_for_split_*: latency vs throughput, same bits).  The parity
# tests use two or three ciphertexts per call: without this they would stop exercising the split kernels.  The suite therefore
# pins the choice to "by parameters only"; the selection itself is covered by the "small_batch_selection" variant of
# test_split_and_whole_polynomial_paths_agree, by test_concurrent_handle_calls_are_combined_without_changing_a_bit (runs with
# the product default) and by running the whole suite with HIPBFV_NO_SMALL_BATCH=0 (tools/gpu_variant_suites.sh).
os.environ.setdefault("HIPBFV_NO_SMALL_BATCH", "1")


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
