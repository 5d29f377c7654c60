import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


# GPU tests run small -> large: the driver runs `pytest -x`, so whatever dies first hides everything after it. The box canary
# and the golden parity tests come first, the 70 011-row persistent-GEMM regime test and the
# bench-contract / multi-process files last.
GPU_FILE_ORDER = [
    "test_00_canary_gpu.py",
    "test_model_gpu.py",            # golden forward / backward of the towers, loss, packed text batches
    "test_ops_gpu.py",              # every kernel against fp64 / the oracle
    "test_pack_meta_gpu.py",
    "test_dropout.py",
    "test_hf_gpu.py", "test_openclip_gpu.py", "test_wukong_gpu.py",
    "test_resnet_gpu.py",
    "test_preprocess.py", "test_dataset.py",
    "test_engine_state_gpu.py",
    "test_bench_regime_gpu.py",     # 70 011-row GEMMs, ViT-B/16 + BERT-base at 64 / 256 pairs
]


def _gpu_rank(item):
    name = os.path.basename(str(item.fspath))
    if name in GPU_FILE_ORDER:
        return GPU_FILE_ORDER.index(name)
    if name.startswith("test_zz_"):
        return len(GPU_FILE_ORDER) + 1
    return GPU_FILE_ORDER.index("test_bench_regime_gpu.py") - 0.5      # unlisted files: before the heavy one


def pytest_collection_modifyitems(config, items):
    # stable sort: CPU tests keep their order (rank of the file they are in, so that files stay together)
    items.sort(key=_gpu_rank)
    # GPU tests never run by accident on a box without a GPU
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
