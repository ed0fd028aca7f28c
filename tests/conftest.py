import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    from stllm_amd import synth
    synth.enable_cache()  # the suites regenerate the same named full-width tensors many times
    # the golden fixtures were generated through a fake LlamaTokenizer whose add_special_tokens is a no-op (tests/golden/ref_shim.py:
    # 32000 words, pad id 0 in every mode); the product's stand-in defaults to HF's behaviour ('[PAD]' = id 32000, 32001 words,
    # st_llm.py:306-310).  Replay the fixtures under the tokenizer they were made with; tests/test_checkpoint_cpu.py covers the HF mode.
    from stllm_amd.tokenizer import IdTokenizer
    IdTokenizer.hf_special_tokens = False


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_device_memory():
    """GPU box only: a full-size model (35 GB of fp32 parameters + 25 GB of packed copies) sits in reference cycles (module <-> cached closures) that only
    the cyclic collector frees, and that collector is paced by Python allocations, not by device memory: five full-size tests in a row ran the 288 GB out."""
    yield
    import torch
    if torch.cuda.is_available():
        import gc
        gc.collect()
        torch.cuda.empty_cache()
