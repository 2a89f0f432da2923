import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def _gpu_present():
    """a HIP device the product can use (device_count does not initialise the GPU); UCF_TEST_ASSUME_GPU=1 overrides"""
    if os.environ.get("UCF_TEST_ASSUME_GPU") == "1":
        return True
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a GPU: every test marked gpu is SKIPPED (the product has no CPU fallback, so it could only
    fail); `-m gpu` on the GPU box runs them.  Selecting them explicitly with -m gpu where there is no GPU still fails loudly"""
    if _gpu_present() or "gpu" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="needs a real MI355X: no HIP device here and the drawdown path has no CPU fallback")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def oracle_quad():
    from oracle_lib import Oracle
    return Oracle(quad=True)


@pytest.fixture(scope="session", autouse=True)
def _parity_report():
    """after a GPU session: the slack every parity gate used (tests/test_gpu_parity.py::PARITY) as JSON"""
    yield
    try:
        import json
        import test_gpu_parity
        if not test_gpu_parity.PARITY:
            return
        out = os.environ.get("UCF_PARITY_OUT", os.path.join(ROOT, "gpurun_out", "parity_r03.json"))
        os.makedirs(os.path.dirname(out), exist_ok=True)
        from unconfined_amd import engine
        rep = {"build_id": engine.build_id(),
               "what": "worst |err| / bound over all rows and radii of a deck, per gate and flavour (1.0 = the gate is exhausted); "
                       "gates: tests/test_gpu_parity.py (vs_binary128_truth = gate 1, vs_reference_out = gate 2), "
                       "tests/test_gpu_contract.py (f4_parameter_batch)",
               "gates": test_gpu_parity.PARITY}
        with open(out, "w") as f:
            json.dump(rep, f, indent=1, sort_keys=True)
    except Exception as exc:      # a report, not a test
        print(f"[parity report] not written: {exc}")
