import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through the C-ABI HIP library)")


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return load


# ---- the CPU-oracle half of tests/test_gpu_fullshape.py::test_learnable_task_reaches_the_same_accuracy ---------------------------------
# 160 optimisation steps of the torch-CPU oracle take ~8 minutes and do not depend on the GPU: they run in a background process from
# the start of a GPU session (the host cores are otherwise idle under the GPU tests) and the test collects the result.
ORACLE_LEARN = {"proc": None, "out": None}


def start_oracle_learn(S: int, steps: int):
    import subprocess
    import tempfile
    if ORACLE_LEARN["proc"] is None:
        out = os.path.join(tempfile.mkdtemp(prefix="zsg_learn_"), f"oracle_{S}_{steps}.pt")
        ORACLE_LEARN["out"] = out
        ORACLE_LEARN["proc"] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "oracle_learn_worker.py"), str(S), str(steps), out],
                                                env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    return ORACLE_LEARN


def pytest_collection_finish(session):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu and any("test_learnable_task_reaches_the_same_accuracy" in it.nodeid for it in session.items):
        S, steps = (int(v) for v in os.environ.get("ZSG_TEST_LEARN", "128,160").split(","))
        start_oracle_learn(S, steps)


def pytest_sessionfinish(session, exitstatus):
    p = ORACLE_LEARN["proc"]
    if p is not None and p.poll() is None:
        p.kill()
