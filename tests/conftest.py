import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    class G:
        forward = np.load(os.path.join(GOLDEN, "forward.npz"))
        parse = np.load(os.path.join(GOLDEN, "parse_paf.npz"))
        yolo = np.load(os.path.join(GOLDEN, "parse_yolo.npz"))
        pafprocess = np.load(os.path.join(GOLDEN, "pafprocess.npz"))
        cpp = np.load(os.path.join(GOLDEN, "paf_to_pose_cpp.npz"))
        keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))
        script = json.load(open(os.path.join(GOLDEN, "script_eval_data.json")))
        script_yolo = json.load(open(os.path.join(GOLDEN, "script_eval_data_yolo.json")))
        script_metrics = json.load(open(os.path.join(GOLDEN, "script_metrics.json")))
        script_metrics_yolo = json.load(open(os.path.join(GOLDEN, "script_metrics_yolo.json")))
    return G


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)
