"""Shared test helpers (golden loading, tolerances)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_CASES = ["ref_cluster_tiny", "ref_cls_tiny", "ref_mnistlike_tiny", "ref_hexa_euclid_tiny", "ref_manhattan_tiny"]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    cfg = json.loads(str(z["config_json"]))
    return z, cfg


def golden_params(z, prefix="param/"):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))
