"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/vitsom_hip.h declares, and rejects bad calls with status codes (no GPU compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "vitsom_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vsom_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_functions():
    fns = header_functions()
    assert "vsom_bmu_cosine_fwd" in fns and "vsom_attention_bwd" in fns and len(fns) >= 25


def test_library_exports_every_declared_symbol():
    import vit_som_amd  # noqa: F401  (loads the library; raises ImportError if absent)
    from vit_som_amd._lib import LIB_PATH, SIGNATURES
    raw = ctypes.CDLL(LIB_PATH)
    for name in header_functions():
        assert hasattr(raw, name), f"{name} declared in include/vitsom_hip.h but not exported"
    assert sorted(SIGNATURES) == header_functions()


def test_version_and_error_string():
    from vit_som_amd._lib import lib
    assert lib.vsom_version() == 100
    assert isinstance(lib.vsom_last_error_string(), bytes)


def test_argument_validation_without_gpu():
    """Bad calls are rejected on the host before any launch (negative VSOM_E* codes)."""
    from vit_som_amd._lib import last_error, lib
    assert lib.vsom_linear_fwd(None, 4, None, None, None, 4, 4, 4, 4, None) == -1
    assert "null" in last_error()
    assert lib.vsom_adamw_step(1, 1, 1, 1, 1, 100, 0.1, 0.9, 0.999, 1e-8, 1, 1.0, 1, None) == -1     # n % 256
    assert lib.vsom_layernorm_fwd(16, 16, 16, 16, 16, 16, 4, 2048, 1e-6, None) == -3               # cols > 1024
    assert lib.vsom_attention_fwd(16, 16, 16, 1, 17, 2, 24, None) == -3                             # head dim 24
    assert lib.vsom_bmu_cosine_fwd(16, 8, 16, 16, 16, None, 16, 2, 3, 8, None, 0, None) == -4       # no workspace
    assert lib.vsom_som_neigh_loss(16, 16, 16, 1.0, None, None, 0.0, None, 16, None, None, None, 2, 3, 7, 16, 64, None) == -3   # distance 7
    # workspace queries are pure host arithmetic
    assert lib.vsom_linear_bwd_weight_workspace_bytes(33280, 576, 192) > 576 * 192 * 4
    assert lib.vsom_bmu_cosine_workspace_bytes(512, 1600, 12288) >= 512 * 1600 * 4
    assert lib.vsom_bmu_cosine_workspace_bytes(0, 1, 1) == 0


def test_no_oracle_import_in_product():
    """The product package must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "vit_som_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no CPU/eager fallback", ""), f"{f} mentions the oracle"
