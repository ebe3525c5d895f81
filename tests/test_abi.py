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


def test_round3_entries_host_side():
    """Host arithmetic and argument checks of the entries added in round 3 (no launch happens)."""
    from vit_som_amd._lib import last_error, lib
    # plane images: 2 planes x 2 bytes per element of the padded operand + one squared-norm partial per row and 64 elements
    R, L = 1600, 12288
    assert lib.vsom_bmu_planes_bytes(R, L) == R * L * 4 + (L // 64) * R * 4
    assert lib.vsom_bmu_planes_bytes(70, 72) == 2 * 3 * 3 * 2048 + ((2 * 70 * 4 + 15) // 16) * 16      # 3 row blocks, 3 k-tiles of 32, padded
    assert lib.vsom_bmu_planes_bytes(0, 8) == 0
    assert lib.vsom_bmu_cosine_x3_planes_supported(512, 1600, 12288) == 1
    assert lib.vsom_bmu_cosine_x3_planes_supported(128, 1600, 12288) == 0           # small batches keep the in-loop split
    assert lib.vsom_bmu_cosine_x3_planes_supported(512, 1600, 12292) == 0           # L % 8
    assert lib.vsom_bmu_cosine_x3_planes_workspace_bytes(128, 1600, 12288) == 0
    assert lib.vsom_bmu_cosine_x3_planes_workspace_bytes(512, 1600, 12288) >= 512 * 1600 * 4
    assert lib.vsom_bmu_planes_from(16, 12288, 512, 12288, None, 0, None) == -4     # no plane buffer
    assert lib.vsom_bmu_cosine_x3_planes_dots(16, 16, 128, 1600, 12288, 16, 1 << 30, None) == -3
    assert "B >= 192" in last_error()
    assert lib.vsom_adamw_step_planes(16, 16, 16, 16, 16, 256 * 40, 0.1, 0.9, 0.999, 1e-8, 1, 1.0, 1, 100, 8, 64, 16,
                                      lib.vsom_bmu_planes_bytes(8, 64), None) == -1                      # slice offset not a multiple of 256
    # LayerNorm backward in two halves
    assert lib.vsom_layernorm_bwd_deferrable(33280, 192) == 1 and lib.vsom_layernorm_bwd_deferrable(100, 192) == 0
    assert lib.vsom_layernorm_bwd_finish_many(None, 0, 1, 192, None) == -1
    assert lib.vsom_layernorm_bwd_finish_many(16, 0, 0, 192, None) == 0             # nothing to do
    # communicator: nothing is held, nothing can be reduced
    import ctypes
    w, r = ctypes.c_int(7), ctypes.c_int(7)
    assert lib.vsom_comm_info(ctypes.byref(w), ctypes.byref(r)) == 0 and (w.value, r.value) == (0, -1)
    assert lib.vsom_comm_allreduce_sum(16, 4, None) == -1 and "communicator" in last_error()
    assert lib.vsom_comm_destroy() == 0
    # launch tape: cut / end / replay without a recording are refused or no-ops, never a crash
    assert lib.vsom_tape_recording() == 0


def test_hook_signature_sees_every_switch():
    from vit_som_amd.tuning import hooks
    base = hooks.signature()
    names = [k for k, _ in base]
    assert {"side_stream", "fwd_split", "bmu_planes", "adamw_planes", "ln_reduce_batched", "bmu_overlap", "launch_tape"} <= set(names)
    try:
        for k, v in base:
            hooks.set(**{k: (not v) if isinstance(v, bool) else 5})
            assert hooks.signature() != base, k
            hooks.reset()
            assert hooks.signature() == base
    finally:
        hooks.reset()


def test_no_oracle_import_in_product():
    """The product package must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "vit_som_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no CPU/eager fallback", ""), f"{f} mentions the oracle"
