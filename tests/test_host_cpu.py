"""Host logic that needs no GPU: module surface, state-dict keys, arenas, optimizer groups,
schedules.  (Compute itself has no CPU path and must fail loudly.)"""
import copy
import math

import numpy as np
import pytest
import torch

from helpers import GOLDEN, golden_params, load_golden


def model(name="ref_cls_tiny"):
    import vit_som_amd
    z, cfg = load_golden(name)
    return vit_som_amd.ViTSOM(copy.deepcopy(cfg), device="cpu"), z, cfg


def test_state_dict_keys_and_shapes_match_reference():
    for name in ("ref_cls_tiny", "ref_cluster_tiny", "ref_mnistlike_tiny"):
        m, z, _ = model(name)
        ref = golden_params(z)
        sd = m.state_dict()
        assert list(sd.keys()) == list(ref.keys()) or set(sd) == set(ref)
        for k in sd:
            assert tuple(sd[k].shape) == tuple(ref[k].shape), k
        m.load_state_dict(ref)
        assert m._it == int(z["iteration"])
        for k, v in m.state_dict().items():
            assert torch.equal(v, ref[k]), k


def test_parameters_are_arena_views():
    m, _, _ = model()
    a = m.arena
    for n, p in m.named_parameters():
        if p.requires_grad:
            off, numel, shape = a.offsets[n]
            assert off % 256 == 0 and p.data_ptr() == a.params.data_ptr() + 4 * off and tuple(p.shape) == shape
    assert a.numel % 256 == 0 and a.wd_chunk.numel() == a.numel // 256
    # frozen sincos tables and buffers are outside the trainable arena
    assert not m.vit.pos_embed.requires_grad and "vit.pos_embed" not in a.offsets
    m2 = m.float()          # _apply keeps the aliasing
    assert all(p.data_ptr() == a.p(n).data_ptr() for n, p in m2._named_trainable())


def test_init_distributions():
    m, z, _ = model()
    ref = golden_params(z)
    assert torch.equal(m.vit.pos_embed, ref["vit.pos_embed"])
    assert torch.equal(m.som_layer.grid_positions, ref["som_layer.grid_positions"])
    W = m.som_layer.prototypes
    assert torch.allclose(W.norm(dim=1), torch.ones(W.shape[0]), atol=1e-6) and float(W.min()) >= 0
    blk = m.vit.blocks[0]
    assert float(blk.attn.qkv.bias.abs().max()) == 0 and float(blk.norm1.weight.min()) == 1
    a = math.sqrt(6.0 / (blk.attn.qkv.weight.shape[0] + blk.attn.qkv.weight.shape[1]))
    assert float(blk.attn.qkv.weight.abs().max()) <= a


def test_optimizer_groups_and_lr_schedule_match_reference():
    m, z, cfg = model()
    (opt,), (sched,) = m.configure_optimizers()
    ref = np.load(GOLDEN + "/ref_lr_schedule.npz")
    assert len(opt.param_groups) == int(ref["n_groups"])
    assert all("lr_scale" in g for g in opt.param_groups[:-1]) and "lr_scale" not in opt.param_groups[-1]
    assert sorted({float(g["weight_decay"]) for g in opt.param_groups}) == [0.0, 0.01, 0.05]
    # per-chunk decay table: ViT >=2-D 0.05, ViT 1-D 0, prototypes/cls_head 0.01, decoder 0 (cls mode)
    wd = m.arena.wd_by_name
    assert wd["vit.blocks.0.attn.qkv.weight"] == 0.05 and wd["vit.blocks.0.attn.qkv.bias"] == 0.0
    assert wd["vit.cls_token"] == 0.05 and wd["som_layer.prototypes"] == 0.01 and wd["cls_head.bias"] == 0.01
    assert wd["vit.decoder_embed.weight"] == 0.0
    lrs = []
    for e in range(cfg["hyperparameters"]["total_epochs"]):
        lrs.append(opt.param_groups[0]["lr"])
        sched.step()
    assert np.allclose(lrs, ref["lrs"], rtol=1e-12)


def test_clustering_mode_keeps_decoder_decay():
    m, _, _ = model("ref_cluster_tiny")
    m.configure_optimizers()
    assert m.arena.wd_by_name["vit.decoder_embed.weight"] == 0.05


def test_schedules():
    m, z, cfg = model("ref_cluster_tiny")
    m.load_state_dict(golden_params(z))
    with pytest.raises(RuntimeError):
        m._gamma_t()
    m.set_schedule(int(z["n_train"]), int(z["est_steps"]))
    m.som_layer.update_temperature(m._it)
    assert abs(m.som_layer.current_temperature - float(z["train/T"])) < 1e-6 * float(z["train/T"])   # reference evaluates T in float32
    hp = cfg["hyperparameters"]
    assert abs(m._gamma_t() - hp["gamma"] * min(1.0, int(z["iteration"]) / (int(z["est_steps"]) // 2))) < 1e-15
    # data-parallel: temperature follows the GLOBAL batch
    m.set_distributed(4, 0)
    assert abs(m.som_layer.total_iterations() - (int(z["n_train"]) / (hp["batch_size"] * 4)) * hp["total_epochs"]) < 1e-9
    assert m.som_layer.index_to_position(torch.tensor([10])).tolist() == [[2.0, 0.0]]    # 3x5 map
    assert np.array_equal(m.som_layer.index_to_position(torch.tensor([10])).numpy(), z["ka/index_to_position_10"])


def test_compute_has_no_cpu_path():
    m, z, _ = model()
    with pytest.raises(ValueError):
        m(torch.from_numpy(z["x"]))
    with pytest.raises(ValueError):
        m.som_layer(torch.randn(2, m.som_layer.latent_dim))
    with pytest.raises(ValueError):
        import vit_som_amd
        _, cfg = load_golden("ref_hexa_euclid_tiny")
        cfg["hyperparameters"]["som"]["distance_fcn"] = "chebyshev"
        vit_som_amd.ViTSOM(cfg, device="cpu")       # som_layer.py:124-125: unknown distance functions raise


def test_patchify_roundtrip():
    m, z, _ = model()
    x = torch.from_numpy(z["x"])
    assert torch.equal(m.vit.unpatchify(m.vit.patchify(x)), x)
