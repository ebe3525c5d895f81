"""the train driver and the Lightning-layout checkpoint."""
import copy
import os

import numpy as np
import pytest
import torch

from helpers import golden_params, load_golden


def test_load_config_and_loader_sharding(tmp_path, monkeypatch):
    from vit_som_amd.train import TensorLoader, load_config
    p = tmp_path / "c.yaml"
    p.write_text("hyperparameters: {batch_size: 4}\ndata: {dataset: cifar-10, num_classes: 10}\n")
    assert load_config(str(p))["data"]["dataset"] == "cifar-10"
    monkeypatch.setenv("DATASET_NAME", "svhn")
    assert load_config(str(p))["data"]["dataset"] == "svhn"               # tools/utils.py:22-24
    x, y = torch.arange(20.).view(20, 1), torch.arange(20)
    a = [b[1] for b in TensorLoader(x, y, 4, rank=0, world_size=2)]
    b = [b[1] for b in TensorLoader(x, y, 4, rank=1, world_size=2)]
    assert sorted(torch.cat(a + b).tolist()) == list(range(20)) and not set(torch.cat(a).tolist()) & set(torch.cat(b).tolist())
    assert len(TensorLoader(x, y, 4, world_size=2).dataset) == 20


def test_checkpoint_layout_roundtrip_cpu(tmp_path):
    import vit_som_amd
    z, cfg = load_golden("ref_cls_tiny")
    m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device="cpu")
    m.load_state_dict(golden_params(z))
    (opt,), (sched,) = m.configure_optimizers()
    m.arena.exp_avg.normal_(); m.arena.exp_avg_sq.uniform_(); opt._step = 5
    path = m.save_checkpoint(str(tmp_path / "x.ckpt"), opt, sched, epoch=3)
    ck = torch.load(path, weights_only=True)
    assert {"epoch", "global_step", "state_dict", "hyper_parameters", "optimizer_states", "lr_schedulers"} <= set(ck)
    assert set(ck["state_dict"]) == set(golden_params(z)) and ck["hyper_parameters"] == cfg
    st = ck["optimizer_states"][0]
    nparams = sum(len(g["params"]) for g in st["param_groups"])
    assert len(st["state"]) == nparams and float(st["state"][0]["step"]) == 5.0
    # torch.optim.AdamW itself accepts this optimizer state (interchange with reference checkpoints)
    leaves = [p.detach().clone().requires_grad_(True) for g in opt.param_groups for p in g["params"]]
    k, groups = 0, []
    for g in opt.param_groups:
        groups.append({"params": leaves[k:k + len(g["params"])], "weight_decay": g["weight_decay"]}); k += len(g["params"])
    ref_opt = torch.optim.AdamW(groups, lr=1e-3)
    ref_opt.load_state_dict(st)
    assert torch.equal(ref_opt.state[leaves[0]]["exp_avg"], m.arena.view(m.arena.exp_avg, opt._param_names_in_group_order()[0]))
    # and back
    m2 = vit_som_amd.ViTSOM.load_from_checkpoint(path, config=cfg, device="cpu")
    for k2, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k2])
    (o2,), _ = m2.configure_optimizers()
    o2.load_state_dict(ref_opt.state_dict())
    assert o2._step == 5
    for n in opt._param_names_in_group_order():        # (arena padding between tensors is not optimizer state)
        assert torch.equal(m2.arena.view(m2.arena.exp_avg, n), m.arena.view(m.arena.exp_avg, n))
        assert torch.equal(m2.arena.view(m2.arena.exp_avg_sq, n), m.arena.view(m.arena.exp_avg_sq, n))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ref_cls_tiny", "ref_cluster_tiny"])
def test_driver_trains_checkpoints_and_evaluates(name, tmp_path):
    from vit_som_amd.train import main, synthetic_loaders
    _, cfg = load_golden(name)
    cfg = copy.deepcopy(cfg)
    cfg["hyperparameters"]["batch_size"] = 32
    cfg["hyperparameters"]["optimizer"]["lr"] = 0.02
    cfg["hyperparameters"]["optimizer"]["warmup_epochs"] = 1
    logs = []
    loaders = lambda c, r, w: synthetic_loaders(c, r, w, n_train=512, n_val=128, n_test=128)
    met = main(cfg, n_runs=2, max_epochs=4, make_loaders=loaders, model_states_dir=str(tmp_path / "states"), log=logs.append)
    losses = [float(l.split("train/total_loss=")[1].split()[0]) for l in logs if "train/total_loss=" in l]
    assert len(losses) == 8 and losses[3] < losses[0] and losses[7] < losses[4]          # learns, in both runs
    assert len(met["run_duration"]) == 2 and any("Aggregated Results" in l for l in logs)
    if cfg["data"]["num_classes"] > 0:
        assert len(met["accuracy"]) == 2 and 0.0 <= met["accuracy"][0] <= 1.0
        assert os.path.exists(tmp_path / "states" / "vit_som_synthetic_best.ckpt")
        assert met["accuracy"][0] > 1.5 / cfg["data"]["num_classes"]                    # clearly above chance on the templates
    else:
        assert len(met["purity"]) == 2 and os.path.exists(tmp_path / "states" / "last.ckpt")
        assert 0.0 < met["purity"][0] <= 1.0 and 0.0 <= met["nmi"][0] <= 1.0
