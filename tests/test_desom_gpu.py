"""DESOM (models/desom.py) as a second client of the SOM kernels: the HIP path against the goldens the
reference itself produced (tests/golden/ref_desom_*.npz) and against the CPU oracle at the shipped
configs' layer sizes.  Tolerances as for ViT-SOM: outputs 1e-4 abs, gradients 1e-4 relative, BMU exact."""
import copy

import pytest
import torch

from helpers import golden_params, load_golden, rel_err

pytestmark = pytest.mark.gpu
CASES = ["ref_desom_tiny", "ref_desom_cls_tiny", "ref_desom_euclid_tiny"]


def build(name):
    import vit_som_amd
    z, cfg = load_golden(name)
    m = vit_som_amd.DESOM(copy.deepcopy(cfg), device="cuda")
    missing, unexpected = m.load_state_dict(golden_params(z), strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    m.set_schedule(int(z["n_train"]))
    m._it = int(z["it"]); m.iteration.fill_(int(z["it"]))
    return m, z, cfg


@pytest.mark.parametrize("name", CASES)
def test_state_dict_keys_are_the_references(name):
    m, z, _ = build(name)
    assert sorted(m.state_dict().keys()) == sorted(k[6:] for k in z.files if k.startswith("param/"))


@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_golden(name):
    m, z, cfg = build(name)
    logits, code, dist, bmu = m(torch.from_numpy(z["x"]).cuda())
    assert torch.allclose(code.cpu(), torch.from_numpy(z["fwd/z"]), atol=1e-4, rtol=0)
    assert torch.allclose(dist.cpu(), torch.from_numpy(z["fwd/dist"]), atol=1e-4, rtol=0)
    assert torch.equal(bmu.cpu(), torch.from_numpy(z["fwd/bmu"])) and bmu.dtype == torch.int64
    if cfg["data"]["num_classes"] > 0:
        assert torch.allclose(logits.cpu(), torch.from_numpy(z["fwd/logits"]), atol=1e-4, rtol=0)
    else:
        assert logits is None


@pytest.mark.parametrize("name", CASES)
def test_training_step_grads_and_adam_trajectory_match_reference_golden(name):
    m, z, cfg = build(name)
    opt = m.configure_optimizers()
    batches = [(z["x"], z["y"]), (z["x1"], z["y1"]), (z["x2"], z["y2"])]
    for s, (x, y) in enumerate(batches):
        opt.zero_grad()
        loss = m.training_step((torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()), s)
        loss.backward()
        if s == 0:
            assert abs(float(loss.detach()) - float(z["train/loss"])) < 2e-5
            assert abs(float(m.som_layer.current_temperature) - float(z["train/T"])) < 1e-6 * float(z["train/T"])
            for n, p in m.named_parameters():
                ref = torch.from_numpy(z["grad/" + n])
                assert rel_err(p.grad.cpu(), ref) < 1e-4 or float((p.grad.cpu() - ref).abs().max()) < 1e-9, n
        else:
            assert abs(float(loss.detach()) - float(z[f"train/loss{s}"])) < 2e-5
        opt.step()
        if s == 0:
            for k, v in m.state_dict().items():
                if v.is_floating_point():
                    assert torch.allclose(v.cpu(), torch.from_numpy(z["after1/" + k]), atol=2e-5), k
    for k, v in m.state_dict().items():
        if v.is_floating_point():
            assert torch.allclose(v.cpu(), torch.from_numpy(z["after3/" + k]), atol=5e-5), k
    assert int(m.iteration) == int(z["it"]) + 3
    m.eval()
    # validation on the trained weights: compare with the oracle (the golden's val/loss was taken at the same point)
    val = m.validation_step((torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["y"]).cuda()), 0)
    assert abs(float(val) - float(z["val/loss"])) < 5e-5


def test_shipped_config_shapes_against_oracle():
    """configs/desom/desom_mnist.yaml layer sizes (784-500-500-2000-10, 8x8 manhattan map, batch 128):
    fused step vs the CPU oracle."""
    import vit_som_amd
    from oracle import desom_oracle as D
    from oracle.gen_golden_desom import make_config
    cfg = make_config(1, 28, [500, 500, 2000, 10], (8, 8), 0, 128)
    torch.manual_seed(0)
    m = vit_som_amd.DESOM(copy.deepcopy(cfg), device="cuda")
    m.set_schedule(60000)
    m._it = 40
    g = torch.Generator().manual_seed(3)
    x = torch.rand(128, 1, 28, 28, generator=g)
    y = torch.zeros(128, dtype=torch.int64)
    P = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    Q = {k: (v.requires_grad_(True) if k in dict(m.named_parameters()) else v) for k, v in P.items()}
    T = D.temperature(cfg, 40, 60000)
    total_ref, parts = D.training_loss(Q, cfg, x, y, T)
    total_ref.backward()
    loss = m.train_step_fused(x.cuda(), y.cuda())
    assert abs(float(loss) - float(total_ref)) < 1e-5
    assert torch.equal(m._ctx[2].bmu.cpu(), parts["bmu"])
    for n, _ in m.named_parameters():
        gref = Q[n].grad
        assert rel_err(m._grad_views[n].cpu(), gref) < 1e-4 or float((m._grad_views[n].cpu() - gref).abs().max()) < 1e-10, n


def test_flowers17_config_shapes_against_oracle():
    """configs/desom/desom_flowers17.yaml: 3x224x224 inputs (150 528 features, a 75 M-parameter first layer),
    17 classes (CE + gamma (SOM + recon)); batch reduced to 16 so the CPU oracle stays in seconds."""
    import vit_som_amd
    from oracle import desom_oracle as D
    from oracle.gen_golden_desom import make_config
    cfg = make_config(3, 224, [500, 500, 2000, 10], (8, 8), 17, 16)
    torch.manual_seed(1)
    m = vit_som_amd.DESOM(copy.deepcopy(cfg), device="cuda")
    m.set_schedule(1360)
    m._it = 7
    g = torch.Generator().manual_seed(4)
    x = torch.rand(16, 3, 224, 224, generator=g)
    y = torch.randint(0, 17, (16,), generator=g)
    P = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    names = dict(m.named_parameters())
    Q = {k: (v.requires_grad_(True) if k in names else v) for k, v in P.items()}
    total_ref, parts = D.training_loss(Q, cfg, x, y, D.temperature(cfg, 7, 1360))
    total_ref.backward()
    loss = m.train_step_fused(x.cuda(), y.cuda())
    assert abs(float(loss) - float(total_ref.detach())) < 2e-5
    assert torch.equal(m._ctx[2].bmu.cpu(), parts["bmu"])
    for n in names:
        gref = Q[n].grad
        got = m._grad_views[n].cpu()
        assert rel_err(got, gref) < 1e-4 or float((got - gref).abs().max()) < 1e-10, n


def test_rejects_what_has_no_kernel():
    import vit_som_amd
    _, cfg = load_golden("ref_desom_tiny")
    bad = copy.deepcopy(cfg); bad["hyperparameters"]["ae"]["batch_norm"] = True
    with pytest.raises(NotImplementedError):
        vit_som_amd.DESOM(bad, device="cuda")
    bad = copy.deepcopy(cfg); bad["hyperparameters"]["optimizer"]["type"] = "adamw"
    with pytest.raises(NotImplementedError):
        vit_som_amd.DESOM(bad, device="cuda").configure_optimizers()
    m = vit_som_amd.DESOM(copy.deepcopy(cfg), device="cuda")
    with pytest.raises(ValueError):
        m(torch.zeros(2, 1, 8, 8))                      # host tensor: there is no CPU path
