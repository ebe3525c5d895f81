"""Golden fixtures for the DESOM client, generated FROM THE REFERENCE ITSELF (build container only).

Imports the unmodified /root/reference/models/{desom,ae,som_layer}.py with the inert stand-ins of
oracle/gen_golden.py (plus a no-op `torchvision.utils.make_grid` and TensorBoard logger, which
desom.py calls inside training_step), runs them on CPU at float32 matmul precision 'highest' on tiny
seeded configs and writes ONLY arrays to tests/golden/ref_desom_*.npz.

Usage (container only):  python oracle/gen_golden_desom.py
"""
import copy
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gen_golden as G  # noqa: E402


def make_config(C, S, encoder_dims, map_size, num_classes, batch_size, distance="manhattan", topology="square",
                gamma=0.001, Tmax=8.0, Tmin=0.1, total_epochs=10, lr=1e-3):
    """Same keys as configs/desom/*.yaml."""
    return {
        "hyperparameters": {
            "model_arch": "desom", "total_epochs": total_epochs, "batch_size": batch_size, "gamma": gamma,
            "som": {"map_size": list(map_size), "lr_max": 0.001, "lr_min": 0.001, "Tmax": Tmax, "Tmin": Tmin,
                    "distance_fcn": distance, "topology": topology},
            "ae": {"encoder_dims": list(encoder_dims), "act": "relu", "batch_norm": False},
            "optimizer": {"type": "adam", "lr": lr, "beta_1": 0.9, "beta_2": 0.999},
        },
        "data": {"dataset": "synthetic", "num_classes": num_classes, "num_channels": C, "input_size": S, "num_workers": 0},
    }


CASES = {
    # clustering (the mnist / fmnist / usps configs): L1 recon + gamma * SOM, manhattan distance
    "ref_desom_tiny": dict(cfg=make_config(1, 8, [24, 16, 6], (3, 4), 0, 7, gamma=0.05), B=7, it=5, n_train=70),
    # classification (the flowers-17 config): CE + gamma * (SOM + recon), 3 channels
    "ref_desom_cls_tiny": dict(cfg=make_config(3, 4, [20, 12, 5], (4, 4), 4, 6, gamma=0.05), B=6, it=2, n_train=60),
    # the other distances through the same client
    "ref_desom_euclid_tiny": dict(cfg=make_config(1, 6, [16, 8], (2, 5), 0, 5, distance="euclidean", topology="hexa",
                                                  gamma=0.05), B=5, it=3, n_train=50),
}


class _Logger:
    class _Exp:
        def add_image(self, *a, **k): pass
    experiment = _Exp()


def run_case(name, spec):
    from models.desom import DESOM
    cfg = copy.deepcopy(spec["cfg"])
    torch.manual_seed(0)
    m = DESOM(cfg)
    torch.set_float32_matmul_precision("highest")
    tr = G._Trainer(spec["n_train"], 0)
    for mod_ in (m, m.som_layer):
        object.__setattr__(mod_, "trainer", tr)
    object.__setattr__(m, "logger", _Logger())
    d = cfg["data"]
    g = torch.Generator().manual_seed(1)
    B = spec["B"]
    x = torch.rand(B, d["num_channels"], d["input_size"], d["input_size"], generator=g)
    y = torch.randint(0, max(d["num_classes"], 1), (B,), generator=g)
    m.iteration.fill_(spec["it"])
    with torch.no_grad():            # spread the latent codes over the map (fresh xavier codes all land on one unit)
        m.autoencoder.encoder[-1].weight.mul_(8.0)
        m.autoencoder.encoder[-1].bias.add_(0.5)
    out = {"x": x.numpy(), "y": y.numpy(), "it": np.int64(spec["it"]), "n_train": np.int64(spec["n_train"])}
    for k, v in m.state_dict().items():
        out["param/" + k] = v.detach().clone().numpy()
    m.train()
    with torch.no_grad():
        logits, z, dist, bmu = m(x.view(B, -1))
    out["fwd/z"], out["fwd/dist"], out["fwd/bmu"] = z.numpy(), dist.numpy(), bmu.numpy()
    if logits is not None:
        out["fwd/logits"] = logits.numpy()
    srt = dist.sort(dim=1).values
    out["fwd/top2_gap"] = (srt[:, 1] - srt[:, 0]).numpy()
    opt = m.configure_optimizers()
    opt.zero_grad()
    loss = m.training_step((x, y), 0)
    loss.backward()
    out["train/loss"] = loss.detach().numpy()
    out["train/T"] = np.float64(float(m.som_layer.current_temperature))
    for n_, p_ in m.named_parameters():
        out["grad/" + n_] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).detach().clone().numpy()
    opt.step()
    for k, v in m.state_dict().items():
        out["after1/" + k] = v.detach().clone().numpy()
    for s in range(2):
        xs = torch.rand(x.shape, generator=g)
        ys = torch.randint(0, max(d["num_classes"], 1), (B,), generator=g)
        out[f"x{s + 1}"], out[f"y{s + 1}"] = xs.numpy(), ys.numpy()
        opt.zero_grad()
        l2 = m.training_step((xs, ys), s + 1)
        l2.backward()
        opt.step()
        out[f"train/loss{s + 1}"] = l2.detach().numpy()
    for k, v in m.state_dict().items():
        out["after3/" + k] = v.detach().clone().numpy()
    m.eval()
    with torch.no_grad():
        out["val/loss"] = m.validation_step((x, y), 0).numpy()
    out["config_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(G.OUT, name + ".npz"), **out)
    print(name, "loss", float(loss), "bmu", bmu.tolist(), "min gap", float(out["fwd/top2_gap"].min()))


def main():
    G._install_stand_ins()
    import torchvision
    import types
    torchvision.utils = types.ModuleType("torchvision.utils")
    torchvision.utils.make_grid = lambda t, *a, **k: t
    sys.modules["torchvision.utils"] = torchvision.utils
    ev = sys.modules["tools.evaluation"]
    for name in ("evaluate_kmeans", "visualize_decoded_prototypes", "visualize_label_heatmap"):
        setattr(ev, name, None)
    only = sys.argv[1:]
    for name, spec in CASES.items():
        if not only or name in only:
            run_case(name, spec)


if __name__ == "__main__":
    main()
