"""CPU oracle for the DESOM client of the SOM kernels: a plain-torch restatement
of /root/reference/models/desom.py + models/ae.py (fully-connected symmetric autoencoder, SOM layer
on the latent code, optional linear classifier).  TEST INFRASTRUCTURE ONLY: imported by tests/
and never by the product package.  Pinned by tests/golden/ref_desom_*.npz, which
oracle/gen_golden_desom.py produced from the reference itself.

Parameters are a flat dict keyed exactly like the reference's state_dict:
    autoencoder.encoder.{0,2,4,...}.{weight,bias}   (nn.Sequential of Linear, act, Linear, ... ae.py:44-63)
    autoencoder.decoder.{0,2,4,...}.{weight,bias}
    som_layer.prototypes                             (som_layer.py:44-56)
    classifier.{weight,bias}                         (desom.py:42, when num_classes > 0)
"""
import torch
import torch.nn.functional as F

from . import vitsom_oracle as V


def layer_dims(cfg):
    """ae.py:27-32: [C*S*S] + encoder_dims, and the reversed list for the decoder."""
    d = cfg["data"]
    enc = [d["num_channels"] * d["input_size"] * d["input_size"]] + list(cfg["hyperparameters"]["ae"]["encoder_dims"])
    return enc, list(reversed(enc))


def _mlp(params, prefix, x, n_layers, relu):
    """ae.py:44-63 with batch_norm False: Linear, act after every layer but the last."""
    for i in range(n_layers):
        x = F.linear(x, params[f"{prefix}.{2 * i}.weight"], params[f"{prefix}.{2 * i}.bias"])
        if i < n_layers - 1 and relu:
            x = torch.relu(x)
    return x


def encode(params, cfg, x_flat):
    enc, _ = layer_dims(cfg)
    return _mlp(params, "autoencoder.encoder", x_flat, len(enc) - 1, cfg["hyperparameters"]["ae"]["act"] == "relu")


def decode(params, cfg, z):
    _, dec = layer_dims(cfg)
    return _mlp(params, "autoencoder.decoder", z, len(dec) - 1, cfg["hyperparameters"]["ae"]["act"] == "relu")


def forward(params, cfg, x):
    """desom.py:52-56 -> (cls_logits | None, x_encoded, distances, bmu_indices)."""
    x_flat = x.reshape(x.shape[0], -1)
    z = encode(params, cfg, x_flat)
    dist = V.som_distances(z, params["som_layer.prototypes"], cfg["hyperparameters"]["som"]["distance_fcn"])
    bmu = dist.argmin(dim=1)
    logits = F.linear(z, params["classifier.weight"], params["classifier.bias"]) if cfg["data"]["num_classes"] > 0 else None
    return logits, z, dist, bmu


def training_loss(params, cfg, x, y, T):
    """desom.py:58-74,135-160: total = recon + gamma*som (clustering) or CE + gamma*(som + recon)."""
    hp = cfg["hyperparameters"]
    logits, z, dist, bmu = forward(params, cfg, x)
    grid = V.grid_positions(hp["som"]["map_size"], hp["som"]["topology"]).to(dist.dtype)
    som = V.som_loss(V.neighbourhood(bmu, grid, T), dist)
    x_flat = x.reshape(x.shape[0], -1)
    recon = (decode(params, cfg, z) - x_flat).abs().mean()
    if cfg["data"]["num_classes"] > 0:
        total = F.cross_entropy(logits, y) + hp["gamma"] * (som + recon)
    else:
        total = recon + hp["gamma"] * som
    return total, {"som": som, "recon": recon, "dist": dist, "bmu": bmu, "z": z, "logits": logits}


def temperature(cfg, iteration, n_train):
    """som_layer.py:127-132 (iteration BEFORE the increment, desom.py:113-118)."""
    hp = cfg["hyperparameters"]
    total = (n_train / hp["batch_size"]) * hp["total_epochs"]
    return hp["som"]["Tmax"] * (hp["som"]["Tmin"] / hp["som"]["Tmax"]) ** (iteration / (total - 1))
