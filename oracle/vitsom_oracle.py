"""CPU oracle for the ViT-SOM training-step hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (plain torch CPU ops, fp32 or fp64) of the
arithmetic of the reference's hot path.  It is imported ONLY by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; the product
package (``vit_som_amd/``) never imports it and has no CPU fallback.

Parity pin: the restatement is checked against outputs of the reference itself, produced in
the build container by ``oracle/gen_golden.py`` (reference modules imported unmodified from
/root/reference) and committed as arrays under ``tests/golden/*.npz``
(``tests/test_oracle_golden.py``).  The only third-party arithmetic on the path that is NOT in
/root/reference is timm's ``PatchEmbed`` (version unpinned by the reference); it is restated
here as ``Conv2d(k=s=p) -> flatten(2).transpose(1,2)`` per its published behaviour and that
single boundary is "parity unpinned" (see DESIGN.md).

Every function cites the reference file:line it follows (paths relative to /root/reference).
Parameters live in a flat ``dict[str, Tensor]`` whose keys are exactly the reference's
``state_dict`` keys (checkpoint row).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# configuration helpers
# --------------------------------------------------------------------------------------
class Dims:
    """Shape bundle derived from the YAML-schema config dict (models/vit_som.py:28-52,
    models/som_layer.py:18-40)."""

    def __init__(self, config: dict):
        hp = config["hyperparameters"]
        vit, som, data = hp["vit"], hp["som"], config["data"]
        self.C = int(data["num_channels"])
        self.img = int(data["input_size"])
        self.p = int(vit["patch_size"])
        self.g = self.img // self.p                  # patches per side
        self.n = self.g * self.g                     # timm PatchEmbed.num_patches
        self.N = self.n + 1                          # tokens incl. CLS
        self.E = int(vit["emb_dim"])
        self.depth = int(vit["depth"])
        self.H = int(vit["heads"])
        self.DE = int(vit["dec_emb_dim"])
        self.ddepth = int(vit["dec_depth"])
        self.DH = int(vit["heads"])                  # decoder_num_heads=vit_hp['heads'] (vit_som.py:47)
        self.mlp = float(vit["mlp_ratio"])
        self.hidden = int(self.E * self.mlp)         # vit.py:51
        self.dhidden = int(self.DE * self.mlp)
        self.num_classes = int(data["num_classes"])
        self.classification = self.num_classes > 0   # vit_som.py:36
        self.map_size = tuple(int(v) for v in som["map_size"])
        self.K = int(np.prod(self.map_size))         # som_layer.py:33
        self.use_reduced = bool(som["use_reduced"])
        self.L = self.E if self.use_reduced else self.E * self.n   # som_layer.py:36-40
        self.distance_fcn = som["distance_fcn"]
        self.topology = som["topology"]
        self.Tmax = float(som["Tmax"])
        self.Tmin = float(som["Tmin"])
        self.gamma = float(hp["gamma"])
        self.batch_size = int(hp["batch_size"])
        self.total_epochs = int(hp["total_epochs"])
        self.smoothing = float(hp["optimizer"]["smoothing"])
        self.patch_dim = self.p * self.p * self.C


# --------------------------------------------------------------------------------------
# A12: 2-D sin-cos position table  (tools/utils.py:131-178)
# --------------------------------------------------------------------------------------
def _sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    # tools/utils.py:159-178 -- float64 throughout
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed(embed_dim: int, grid_size: int, cls_token: bool = True) -> torch.Tensor:
    """tools/utils.py:131-157.  np.meshgrid(grid_w, grid_h): grid[0] varies along the fast
    (column) axis, so the first half of the channels encodes the COLUMN index."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    emb = np.concatenate([_sincos_1d(embed_dim // 2, grid[0]), _sincos_1d(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return torch.from_numpy(emb).float().unsqueeze(0)          # vit.py:103  [1, n+1, D]


# --------------------------------------------------------------------------------------
# A6: SOM grid  (models/som_layer.py:60-81)
# --------------------------------------------------------------------------------------
def grid_positions(map_size, topology: str) -> torch.Tensor:
    rows, cols = map_size
    if topology == "square":                                   # som_layer.py:61-67
        gy, gx = torch.meshgrid(torch.arange(rows), torch.arange(cols), indexing="ij")
        return torch.stack([gy, gx], dim=-1).view(-1, 2).float()
    if topology == "hexa":                                     # som_layer.py:68-77
        pos = torch.zeros(rows * cols, 2)
        for i in range(rows * cols):
            r, c = i // cols, i % cols
            pos[i, 0] = c + (0.5 if r % 2 == 1 else 0.0)
            pos[i, 1] = r * np.sqrt(3) / 2
        return pos
    raise ValueError(f"Unsupported topology: {topology}")


def index_to_position(indices: torch.Tensor, map_size) -> torch.Tensor:
    """som_layer.py:134-135 (known answer: index 10 on an 8-column map -> (1, 2),
    experiments/tests/unit_test.py:11-16)."""
    return torch.stack((indices // map_size[1], indices % map_size[1]), dim=1).float()


# --------------------------------------------------------------------------------------
# A11: parameter construction (same distributions as the reference; own RNG stream)
# --------------------------------------------------------------------------------------
def _xavier(shape, gen, fan_in=None, fan_out=None) -> torch.Tensor:
    fo, fi = (shape[0], int(np.prod(shape[1:]))) if fan_in is None else (fan_out, fan_in)
    a = math.sqrt(6.0 / (fi + fo))
    return (torch.rand(shape, generator=gen) * 2 - 1) * a


def block_keys(prefix: str):
    return [f"{prefix}.{s}" for s in (
        "norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
        "attn.proj.bias", "norm2.weight", "norm2.bias", "mlp.0.weight", "mlp.0.bias",
        "mlp.2.weight", "mlp.2.bias")]


def init_params(config: dict, seed: int = 0, dtype=torch.float32) -> Params:
    """Random-init parameters with the reference's distributions and state-dict keys.

    vit.py:100-125: Linear -> xavier_uniform weight / zero bias; LayerNorm -> (1, 0);
    patch_embed.proj.weight -> xavier_uniform on the [E, C*p*p] view; its bias keeps torch's
    Conv2d default U(-1/sqrt(fan_in), 1/sqrt(fan_in)); cls_token -> N(0, 0.02);
    pos embeds -> frozen sincos.  som_layer.py:44-56: prototypes = rand(K, L), row-normalised
    for cosine.  vit_som.py:58-59: cls_head.weight -> N(0, 0.02), bias -> Linear default."""
    d = Dims(config)
    g = torch.Generator().manual_seed(seed)
    P: Params = {}
    P["iteration"] = torch.tensor(0)
    P["vit.cls_token"] = torch.randn(1, 1, d.E, generator=g) * 0.02
    P["vit.pos_embed"] = sincos_pos_embed(d.E, d.g)
    P["vit.decoder_pos_embed"] = sincos_pos_embed(d.DE, d.g)
    P["vit.patch_embed.proj.weight"] = _xavier((d.E, d.patch_dim), g).view(d.E, d.C, d.p, d.p)
    bnd = 1.0 / math.sqrt(d.patch_dim)
    P["vit.patch_embed.proj.bias"] = (torch.rand(d.E, generator=g) * 2 - 1) * bnd

    def block(prefix, dim, hidden):
        P[f"{prefix}.norm1.weight"] = torch.ones(dim)
        P[f"{prefix}.norm1.bias"] = torch.zeros(dim)
        P[f"{prefix}.attn.qkv.weight"] = _xavier((3 * dim, dim), g)
        P[f"{prefix}.attn.qkv.bias"] = torch.zeros(3 * dim)
        P[f"{prefix}.attn.proj.weight"] = _xavier((dim, dim), g)
        P[f"{prefix}.attn.proj.bias"] = torch.zeros(dim)
        P[f"{prefix}.norm2.weight"] = torch.ones(dim)
        P[f"{prefix}.norm2.bias"] = torch.zeros(dim)
        P[f"{prefix}.mlp.0.weight"] = _xavier((hidden, dim), g)
        P[f"{prefix}.mlp.0.bias"] = torch.zeros(hidden)
        P[f"{prefix}.mlp.2.weight"] = _xavier((dim, hidden), g)
        P[f"{prefix}.mlp.2.bias"] = torch.zeros(dim)

    for i in range(d.depth):
        block(f"vit.blocks.{i}", d.E, d.hidden)
    P["vit.norm.weight"], P["vit.norm.bias"] = torch.ones(d.E), torch.zeros(d.E)
    P["vit.decoder_embed.weight"] = _xavier((d.DE, d.E), g)
    P["vit.decoder_embed.bias"] = torch.zeros(d.DE)
    for i in range(d.ddepth):
        block(f"vit.decoder_blocks.{i}", d.DE, d.dhidden)
    P["vit.decoder_norm.weight"], P["vit.decoder_norm.bias"] = torch.ones(d.DE), torch.zeros(d.DE)
    P["vit.decoder_pred.weight"] = _xavier((d.patch_dim, d.DE), g)
    P["vit.decoder_pred.bias"] = torch.zeros(d.patch_dim)
    W = torch.rand(d.K, d.L, generator=g)
    if d.distance_fcn == "cosine":
        W = F.normalize(W, p=2, dim=1)
    P["som_layer.prototypes"] = W
    P["som_layer.grid_positions"] = grid_positions(d.map_size, d.topology)
    if d.classification:
        P["cls_head.weight"] = torch.randn(d.num_classes, d.E, generator=g) * 0.02
        bnd = 1.0 / math.sqrt(d.E)
        P["cls_head.bias"] = (torch.rand(d.num_classes, generator=g) * 2 - 1) * bnd
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in P.items()}


FROZEN = ("iteration", "vit.pos_embed", "vit.decoder_pos_embed", "som_layer.grid_positions")


def trainable_keys(P: Params):
    return [k for k in P if k not in FROZEN]


# --------------------------------------------------------------------------------------
# A4 / A3: attention and transformer block  (models/vit.py:28-63)
# --------------------------------------------------------------------------------------
def attention(P: Params, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    B, N, C = x.shape                                                         # vit.py:29
    hd = C // heads
    qkv = F.linear(x, P[f"{pre}.qkv.weight"], P[f"{pre}.qkv.bias"])          # vit.py:30
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5)                           # vit.py:33
    attn = attn.softmax(dim=-1)                                               # vit.py:34 (dropout p=0)
    out = (attn @ v).transpose(1, 2).reshape(B, N, C)                         # vit.py:37
    return F.linear(out, P[f"{pre}.proj.weight"], P[f"{pre}.proj.bias"])     # vit.py:38


def block(P: Params, pre: str, x: torch.Tensor, heads: int, eps: float = 1e-6) -> torch.Tensor:
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), P[f"{pre}.norm1.weight"], P[f"{pre}.norm1.bias"], eps)   # vit.py:60
    x = x + attention(P, f"{pre}.attn", h, heads)                                      # vit.py:61
    h = F.layer_norm(x, (C,), P[f"{pre}.norm2.weight"], P[f"{pre}.norm2.bias"], eps)
    h = F.linear(h, P[f"{pre}.mlp.0.weight"], P[f"{pre}.mlp.0.bias"])                  # vit.py:53
    h = F.gelu(h)                                                                      # exact erf GELU, vit.py:54
    h = F.linear(h, P[f"{pre}.mlp.2.weight"], P[f"{pre}.mlp.2.bias"])
    return x + h                                                                       # vit.py:62


def patchify(imgs: torch.Tensor, p: int) -> torch.Tensor:
    """vit.py:127-139: [B,C,H,W] -> [B, h*w, p*p*C] with the channel index fastest."""
    B, C, Himg, _ = imgs.shape
    h = w = Himg // p
    x = imgs.reshape(B, C, h, p, w, p)
    return torch.einsum("nchpwq->nhwpqc", x).reshape(B, h * w, p * p * C)


def unpatchify(x: torch.Tensor, p: int) -> torch.Tensor:
    """vit.py:141-153."""
    B = x.shape[0]
    h = w = int(x.shape[1] ** 0.5)
    c = x.shape[2] // (p * p)
    x = x.reshape(B, h, w, p, p, c)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(B, c, h * p, w * p)


def patch_embed(P: Params, x: torch.Tensor, p: int) -> torch.Tensor:
    """timm PatchEmbed as used at vit.py:76,207: Conv2d(C,E,k=p,s=p,bias) -> flatten(2) ->
    transpose(1,2).  (third-party, unpinned: restated from published behaviour)"""
    y = F.conv2d(x, P["vit.patch_embed.proj.weight"], P["vit.patch_embed.proj.bias"], stride=p)
    return y.flatten(2).transpose(1, 2)


def vit_forward(P: Params, x: torch.Tensor, d: Dims):
    """ViTAutoencoder.forward, vit.py:202-240 -> (cls_token_out, patch_tokens_out, recon_img)."""
    t = patch_embed(P, x, d.p)                                                # vit.py:207
    t = t + P["vit.pos_embed"][:, 1:, :]                                      # vit.py:208
    cls = P["vit.cls_token"] + P["vit.pos_embed"][:, :1, :]                   # vit.py:210
    t = torch.cat((cls.expand(t.shape[0], -1, -1), t), dim=1)                 # vit.py:211-212
    for i in range(d.depth):
        t = block(P, f"vit.blocks.{i}", t, d.H)                               # vit.py:215-218
    t = F.layer_norm(t, (d.E,), P["vit.norm.weight"], P["vit.norm.bias"], 1e-6)   # vit.py:219
    cls_out, patches = t[:, 0], t[:, 1:]                                      # vit.py:221-222
    dec = F.linear(t, P["vit.decoder_embed.weight"], P["vit.decoder_embed.bias"])   # vit.py:225
    dec = dec + P["vit.decoder_pos_embed"]                                    # vit.py:226
    for i in range(d.ddepth):
        dec = block(P, f"vit.decoder_blocks.{i}", dec, d.DH)                  # vit.py:229-230
    dec = F.layer_norm(dec, (d.DE,), P["vit.decoder_norm.weight"], P["vit.decoder_norm.bias"], 1e-6)
    pred = F.linear(dec, P["vit.decoder_pred.weight"], P["vit.decoder_pred.bias"])[:, 1:, :]  # vit.py:234
    return cls_out, patches, unpatchify(pred, d.p)                            # vit.py:236


def attention_map(P: Params, pre: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """The [B, heads, N, N] softmax probabilities Attention.forward returns with return_attn=True (vit.py:33-34,41-42);
    x is the block's LayerNorm-ed input."""
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, P[f"{pre}.qkv.weight"], P[f"{pre}.qkv.bias"]).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    return ((qkv[0] @ qkv[1].transpose(-2, -1)) * (hd ** -0.5)).softmax(dim=-1)


def _blocks_with_maps(P: Params, prefix: str, depth: int, t: torch.Tensor, heads: int):
    maps = []
    for i in range(depth):
        pre = f"{prefix}.{i}"
        h = F.layer_norm(t, (t.shape[-1],), P[f"{pre}.norm1.weight"], P[f"{pre}.norm1.bias"], 1e-6)
        maps.append(attention_map(P, f"{pre}.attn", h, heads))
        t = block(P, pre, t, heads)
    return t, maps


def vit_forward_features(P: Params, x: torch.Tensor, d: Dims):
    """ViTAutoencoder.forward_features(return_attns=True), vit.py:155-179 -> (cls_token_out, [attention maps])."""
    t = patch_embed(P, x, d.p) + P["vit.pos_embed"][:, 1:, :]
    cls = P["vit.cls_token"] + P["vit.pos_embed"][:, :1, :]
    t = torch.cat((cls.expand(t.shape[0], -1, -1), t), dim=1)
    t, maps = _blocks_with_maps(P, "vit.blocks", d.depth, t, d.H)
    t = F.layer_norm(t, (d.E,), P["vit.norm.weight"], P["vit.norm.bias"], 1e-6)
    return t[:, 0], maps


def vit_forward_decoder(P: Params, tokens: torch.Tensor, d: Dims):
    """ViTAutoencoder.forward_decoder(return_attn=True), vit.py:182-200 -> (decoded_patches [B,n,p*p*C], [maps]) for an
    arbitrary token tensor [B, n+1, E] (caller: tools/evaluation.py:209-222)."""
    dec = F.linear(tokens, P["vit.decoder_embed.weight"], P["vit.decoder_embed.bias"])      # vit.py:186
    dec = dec + P["vit.decoder_pos_embed"]                                                  # vit.py:187
    dec, maps = _blocks_with_maps(P, "vit.decoder_blocks", d.ddepth, dec, d.DH)             # vit.py:189-195
    dec = F.layer_norm(dec, (d.DE,), P["vit.decoder_norm.weight"], P["vit.decoder_norm.bias"], 1e-6)
    return F.linear(dec, P["vit.decoder_pred.weight"], P["vit.decoder_pred.bias"])[:, 1:, :], maps   # vit.py:197


# --------------------------------------------------------------------------------------
# A5-A7: SOM layer  (models/som_layer.py:83-152)
# --------------------------------------------------------------------------------------
def som_distances(x: torch.Tensor, W: torch.Tensor, fcn: str = "cosine") -> torch.Tensor:
    """som_layer.py:111-125."""
    if fcn == "cosine":
        xn = F.normalize(x, p=2, dim=1)                                       # eps 1e-12
        wn = F.normalize(W, p=2, dim=1)
        return 1 - torch.mm(xn, wn.T)                                         # som_layer.py:122
    if fcn == "euclidean":
        return torch.cdist(x, W, p=2)
    if fcn == "manhattan":
        return torch.cdist(x, W, p=1)
    raise ValueError(f"Unsupported distance function: {fcn}")


def som_forward(x: torch.Tensor, W: torch.Tensor, fcn: str = "cosine"):
    """som_layer.py:83-89: first-minimum argmin, int64."""
    if x.dim() > 2:
        x = x.flatten(start_dim=1)
    dist = som_distances(x, W, fcn)
    return dist, torch.argmin(dist, dim=1)


def temperature(iteration, Tmax: float, Tmin: float, n_train: int, batch_size: int, total_epochs: int):
    """som_layer.py:127-132 (total_iterations is a float; iteration is the value BEFORE the
    step's increment, vit_som.py:84,104)."""
    total_iterations = (n_train / batch_size) * total_epochs
    return Tmax * (Tmin / Tmax) ** (iteration / (total_iterations - 1))


def neighbourhood(bmu: torch.Tensor, grid: torch.Tensor, T) -> torch.Tensor:
    """som_layer.py:144-152: h_ik = exp(-||g_k - g_bmu(i)||^2 / (2 T^2))."""
    bpos = grid[bmu]
    dgrid = torch.norm(grid.unsqueeze(0) - bpos.unsqueeze(1), dim=2)
    return torch.exp(-dgrid ** 2 / (2 * T ** 2))


def som_loss(h: torch.Tensor, dist: torch.Tensor) -> torch.Tensor:
    """som_layer.py:137-142."""
    return torch.mean(h * dist)


# --------------------------------------------------------------------------------------
# A1 / A8 / A9: model forward and step losses  (models/vit_som.py:67-125)
# --------------------------------------------------------------------------------------
def forward(P: Params, x: torch.Tensor, d: Dims):
    """ViTSOM.forward, vit_som.py:67-78 -> (cls_token, recon_img, logits|None, distances, bmu)."""
    cls, patches, recon = vit_forward(P, x, d)
    som_in = cls if d.use_reduced else patches.flatten(start_dim=1)           # vit_som.py:70-73
    dist, bmu = som_forward(som_in, P["som_layer.prototypes"], d.distance_fcn)
    logits = F.linear(cls, P["cls_head.weight"], P["cls_head.bias"]) if d.classification else None
    return cls, recon, logits, dist, bmu


def gamma_ramp(gamma: float, iteration: int, estimated_stepping_batches: int) -> float:
    """vit_som.py:89-90."""
    ramp_up_end_step = estimated_stepping_batches // 2
    return gamma * min(1.0, iteration / ramp_up_end_step)


def training_loss(P: Params, x, y, d: Dims, iteration: int, n_train: int, est_steps: int):
    """ViTSOM.training_step, vit_som.py:80-105.  Returns (total, dict of parts)."""
    cls, recon, logits, dist, bmu = forward(P, x, d)
    T = temperature(iteration, d.Tmax, d.Tmin, n_train, d.batch_size, d.total_epochs)   # :84
    h = neighbourhood(bmu, P["som_layer.grid_positions"], T)                            # :85
    ls = som_loss(h, dist)                                                              # :86
    g = gamma_ramp(d.gamma, iteration, est_steps)
    if d.classification:
        main = F.cross_entropy(logits, y.view(-1), label_smoothing=d.smoothing)         # :96
    else:
        main = F.l1_loss(recon, x)                                                      # :100
    total = main + g * ls
    return total, {"main": main, "som": ls, "gamma_t": g, "T": T, "bmu": bmu, "dist": dist,
                   "cls": cls, "recon": recon, "logits": logits, "h": h}


def validation_loss(P: Params, x, y, d: Dims, T: float):
    """ViTSOM.validation_step, vit_som.py:107-125 (full gamma, current temperature)."""
    cls, recon, logits, dist, bmu = forward(P, x, d)
    h = neighbourhood(bmu, P["som_layer.grid_positions"], T)
    ls = som_loss(h, dist)
    if d.classification:
        main = F.cross_entropy(logits, y.view(-1), label_smoothing=d.smoothing)
        acc = (logits.argmax(dim=-1) == y.view(-1)).float().mean()
    else:
        main, acc = F.l1_loss(recon, x), None
    return main + d.gamma * ls, {"main": main, "som": ls, "acc": acc, "bmu": bmu}


def loss_and_grads(P: Params, x, y, d: Dims, iteration: int, n_train: int, est_steps: int):
    """Autograd of training_loss w.r.t. every trainable parameter (what Lightning's
    loss.backward() produces).  Parameters unused by the loss (the decoder in
    classification mode) get a zero gradient here."""
    Q = {k: (v.detach().clone().requires_grad_(True) if k not in FROZEN else v) for k, v in P.items()}
    total, parts = training_loss(Q, x, y, d, iteration, n_train, est_steps)
    keys = trainable_keys(Q)
    grads = torch.autograd.grad(total, [Q[k] for k in keys], allow_unused=True)
    G = {k: (g if g is not None else torch.zeros_like(Q[k])) for k, g in zip(keys, grads)}
    return total.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in parts.items()}, G


# --------------------------------------------------------------------------------------
# A10: optimiser semantics  (models/vit_som.py:127-163, tools/utils.py:28-84)
# --------------------------------------------------------------------------------------
def weight_decay_of(key: str, P: Params, wd: float) -> float:
    """ViT params: wd for ndim>=2, 0 for 1-D (utils.py:44-49; note cls_token is 3-D -> wd).
    prototypes / cls_head.{weight,bias}: the extra group has no weight_decay key ->
    AdamW default 0.01 (vit_som.py:140-144)."""
    if key.startswith("vit."):
        return 0.0 if P[key].ndim == 1 else wd
    return 0.01


def lr_multiplier(epoch: int, min_lr: float, warmup_epochs: float, total_epochs: int) -> float:
    """vit_som.py:160 -- min_lr is a multiplier floor; layer-wise lr_scale is inert."""
    return max(min_lr, min((epoch + 1) / (warmup_epochs + 1e-8),
                           0.5 * (math.cos(epoch / total_epochs * math.pi) + 1)))


def base_lr(config: dict) -> float:
    hp = config["hyperparameters"]
    return hp["optimizer"]["lr"] * hp["batch_size"] / 256                     # vit_som.py:149


def make_torch_optimizer(P: Params, config: dict, epoch: int = 0):
    """torch.optim.AdamW over leaf copies of P with the reference's effective groups."""
    opt = config["hyperparameters"]["optimizer"]
    keys = trainable_keys(P)
    leaves = {k: P[k].detach().clone().requires_grad_(True) for k in keys}
    groups: Dict[float, list] = {}
    for k in keys:
        groups.setdefault(weight_decay_of(k, P, opt["weight_decay"]), []).append(leaves[k])
    lr = base_lr(config) * lr_multiplier(epoch, opt["min_lr"], opt["warmup_epochs"],
                                          config["hyperparameters"]["total_epochs"])
    cls = torch.optim.AdamW if opt["type"] == "adamw" else torch.optim.Adam
    o = cls([{"params": v, "weight_decay": wd} for wd, v in groups.items()], lr=lr,
            betas=(opt["beta_1"], opt["beta_2"]))
    return o, leaves


def adamw_reference(p, g, m, v, step: int, lr: float, b1: float, b2: float, eps: float, wd: float):
    """One decoupled AdamW update, written out (torch.optim.AdamW single-tensor semantics)."""
    p = p * (1 - lr * wd)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


# --------------------------------------------------------------------------------------
# synthetic inputs and the CPU-baseline step
# --------------------------------------------------------------------------------------
def synthetic_batch(d: Dims, B: int, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, d.C, d.img, d.img, generator=g)
    y = torch.randint(0, max(d.num_classes, 1), (B,), generator=g)
    return x, y


class CPUStep:
    """fwd + bwd + AdamW on the host cores: the ``cpu_baseline`` leg of bench.py."""

    def __init__(self, config: dict, seed: int = 0):
        self.config, self.d = config, Dims(config)
        P = init_params(config, seed)
        self.opt, self.leaves = make_torch_optimizer(P, config)
        self.P = dict(P)
        self.P.update(self.leaves)
        self.iteration = 0

    def step(self, x, y, n_train: int, est_steps: int) -> float:
        self.opt.zero_grad(set_to_none=True)
        total, _ = training_loss(self.P, x, y, self.d, self.iteration, n_train, est_steps)
        total.backward()
        self.opt.step()
        self.iteration += 1
        return float(total.detach())
