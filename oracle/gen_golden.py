"""Generate golden fixtures FROM THE REFERENCE ITSELF (build container only).

Imports the reference's unmodified hot-path modules from their read-only location
(/root/reference/models/{vit_som,vit,som_layer}.py, tools/utils.py), runs them on CPU at
float32 matmul precision 'highest' (the reference's constructor sets 'medium',
models/vit_som.py:23 -- reset here) on tiny seeded configs, and writes ONLY
arrays (inputs, parameters, outputs, gradients) to tests/golden/*.npz.  No reference source or
bytecode is copied anywhere.

Packages the reference imports that are absent from this image (pytorch_lightning, timm,
torchvision, tkinter, and the off-path tools.evaluation deps) are replaced by the inert
stand-ins of SURVEY.md Appendix A.  None of them carries arithmetic except timm's
``PatchEmbed`` (8 lines: Conv2d k=s=p -> flatten(2).transpose(1,2)); that boundary is
"parity unpinned" (DESIGN.md).

Usage (container only; /root/reference does not exist on the GPU box):
    python oracle/gen_golden.py            # rewrites tests/golden/ref_*.npz
"""
import copy
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_stand_ins():
    sys.dont_write_bytecode = True

    def mod(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k): pass
        def log(self, *a, **k): pass
        def log_dict(self, *a, **k): pass
        @property
        def device(self): return next(self.parameters()).device

    mod("pytorch_lightning", LightningModule=LightningModule, seed_everything=torch.manual_seed)

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
            super().__init__()
            self.patch_size = (patch_size, patch_size)
            self.num_patches = (img_size // patch_size) ** 2
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=True)
        def forward(self, x): return self.proj(x).flatten(2).transpose(1, 2)

    mod("timm"); mod("timm.models")
    mod("timm.models.vision_transformer", PatchEmbed=PatchEmbed, Block=object)
    mod("tkinter", Y="y")
    mod("torchvision", transforms=mod("torchvision.transforms", RandomResizedCrop=object,
                                      functional=mod("torchvision.transforms.functional")))
    mod("tools.evaluation", evaluate_clustering=None, evaluate_classification=None)
    sys.path.insert(0, REF)


class _Trainer:
    class _DL:
        def __init__(self, n): self.dataset = range(n)
    def __init__(self, n_train, est):
        self.train_dataloader = self._DL(n_train)
        self.estimated_stepping_batches = est


def make_config(C, img, p, E, depth, heads, DE, ddepth, map_size, num_classes, batch_size,
                distance="cosine", topology="square", gamma=0.01, Tmax=4.0, Tmin=0.1,
                total_epochs=10, lr=5e-4, mlp_ratio=4):
    return {
        "hyperparameters": {
            "model_arch": "vit_som", "total_epochs": total_epochs, "batch_size": batch_size, "gamma": gamma,
            "som": {"map_size": list(map_size), "Tmax": Tmax, "Tmin": Tmin, "distance_fcn": distance,
                    "topology": topology, "use_reduced": False},
            "vit": {"patch_size": p, "emb_dim": E, "depth": depth, "dec_emb_dim": DE, "dec_depth": ddepth,
                    "heads": heads, "mlp_ratio": mlp_ratio, "qkv_bias": True, "qk_norm": False, "proj_drop": 0,
                    "attn_drop": 0, "drop_path": 0.1, "global_pool": False},
            "optimizer": {"type": "adamw", "lr": lr, "min_lr": 1e-6, "beta_1": 0.9, "beta_2": 0.999,
                          "scheduler": "cosine_annealing", "warmup_epochs": 2, "weight_decay": 0.05,
                          "layer_decay": 0.75, "smoothing": 0.1},
        },
        "data": {"dataset": "synthetic", "num_classes": num_classes, "num_channels": C, "input_size": img,
                 "num_workers": 0},
    }


CASES = {
    # clustering (L1 recon + SOM), 1 channel, non-square map
    "ref_cluster_tiny": dict(cfg=make_config(1, 8, 2, 16, 2, 2, 8, 1, (3, 5), 0, 6), B=6, it=7, n_train=60, est=40),
    # classification (CE label-smoothing + SOM), 3 channels, heads=3
    "ref_cls_tiny": dict(cfg=make_config(3, 8, 4, 24, 2, 3, 12, 1, (4, 4), 5, 5), B=5, it=3, n_train=50, est=30),
    # MNIST-family awkward shapes: E=16, heads 2 (hd 8), decoder DE=4 (hd 2), p=2 on 12x12 -> N=37
    "ref_mnistlike_tiny": dict(cfg=make_config(1, 12, 2, 16, 2, 2, 4, 2, (5, 4), 0, 4, gamma=0.005, Tmax=20.0,
                                               Tmin=0.001), B=4, it=11, n_train=40, est=20),
    # the other SOMLayer variants: hexa topology + euclidean distance
    "ref_hexa_euclid_tiny": dict(cfg=make_config(1, 8, 2, 16, 1, 2, 8, 1, (4, 3), 0, 6, distance="euclidean",
                                                 topology="hexa"), B=6, it=5, n_train=60, est=40),
    # manhattan distance (torch.cdist p=1, som_layer.py:115-116; the DESOM configs' distance)
    "ref_manhattan_tiny": dict(cfg=make_config(1, 8, 2, 16, 1, 2, 8, 1, (3, 4), 0, 6, distance="manhattan"),
                               B=5, it=4, n_train=60, est=40),
}


def run_case(name, spec):
    from models.vit_som import ViTSOM
    cfg = copy.deepcopy(spec["cfg"])
    torch.manual_seed(0)
    m = ViTSOM(cfg)
    torch.set_float32_matmul_precision("highest")
    tr = _Trainer(spec["n_train"], spec["est"])
    object.__setattr__(m, "trainer", tr)
    object.__setattr__(m.som_layer, "trainer", tr)
    m.iteration.fill_(spec["it"])
    # perturb LayerNorm affine + biases so those gradients/paths are non-trivial
    g = torch.Generator().manual_seed(123)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if p_.requires_grad and p_.ndim == 1:
                p_.add_(0.1 * torch.randn(p_.shape, generator=g))
    d = cfg["data"]
    x = torch.randn(spec["B"], d["num_channels"], d["input_size"], d["input_size"], generator=g)
    y = torch.randint(0, max(d["num_classes"], 1), (spec["B"],), generator=g)

    out = {}
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k, v in sd0.items():
        out["param/" + k] = v.numpy()
    out["x"], out["y"] = x.numpy(), y.numpy()
    out["iteration"] = np.int64(spec["it"]); out["n_train"] = np.int64(spec["n_train"]); out["est_steps"] = np.int64(spec["est"])

    m.eval()
    with torch.no_grad():
        cls, recon, logits, dist, bmu = m(x)
    out["fwd/cls"], out["fwd/recon"], out["fwd/dist"], out["fwd/bmu"] = (t.clone().numpy() for t in (cls, recon, dist, bmu))
    if logits is not None:
        out["fwd/logits"] = logits.numpy()
    srt = torch.sort(dist, dim=1).values
    out["fwd/top2_gap"] = (srt[:, 1] - srt[:, 0]).numpy()

    # one training step through the reference's own configure_optimizers()
    m.train()
    (opt,), (sched,) = m.configure_optimizers()
    opt.zero_grad()
    loss = m.training_step((x, y), 0)
    loss.backward()
    out["train/loss"] = loss.detach().numpy()
    out["train/T"] = np.float64(float(m.som_layer.current_temperature))
    h = m.som_layer.compute_weights(bmu)
    out["train/h"] = h.detach().clone().numpy()
    out["train/som_loss"] = m.som_layer.som_loss(h, dist).detach().numpy()
    for n_, p_ in m.named_parameters():
        if p_.requires_grad:
            out["grad/" + n_] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).detach().clone().numpy()
            out["gradnone/" + n_] = np.bool_(p_.grad is None)
    out["opt/lr"] = np.float64(opt.param_groups[0]["lr"])
    out["opt/wds"] = np.array(sorted({float(gp["weight_decay"]) for gp in opt.param_groups}))
    opt.step()
    for k, v in m.state_dict().items():
        out["after1/" + k] = v.detach().clone().numpy()
    # two more steps on fresh batches (trajectory check)
    for s in range(2):
        xs = torch.randn(x.shape, generator=g)
        ys = torch.randint(0, max(d["num_classes"], 1), (spec["B"],), generator=g)
        out[f"x{s + 1}"], out[f"y{s + 1}"] = xs.numpy(), ys.numpy()
        opt.zero_grad()
        l2 = m.training_step((xs, ys), s + 1)
        l2.backward()
        opt.step()
        out[f"train/loss{s + 1}"] = l2.detach().numpy()
    for k, v in m.state_dict().items():
        out["after3/" + k] = v.detach().clone().numpy()
    # validation step (full gamma, current temperature)
    m.eval()
    with torch.no_grad():
        out["val/loss"] = m.validation_step((x, y), 0).numpy()
    # known answers
    out["ka/index_to_position_10"] = m.som_layer.index_to_position(torch.tensor([10])).numpy()
    out["ka/pos_embed_0_2_2"] = m.vit.pos_embed[0, 2, :2].numpy()
    import json
    out["config_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "loss", float(loss), "bmu", bmu.tolist(), "min gap", float(out["fwd/top2_gap"].min()))


def run_decoder_case(name, spec):
    """Attribute surface other reference code reaches into: ViTAutoencoder.forward_decoder on an arbitrary
    token tensor (tools/evaluation.py:209-222 decode_prototype; the reference's return_attn=True branch, vit.py:190-193,
    is the one that runs), forward_features, and the attention maps of forward(return_attns=True) (vit.py:214-218,238-239)."""
    from models.vit_som import ViTSOM
    cfg = copy.deepcopy(spec["cfg"])
    torch.manual_seed(0)
    m = ViTSOM(cfg)
    torch.set_float32_matmul_precision("highest")
    g = torch.Generator().manual_seed(321)
    with torch.no_grad():
        for n_, p_ in m.named_parameters():
            if p_.requires_grad and p_.ndim == 1:
                p_.add_(0.1 * torch.randn(p_.shape, generator=g))
    d = cfg["data"]
    vit = m.vit
    n, E = vit.patch_embed.num_patches, cfg["hyperparameters"]["vit"]["emb_dim"]
    x = torch.randn(3, d["num_channels"], d["input_size"], d["input_size"], generator=g)
    tokens = torch.randn(2, n + 1, E, generator=g)
    tokens[:, 0] = 0.0                                  # decode_prototype's zero CLS placeholder
    out = {"param/" + k: v.detach().clone().numpy() for k, v in m.state_dict().items()}
    out["x"], out["tokens"] = x.numpy(), tokens.numpy()
    m.eval()
    with torch.no_grad():
        patches, attns = vit.forward_decoder(tokens, return_attn=True)
        out["dec/patches"] = patches.numpy()
        out["dec/recon"] = vit.unpatchify(patches).numpy()
        for i, a_ in enumerate(attns):
            out[f"dec/attn{i}"] = a_.numpy()
        cls, patch_tokens, recon, eattns = vit(x, return_attns=True)
        out["fwd/cls"], out["fwd/patches"], out["fwd/recon"] = cls.numpy(), patch_tokens.numpy(), recon.numpy()
        for i, a_ in enumerate(eattns):
            out[f"fwd/attn{i}"] = a_.numpy()
        fcls, fattns = vit.forward_features(x, return_attns=True)
        out["ff/cls"] = fcls.numpy()
        out["ff/n_attn"] = np.int64(len(fattns))
    import json
    out["config_json"] = np.array(json.dumps(cfg))
    np.savez_compressed(os.path.join(OUT, "ref_decoder_" + name[4:] + ".npz"), **out)
    print("ref_decoder_" + name[4:], "patches", tuple(patches.shape), "attn maps", len(attns), "+", len(eattns))


DECODER_CASES = ["ref_cluster_tiny", "ref_mnistlike_tiny", "ref_cls_tiny"]


def main():
    _install_stand_ins()
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]                      # optional case names: regenerate just those fixtures
    for name, spec in CASES.items():
        if not only or name in only:
            run_case(name, spec)
    for name in DECODER_CASES:
        if not only or ("ref_decoder_" + name[4:]) in only:
            run_decoder_case(name, CASES[name])
    if only and "ref_lr_schedule" not in only:
        return
    # scheduler known answers from the reference's own LambdaLR lambda (vit_som.py:160)
    from models.vit_som import ViTSOM
    cfg = copy.deepcopy(CASES["ref_cls_tiny"]["cfg"])
    m = ViTSOM(cfg)
    torch.set_float32_matmul_precision("highest")
    (opt,), (sched,) = m.configure_optimizers()
    lrs = []
    for e in range(cfg["hyperparameters"]["total_epochs"]):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step(); sched.step()
    np.savez_compressed(os.path.join(OUT, "ref_lr_schedule.npz"), lrs=np.array(lrs),
                        n_groups=np.int64(len(opt.param_groups)))
    print("lr schedule", lrs[:4], "groups", len(opt.param_groups))


if __name__ == "__main__":
    main()
