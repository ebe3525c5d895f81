"""Host-side mirror of the reference's DESOM model (models/desom.py:13-174, models/ae.py:9-63): a
fully-connected symmetric autoencoder, the SOM layer on its latent code and an optional linear
classifier -- the second client of the SOM kernels.  Same constructor (the
YAML-schema ``config`` dict of configs/desom/*.yaml), method names, return tuples and state_dict
keys as the reference; all arithmetic runs in libvitsom_hip.so through ``ops``; no CPU path.

  DESOM        forward / training_step / validation_step / configure_optimizers / update
  Autoencoder  encoder / decoder as nn.Sequential of Linear (+ ReLU) so that the keys are
               ``autoencoder.encoder.0.weight`` ... exactly as ae.py:44-63 builds them
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import ops
from .arena import ParamArena
from .model import _HAVE_PL, FusedAdamW, SOMLayer, _Acts, _ArenaOwner, _Base, _StepLoss


class Autoencoder(nn.Module):
    """models/ae.py:9-63 (fully-connected, symmetric, xavier-uniform weights, nn.Linear default biases)."""

    def __init__(self, config):
        super().__init__()
        ae, d = config["hyperparameters"]["ae"], config["data"]
        if ae["batch_norm"]:
            raise NotImplementedError("ae.batch_norm=True has no HIP kernel (every shipped DESOM config sets it False)")
        self.relu = ae["act"] == "relu"                                           # ae.py:24: anything else is Identity
        input_dim = d["num_channels"] * d["input_size"] * d["input_size"]
        self.encoder_dims = [input_dim] + list(ae["encoder_dims"])                # ae.py:27-28
        self.encoder = self._build(self.encoder_dims)
        self.decoder = self._build(list(reversed(self.encoder_dims)))

    def _build(self, dims: List[int]) -> nn.Sequential:
        layers: List[nn.Module] = []
        n = len(dims) - 1
        for i in range(n):
            lin = nn.Linear(dims[i], dims[i + 1])
            nn.init.xavier_uniform_(lin.weight)
            layers.append(lin)
            if i < n - 1:
                layers.append(nn.ReLU() if self.relu else nn.Identity())
            elif dims is not self.encoder_dims:
                layers.append(nn.Identity())                                      # decoder output_act, ae.py:58-59
        return nn.Sequential(*layers)

    @staticmethod
    def linears(seq: nn.Sequential):
        return [(str(i), m) for i, m in enumerate(seq) if isinstance(m, nn.Linear)]


class DESOM(_ArenaOwner, _Base):
    """Deep Embedded Self-Organizing Map (models/desom.py:13-174), MI355X-native."""

    def __init__(self, config, device=None):
        super().__init__()
        self.config = config
        if _HAVE_PL:
            self.save_hyperparameters(config)
        hp, d = config["hyperparameters"], config["data"]
        self.total_epochs, self.batch_size, self.gamma = hp["total_epochs"], hp["batch_size"], hp["gamma"]
        self.encoder_dims = hp["ae"]["encoder_dims"]
        o = hp["optimizer"]
        self.opt_type, self.opt_lr, self.beta_1, self.beta_2 = o["type"], o["lr"], o["beta_1"], o["beta_2"]
        self.num_classes = d["num_classes"]
        self.classification = self.num_classes > 0
        self.autoencoder = Autoencoder(config)
        self.som_layer = SOMLayer(config)
        self.classifier = nn.Linear(self.encoder_dims[-1], self.num_classes) if self.classification else nn.Identity()
        self.register_buffer("iteration", torch.tensor(0))
        self._it = 0
        self._last: Dict[str, torch.Tensor] = {}
        self._bufs: Optional[_Acts] = None
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self._pack(torch.device(device))

    # FusedAdamW protocol (the reference's Adam has no weight decay: _default_weight_decay stays 0)
    def _decoder_param_names(self):
        return []

    def set_schedule(self, n_train: int, estimated_stepping_batches: int = 0):
        """Trainer-less replacement for len(trainer.train_dataloader.dataset) (som_layer.py:131)."""
        self.som_layer._n_train = int(n_train)

    # -- buffers ------------------------------------------------------------------------------
    def _buffers_for(self, B: int, dev) -> _Acts:
        a = self._bufs
        if a is not None and a.B == B and a.device == dev:
            return a
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)   # noqa: E731
        a = _Acts()
        a.B, a.device = B, dev
        enc, dec = self.autoencoder.encoder_dims, list(reversed(self.autoencoder.encoder_dims))
        # per layer: output activation and (for hidden layers) the activation derivative
        a.enc_act = [f(B, n) for n in enc[1:]]
        a.enc_der = [f(B, n) for n in enc[1:-1]]
        a.dec_act = [f(B, n) for n in dec[1:]]
        a.dec_der = [f(B, n) for n in dec[1:-1]]
        widest = max(enc)
        a.ga, a.gb = f(B * widest), f(B * widest)          # ping-pong gradient buffers
        a.dz = f(B, enc[-1])
        a.dpred = f(B, enc[0])
        a.recon_sum, a.ce_sum = f(1), f(1)
        if self.classification:
            a.logits, a.dlogits = f(B, self.num_classes), f(B, self.num_classes)
        self._bufs = a
        return a

    def _check_input(self, x):
        if not x.is_cuda:
            raise ValueError("DESOM: input must live on the GPU (there is no CPU path)")
        x = x.reshape(x.shape[0], -1)
        if x.dtype != torch.float32:
            x = x.float()
        if x.shape[1] != self.autoencoder.encoder_dims[0]:
            raise ValueError(f"DESOM: expected {self.autoencoder.encoder_dims[0]} input features, got {x.shape[1]}")
        return x.contiguous()

    def _mlp_fwd(self, seq, x, acts, ders):
        lin = Autoencoder.linears(seq)
        for i, (_, m) in enumerate(lin):
            if i < len(lin) - 1 and self.autoencoder.relu:
                ops.linear_relu_fwd(x, m.weight, m.bias, ders[i], acts[i])
            else:
                ops.linear_fwd(x, m.weight, m.bias, acts[i])
                if i < len(lin) - 1:
                    ops.fill(ders[i], 1.0)                  # Identity activation: derivative 1
            x = acts[i]
        return x

    def _mlp_bwd(self, seq, prefix, x_in, acts, ders, gout, a: _Acts, need_dx: bool):
        """gout: gradient w.r.t. the last layer's output; writes weight / bias gradients; returns the
        gradient w.r.t. x_in (or None)."""
        lin = Autoencoder.linears(seq)
        for i in reversed(range(len(lin))):
            idx, m = lin[i]
            inp = acts[i - 1] if i > 0 else x_in
            ops.linear_bwd_weight(gout, inp, self._grad_views[f"{prefix}.{idx}.weight"], self._grad_views[f"{prefix}.{idx}.bias"])
            if i == 0 and not need_dx:
                return None
            buf = a.ga if gout.data_ptr() != a.ga.data_ptr() else a.gb
            dx = buf[:inp.numel()].view(inp.shape)
            ops.linear_bwd_input(gout, m.weight, dx, gelu_grad=ders[i - 1] if i > 0 else None)
            gout = dx
        return gout

    # -- reference API ------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x):
        """desom.py:52-56 -> (cls_logits | None, x_encoded, distances, bmu_indices[int64])."""
        x = self._check_input(x)
        a = self._buffers_for(x.shape[0], x.device)
        z = self._mlp_fwd(self.autoencoder.encoder, x, a.enc_act, a.enc_der)
        dist, bmu = self.som_layer(z)
        logits = None
        if self.classification:
            ops.linear_fwd(z, self.classifier.weight, self.classifier.bias, a.logits)
            logits = a.logits.clone()
        return logits, z.clone(), dist, bmu

    @torch.no_grad()
    def predict(self, x):
        """Inference fast path for tools/evaluation.py (evaluate_clustering / evaluate_classification read
        only the BMU indices / logits, evaluation.py:38-39,115-116): encoder + SOM (+ classifier), no
        decoder.  Returns (bmu_indices [B] int64, logits [B,C] | None) as views of internal buffers."""
        x = self._check_input(x)
        a = self._buffers_for(x.shape[0], x.device)
        s = self.som_layer._buffers_for(x.shape[0], x.device)
        z = self._mlp_fwd(self.autoencoder.encoder, x, a.enc_act, a.enc_der)
        self.som_layer._distances_into(z, s)
        if self.classification:
            ops.linear_fwd(z, self.classifier.weight, self.classifier.bias, a.logits)
        return s.bmu, (a.logits if self.classification else None)

    @torch.no_grad()
    def _forward_losses(self, x, y, gamma_t, T, want_grad: bool):
        """All forward kernels + losses (+ the loss-side gradients when want_grad).  ``gamma_t`` is the
        constant gamma of desom.py:27 (DESOM has no ramp); returns the total loss tensor."""
        x = self._check_input(x)
        B, K = x.shape[0], self.som_layer.n_prototypes
        a = self._buffers_for(B, x.device)
        s = self.som_layer._buffers_for(B, x.device)
        z = self._mlp_fwd(self.autoencoder.encoder, x, a.enc_act, a.enc_der)
        self.som_layer._distances_into(z, s)
        pred = self._mlp_fwd(self.autoencoder.decoder, z, a.dec_act, a.dec_der)
        g = float(gamma_t)
        recon_w = g if self.classification else 1.0                      # desom.py:148-153
        mode = self.som_layer._dist_mode
        if want_grad:
            ops.som_neigh_loss(s.dist, s.bmu, self.som_layer.grid_positions, T, s.loss_sum, inv_nx=s.inx, inv_nw=s.inw,
                               grad_scale=g / (B * K), coef=s.coef, row_dot=s.row_dot, col_dot=s.col_dot, distance=mode)
        else:
            ops.som_neigh_loss(s.dist, s.bmu, self.som_layer.grid_positions, T, s.loss_sum, distance=mode)
        ops.l1_loss(pred, x, a.recon_sum, dpred=a.dpred if want_grad else None, grad_scale=recon_w / pred.numel())
        som = s.loss_sum[0] / (B * K)
        recon = a.recon_sum[0] / pred.numel()
        if self.classification:
            ops.linear_fwd(z, self.classifier.weight, self.classifier.bias, a.logits)
            yv = y.view(-1)
            if yv.dtype != torch.int64:
                yv = yv.long()
            ops.cross_entropy_ls(a.logits, yv.contiguous(), 0.0, a.ce_sum, dlogits=a.dlogits if want_grad else None,
                                 grad_scale=1.0 / B)
            total = a.ce_sum[0] / B + g * (som + recon)
        else:
            total = recon + g * som
        self._ctx = (x, a, s)
        self._forward_id, self._seeds_consumed = self._forward_id + 1, False
        self._last = {"recon": recon, "som": som, "total": total}
        return total

    @torch.no_grad()
    def _scale_seeds(self, gout):
        """Multiply the loss-side gradient seeds by the scalar `gout` (0-dim device tensor)."""
        _, a, s = self._ctx
        gout = gout.detach().reshape(1).float().contiguous()
        for buf in (s.coef, s.row_dot, s.col_dot, a.dpred) + ((a.dlogits,) if self.classification else ()):
            ops.scale_by(buf, gout)

    @torch.no_grad()
    def _backward(self):
        x, a, s = self._ctx
        self._grads_reduced = False
        self._exchange_reset()
        z = a.enc_act[-1]
        # decoder: d total / d pred sits in a.dpred
        dz_dec = self._mlp_bwd(self.autoencoder.decoder, "autoencoder.decoder", z, a.dec_act, a.dec_der, a.dpred, a, True)
        a.dz.copy_(dz_dec)
        W = self.som_layer.prototypes
        gW = self._grad_views["som_layer.prototypes"]
        if self.som_layer._dist_mode == ops.DIST_MANHATTAN:
            ops.som_bwd_manhattan(z, W, s.coef, gW, a.dz, accumulate_gx=True)
        else:
            ops.som_bwd(z, W, s.coef, s.row_dot, s.col_dot, gW, a.dz, accumulate_gx=True)
        if self._overlap_enabled():
            off, n, _ = self.arena.offsets["som_layer.prototypes"]
            self._reduce_early(off, off + (n + 255) // 256 * 256,
                               streams=[torch.cuda.current_stream()] if gW.is_cuda else [])
        if self.classification:
            ops.linear_bwd_weight(a.dlogits, z, self._grad_views["classifier.weight"], self._grad_views["classifier.bias"])
            ops.linear_bwd_input(a.dlogits, self.classifier.weight, a.dz, accumulate=True)
        self._mlp_bwd(self.autoencoder.encoder, "autoencoder.encoder", x, a.enc_act, a.enc_der, a.dz, a, False)

    def update(self):
        """desom.py:113-118: temperature from the iteration BEFORE its increment."""
        self.som_layer.update_temperature(self._it)
        self._it += 1
        self.iteration += 1

    def training_step(self, batch, batch_idx):
        """desom.py:58-74.  Returns a scalar tensor; ``.backward()`` runs the HIP backward."""
        x, y = batch
        self.update()
        if self._anchor is None:
            self._anchor = torch.zeros((), device=self.arena.device, requires_grad=True)
        return _StepLoss.apply(self._anchor, self, x, y, self.gamma, float(self.som_layer.current_temperature))

    def train_step_fused(self, x, y):
        """Same step without the autograd bridge (the caller then runs optimizer.step())."""
        self.update()
        total = self._forward_losses(x, y, self.gamma, float(self.som_layer.current_temperature), want_grad=True)
        self._backward()
        return total

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """desom.py:76-93 (no temperature update)."""
        x, y = batch
        return self._forward_losses(x, y, self.gamma, float(self.som_layer.current_temperature), want_grad=False).clone()

    def configure_optimizers(self):
        """desom.py:95-111: Adam(lr, betas) over all parameters.  (The reference's 'adamw' branch reads
        attributes it never sets -- weight_decay, warmup_epochs -- and cannot run; it is refused here.)"""
        if self.opt_type != "adam":
            raise NotImplementedError("DESOM: optimizer.type must be 'adam' (the reference's 'adamw' branch raises AttributeError)")
        params = [p for _, p in self._named_trainable()]
        return FusedAdamW(self, [{"params": params, "weight_decay": 0.0}], lr=self.opt_lr, betas=(self.beta_1, self.beta_2),
                          adamw=False)
