"""Host-side mirror of the reference's module surface for the ViT-SOM training-step hot path.

Classes keep the reference's names, constructor (the YAML-schema ``config`` dict), method
names, positional return tuples / dtypes and ``state_dict`` keys:

  ViTSOM          models/vit_som.py:17-187   forward / training_step / validation_step /
                                             configure_optimizers
  ViTAutoencoder  models/vit.py:66-240       forward / forward_features / patchify / unpatchify
  SOMLayer        models/som_layer.py:8-152  forward / compute_distances / update_temperature /
                                             compute_weights / som_loss / index_to_position

All arithmetic runs in libvitsom_hip.so (hand-written gfx950 kernels) through ``ops``; this file
only owns memory (flat parameter arenas, activation buffers), ordering, schedules and the
data-parallel exchange (one RCCL all-reduce over the gradient arena).  There is no CPU path.
"""
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import Event, on_stream, stream_wait_stream
from .arena import ParamArena
from .tuning import hooks

try:  # the reference subclasses pl.LightningModule; do the same when Lightning is importable
    import pytorch_lightning as pl  # type: ignore
    _Base = pl.LightningModule
    _HAVE_PL = True
except Exception:  # pragma: no cover - Lightning is not in this image
    _Base = nn.Module
    _HAVE_PL = False


# ------------------------------------------------------------------------------------ helpers
def _sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim: int, grid_size: int, cls_token: bool = False) -> np.ndarray:
    """tools/utils.py:131-178 (float64; w-coordinate first, CLS row zeros)."""
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid_size, grid_size)
    emb = np.concatenate([_sincos_1d(embed_dim // 2, grid[0]), _sincos_1d(embed_dim // 2, grid[1])], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def get_layer_id_for_vit(name: str, num_layers: int) -> int:
    """tools/utils.py:73-84."""
    if name in ("cls_token", "pos_embed") or name.startswith("patch_embed"):
        return 0
    if name.startswith("blocks"):
        return int(name.split(".")[1]) + 1
    return num_layers


def param_groups_lrd(model, weight_decay=0.05, no_weight_decay_list=(), layer_decay=0.75):
    """tools/utils.py:28-71: layer/decay groups carrying an ``lr_scale`` key (inert downstream,
    SURVEY.md 3.3 -- kept so optimizer.param_groups looks like the reference's)."""
    groups: Dict[str, dict] = {}
    num_layers = len(model.blocks) + 1
    scales = [layer_decay ** (num_layers - i) for i in range(num_layers + 1)]
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim == 1 or n in no_weight_decay_list:
            g_decay, this_decay = "no_decay", 0.0
        else:
            g_decay, this_decay = "decay", weight_decay
        layer_id = get_layer_id_for_vit(n, num_layers)
        name = "layer_%d_%s" % (layer_id, g_decay)
        if name not in groups:
            groups[name] = {"lr_scale": scales[layer_id], "weight_decay": this_decay, "params": []}
        groups[name]["params"].append(p)
    return list(groups.values())


def _xavier_(t: torch.Tensor, fan_out: int, fan_in: int):
    a = math.sqrt(6.0 / (fan_in + fan_out))
    return t.uniform_(-a, a)


class _Affine(nn.Module):
    """Holder with ``weight`` / ``bias`` Parameters (Linear, LayerNorm, Conv2d-as-proj)."""

    def __init__(self, wshape, bshape):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(wshape))
        self.bias = nn.Parameter(torch.empty(bshape))


class _Attention(nn.Module):          # models/vit.py:16-26
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.scale = (dim // heads) ** -0.5
        self.qkv = _Affine((3 * dim, dim), (3 * dim,))
        self.proj = _Affine((dim, dim), (dim,))


class Block(nn.Module):               # models/vit.py:45-57
    def __init__(self, dim, heads, mlp_ratio):
        super().__init__()
        hidden = int(dim * mlp_ratio)
        self.dim, self.heads, self.hidden = dim, heads, hidden
        self.norm1 = _Affine((dim,), (dim,))
        self.attn = _Attention(dim, heads)
        self.norm2 = _Affine((dim,), (dim,))
        self.mlp = nn.ModuleDict({"0": _Affine((hidden, dim), (hidden,)), "2": _Affine((dim, hidden), (dim,))})


class _PatchEmbed(nn.Module):         # timm PatchEmbed attribute surface used by the reference
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = _Affine((embed_dim, in_chans, patch_size, patch_size), (embed_dim,))


def _trainable_order(vit: "ViTAutoencoder"):
    """(state-dict name, parameter) in forward order: weight then bias of each layer adjacent."""
    return [(n, p) for n, p in vit.named_parameters() if p.requires_grad]


class _Acts:
    """Activation / gradient buffers for one batch size (allocated once, reused every step)."""
    pass


_LOSS_RING = 16     # the loss terms of a step stay readable until this many further steps have run


# ------------------------------------------------------------------------------------ ViT autoencoder
class ViTAutoencoder(nn.Module):
    """MAE-style unmasked ViT autoencoder (models/vit.py:66-240); compute on the HIP kernels."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.0, norm_layer=None,
                 norm_pix_loss=False, eps: float = 1e-6):
        super().__init__()
        self.img_size, self.in_chans, self.eps = img_size, in_chans, eps
        self.embed_dim, self.decoder_embed_dim = embed_dim, decoder_embed_dim
        self.num_heads, self.decoder_num_heads = num_heads, decoder_num_heads
        self.patch_embed = _PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = _Affine((embed_dim,), (embed_dim,))
        self.decoder_embed = _Affine((decoder_embed_dim, embed_dim), (decoder_embed_dim,))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, n + 1, decoder_embed_dim), requires_grad=False)
        self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio)
                                             for _ in range(decoder_depth)])
        self.decoder_norm = _Affine((decoder_embed_dim,), (decoder_embed_dim,))
        self.decoder_pred = _Affine((patch_size ** 2 * in_chans, decoder_embed_dim), (patch_size ** 2 * in_chans,))
        self.initialize_weights()
        self._acts: Dict[int, _Acts] = {}

    # -- init: same distributions as vit.py:100-125 ------------------------------------------
    def initialize_weights(self):
        g = int(self.patch_embed.num_patches ** 0.5)
        with torch.no_grad():
            self.pos_embed.copy_(torch.from_numpy(get_2d_sincos_pos_embed(self.embed_dim, g, True)).float().unsqueeze(0))
            self.decoder_pos_embed.copy_(
                torch.from_numpy(get_2d_sincos_pos_embed(self.decoder_embed_dim, g, True)).float().unsqueeze(0))
            w = self.patch_embed.proj.weight
            _xavier_(w, w.shape[0], w[0].numel())
            bound = 1.0 / math.sqrt(w[0].numel())                    # Conv2d default bias init (untouched by _init_weights)
            self.patch_embed.proj.bias.uniform_(-bound, bound)
            self.cls_token.normal_(std=0.02)
            for name, m in self.named_modules():
                if not isinstance(m, _Affine) or m is self.patch_embed.proj:
                    continue
                if m.weight.ndim == 2:                                # nn.Linear: xavier_uniform / zero bias
                    _xavier_(m.weight, m.weight.shape[0], m.weight.shape[1])
                    m.bias.zero_()
                else:                                                 # nn.LayerNorm
                    m.weight.fill_(1.0)
                    m.bias.zero_()

    # -- pure index shuffles kept for API parity (torch view ops: no arithmetic) -------------
    def patchify(self, imgs):
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        h = w = imgs.shape[2] // p
        c = imgs.shape[1]
        x = imgs.reshape(imgs.shape[0], c, h, p, w, p)
        return torch.einsum("nchpwq->nhwpqc", x).reshape(imgs.shape[0], h * w, p ** 2 * c)

    def unpatchify(self, x):
        p = self.patch_embed.patch_size[0]
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        c = x.shape[2] // (p * p)
        x = x.reshape(x.shape[0], h, w, p, p, c)
        return torch.einsum("nhwpqc->nchpwq", x).reshape(x.shape[0], c, h * p, w * p)

    # -- buffers ------------------------------------------------------------------------------
    def _buffers_for(self, B: int, device) -> _Acts:
        a = self._acts.get(B)
        if a is not None and a.device == device:
            return a
        E, DE = self.embed_dim, self.decoder_embed_dim
        n = self.patch_embed.num_patches
        N, T = n + 1, B * (n + 1)
        p = self.patch_embed.patch_size[0]
        pd = p * p * self.in_chans
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)   # noqa: E731
        a = _Acts()
        a.device, a.B, a.N, a.T = device, B, N, T
        a.xp = f(B * n, pd)
        a.tok0 = f(T, E)

        def layer(dim, heads, hidden):
            L = _Acts()
            L.a1, L.mean1, L.rstd1 = f(T, dim), f(T), f(T)
            L.qkv, L.ao, L.lse = f(T, 3 * dim), f(T, dim), f(B, heads, N)
            L.x1, L.a2, L.mean2, L.rstd2 = f(T, dim), f(T, dim), f(T), f(T)
            L.hpre, L.hact, L.x2 = f(T, hidden), f(T, hidden), f(T, dim)
            return L
        a.enc = [layer(E, self.num_heads, b.hidden) for b in self.blocks]
        a.xe, a.mean_e, a.rstd_e = f(T, E), f(T), f(T)
        a.dec0 = f(T, DE)
        a.dec = [layer(DE, self.decoder_num_heads, b.hidden) for b in self.decoder_blocks]
        a.dn, a.mean_d, a.rstd_d = f(T, DE), f(T), f(T)
        a.pred = f(T, pd)
        # backward temporaries (shared by all layers; sized for the wider of encoder / decoder)
        W = max(E, DE)
        Hd = max([b.hidden for b in self.blocks] + [b.hidden for b in self.decoder_blocks])
        # five rotating [T, dim] gradient buffers and two dh / dqkv sets: a buffer the side stream reads
        # in one block is rewritten two blocks later at the earliest (see _side_join)
        a.g = [f(T * W) for _ in range(5)]
        a.dh2, a.dqkv2, a.da = [f(T * Hd), f(T * Hd)], [f(T * 3 * W), f(T * 3 * W)], f(T * W)
        a.delta = f(B * max(self.num_heads, self.decoder_num_heads) * N)
        a.dpred = f(T, pd)
        a.d_xe = f(T, E)
        a.version = 0            # bumped whenever the activation buffers are rewritten (staleness guard of the autograd bridges)
        # at most two batch sizes stay allocated (the training batch and, e.g., decode_prototype's batch of one)
        keep = list(self._acts.items())[-1:]
        self._acts = dict(keep + [(B, a)])
        return a

    # -- forward ------------------------------------------------------------------------------
    def _block_fwd(self, blk: Block, L: _Acts, x_in: torch.Tensor, B: int, N: int):
        T = B * N
        ops.layernorm_fwd(x_in, blk.norm1.weight, blk.norm1.bias, L.a1, L.mean1, L.rstd1, self.eps)
        ops.linear_fwd(L.a1, blk.attn.qkv.weight, blk.attn.qkv.bias, L.qkv)
        ops.attention_fwd(L.qkv, L.ao, L.lse, B, N, blk.heads, blk.dim // blk.heads)
        ops.linear_residual_fwd(L.ao, blk.attn.proj.weight, blk.attn.proj.bias, x_in, T, L.x1)
        ops.layernorm_fwd(L.x1, blk.norm2.weight, blk.norm2.bias, L.a2, L.mean2, L.rstd2, self.eps)
        ops.linear_gelu_fwd(L.a2, blk.mlp["0"].weight, blk.mlp["0"].bias, L.hpre, L.hact)
        ops.linear_residual_fwd(L.hact, blk.mlp["2"].weight, blk.mlp["2"].bias, L.x1, T, L.x2)
        return L.x2

    def _encode(self, x: torch.Tensor, a: _Acts):
        E = self.embed_dim
        p = self.patch_embed.patch_size[0]
        a.version += 1
        ops.patch_embed_fwd(x, self.patch_embed.proj.weight.view(E, -1), self.patch_embed.proj.bias, self.pos_embed[0],
                            self.cls_token.view(E), a.tok0, a.xp, p)
        cur = a.tok0
        side = None
        if cur.is_cuda and a.B % 2 == 0 and a.B >= 64 and hooks.fwd_split:
            # the owner (ViTSOM) lends the stream its backward uses for the weight gradients -- idle during
            # the forward; a stream of its own would compete for the few hardware queues of the process
            # (measured: erratic, sometimes slower than one chain)
            side = self.__dict__.get("_lent_stream")
            if side is None or side.device != cur.device:
                side = self.__dict__.get("_fwd_side")
                if side is None or side.device != cur.device:
                    side = torch.cuda.Stream(device=cur.device)
        self.__dict__["_fwd_side"] = side
        if side is not None:
            # The forward is one dependent chain per image: the two halves of the batch run as two chains
            # on two streams (row-sliced views of the same buffers, so the results are the same bits and
            # the backward sees one batch); staggered against each other, one chain's latency-bound
            # kernels (attention, LayerNorm) run under the other's GEMMs.
            Bh, Th = a.B // 2, a.T // 2
            cuts = a.__dict__.get("_enc_halves")
            if cuts is None:
                def cut(L, h):
                    Lh = _Acts()
                    for k, v in L.__dict__.items():
                        Lh.__dict__[k] = v[h * Bh:(h + 1) * Bh] if k == "lse" else v[h * Th:(h + 1) * Th]
                    return Lh
                cuts = a.__dict__["_enc_halves"] = [[cut(L, h) for L in a.enc] for h in (0, 1)]
            self._event().record().wait(side)
            # All blocks by default (A/B in one process, round 2: 0 / 6 / 12 of 12 blocks split -> 11.77 / 11.81 /
            # 11.68 ms per step; round 1 kept it to half because the f32-MFMA BMU pass ran slower right after a dense
            # forward -- the bf16 BMU pass does not).
            nsplit = len(self.blocks) if hooks.fwd_split_blocks is None else int(hooks.fwd_split_blocks)
            nsplit = max(0, min(nsplit, len(self.blocks)))
            # enqueue the two chains alternately, block by block: the host feeds both streams at the same pace (all
            # of chain 0 first left the second stream idle for the ~0.7 ms the host needs to enqueue six blocks)
            c0, c1 = a.tok0[:Th], a.tok0[Th:]
            for i in range(nsplit):
                c0 = self._block_fwd(self.blocks[i], cuts[0][i], c0, Bh, a.N)
                with on_stream(side):
                    c1 = self._block_fwd(self.blocks[i], cuts[1][i], c1, Bh, a.N)
            self._event().record(side).wait()
            cur = a.enc[nsplit - 1].x2 if nsplit > 0 else a.tok0
            for blk, L in zip(self.blocks[nsplit:], a.enc[nsplit:]):
                cur = self._block_fwd(blk, L, cur, a.B, a.N)
        else:
            for blk, L in zip(self.blocks, a.enc):
                cur = self._block_fwd(blk, L, cur, a.B, a.N)
        ops.layernorm_fwd(cur, self.norm.weight, self.norm.bias, a.xe, a.mean_e, a.rstd_e, self.eps)
        return a.xe

    def _decode(self, a: _Acts):
        ops.linear_residual_fwd(a.xe, self.decoder_embed.weight, self.decoder_embed.bias, self.decoder_pos_embed[0],
                                a.N, a.dec0)
        cur = a.dec0
        for blk, L in zip(self.decoder_blocks, a.dec):
            cur = self._block_fwd(blk, L, cur, a.B, a.N)
        ops.layernorm_fwd(cur, self.decoder_norm.weight, self.decoder_norm.bias, a.dn, a.mean_d, a.rstd_d, self.eps)
        ops.linear_fwd(a.dn, self.decoder_pred.weight, self.decoder_pred.bias, a.pred)
        return a.pred

    def _check_input(self, x):
        if x.dim() != 4 or x.shape[1] != self.in_chans or x.shape[2] != self.img_size or x.shape[3] != self.img_size:
            raise ValueError(f"expected input [B,{self.in_chans},{self.img_size},{self.img_size}], got {tuple(x.shape)}")
        return x.contiguous().float()

    def _attention_maps(self, blocks, layers, a: _Acts):
        """[B, heads, N, N] softmax probabilities of every block (vit.py:33-34,41-42), formed from the saved qkv / lse."""
        out = []
        for blk, L in zip(blocks, layers):
            probs = torch.empty(a.B, blk.heads, a.N, a.N, dtype=torch.float32, device=a.device)
            ops.attention_probs(L.qkv, L.lse, probs, a.B, a.N, blk.heads, blk.dim // blk.heads)
            out.append(probs)
        return out

    def _wants_grad(self, *inputs):
        return torch.is_grad_enabled() and (any(p.requires_grad for _, p in _trainable_order(self))
                                            or any(t.requires_grad for t in inputs))

    def forward_features(self, x, return_attns=False):
        """vit.py:155-179 -> (cls_token_out [B,E], attns | None); differentiable w.r.t. the encoder parameters
        under autograd (``_VitForwardFn`` in features mode)."""
        x = self._check_input(x)
        if self._wants_grad(x):
            cls = _VitForwardFn.apply(x, self, "features", *[p for _, p in _trainable_order(self)])
            a = self._acts[x.shape[0]]
        else:
            with torch.no_grad():
                a = self._buffers_for(x.shape[0], x.device)
                cls = self._encode(x, a).view(a.B, a.N, self.embed_dim)[:, 0].clone()
        with torch.no_grad():
            attns = self._attention_maps(self.blocks, a.enc, a) if return_attns else None
        return cls, attns

    def forward(self, x, return_attns=False):
        """vit.py:202-240 -> (cls_token_out [B,E], patch_tokens_out [B,n,E], recon_img [B,C,S,S]) (+ the encoder's
        attention maps as a fourth element when return_attns, vit.py:238-239).  With autograd enabled the three
        outputs are differentiable w.r.t. every trainable parameter (``_VitForwardFn``: the stand-alone use of the
        sub-module; the fused training step of ViTSOM does not go through here)."""
        x = self._check_input(x)
        if self._wants_grad(x):
            out = _VitForwardFn.apply(x, self, "full", *[p for _, p in _trainable_order(self)])
        else:
            with torch.no_grad():
                out = self._forward_impl(x)
        if return_attns:
            with torch.no_grad():
                a = self._acts[x.shape[0]]
                return tuple(out) + (self._attention_maps(self.blocks, a.enc, a),)
        return out

    def _forward_impl(self, x):
        a = self._buffers_for(x.shape[0], x.device)
        xe = self._encode(x, a).view(a.B, a.N, self.embed_dim)
        self._decode(a)
        recon = torch.empty_like(x)
        scratch1 = torch.empty(1, dtype=torch.float32, device=x.device)
        ops.l1_unpatchify(a.pred, x, scratch1, recon=recon, p=self.patch_embed.patch_size[0])
        return xe[:, 0].clone(), xe[:, 1:].clone(), recon

    def forward_decoder(self, x, return_attn=False):
        """vit.py:182-200 -> (decoded_patches [B,n,p*p*C], attns | None): decoder_embed -> + decoder_pos_embed -> decoder
        blocks -> decoder_norm -> decoder_pred[:, 1:] on an ARBITRARY token tensor x [B,n+1,E] (tools/evaluation.py:209-222
        feeds a prototype behind a zero CLS row).  The reference's return_attn=False branch assigns the block's
        (x, attn) tuple to `decoded` (vit.py:195) and fails at decoder_norm; this is what it means.  Differentiable
        w.r.t. the decoder parameters and x under autograd."""
        n, E = self.patch_embed.num_patches, self.embed_dim
        if x.dim() != 3 or x.shape[1] != n + 1 or x.shape[2] != E:
            raise ValueError(f"forward_decoder: expected tokens [B,{n + 1},{E}], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise ValueError("forward_decoder: input must live on the MI355X (there is no CPU path)")
        x = x.float()
        if self._wants_grad(x):
            patches = _VitDecoderFn.apply(x, self, *[p for _, p in _trainable_order(self)])
            a = self._acts[x.shape[0]]
        else:
            with torch.no_grad():
                a = self._decode_tokens(x)
                patches = a.pred.view(a.B, a.N, -1)[:, 1:].clone()
        with torch.no_grad():
            attns = self._attention_maps(self.decoder_blocks, a.dec, a) if return_attn else None
        return patches, attns

    def _decode_tokens(self, x):
        a = self._buffers_for(x.shape[0], x.device)
        a.version += 1
        a.xe.view(a.B, a.N, self.embed_dim).copy_(x)
        self._decode(a)
        return a

    # -- backward -----------------------------------------------------------------------------
    @staticmethod
    def _dx(WT, dy, weight, dx, **kw):
        """dX = dY W: from the transposed weight copy when the owner keeps one (NT kernel family)."""
        wt = WT(weight) if WT is not None else None
        if wt is not None:
            return ops.linear_bwd_input_t(dy, wt, dx, **kw)
        return ops.linear_bwd_input(dy, weight, dx, **kw)

    # Weight-gradient GEMMs (and their slab reductions) are off the backward's critical path: nothing
    # reads dW before the optimizer.  With a side stream set (ViTSOM does, on the GPU) they run
    # concurrently with the dX / LayerNorm / attention chain and fill its tail rounds and the
    # small-grid gaps.  Ordering: (1) a side GEMM waits for the main-stream kernel that produced
    # its dY; (2) the dY buffers (gout / g1 from a ring of five, dh / dqkv from two sets) are rewritten
    # two blocks later at the earliest, and the entry of block j waits for the side work of block
    # j+2 (_side_join) -- a wait that has normally long been satisfied, so the main stream does not
    # stall on the ~15 us cross-stream signalling latency a wait on the PREVIOUS block costs;
    # (3) the owner joins the side stream before anything reads the gradients.  The saved
    # activations the GEMMs read are not written during a backward pass.
    _side = None

    def _event(self):
        """Pooled library events (re-recording one is safe once the waits on its previous record are enqueued; the pool is
        far longer than the few events whose wait is deferred by a block or two)."""
        return Event.pooled()

    def _dw(self, dy, x, gw, gb):
        side = self._side
        if side is None:
            return ops.linear_bwd_weight(dy, x, gw, gb)
        self._event().record().wait(side)       # recorded on the main (current) stream: dy is final here
        with on_stream(side):
            ops.linear_bwd_weight(dy, x, gw, gb)

    def _side_join(self, keep: int = 1):
        """Main stream waits for the side work of all but the `keep` most recent blocks."""
        pend = self.__dict__.setdefault("_side_pending", [])
        if self._side is None:
            pend.clear()
            return
        while len(pend) > keep:
            pend.pop(0).wait()

    def _side_mark(self):
        if self._side is not None:
            self.__dict__.setdefault("_side_pending", []).append(self._event().record(self._side))

    def _ln_bwd(self, dy, x, mean, rstd, gamma, resid, dx, dgamma, dbeta):
        """LayerNorm backward; with a job list lent by the owner (ViTSOM._backward) the dgamma / dbeta reduction is
        left to the owner's next flush."""
        jobs = self.__dict__.get("_ln_jobs")
        if jobs is not None and ops.layernorm_bwd_deferrable(*x.shape):
            return jobs.bwd(dy, x, mean, rstd, gamma, resid, dx, dgamma, dbeta)
        return ops.layernorm_bwd(dy, x, mean, rstd, gamma, resid, dx, dgamma, dbeta)

    def _block_bwd(self, blk: Block, L: _Acts, x_in, gout, a: _Acts, G, prefix: str, bufs, WT=None, parity: int = 0):
        """gout: gradient w.r.t. the block output [T,dim]; returns gradient w.r.t. x_in (in bufs)."""
        T, dim, hid = a.T, blk.dim, blk.hidden
        self._side_join(keep=1)            # side work of the block before the previous one must be done
        g1, g0 = bufs
        dh = a.dh2[parity][:T * hid].view(T, hid)
        da = a.da[:T * dim].view(T, dim)
        dqkv = a.dqkv2[parity][:T * 3 * dim].view(T, 3 * dim)
        self._dw(gout, L.hact, G(f"{prefix}.mlp.2.weight"), G(f"{prefix}.mlp.2.bias"))
        self._dx(WT, gout, blk.mlp["2"].weight, dh, gelu_grad=L.hpre)
        self._dw(dh, L.a2, G(f"{prefix}.mlp.0.weight"), G(f"{prefix}.mlp.0.bias"))
        self._dx(WT, dh, blk.mlp["0"].weight, da)
        self._ln_bwd(da, L.x1, L.mean2, L.rstd2, blk.norm2.weight, gout, g1, G(f"{prefix}.norm2.weight"),
                          G(f"{prefix}.norm2.bias"))
        self._dw(g1, L.ao, G(f"{prefix}.attn.proj.weight"), G(f"{prefix}.attn.proj.bias"))
        self._dx(WT, g1, blk.attn.proj.weight, da)
        ops.attention_bwd(L.qkv, L.ao, da, L.lse, dqkv, a.delta, a.B, a.N, blk.heads, dim // blk.heads)
        self._dw(dqkv, L.a1, G(f"{prefix}.attn.qkv.weight"), G(f"{prefix}.attn.qkv.bias"))
        self._dx(WT, dqkv, blk.attn.qkv.weight, da)
        self._ln_bwd(da, x_in, L.mean1, L.rstd1, blk.norm1.weight, g1, g0, G(f"{prefix}.norm1.weight"),
                          G(f"{prefix}.norm1.bias"))
        self._side_mark()
        return g0

    def _views(self, a: _Acts, dim: int):
        return [b[:a.T * dim].view(a.T, dim) for b in a.g]

    def _decoder_bwd(self, a: _Acts, G, WT=None, before_dxe=None):
        """a.dpred holds dL/dpred; writes decoder grads and dL/d(xe) into a.d_xe -- overwriting it, or,
        when `before_dxe` is given, calling it and then ADDING to what a.d_xe holds (the SOM input
        gradient written concurrently on another stream; `before_dxe` waits for it)."""
        DE = self.decoder_embed_dim
        ring = self._views(a, DE)
        gA = ring[0]
        ops.linear_bwd_weight(a.dpred, a.dn, G("decoder_pred.weight"), G("decoder_pred.bias"))
        dn_grad = a.da[:a.T * DE].view(a.T, DE)
        self._dx(WT, a.dpred, self.decoder_pred.weight, dn_grad)
        x_last = a.dec[-1].x2 if a.dec else a.dec0
        self._ln_bwd(dn_grad, x_last, a.mean_d, a.rstd_d, self.decoder_norm.weight, None, gA,
                          G("decoder_norm.weight"), G("decoder_norm.bias"))
        gout, pos = gA, 0
        for j, i in enumerate(reversed(range(len(self.decoder_blocks)))):
            x_in = a.dec[i - 1].x2 if i > 0 else a.dec0
            bufs = [ring[(pos + 1) % 5], ring[(pos + 2) % 5]]
            gout = self._block_bwd(self.decoder_blocks[i], a.dec[i], x_in, gout, a, G, f"decoder_blocks.{i}", bufs, WT, j & 1)
            pos = (pos + 2) % 5
        ops.linear_bwd_weight(gout, a.xe, G("decoder_embed.weight"), G("decoder_embed.bias"))
        if before_dxe is not None:
            before_dxe()
        self._dx(WT, gout, self.decoder_embed.weight, a.d_xe, accumulate=before_dxe is not None)

    def _encoder_bwd(self, a: _Acts, G, WT=None, on_block=None):
        """a.d_xe holds dL/d(xe); writes every encoder gradient.  on_block(i) is called once block i's
        backward (main chain and weight-gradient side work) has been enqueued."""
        self._side_join(keep=0)            # the decoder's blocks may still be reading the shared buffers
        E = self.embed_dim
        ring = self._views(a, E)
        gA = ring[0]
        x_last = a.enc[-1].x2 if a.enc else a.tok0
        self._ln_bwd(a.d_xe, x_last, a.mean_e, a.rstd_e, self.norm.weight, None, gA, G("norm.weight"), G("norm.bias"))
        gout, pos = gA, 0
        for j, i in enumerate(reversed(range(len(self.blocks)))):
            x_in = a.enc[i - 1].x2 if i > 0 else a.tok0
            bufs = [ring[(pos + 1) % 5], ring[(pos + 2) % 5]]
            gout = self._block_bwd(self.blocks[i], a.enc[i], x_in, gout, a, G, f"blocks.{i}", bufs, WT, j & 1)
            pos = (pos + 2) % 5
            if on_block is not None:
                on_block(i)
        p = self.patch_embed.patch_size[0]
        ops.patch_embed_bwd(gout, a.xp, G("patch_embed.proj.weight").view(E, -1), G("patch_embed.proj.bias"),
                            G("cls_token").view(E), a.B, self.in_chans, self.img_size, p, E)


# ------------------------------------------------------------------------------------ per-module autograd
def _stale(vit, B, version):
    a = vit._acts.get(B)
    if a is None or a.version != version:
        raise RuntimeError("ViTAutoencoder: backward() after the activation buffers of this batch size were rewritten "
                           "(another forward / training_step / validation_step ran in between); call backward first")
    return a


class _VitForwardFn(torch.autograd.Function):
    """ViTAutoencoder.forward / forward_features for stand-alone use under autograd (models/vit.py:155-179,202-240):
    forward = the HIP forward kernels; backward = the same HIP backward kernels the fused step uses, fed with the
    upstream gradients of (cls, patches, recon) -- or of cls alone in "features" mode.  The gradient w.r.t. the input
    IMAGE is not produced (nothing on the path needs it): an input that requires grad is refused."""

    @staticmethod
    def forward(ctx, x, vit, mode, *params):
        if x.requires_grad:
            raise RuntimeError("ViTAutoencoder: the gradient w.r.t. the input image is not implemented")
        ctx.vit, ctx.B, ctx.mode = vit, x.shape[0], mode
        with torch.no_grad():
            if mode == "features":
                a = vit._buffers_for(x.shape[0], x.device)
                out = vit._encode(x, a).view(a.B, a.N, vit.embed_dim)[:, 0].clone()
            else:
                out = vit._forward_impl(x)
        ctx.version = vit._acts[ctx.B].version
        return out

    @staticmethod
    def backward(ctx, g_cls, g_patches=None, g_recon=None):
        vit = ctx.vit
        a = _stale(vit, ctx.B, ctx.version)
        named = _trainable_order(vit)
        with torch.no_grad():
            grads = {n: torch.zeros_like(p) for n, p in named}
            G = grads.__getitem__
            E, N, B = vit.embed_dim, a.N, a.B
            side, vit._side = vit._side, None                     # single stream: this is not the fused step
            try:
                if g_recon is not None:
                    dp = a.dpred.view(B, N, -1)
                    dp[:, 0].zero_()
                    dp[:, 1:].copy_(vit.patchify(g_recon.float()))
                    vit._decoder_bwd(a, G)
                else:
                    a.d_xe.zero_()
                d = a.d_xe.view(B, N, E)
                if g_cls is not None:
                    d[:, 0].add_(g_cls)
                if g_patches is not None:
                    d[:, 1:].add_(g_patches)
                vit._encoder_bwd(a, G)
            finally:
                vit._side = side
        return (None, None, None) + tuple(grads[n] for n, _ in named)


class _VitDecoderFn(torch.autograd.Function):
    """ViTAutoencoder.forward_decoder under autograd (models/vit.py:182-200): gradients to the decoder parameters and
    to the token tensor it was fed."""

    @staticmethod
    def forward(ctx, x, vit, *params):
        ctx.vit, ctx.B = vit, x.shape[0]
        with torch.no_grad():
            a = vit._decode_tokens(x)
            out = a.pred.view(a.B, a.N, -1)[:, 1:].clone()
        ctx.version = a.version
        return out

    @staticmethod
    def backward(ctx, g_patches):
        vit = ctx.vit
        a = _stale(vit, ctx.B, ctx.version)
        named = _trainable_order(vit)
        with torch.no_grad():
            grads = {n: torch.zeros_like(p) for n, p in named}
            side, vit._side = vit._side, None
            try:
                dp = a.dpred.view(a.B, a.N, -1)
                dp[:, 0].zero_()
                dp[:, 1:].copy_(g_patches.float())
                vit._decoder_bwd(a, grads.__getitem__)
            finally:
                vit._side = side
            gx = a.d_xe.view(a.B, a.N, vit.embed_dim).clone()
        return (gx, None) + tuple(grads[n] for n, _ in named)


class _SomDistancesFn(torch.autograd.Function):
    """SOMLayer.forward under autograd (som_layer.py:83-89, 111-125): distances differentiable w.r.t. the input rows
    and the prototypes; the BMU indices are returned alongside (non-differentiable, argmin)."""

    @staticmethod
    def forward(ctx, x, W, layer):
        with torch.no_grad():
            s = layer._buffers_for(x.shape[0], x.device)
            layer._distances_into(x, s)
            dist, bmu = s.dist.clone(), s.bmu.clone()
            ctx.save_for_backward(x, W, dist, s.inx.clone(), s.inw.clone())
        ctx.mode = layer._dist_mode
        ctx.mark_non_differentiable(bmu)
        return dist, bmu

    @staticmethod
    def backward(ctx, g_dist, _g_bmu):
        x, W, dist, inx, inw = ctx.saved_tensors
        B, K = dist.shape
        with torch.no_grad():
            f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dist.device)   # noqa: E731
            coef, row_dot, col_dot, tmp = f(B, K), f(B), f(K), f(1)
            # backward coefficients of sum(g * dist): the upstream gradient plays the role of the weights
            ops.som_weighted_loss(dist, g_dist.float().contiguous(), tmp, inv_nx=inx, inv_nw=inw, grad_scale=1.0, coef=coef,
                                  row_dot=row_dot, col_dot=col_dot, distance=ctx.mode)
            gW, gX = torch.empty_like(W), torch.empty_like(x)
            if ctx.mode == ops.DIST_MANHATTAN:
                ops.som_bwd_manhattan(x, W, coef, gW, gX, accumulate_gx=False)
            else:
                ops.som_bwd(x, W, coef, row_dot, col_dot, gW, gX, accumulate_gx=False)
        return gX, gW, None


class _SomLossFn(torch.autograd.Function):
    """mean(weights * distances) (som_layer.py:137-142) with gradients to both arguments."""

    @staticmethod
    def forward(ctx, weights, distances):
        ctx.save_for_backward(weights, distances)
        tmp = torch.empty(1, dtype=torch.float32, device=distances.device)
        ops.som_weighted_loss(distances, weights, tmp)
        out = torch.empty((), dtype=torch.float32, device=distances.device)
        ops.scaled_mul(out.view(1), tmp, factor=1.0 / distances.numel())
        return out

    @staticmethod
    def backward(ctx, gout):
        weights, distances = ctx.saved_tensors
        g = gout.detach().reshape(1).float().contiguous()
        inv = 1.0 / distances.numel()
        gw = ops.scaled_mul(torch.empty_like(weights), distances, scale_dev=g, factor=inv) if ctx.needs_input_grad[0] else None
        gd = ops.scaled_mul(torch.empty_like(distances), weights, scale_dev=g, factor=inv) if ctx.needs_input_grad[1] else None
        return gw, gd


# ------------------------------------------------------------------------------------ SOM layer
class SOMLayer(_Base):
    """models/som_layer.py:8-152 on the HIP kernels (cosine / euclidean / manhattan distance; square /
    hexa topology; clients: ViTSOM and DESOM)."""

    def __init__(self, config):
        super().__init__()
        hp = config["hyperparameters"]
        self.model_arch = hp["model_arch"]
        som_hp, data_hp = hp["som"], config["data"]
        vit_hp = hp["vit"] if self.model_arch == "vit_som" else None
        self.total_epochs, self.batch_size = hp["total_epochs"], hp["batch_size"]
        self.map_size = som_hp["map_size"]
        self.Tmax, self.Tmin = som_hp["Tmax"], som_hp["Tmin"]
        self.topology, self.distance_fcn = som_hp["topology"], som_hp["distance_fcn"]
        self.n_prototypes = int(np.prod(self.map_size))
        modes = {"cosine": ops.DIST_COSINE, "euclidean": ops.DIST_EUCLIDEAN, "manhattan": ops.DIST_MANHATTAN}
        if self.distance_fcn not in modes:                              # som_layer.py:111-125 raises the same way
            raise ValueError(f"Unsupported distance function: {self.distance_fcn}")
        self._dist_mode = modes[self.distance_fcn]
        if self.model_arch == "vit_som":                                       # som_layer.py:35-40
            self.use_reduced = som_hp["use_reduced"]
            latent_dim = vit_hp["emb_dim"]
            if not self.use_reduced:
                latent_dim *= (data_hp["input_size"] // vit_hp["patch_size"]) ** 2
        else:                                                                  # DESOM: the autoencoder's code
            self.use_reduced = False
            latent_dim = hp["ae"]["encoder_dims"][-1]
        self.latent_dim = latent_dim
        self.current_temperature = self.Tmax
        proto = torch.rand(self.n_prototypes, latent_dim)                      # som_layer.py:44-56
        if self.distance_fcn == "cosine":
            proto = torch.nn.functional.normalize(proto, p=2, dim=1)
        self.prototypes = nn.Parameter(proto)
        self.create_grid_positions()
        self._world_size = 1
        self._n_train: Optional[int] = None
        self._bufs: Dict[int, _Acts] = {}
        # pre-split plane image of the prototypes for the BMU contraction (ops.bmu_planes_*): valid while its stamp
        # equals _w_stamp().  FusedAdamW rewrites it in the pass that updates the prototypes; anything else that
        # changes them is seen through torch's version counter, the storage address or _raw_updates.
        self._wplanes: Optional[torch.Tensor] = None
        self._wplanes_stamp = None
        self._raw_updates = 0           # updates of the prototypes that bypass torch (raw-pointer kernels)
        self._planes_used = False       # a forward took the planes path: the optimizer keeps the image current

    # ---- plane image of the prototypes ---------------------------------------------------
    def _w_stamp(self):
        W = self.prototypes
        return (W.data_ptr(), W._version, self._raw_updates, tuple(W.shape))

    def invalidate_planes(self):
        """Call after changing the prototypes behind torch's back (writes through ``.data`` or a raw pointer)."""
        self._wplanes_stamp = None

    def _planes_shape_ok(self, B: int) -> bool:
        W = self.prototypes
        return bool(hooks.bmu_planes and self._dist_mode == ops.DIST_COSINE and W.is_cuda and ops.get_gemm_mode() != ops.GEMM_F32
                    and ops.bmu_planes_supported(B, W.shape[0], W.shape[1]))

    def _w_planes(self) -> torch.Tensor:
        """The prototypes' plane buffer, re-split here if it does not describe them any more."""
        W = self.prototypes
        if self._wplanes is None or self._wplanes.device != W.device or self._wplanes.numel() != ops.lib.vsom_bmu_planes_bytes(*W.shape):
            self._wplanes = ops.bmu_planes_alloc(W.shape[0], W.shape[1], W.device)
            self._wplanes_stamp = None
        if self._wplanes_stamp != self._w_stamp():
            ops.bmu_planes_from(W.detach(), self._wplanes)
            self._wplanes_stamp = self._w_stamp()
        return self._wplanes

    def _w_planes_async(self, side_stream, force: bool):
        """Bring the prototypes' image up to date on `side_stream`, behind everything the launch stream holds so far
        (the optimizer step that wrote the prototypes, the last contraction that read the image) -- when it is stale,
        or always with `force` (a recorded training step must contain the launch whatever the state it was recorded
        in).  Returns the event the consumer has to wait for, or None when nothing was launched."""
        W = self.prototypes
        if self._wplanes is None or self._wplanes.device != W.device or self._wplanes.numel() != ops.lib.vsom_bmu_planes_bytes(*W.shape):
            self._wplanes = ops.bmu_planes_alloc(W.shape[0], W.shape[1], W.device)
            self._wplanes_stamp = None
        if not force and self._wplanes_stamp == self._w_stamp():
            return None
        Event.pooled().record().wait(side_stream)
        with on_stream(side_stream):
            ops.bmu_planes_from(W.detach(), self._wplanes)
        self._wplanes_stamp = self._w_stamp()
        return Event.pooled().record(side_stream)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate_planes()

    def create_grid_positions(self):                                   # som_layer.py:60-81
        if self.topology == "square":
            gy, gx = torch.meshgrid(torch.arange(self.map_size[0]), torch.arange(self.map_size[1]), indexing="ij")
            positions = torch.stack([gy, gx], dim=-1).view(-1, 2).float()
        elif self.topology == "hexa":
            rows, cols = self.map_size
            positions = torch.zeros(self.n_prototypes, 2)
            for i in range(self.n_prototypes):
                row, col = i // cols, i % cols
                positions[i, 0] = col + (0.5 if row % 2 == 1 else 0.0)
                positions[i, 1] = row * np.sqrt(3) / 2
        else:
            raise ValueError(f"Unsupported topology: {self.topology}")
        self.register_buffer("grid_positions", positions)

    def _buffers_for(self, B: int, device) -> _Acts:
        s = self._bufs.get(B)
        if s is not None and s.device == device:
            return s
        K = self.n_prototypes
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=device)   # noqa: E731
        s = _Acts()
        s.device = device
        s.inx, s.inw = f(B), f(K)
        s.dist, s.bmu = f(B, K), torch.empty(B, dtype=torch.int64, device=device)
        s.reranked = torch.zeros(1, dtype=torch.int32, device=device)      # rows whose BMU needed the exact re-rank (cumulative)
        s.coef, s.row_dot, s.col_dot = f(B, K), f(B), f(K)
        s.loss_sum = f(1)
        # like the ViT's activation buffers: at most two batch sizes stay allocated (training and validation batches alternate)
        self._bufs = dict(list(self._bufs.items())[-1:] + [(B, s)])
        return s

    # reference API -----------------------------------------------------------------------
    @torch.no_grad()
    def compute_distances(self, x):                                    # som_layer.py:111-125
        if x.dim() > 2:
            x = x.flatten(start_dim=1)
        s = self._buffers_for(x.shape[0], x.device)
        self._distances_into(x, s)
        return s.dist.clone()

    def _distances_into(self, x2d, s: _Acts):
        if self._dist_mode == ops.DIST_COSINE:
            W = self.prototypes
            if (ops.get_gemm_mode() != ops.GEMM_F32 and W.shape[0] <= 2048 and W.shape[1] % 4 == 0
                    and x2d.stride(0) % 4 == 0 and x2d.data_ptr() % 16 == 0):
                # norms + reduced-precision contraction + exact re-rank in one pass over X and W
                if self._planes_shape_ok(x2d.shape[0]):
                    # ... on pre-split operands: the prototypes' image is kept by the optimizer, the samples' written here
                    self._planes_used = True
                    if getattr(s, "xplanes", None) is None:
                        s.xplanes = ops.bmu_planes_alloc(x2d.shape[0], x2d.shape[1], x2d.device)
                    wplanes = self._w_planes()
                    ops.bmu_planes_from(x2d, s.xplanes)
                    ops.bmu_cosine_x3_planes_fwd(x2d, W, s.xplanes, wplanes, s.dist, s.bmu, s.inx, s.inw, s.reranked)
                else:
                    ops.bmu_cosine_x3_fwd(x2d, W, s.dist, s.bmu, s.inx, s.inw, s.reranked)
            else:
                ops.row_inv_norm(x2d, s.inx)
                ops.row_inv_norm(W, s.inw)
                ops.bmu_cosine_fwd(x2d, W, s.inx, s.inw, s.dist, s.bmu)
        elif self._dist_mode == ops.DIST_MANHATTAN:
            ops.bmu_manhattan_fwd(x2d, self.prototypes, s.dist, s.bmu)
        else:                                   # euclidean: inx / inw hold the squared norms
            ops.row_sqnorm(x2d, s.inx)
            ops.row_sqnorm(self.prototypes, s.inw)
            ops.bmu_euclid_fwd(x2d, self.prototypes, s.inx, s.inw, s.dist, s.bmu)

    def forward(self, x):                                              # som_layer.py:83-89
        """-> (distances [B,K], bmu_indices [B] int64).  With autograd enabled the distances are differentiable
        w.r.t. `x` and the prototypes (``_SomDistancesFn``); the fused training step does not go through here."""
        if x.dim() > 2:
            x = x.flatten(start_dim=1)
        x = x.float()
        if x.stride(-1) != 1:
            x = x.contiguous()
        if torch.is_grad_enabled() and (x.requires_grad or self.prototypes.requires_grad):
            return _SomDistancesFn.apply(x, self.prototypes, self)
        with torch.no_grad():
            s = self._buffers_for(x.shape[0], x.device)
            self._distances_into(x, s)
            return s.dist.clone(), s.bmu.clone()

    def total_iterations(self) -> float:
        n = self._n_train
        if n is None:
            tr = getattr(self, "_trainer_ref", None)
            if tr is None:
                raise RuntimeError("SOMLayer: call ViTSOM.set_schedule(n_train, estimated_stepping_batches) "
                                   "or attach a trainer before training_step")
            n = len(tr.train_dataloader.dataset)
        # single-process semantics on the GLOBAL batch (the reference divides by the per-rank
        # batch size only, som_layer.py:131 -- SURVEY.md section 5, defect (b))
        return (n / (self.batch_size * self._world_size)) * self.total_epochs

    def update_temperature(self, iteration):                           # som_layer.py:127-132
        it = float(iteration)
        self.current_temperature = self.Tmax * (self.Tmin / self.Tmax) ** (it / (self.total_iterations() - 1))

    def index_to_position(self, indices):                              # som_layer.py:134-135
        return torch.stack((indices // self.map_size[1], indices % self.map_size[1]), dim=1).float()

    @torch.no_grad()
    def compute_weights(self, bmu_indices):                            # som_layer.py:144-152
        B, K = bmu_indices.shape[0], self.n_prototypes
        dev = bmu_indices.device
        h = torch.empty(B, K, dtype=torch.float32, device=dev)
        zero_d = torch.zeros(B, K, dtype=torch.float32, device=dev)
        tmp = torch.empty(1, dtype=torch.float32, device=dev)
        ops.som_neigh_loss(zero_d, bmu_indices.contiguous(), self.grid_positions, float(self.current_temperature), tmp, h=h,
                           distance=self._dist_mode)
        return h

    def som_loss(self, weights, distances):                            # som_layer.py:137-142
        """mean(weights * distances) for ANY weights tensor, differentiable in both arguments."""
        if weights.shape != distances.shape:
            raise ValueError(f"som_loss: weights {tuple(weights.shape)} and distances {tuple(distances.shape)} differ")
        return _SomLossFn.apply(weights.float().contiguous(), distances.float().contiguous())


# ------------------------------------------------------------------------------------ optimiser
class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW / Adam semantics (vit_som.py:146-157) as ONE kernel over the flat arena.

    ``param_groups`` mirror the reference's (layer/decay groups with the inert ``lr_scale`` key
    plus the prototypes/cls_head group with AdamW's default weight_decay=0.01).  All groups
    share one lr (the reference's single-lambda LambdaLR scales them equally).  ``step()``
    first sums the gradient arena across ranks (RCCL all-reduce) when world_size > 1."""

    def __init__(self, model: "ViTSOM", param_groups, lr, betas, adamw=True, eps=1e-8):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0.01 if adamw else 0.0)
        super().__init__(param_groups, defaults)
        self._model = model
        self._adamw = adamw
        self._step = 0
        # per-chunk weight decay follows the groups
        name_of = {id(p): n for n, p in model._named_trainable()}
        for g in self.param_groups:
            for p in g["params"]:
                model.arena.set_weight_decay(name_of[id(p)], float(g["weight_decay"]))
        if model.classification:
            # the reference leaves the decoder without gradients in classification mode, so
            # torch's AdamW never touches it (no decay either) -- SURVEY.md section 5 defect (a)
            for n in model._decoder_param_names():
                model.arena.set_weight_decay(n, 0.0)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:               # Lightning's automatic optimization passes training_step + backward here
            with torch.enable_grad():
                loss = closure()
        m = self._model
        m.allreduce_gradients()
        self._step += 1
        g0 = self.param_groups[0]
        lrs = {float(g["lr"]) for g in self.param_groups}
        if len(lrs) != 1:
            raise RuntimeError("FusedAdamW: per-group learning rates differ; the arena kernel uses one lr")
        b1, b2 = g0["betas"]
        som = getattr(m, "som_layer", None)
        planes = None
        if som is not None and som._planes_used and hooks.bmu_planes and hooks.adamw_planes:
            # the prototypes' plane image for the next BMU pass leaves the same kernel that updates them
            name = next((n for n, q in m._named_trainable() if q is som.prototypes), None)
            W = som.prototypes
            if name is not None and W.dim() == 2 and W.shape[1] % 8 == 0 and ops.get_gemm_mode() != ops.GEMM_F32:
                if som._wplanes is None or som._wplanes.device != W.device:
                    som._wplanes = ops.bmu_planes_alloc(W.shape[0], W.shape[1], W.device)
                planes = (m.arena.offsets[name][0], W.shape[0], W.shape[1], som._wplanes)
        ops.adamw_step(m.arena.params, m.arena.grads, m.arena.exp_avg, m.arena.exp_avg_sq, m.arena.wd_chunk,
                       float(g0["lr"]), b1, b2, g0["eps"], self._step, grad_scale=1.0 / m.world_size, adamw=self._adamw,
                       planes=planes)
        if som is not None:
            som._raw_updates += 1                                # the kernel writes through raw pointers
            som._wplanes_stamp = som._w_stamp() if planes is not None else None
        return loss

    def zero_grad(self, set_to_none: bool = True):
        # gradients are fully overwritten by every backward pass; nothing to clear
        return None

    # torch.optim.AdamW-compatible state layout (per-parameter 'step' / 'exp_avg' / 'exp_avg_sq' in
    # param_groups order), so optimizer states interchange with reference-written checkpoints
    def _param_names_in_group_order(self):
        name_of = {id(p): n for n, p in self._model._named_trainable()}
        return [name_of[id(p)] for g in self.param_groups for p in g["params"]]

    def state_dict(self):
        a = self._model.arena
        names = self._param_names_in_group_order()
        state = {}
        if self._step > 0:
            for i, n in enumerate(names):
                state[i] = {"step": torch.tensor(float(self._step)), "exp_avg": a.view(a.exp_avg, n).clone(),
                            "exp_avg_sq": a.view(a.exp_avg_sq, n).clone()}
        groups, k = [], 0
        for g in self.param_groups:
            pg = {key: v for key, v in g.items() if key != "params"}
            pg["params"] = list(range(k, k + len(g["params"])))
            k += len(g["params"])
            groups.append(pg)
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        a = self._model.arena
        names = self._param_names_in_group_order()
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("FusedAdamW.load_state_dict: different number of parameter groups")
        for g, sg in zip(self.param_groups, sd["param_groups"]):
            for key, v in sg.items():
                if key != "params":
                    g[key] = v
        steps = set()
        a.exp_avg.zero_(); a.exp_avg_sq.zero_()
        for i, st in sd["state"].items():
            n = names[int(i)]
            a.view(a.exp_avg, n).copy_(st["exp_avg"])
            a.view(a.exp_avg_sq, n).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("FusedAdamW.load_state_dict: parameters carry different step counts")
        self._step = steps.pop() if steps else 0


# ------------------------------------------------------------------------------------ autograd bridge
class _StepLoss(torch.autograd.Function):
    """Makes the fused step look like one differentiable scalar to torch / Lightning:
    forward = all HIP forward kernels + losses, backward = all HIP backward kernels writing the
    gradient arena.  The incoming scalar gradient multiplies the three loss-side seeds (dL/dpred,
    dL/dlogits, the SOM coefficients) BEFORE the backward kernels run -- the backward is linear in
    them -- so nothing touches the arena after the overlapped all-reduces have started."""

    @staticmethod
    def forward(ctx, anchor, model, x, y, gamma_t, T):
        ctx.model = model
        ctx.taped = hasattr(model, "_tape_usable") and model._tape_usable(x)
        if ctx.taped:
            out, ctx.backward_done = model._taped_forward(x, y, gamma_t, T)
            out = out.clone()
        else:
            out = model._forward_losses(x, y, gamma_t, T, want_grad=True).clone()
        ctx.forward_id = model._forward_id
        return out

    @staticmethod
    def backward(ctx, gout):
        m = ctx.model
        if ctx.forward_id != m._forward_id or m._seeds_consumed:
            raise RuntimeError("ViTSOM: backward() called twice for one training_step (or after a later forward): the "
                               "step's buffers and gradient seeds are single-use; gradient accumulation is not supported")
        m._seeds_consumed = True
        if ctx.taped and m._ctx[1].__dict__.get("tape") is not None:
            m._taped_backward(gout)        # segment 2 scales the seeds by gout, segment 3 is the backward
        else:
            m._scale_seeds(gout)           # 1.0 under a plain loss.backward()
            m._backward()
        m._expose_grads()
        return None, None, None, None, None, None


class _StepTape:
    """Handle of a recorded step (vsom_tape_*): destroyed with the activation buffers it points into."""

    def __init__(self, tid, nseg, key, started, comm_dirty, side, som_bufs):
        if nseg != 4:
            ops.tape_destroy(tid)
            raise RuntimeError(f"launch tape: expected 4 segments, recorded {nseg}")
        self.id, self.key, self.started, self.comm_dirty, self.side, self.som_bufs = tid, key, started, comm_dirty, side, som_bufs

    def close(self):
        if self.id:
            ops.tape_destroy(self.id)
            self.id = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def init_vsom_comm(world_size: int, rank: int, unique_id: Optional[bytes] = None):
    """One RCCL communicator per process behind the C-ABI (vsom_comm_init).  The unique id comes from rank 0; with no
    `unique_id` given it travels over the torch.distributed process group the launcher set up (host-side plumbing)."""
    w, r = ops.comm_info()
    if w == world_size and r == rank:
        return
    if w != 0:
        ops.comm_destroy()
    if unique_id is None:
        if world_size == 1:
            unique_id = ops.comm_unique_id()
        else:
            import torch.distributed as dist
            box = [ops.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            unique_id = box[0]
    ops.comm_init(unique_id, world_size, rank)


def _vsom_comm_selftest(world_size: int, device) -> bool:
    """One small sum all-reduce through the library's communicator, checked against the closed form: rank r contributes
    r + 1 in every element, the sum is world (world + 1) / 2."""
    _, rank = ops.comm_info()
    buf = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=device)
    ops.comm_allreduce_sum(buf)
    torch.cuda.synchronize(device)
    return bool((buf == world_size * (world_size + 1) / 2).all().item())


# ------------------------------------------------------------------------------------ arena owner
_STEP_STREAMS: Dict[int, tuple] = {}        # device index -> (side stream, SOM stream), shared by every model of the process


class _ArenaOwner:
    """What every model on this path shares: trainable tensors packed into flat arenas
    (arena.py), gradients exposed as views, and the data-parallel exchange over the gradient
    arena.  Subclasses provide ``som_layer`` and may override the two hooks."""

    arena: Optional[ParamArena] = None
    world_size, rank = 1, 0
    _grads_reduced = False
    _forward_id, _seeds_consumed = 0, False      # one backward per forward of the fused step (_StepLoss)

    def _default_weight_decay(self, name: str, p) -> float:
        return 0.0

    def _after_pack(self):
        pass

    def _named_trainable(self):
        return [(n, p) for n, p in self.named_parameters() if p.requires_grad]

    def _pack(self, device):
        """(Re)build the flat arenas on `device` and re-point every Parameter at its view."""
        old_wd = self.arena.wd_by_name if self.arena is not None else {}
        named = self._named_trainable()
        specs = []
        for n, p in named:
            wd = old_wd[n] if n in old_wd else self._default_weight_decay(n, p)
            specs.append((n, tuple(p.shape), wd))
        arena = ParamArena(specs, device)
        with torch.no_grad():
            for n, p in named:
                v = arena.p(n)
                v.copy_(p.detach().to(device))
                p.data = v
            for n, b in list(self.named_buffers()) + [(n, p) for n, p in self.named_parameters() if not p.requires_grad]:
                if b.device != device:
                    b.data = b.data.to(device)
        if self.arena is not None and self.arena.device == device:
            arena.exp_avg.copy_(self.arena.exp_avg)
            arena.exp_avg_sq.copy_(self.arena.exp_avg_sq)
        self.arena = arena
        self._anchor = None
        self._grad_views = {n: arena.g(n) for n, _ in named}
        self._after_pack()

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        dev = next(self.parameters()).device
        aliased = all(p.data_ptr() == self.arena.p(n).data_ptr() for n, p in self._named_trainable())
        if not aliased or dev != self.arena.device:
            self._pack(dev)
        return self

    def _G(self, prefix: str):
        return lambda name: self._grad_views[prefix + name]

    def _expose_grads(self):
        for n, p in self._named_trainable():
            p.grad = self._grad_views[n]

    _use_vsom_comm = False

    def set_distributed(self, world_size: int, rank: int = 0, backend: Optional[str] = None):
        """backend: "rccl" = the library's own communicator (vsom_comm_*; the default on the GPU unless torch.distributed
        runs on gloo), "torch" = torch.distributed's all_reduce (gloo on CPU tensors, or its "nccl" = RCCL)."""
        self.world_size, self.rank = int(world_size), int(rank)
        self.som_layer._world_size = int(world_size)
        self._backend_defaulted = backend is None
        if backend is None:
            backend = "torch"
            if self.world_size > 1 and self.arena is not None and self.arena.grads.is_cuda:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized() and dist.get_backend() != "gloo":
                    backend = "rccl"
        if backend not in ("rccl", "torch"):
            raise ValueError(f"set_distributed: unknown backend {backend!r}")
        self._use_vsom_comm = backend == "rccl"
        if self._use_vsom_comm:
            chosen_by_default = getattr(self, "_backend_defaulted", False)
            try:
                init_vsom_comm(self.world_size, self.rank)
                ok = self.world_size == 1 or _vsom_comm_selftest(self.world_size, self.arena.grads.device)
                err = None if ok else "self-test all-reduce gave a wrong sum"
            except Exception as e:                       # noqa: BLE001 -- a collective backend that does not come up
                if not chosen_by_default:
                    raise
                ok, err = False, repr(e)
            if chosen_by_default and self.world_size > 1:
                # every rank takes the same path: agree on it through the process group that is known to work
                import torch.distributed as dist
                flag = torch.tensor([1.0 if ok else 0.0], device=self.arena.grads.device)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                all_ok = bool(flag.item() > 0.5)
                if not all_ok:
                    import warnings
                    warnings.warn(f"vit_som_amd: the library's RCCL communicator did not come up on every rank ({err}); "
                                  f"the gradient exchange uses torch.distributed instead")
                    self._use_vsom_comm = False
                    if ops.comm_info()[0] != 0:
                        ops.comm_destroy()
            elif not ok:
                raise RuntimeError(f"set_distributed: vsom_comm {err}")

    # -- data-parallel exchange: sum all-reduce over the gradient arena, in pieces -----------------
    # Each piece is a contiguous arena slice whose gradients are final at a known point of the backward
    # pass: the [K, L] prototype accumulator right after the SOM backward (79 of 100 MB at CIFAR shapes),
    # the decoder after the decoder backward, the encoder in buckets of a few blocks in reverse layer
    # order.  A piece is issued from a stream of its own that first waits for the events of the streams
    # that wrote it (main chain + weight-gradient side stream), so the collective (RCCL runs it on its
    # own stream) overlaps the rest of the backward; allreduce_gradients() reduces what is left and
    # makes the consumer stream wait for every piece.  Under torch.distributed "nccl" == RCCL over xGMI.
    def _overlap_enabled(self) -> bool:
        return self.world_size > 1 and hooks.overlap_allreduce

    def _exchange_reset(self):
        """Forget the pieces of the previous exchange.  Pieces still in flight (a backward pass whose gradients were
        never consumed by allreduce_gradients() / optimizer.step()) are waited for first: the new backward is about
        to overwrite the arena slices they are reducing."""
        for w in getattr(self, "_works", ()):
            w.wait()
        if getattr(self, "_comm_dirty", False) and self.arena is not None and self.arena.grads.is_cuda:
            stream_wait_stream(None, self._comm)
        self._works, self._started, self._comm_dirty = [], [], False

    def _arena_span(self, first: str, last: str):
        """[lo, hi) of the arena slice from parameter `first` through parameter `last` (padded)."""
        lo = self.arena.offsets[first][0]
        off, n, _ = self.arena.offsets[last]
        return lo, off + (n + 255) // 256 * 256

    def _reduce_async(self, lo: int, hi: int, after=()):
        """Start the sum all-reduce of grads[lo:hi]; `after` = events the piece must wait for."""
        import torch.distributed as dist
        g = self.arena.grads
        if hi <= lo:
            return
        if g.is_cuda:
            comm = getattr(self, "_comm", None)
            if comm is None or comm.device != g.device:
                comm = self._comm = torch.cuda.Stream(device=g.device)
            for ev in after:
                ev.wait(comm)
            if self._use_vsom_comm:
                # the library's own RCCL communicator (vsom_comm_*): the collective is enqueued on `comm` like a kernel
                with on_stream(comm):
                    ops.comm_allreduce_sum(g[lo:hi])
                self._comm_dirty = True
            else:
                with torch.cuda.stream(comm):
                    self._works.append(dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        else:
            self._works.append(dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self._started.append((lo, hi))

    def _reduce_early(self, lo: int, hi: int, streams=()):
        """Called inside the backward pass once grads[lo:hi] is final on the given streams."""
        if not self._overlap_enabled():
            return
        evs = []
        if self.arena.grads.is_cuda:
            for st in streams:
                evs.append(Event.pooled().record(st))
        self._reduce_async(lo, hi, evs)

    def allreduce_gradients(self):
        """Reduce every arena slice not yet in flight, then make the current stream wait for all pieces.
        Idempotent until the next backward pass; AdamW divides by world_size."""
        if self.world_size <= 1 or self._grads_reduced:
            return
        self._grads_reduced = True
        g = self.arena.grads
        if not hasattr(self, "_works"):
            self._exchange_reset()
        evs = []
        if g.is_cuda:
            evs.append(Event.pooled().record())     # current stream: every gradient is final here
        pos = 0
        for lo, hi in sorted(self._started) + [(g.numel(), g.numel())]:
            if lo > pos:
                self._reduce_async(pos, lo, evs)
            pos = max(pos, hi)
        self._exchange_reset()                  # torch "nccl" / vsom_comm: the current stream waits; gloo: the host does

    def broadcast_parameters(self, src: int = 0):
        """Replicas are built from the same seed; this makes it explicit (DDP broadcasts at construction)."""
        if self.world_size > 1:
            import torch.distributed as dist
            dist.broadcast(self.arena.params, src=src)


# ------------------------------------------------------------------------------------ ViT-SOM
class ViTSOM(_ArenaOwner, _Base):
    """Vision Transformer Self-Organizing Map (models/vit_som.py:17-187), MI355X-native."""

    def __init__(self, config, device=None):
        super().__init__()
        # NOTE: unlike vit_som.py:23 this does NOT lower torch's global float32 matmul precision:
        # every contraction here is exact fp32 on MFMA.
        self.config = config
        if _HAVE_PL:
            self.save_hyperparameters(config)
        hp, data_hp = config["hyperparameters"], config["data"]
        vit_hp, opt_hp, som_hp = hp["vit"], hp["optimizer"], hp["som"]
        self.gamma = hp["gamma"]
        self.use_reduced = som_hp["use_reduced"]
        self.classification = data_hp["num_classes"] > 0
        self.vit = ViTAutoencoder(
            img_size=data_hp["input_size"], patch_size=vit_hp["patch_size"], in_chans=data_hp["num_channels"],
            embed_dim=vit_hp["emb_dim"], depth=vit_hp["depth"], num_heads=vit_hp["heads"],
            decoder_embed_dim=vit_hp["dec_emb_dim"], decoder_depth=vit_hp["dec_depth"],
            decoder_num_heads=vit_hp["heads"], mlp_ratio=vit_hp["mlp_ratio"], eps=1e-6)
        self.som_layer = SOMLayer(config)
        if self.classification:
            self.cls_head = _Affine((data_hp["num_classes"], vit_hp["emb_dim"]), (data_hp["num_classes"],))
            with torch.no_grad():
                self.cls_head.weight.normal_(std=0.02)
                bound = 1.0 / math.sqrt(vit_hp["emb_dim"])
                self.cls_head.bias.uniform_(-bound, bound)
        self.smoothing = float(opt_hp["smoothing"])
        self.register_buffer("iteration", torch.tensor(0))
        self._it = 0
        self._n_train: Optional[int] = None
        self._est_steps: Optional[int] = None
        self.world_size, self.rank = 1, 0
        self._grads_reduced = False
        self._forward_id, self._seeds_consumed = 0, False
        self._last: Dict[str, torch.Tensor] = {}
        self.arena: Optional[ParamArena] = None
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self._pack(torch.device(device))

    # -- arenas -------------------------------------------------------------------------------
    def _default_weight_decay(self, name: str, p) -> float:
        if name.startswith("vit."):
            return 0.0 if p.ndim == 1 else 0.05
        return 0.01

    def _after_pack(self):
        self._build_weight_transposes()

    def _decoder_param_names(self):
        return [n for n, _ in self._named_trainable() if n.startswith("vit.decoder_")]

    def _build_weight_transposes(self):
        """Transposed copies W^T of the ViT Linear weights whose input gradient is needed, so that
        dX = dY W runs on the forward's kernel family (both operands contiguous along the reduction).
        One flat buffer + a device table; refreshed by ONE batched transpose per backward pass."""
        arena, dev = self.arena, self.arena.device
        rows, views, off = [], {}, 0
        for n, p in self._named_trainable():
            if not (n.startswith("vit.") and p.ndim == 2 and n.endswith(".weight")) or "patch_embed" in n:
                continue
            N, K = p.shape
            if N % 4 or K % 4:
                continue
            src = (arena.p(n).data_ptr() - arena.params.data_ptr()) // 4
            rows.append((src, off, N, K))
            views[arena.p(n).data_ptr()] = (off, K, N)
            off += -(-N * K // 64) * 64
        self._wt_flat = torch.empty(max(off, 1), dtype=torch.float32, device=dev)
        self._wt_table = torch.tensor(rows, dtype=torch.int64, device=dev).view(-1, 4) if rows else None
        self._wt_views = {k: self._wt_flat[o:o + a * b].view(a, b) for k, (o, a, b) in views.items()}
        self._wt_max = (max(r[2] for r in rows), max(r[3] for r in rows)) if rows else (1, 1)

    def _refresh_weight_transposes(self):
        if self._wt_table is not None and self._wt_flat.is_cuda:
            ops.transpose_many(self.arena.params, self._wt_flat, self._wt_table, *self._wt_max)

    def _WT(self, weight):
        return self._wt_views.get(weight.data_ptr())

    # -- schedules ----------------------------------------------------------------------------
    def set_schedule(self, n_train: int, estimated_stepping_batches: int):
        """Trainer-less replacement for len(trainer.train_dataloader.dataset) and
        trainer.estimated_stepping_batches (som_layer.py:131, vit_som.py:89)."""
        self._n_train, self._est_steps = int(n_train), int(estimated_stepping_batches)
        self.som_layer._n_train = int(n_train)

    def _estimated_steps(self) -> int:
        if self._est_steps is not None:
            return self._est_steps
        tr = getattr(self, "_trainer", None) if not _HAVE_PL else getattr(self, "trainer", None)
        if tr is None:
            raise RuntimeError("ViTSOM: call set_schedule(n_train, estimated_stepping_batches) or attach a trainer")
        self.som_layer._trainer_ref = tr
        return int(tr.estimated_stepping_batches)

    def _gamma_t(self) -> float:                                        # vit_som.py:89-90 (host int, no .item() sync)
        ramp_up_end_step = self._estimated_steps() // 2
        return self.config["hyperparameters"]["gamma"] * min(1.0, self._it / ramp_up_end_step)

    def _log(self, *a, **k):
        """self.log / self.log_dict when a Lightning trainer is attached (vit_som.py:95-101); a no-op otherwise.
        Errors raised by Lightning's logger propagate."""
        if _HAVE_PL and getattr(self, "_trainer", None) is not None:
            self.log_dict(*a, **k) if isinstance(a[0], dict) else self.log(*a, **k)

    # -- fused forward + losses ---------------------------------------------------------------
    def _som_input(self, a: _Acts):
        E, N, B = self.vit.embed_dim, a.N, a.B
        if self.use_reduced:
            return torch.as_strided(a.xe, (B, E), (N * E, 1), a.xe.storage_offset())
        return torch.as_strided(a.xe, (B, (N - 1) * E), (N * E, 1), a.xe.storage_offset() + E)

    def _cls_view(self, buf: torch.Tensor, a: _Acts):
        E = self.vit.embed_dim
        return torch.as_strided(buf, (a.B, E), (a.N * E, 1), buf.storage_offset())

    @torch.no_grad()
    def _run_forward(self, x, need_decoder: bool, fresh_w: bool = False):
        x = self.vit._check_input(x)
        if not x.is_cuda:
            raise ValueError("ViTSOM: input must live on the MI355X (there is no CPU path)")
        a = self.vit._buffers_for(x.shape[0], x.device)
        self._ensure_streams(x.device)
        # the prototypes' plane image for the BMU pass: re-split on the SOM stream while the encoder runs
        w_ready = self.som_layer._w_planes_async(self._som_stream, fresh_w) if self.som_layer._planes_shape_ok(a.B) else None
        self.vit._encode(x, a)
        s = self.som_layer._buffers_for(a.B, x.device)
        if need_decoder and hooks.bmu_overlap and hooks.side_stream:
            # The BMU pass and the decoder both start from the encoder output and do not meet before the losses: the pass
            # runs on the SOM stream (behind the prototypes' image, which is written there) under the decoder's
            # latency-bound kernels.
            som = self._som_stream
            Event.pooled().record().wait(som)
            with on_stream(som):
                self.som_layer._distances_into(self._som_input(a), s)
            self.vit._decode(a)
            Event.pooled().record(som).wait()
        else:
            if need_decoder:
                self.vit._decode(a)
            if w_ready is not None:
                w_ready.wait()
            self.som_layer._distances_into(self._som_input(a), s)
        if self.classification:
            if not hasattr(a, "logits"):
                a.logits = torch.empty(a.B, self.cls_head.weight.shape[0], dtype=torch.float32, device=x.device)
                a.dlogits = torch.empty_like(a.logits)
            ops.linear_fwd(self._cls_view(a.xe, a), self.cls_head.weight, self.cls_head.bias, a.logits)
        return x, a, s

    @torch.no_grad()
    def forward(self, x):
        """vit_som.py:67-78 -> (cls_token, recon_img, logits | None, distances, bmu_indices[int64])."""
        x, a, s = self._run_forward(x, need_decoder=True)
        recon = torch.empty_like(x)
        tmp = torch.empty(1, dtype=torch.float32, device=x.device)
        ops.l1_unpatchify(a.pred, x, tmp, recon=recon, p=self.vit.patch_embed.patch_size[0])
        cls = self._cls_view(a.xe, a).clone()
        logits = a.logits.clone() if self.classification else None
        return cls, recon, logits, s.dist.clone(), s.bmu.clone()

    @torch.no_grad()
    def predict(self, x):
        """Inference fast path for tools/evaluation.py: encoder + SOM (+ cls head) only -- the decoder,
        whose output evaluate_clustering / evaluate_classification never read, is skipped.  Returns
        (bmu_indices [B] int64, logits [B,C] | None) as views of internal buffers (valid until the
        next call)."""
        x, a, s = self._run_forward(x, need_decoder=False)
        return s.bmu, (a.logits if self.classification else None)

    # The two calls of the step whose arguments change from step to step (temperature, gamma ramp): the host issues them
    # itself, also when the rest of the step is replayed from a launch tape (tape holes).
    def _call_neigh(self, s: _Acts, gamma_t: float, T: float, B: int, want_grad: bool):
        K = self.som_layer.n_prototypes
        if want_grad:
            ops.som_neigh_loss(s.dist, s.bmu, self.som_layer.grid_positions, T, s.loss_sum, inv_nx=s.inx, inv_nw=s.inw,
                               grad_scale=gamma_t / (B * K), coef=s.coef, row_dot=s.row_dot, col_dot=s.col_dot,
                               distance=self.som_layer._dist_mode)
        else:
            ops.som_neigh_loss(s.dist, s.bmu, self.som_layer.grid_positions, T, s.loss_sum,
                               distance=self.som_layer._dist_mode)

    def _call_parts(self, a: _Acts, s: _Acts, gamma_t: float, T: float, B: int, numel_x: int, want_grad: bool):
        # total = main + gamma_t * som (and the two terms by themselves, for logging) from the two device-side sums, in one
        # tiny kernel that also advances the `iteration` buffer of a training step (vit_som.py:104): no ATen kernel in
        # the step.  The three values land in their own slot of a small ring, so `_last` and the returned loss stay
        # valid for the next _LOSS_RING - 1 steps (plain tensors: .get / `in` / iteration / ** all see them).
        K = self.som_layer.n_prototypes
        main_scale = 1.0 / B if self.classification else 1.0 / numel_x
        a.loss_slot = (a.loss_slot + 1) % _LOSS_RING
        parts = a.loss_ring[a.loss_slot]
        ops.loss_parts(parts, a.main_sum, main_scale, s.loss_sum, gamma_t / (B * K), 1.0 / (B * K),
                       counter=self.iteration if want_grad else None)
        self._last = {"total": parts[0], "main": parts[1], "som": parts[2], "gamma_t": gamma_t, "T": T}
        return parts[0]

    def _loss_buffers(self, a: _Acts, dev):
        if not hasattr(a, "main_sum"):
            a.main_sum = torch.empty(1, dtype=torch.float32, device=dev)
            a.loss_ring = torch.zeros(_LOSS_RING, 4, dtype=torch.float32, device=dev)
            a.loss_slot = 0

    @torch.no_grad()
    def _forward_losses(self, x, y, gamma_t: float, T: float, want_grad: bool):
        """All forward kernels + both losses (+ loss-side gradients when want_grad).  Returns the
        total loss as a 0-dim device tensor; parts land in self._last."""
        # (a training step always re-splits the prototypes unless the optimizer keeps their image: a recorded step must
        # not depend on the state it was recorded in)
        x, a, s = self._run_forward(x, need_decoder=not self.classification, fresh_w=want_grad and not hooks.adamw_planes)
        B = a.B
        self._ctx = (x, a, s)
        self._forward_id, self._seeds_consumed = self._forward_id + 1, False
        self._loss_buffers(a, x.device)
        with ops.tape_hole():
            self._call_neigh(s, gamma_t, T, B, want_grad)
        if self.classification:
            yv = y.view(-1)
            if yv.dtype != torch.int64:
                yv = yv.long()
            ops.cross_entropy_ls(a.logits, yv.contiguous(), self.smoothing, a.main_sum,
                                 dlogits=a.dlogits if want_grad else None, grad_scale=1.0 / B)
        else:
            ops.l1_unpatchify(a.pred, x, a.main_sum, dpred=a.dpred if want_grad else None, grad_scale=1.0 / x.numel(),
                              p=self.vit.patch_embed.patch_size[0])
        with ops.tape_hole():
            total = self._call_parts(a, s, gamma_t, T, B, x.numel(), want_grad)
        return total

    def _ensure_streams(self, device):
        """The two extra HIP streams of the step (kept to two: a process has few hardware queues)."""
        if getattr(self, "_side_stream", None) is None or self._side_stream.device != device:
            # one pair per device for the whole process: which hardware queue a stream lands on depends on how many
            # streams the process has created, and two of a model's streams on one queue serialise (measured: the 3rd, 5th
            # ... model of a process ran its step 1.4x slower at batch 128)
            key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
            pair = _STEP_STREAMS.get(key)
            if pair is None:
                pair = _STEP_STREAMS[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
            self._side_stream = pair[0]      # weight-gradient GEMMs; second forward chain
            self._som_stream = pair[1]       # SOM backward + early all-reduce; the prototypes' plane image in the forward
        self.vit.__dict__["_lent_stream"] = self._side_stream

    @torch.no_grad()
    def _scale_seeds(self, gout):
        """Multiply the loss-side gradient seeds by the scalar `gout` (a 0-dim device tensor)."""
        _, a, s = self._ctx
        gout = gout.detach().reshape(1).float().contiguous()
        for buf in ((s.coef, s.row_dot, s.col_dot) + ((a.dlogits,) if self.classification else (a.dpred,))):
            ops.scale_by(buf, gout)

    def _exchange_buckets(self):
        """Arena slices reduced early, in the order the backward finishes them: name -> (lo, hi)."""
        b = self.__dict__.get("_bucket_cache")
        if b is not None and b[0] is self.arena:
            return b[1]
        names = [n for n, _ in self._named_trainable()]
        out = {"som": (self.arena.offsets["som_layer.prototypes"][0], self.arena.numel)}
        dec = [n for n in names if n.startswith("vit.decoder_")]
        if dec and not self.classification:
            out["decoder"] = self._arena_span(dec[0], dec[-1])
        D = len(self.vit.blocks)
        step = max(1, int(hooks.bucket_blocks))
        hi_name = "vit.norm.bias"
        for i in range(D - step, 0, -step):                 # blocks [i, i + step) (+ the final norm for the top bucket)
            out[f"enc{i}"] = self._arena_span(f"vit.blocks.{i}.norm1.weight", hi_name)
            hi_name = f"vit.blocks.{i - 1}.mlp.2.bias"
        self.__dict__["_bucket_cache"] = (self.arena, out)
        return out

    @torch.no_grad()
    def _backward(self):
        """All backward kernels; overwrites the whole gradient arena (no accumulation)."""
        x, a, s = self._ctx
        self._grads_reduced = False
        self._exchange_reset()
        if x.is_cuda and hooks.side_stream:
            self._ensure_streams(x.device)
            self.vit._side = self._side_stream
        else:
            self.vit._side = None
        Gv = self._G("vit.")
        self._refresh_weight_transposes()
        # the LayerNorm backwards leave their dgamma / dbeta reductions to one launch per exchange piece (or one in all)
        jobs = None
        if x.is_cuda and hooks.ln_reduce_batched:
            jobs = a.__dict__.get("ln_jobs")
            if jobs is None:
                jobs = a.ln_jobs = ops.LayerNormJobs(x.device)
            jobs.begin()
        self.vit.__dict__["_ln_jobs"] = jobs
        try:
            self._backward_body(x, a, s, Gv, jobs)
        finally:
            self.vit.__dict__["_ln_jobs"] = None

    def _backward_body(self, x, a, s, Gv, jobs):
        X = self._som_input(a)
        E, N = self.vit.embed_dim, a.N
        if self.use_reduced:
            gX = torch.as_strided(a.d_xe, (a.B, E), (N * E, 1), a.d_xe.storage_offset())
        else:
            gX = torch.as_strided(a.d_xe, (a.B, (N - 1) * E), (N * E, 1), a.d_xe.storage_offset() + E)
        buckets = self._exchange_buckets() if self._overlap_enabled() else {}
        main = torch.cuda.current_stream() if x.is_cuda else None

        def som_backward(gx_out, accumulate):
            if self.som_layer._dist_mode == ops.DIST_MANHATTAN:
                ops.som_bwd_manhattan(X, self.som_layer.prototypes, s.coef, self._grad_views["som_layer.prototypes"], gx_out,
                                      accumulate_gx=accumulate)
            else:
                ops.som_bwd(X, self.som_layer.prototypes, s.coef, s.row_dot, s.col_dot,
                            self._grad_views["som_layer.prototypes"], gx_out, accumulate_gx=accumulate)

        def streams_now():
            return [st for st in (main, self.vit._side) if st is not None]

        def flush():
            if jobs is not None:
                jobs.flush()

        side = self._som_stream if self.vit._side is not None else None
        if self.classification or side is None:
            if self.classification:
                ops.fill(a.d_xe, 0.0)
                # decoder is unused by the classification loss: its gradients are exactly zero
                for n in self._decoder_param_names():
                    ops.fill(self._grad_views[n], 0.0)
                ops.linear_bwd_weight(a.dlogits, self._cls_view(a.xe, a), self._grad_views["cls_head.weight"],
                                      self._grad_views["cls_head.bias"])
                ops.linear_bwd_input(a.dlogits, self.cls_head.weight, self._cls_view(a.d_xe, a), accumulate=True)
            else:
                self.vit._decoder_bwd(a, Gv, self._WT)
                if "decoder" in buckets:
                    flush()
                    self._reduce_early(*buckets["decoder"], streams=streams_now())
            som_backward(gX, True)
            if "som" in buckets:
                self._reduce_early(*buckets["som"], streams=streams_now())
        else:
            # The SOM backward depends only on the forward (coef, X, W), so it runs on a stream of its
            # own under the decoder backward and writes its input gradient straight into (the zeroed)
            # d_xe; the decoder's last GEMM waits for it and accumulates on top.  The prototype
            # all-reduce is issued behind it: it starts the moment gW is final, before the decoder
            # backward has finished.
            self.vit._event().record().wait(side)
            with on_stream(side):
                ops.fill(a.d_xe, 0.0)
                som_backward(gX, False)
            if "som" in buckets:
                self._reduce_early(*buckets["som"], streams=[side])
            # a dedicated event: its wait is deferred to the end of the decoder backward, by which time a pooled
            # (round-robin) event could have been re-recorded for something else (deep decoders)
            som_done = self.__dict__.get("_som_done_ev")
            if som_done is None:
                som_done = self.__dict__["_som_done_ev"] = Event()
            som_done.record(side)
            self.vit._decoder_bwd(a, Gv, self._WT, before_dxe=lambda: som_done.wait())
            if "decoder" in buckets:
                flush()
                self._reduce_early(*buckets["decoder"], streams=streams_now())

        def on_block(i):
            b = buckets.get(f"enc{i}")
            if b is not None:
                flush()
                self._reduce_early(*b, streams=streams_now())

        self.vit._encoder_bwd(a, Gv, self._WT, on_block if buckets else None)
        flush()
        if self.vit._side is not None:
            stream_wait_stream(None, self.vit._side)     # every gradient is final from here on
            self.vit.__dict__.setdefault("_side_pending", []).clear()

    # -- data-parallel exchange ----------------------------------------------------------------
    # -- reference API ---------------------------------------------------------------------------
    def _schedules_for_step(self):
        self.som_layer.update_temperature(self._it)                     # vit_som.py:84 (iteration BEFORE increment)
        return self._gamma_t(), float(self.som_layer.current_temperature)

    # -- launch tape: the step's ~420 launches recorded once (while they run) and re-issued from C -------------------
    # Segments: 0 = forward up to the distances, 1 = main loss, 2 = loss-seed scaling (autograd bridge only), 3 = the whole
    # backward; the neighbourhood kernel and the loss combination sit in the holes between 0 | 1 | 2 and are issued from here
    # with this step's temperature and gamma.  Inputs are staged into fixed buffers; every other buffer of the step is
    # persistent per batch size, so the tape lives and dies with the activation buffers (`a`).
    def _tape_key(self):
        return (ops.get_gemm_mode(), ops.get_attention_fused(), hooks.signature(),
                self.world_size, self._use_vsom_comm, id(self.arena), id(self.som_layer._wplanes))

    def _tape_usable(self, x) -> bool:
        return bool(hooks.launch_tape and x.is_cuda and (self.world_size == 1 or self._use_vsom_comm) and ops.tape_recording() == 0)

    def _stage_inputs(self, x, y, a: _Acts):
        if not hasattr(a, "x_in"):
            a.x_in = torch.empty(a.B, self.vit.in_chans, self.vit.img_size, self.vit.img_size, dtype=torch.float32, device=a.device)
            a.y_in = torch.zeros(a.B, dtype=torch.int64, device=a.device)
            a.gout_in = torch.ones(1, dtype=torch.float32, device=a.device)
            a.steps_seen = 0
        a.x_in.copy_(x)
        if self.classification:
            a.y_in.copy_(y.view(-1))
        return a.x_in, a.y_in

    @torch.no_grad()
    def _taped_forward(self, x, y, gamma_t: float, T: float):
        """Forward + losses (+ the backward too while the tape is being recorded).  Returns (total, backward_done)."""
        x = self.vit._check_input(x)
        a = self.vit._buffers_for(x.shape[0], x.device)
        xs, ys = self._stage_inputs(x, y, a)
        tape = a.__dict__.get("tape")
        # the tape holds raw pointers: it is valid only for the buffers (ViT activations `a`, SOM buffers `s`, arenas) and the
        # switches it was recorded with
        if tape is not None and (tape.key != self._tape_key() or tape.som_bufs is not self.som_layer._bufs.get(a.B)):
            tape.close()
            tape = a.tape = None
        if hooks.adamw_planes and self.som_layer._planes_shape_ok(a.B):
            self.som_layer._w_planes()                # outside the tape: launches only when the optimizer-kept image is stale
        if tape is None:
            a.steps_seen += 1
            if a.steps_seen <= 2:                     # host-driven: scratch buffers and lazily built tables settle first
                return self._forward_losses(xs, ys, gamma_t, T, want_grad=True), False
            tid = ops.tape_begin()
            try:
                total = self._forward_losses(xs, ys, gamma_t, T, want_grad=True)          # segments 0 | hole | 1 | hole | 2 ...
                a.gout_in.fill_(1.0)
                self._scale_seeds(a.gout_in)                                              # ... segment 2 (x 1.0: exact no-op)
                ops.tape_cut()
                self._backward()                                                          # segment 3
            finally:
                nseg = ops.tape_end()
            a.tape = _StepTape(tid, nseg, self._tape_key(), list(self._started), self._comm_dirty, self.vit._side, self._ctx[2])
            return total, True
        # replay: the host-side state a host-driven step leaves behind, then segment | hole | segment | hole
        s = tape.som_bufs
        a.version += 1
        self._ctx = (xs, a, s)
        self._forward_id, self._seeds_consumed = self._forward_id + 1, False
        ops.tape_replay(tape.id, 0)
        self._call_neigh(s, gamma_t, T, a.B, True)
        ops.tape_replay(tape.id, 1)
        return self._call_parts(a, s, gamma_t, T, a.B, xs.numel(), True), False

    @torch.no_grad()
    def _taped_backward(self, gout=None):
        """The backward of the last taped forward (segment 3; segment 2 first when a loss seed other than 1 comes in)."""
        _, a, _ = self._ctx
        tape = a.tape
        self._grads_reduced = False
        self._exchange_reset()
        if gout is not None:
            a.gout_in.copy_(gout.detach().reshape(1))
            ops.tape_replay(tape.id, 2)
        ops.tape_replay(tape.id, 3)
        self._started, self._comm_dirty, self.vit._side = list(tape.started), tape.comm_dirty, tape.side

    def training_step(self, batch, batch_idx):
        """vit_som.py:80-105.  Returns a scalar tensor; ``.backward()`` runs the HIP backward."""
        x, y = batch
        self._estimated_steps()
        gamma_t, T = self._schedules_for_step()
        if self._anchor is None:
            self._anchor = torch.zeros((), device=self.arena.device, requires_grad=True)
        total = _StepLoss.apply(self._anchor, self, x, y, gamma_t, T)
        self._advance()
        if _HAVE_PL and getattr(self, "_trainer", None) is not None:          # vit_som.py:91,95-102
            main = "train/cls_loss" if self.classification else "train/recon_loss"
            self._log("hp/gamma", gamma_t)
            self._log({main: self._last["main"], "train/som_loss": self._last["som"], "train/total_loss": self._last["total"]})
        return total

    def train_step_fused(self, x, y):
        """Same step without the autograd bridge: forward + losses + backward into the gradient
        arena (the caller then runs optimizer.step()).  Returns the loss tensor."""
        self._estimated_steps()
        gamma_t, T = self._schedules_for_step()
        if self._tape_usable(x):
            total, backward_done = self._taped_forward(x, y, gamma_t, T)
            if not backward_done:
                if self._ctx[1].__dict__.get("tape") is not None:
                    self._taped_backward()
                else:
                    self._backward()
        else:
            total = self._forward_losses(x, y, gamma_t, T, want_grad=True)
            self._backward()
        self._advance()
        return total

    def _advance(self):
        self._it += 1                                                   # the device buffer moved with the loss (_forward_losses)

    def validation_step(self, batch, batch_idx):
        """vit_som.py:107-125 (full gamma, current temperature, no schedule update)."""
        x, y = batch
        total = self._forward_losses(x, y, float(self.gamma), float(self.som_layer.current_temperature), want_grad=False)
        if self.classification:
            a = self._ctx[1]
            self._last["acc"] = (a.logits.argmax(dim=-1) == y.view(-1)).float().mean()
        if _HAVE_PL and getattr(self, "_trainer", None) is not None:          # vit_som.py:116-123
            main = "val/cls_loss" if self.classification else "val/recon_loss"
            logs = {main: self._last["main"], "val/som_loss": self._last["som"], "val/total_loss": self._last["total"]}
            if self.classification:
                logs["val/accuracy"] = self._last["acc"]
            self._log(logs)
        return total.clone()

    def configure_optimizers(self):
        """vit_som.py:127-163: AdamW/Adam (lr * batch_size / 256), reference param groups, per-epoch
        LambdaLR with the warm-up / cosine multiplier floored at min_lr."""
        hp = self.config["hyperparameters"]
        opt_hp = hp["optimizer"]
        groups = param_groups_lrd(self.vit, weight_decay=opt_hp["weight_decay"], layer_decay=opt_hp["layer_decay"])
        other = list(self.som_layer.parameters())
        if self.classification:
            other.extend(list(self.cls_head.parameters()))
        groups.append({"params": other})
        if opt_hp["type"] not in ("adamw", "adam"):
            raise ValueError(f"unsupported optimizer type {opt_hp['type']!r}")
        optimizer = FusedAdamW(self, groups, lr=opt_hp["lr"] * hp["batch_size"] / 256,
                               betas=(opt_hp["beta_1"], opt_hp["beta_2"]), adamw=(opt_hp["type"] == "adamw"))
        if opt_hp["scheduler"] != "cosine_annealing":
            raise ValueError(f"unsupported scheduler {opt_hp['scheduler']!r}")
        lr_func = lambda epoch: max(opt_hp["min_lr"], min((epoch + 1) / (opt_hp["warmup_epochs"] + 1e-8),   # noqa: E731
                                                           0.5 * (math.cos(epoch / hp["total_epochs"] * math.pi) + 1)))
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lr_func)
        return [optimizer], [scheduler]

    # -- checkpoints: Lightning's .ckpt layout --------------------------------------
    def save_checkpoint(self, path, optimizer=None, scheduler=None, epoch=0, global_step=None):
        """Write a file with the keys a Lightning ModelCheckpoint writes (train_vit_som.py:81-84):
        state_dict (reference key names), hyper_parameters (= the config dict, vit_som.py:26),
        optimizer_states / lr_schedulers, epoch, global_step."""
        ckpt = {
            "epoch": int(epoch), "global_step": int(self._it if global_step is None else global_step),
            "pytorch-lightning_version": "2.2.1", "hparams_name": "config",
            "state_dict": {k: v.detach().cpu().clone() for k, v in self.state_dict().items()},
            "hyper_parameters": self.config,
            "optimizer_states": [optimizer.state_dict()] if optimizer is not None else [],
            "lr_schedulers": [scheduler.state_dict()] if scheduler is not None else [],
        }
        for st in ckpt["optimizer_states"]:
            for s in st["state"].values():
                for k2 in ("exp_avg", "exp_avg_sq"):
                    s[k2] = s[k2].cpu()
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        torch.save(ckpt, path)
        return path

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, config=None, device=None, map_location=None):
        """ViTSOM.load_from_checkpoint(path, config=config) (train_vit_som.py:111).  Only loaders
        that execute nothing from the file are used (torch.load(weights_only=True))."""
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        if config is None:
            config = ckpt.get("hyper_parameters")
            if config is None:
                raise ValueError("checkpoint carries no hyper_parameters; pass config=")
        model = cls(config, device=device)
        model.load_state_dict(ckpt["state_dict"])
        model._loaded_checkpoint = ckpt
        return model

    def on_train_end(self):                                             # vit_som.py:165-172
        print(f"Peak GPU memory usage: {torch.cuda.max_memory_allocated() / 1e9:.4f} GB")

    def get_latent_representation(self, x):
        """vit_som.py:174-187 (the reference unpacks 4 of 3 values; this returns what it meant)."""
        with torch.no_grad():
            cls_token, patches, _ = self.vit(x)
            return cls_token if self.use_reduced else patches.flatten(start_dim=1)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        key = prefix + "iteration"
        if key in state_dict:
            self._it = int(state_dict[key])
