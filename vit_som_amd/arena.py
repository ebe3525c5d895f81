"""Flat fp32 parameter arenas.

Every trainable tensor of the model is a view into ONE contiguous buffer (and its gradient /
Adam moments into three parallel ones), each tensor starting on a 256-float (1 KiB) boundary:
 * the data-parallel exchange is a single RCCL all-reduce over ``grads`` (ViT gradients and the
   [K, L] prototype accumulators together);
 * AdamW is a single kernel over the arena, weight decay looked up per 256-float chunk;
 * ``state_dict`` keys / shapes stay exactly the reference's (the views are the Parameters).
"""
from typing import Dict, List, Tuple

import torch

CHUNK = 256


class ParamArena:
    def __init__(self, specs: List[Tuple[str, Tuple[int, ...], float]], device):
        """specs: (name, shape, weight_decay) in arena order."""
        self.device = torch.device(device)
        self.offsets: Dict[str, Tuple[int, int, Tuple[int, ...]]] = {}
        off = 0
        wd_chunks: List[float] = []
        for name, shape, wd in specs:
            n = 1
            for s in shape:
                n *= int(s)
            padded = (n + CHUNK - 1) // CHUNK * CHUNK
            self.offsets[name] = (off, n, tuple(int(s) for s in shape))
            wd_chunks += [float(wd)] * (padded // CHUNK)
            off += padded
        self.numel = off
        self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.wd_chunk = torch.tensor(wd_chunks, dtype=torch.float32, device=self.device)
        self.wd_by_name = {name: float(wd) for name, _, wd in specs}

    def view(self, buf: torch.Tensor, name: str) -> torch.Tensor:
        off, n, shape = self.offsets[name]
        return buf[off:off + n].view(shape)

    def p(self, name: str) -> torch.Tensor:
        return self.view(self.params, name)

    def g(self, name: str) -> torch.Tensor:
        return self.view(self.grads, name)

    def set_weight_decay(self, name: str, wd: float):
        off, n, _ = self.offsets[name]
        padded = (n + CHUNK - 1) // CHUNK * CHUNK
        self.wd_chunk[off // CHUNK:(off + padded) // CHUNK] = wd
        self.wd_by_name[name] = wd
