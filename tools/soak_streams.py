import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vit_som_amd import ViTSOM
finals = []
for side in ("0", "1", "1"):
    from vit_som_amd.tuning import hooks
    hooks.set(side_stream=side == "1", fwd_split=side == "1")
    torch.manual_seed(0)
    m = ViTSOM(bench.c3_config(512), device="cuda")
    m.set_schedule(50000, 10000); m._it = 1000
    (opt,), _ = m.configure_optimizers()
    g = torch.Generator().manual_seed(7)
    for s in range(15):
        x = torch.randn(512, 3, 32, 32, generator=g).cuda(); y = torch.zeros(512, dtype=torch.int64, device="cuda")
        loss = m.train_step_fused(x, y); opt.step()
    finals.append((m.arena.params.clone(), float(loss)))
    print("side", side, "loss", float(loss))
print("bitwise equal off/on:", torch.equal(finals[0][0], finals[1][0]), " on/on:", torch.equal(finals[1][0], finals[2][0]))
