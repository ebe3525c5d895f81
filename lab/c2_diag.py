import copy, sys, torch
sys.path.insert(0, "/root/repo")
import vit_som_amd
from oracle.gen_golden import make_config
from vit_som_amd.tuning import hooks
DEV = "cuda"
cfg = make_config(3, 32, 4, 192, 12, 3, 96, 2, (24, 24), 0, 512, gamma=0.01, Tmax=4.0, Tmin=0.1)
g = torch.Generator().manual_seed(17)
x = torch.randn(512, 3, 32, 32, generator=g).to(DEV); y = torch.zeros(512, dtype=torch.int64, device=DEV)
def run(scale, **hk):
    hooks.set(**hk)
    torch.manual_seed(0)
    m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device=DEV)
    m.set_schedule(60000, 9000); m._it = 800
    loss = m.training_step((x, y), 0)
    (loss * scale).backward()
    torch.cuda.synchronize()
    hooks.reset()
    return m, m.arena.grads.clone()
def diff(m, a, b, tag):
    bad = []
    for n, (off, cnt, _) in m.arena.offsets.items():
        if not torch.equal(a[off:off + cnt], b[off:off + cnt]):
            d = (a[off:off + cnt] - b[off:off + cnt]).abs().max().item(); bad.append((n, d, a[off:off+cnt].abs().max().item()))
    print(tag, "differing tensors:", len(bad), bad[:8])
m, g1 = run(1.0); _, g1b = run(1.0); _, g2 = run(2.0)
diff(m, g1, g1b, "same seed twice:")
diff(m, 2 * g1, g2, "2x seed:")
_, s1 = run(1.0, side_stream=False); _, s2 = run(2.0, side_stream=False)
diff(m, 2 * s1, s2, "2x seed, no side streams:")
diff(m, g1, s1, "side vs no side:")
