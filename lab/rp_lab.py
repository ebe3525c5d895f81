"""Row-panel engine (csrc/linear_rp.hip) against the tile engine (gemm_x6.h) on the eight NT GEMMs of one encoder layer at the
c3 shapes: relative Frobenius error vs fp64 for both, time alone for both.   python lab/rp_lab.py [splits ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vit_som_amd import ops
T, E, H4 = int(os.environ.get("RP_T", 512 * 65)), 192, 768
if len(sys.argv) > 1 and sys.argv[1].startswith("T="):
    T = int(sys.argv.pop(1)[2:])
dev = "cuda"
torch.manual_seed(0)
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
r = lambda *s: torch.randn(*s, device=dev)
def image(W, transpose):
    """W [n, k] nn.Linear weight; forward image (transpose=0): n_out = n, k_red = k; input-gradient image: n_out = k, k_red = n"""
    n_out, k_red = (W.shape[1], W.shape[0]) if transpose else W.shape
    kind = ops.weight_image_kind(n_out, k_red)
    assert kind, (n_out, k_red)
    nb = ops.weight_image_bytes(n_out, k_red)
    img = torch.empty(nb, dtype=torch.uint8, device=dev)
    table = torch.tensor([[0, 0, n_out, k_red, transpose, kind]], dtype=torch.int64, device=dev)
    ops.weight_images_prepare(W.contiguous().view(-1), img, table, ((n_out + 31) // 32) * ((k_red + 31) // 32 * 2))
    return img
def rel(a, b): return float((a.double() - b).norm() / b.norm())
x, x3, x4 = r(T, E), r(T, 3 * E), r(T, H4)
Wqkv, Wp, W1, W2 = r(3 * E, E) * 0.05, r(E, E) * 0.05, r(H4, E) * 0.05, r(E, H4) * 0.05
b3, b1, b4 = r(3 * E), r(E), r(H4)
res = r(T, E)
gam, bet = 1 + 0.1 * r(E), 0.1 * r(E)
gp = r(T, H4)
cases = []
def add(name, X, W, transpose, bias, epi, N, **kw): cases.append((name, X, W, transpose, bias, epi, N, kw))
add("fwd qkv", x, Wqkv, 0, b3, ops.EPI_BIAS, 576)
add("fwd qkv +LN", x, Wqkv, 0, b3, ops.EPI_BIAS, 576, ln=True)
add("fwd proj +res", x, Wp, 0, b1, ops.EPI_BIAS_RES, 192, R=res)
add("fwd fc1 +gelu", x, W1, 0, b4, ops.EPI_BIAS_GELU, 768, two=True)
add("fwd fc1 +gelu +LN", x, W1, 0, b4, ops.EPI_BIAS_GELU, 768, two=True, ln=True)
add("fwd fc2 +res", x4, W2, 0, b1, ops.EPI_BIAS_RES, 192, R=res)
add("dX fc2 x gelu'", x, W2, 1, None, ops.EPI_GELU_BWD, 768, R=gp)
add("dX fc1", x4, W1, 1, None, ops.EPI_NONE, 192)
add("dX proj", x, Wp, 1, None, ops.EPI_NONE, 192)
add("dX qkv", x3, Wqkv, 1, None, ops.EPI_NONE, 192)
tot_new = tot_old = 0.0
for name, X, W, tr, bias, epi, N, kw in cases:
    img = image(W, tr)
    out, out2 = torch.empty(T, N, device=dev), (torch.empty(T, N, device=dev) if kw.get("two") else None)
    mean, rstd = torch.empty(T, device=dev), torch.empty(T, device=dev)
    ln = (gam, bet, 1e-6, mean, rstd, None) if kw.get("ln") else None
    fn = lambda: ops.linear_planes(X, img, bias, out, epilogue=epi, R=kw.get("R"), out2=out2, ln=ln)
    fn(); torch.cuda.synchronize()
    # fp64 reference
    A = X.double()
    if ln is not None:
        A = torch.nn.functional.layer_norm(A, (E,), gam.double(), bet.double(), 1e-6)
    Bm = (W.double() if tr else W.double().t())            # [K_red, N_out]
    acc = A @ Bm
    if bias is not None: acc = acc + bias.double()
    if epi == ops.EPI_BIAS_RES: ref = acc + kw["R"].double()
    elif epi == ops.EPI_GELU_BWD: ref = acc * kw["R"].double()
    elif epi == ops.EPI_BIAS_GELU: ref = torch.nn.functional.gelu(acc)
    else: ref = acc
    got = out2 if epi == ops.EPI_BIAS_GELU else out
    e_new = rel(got, ref)
    # old engine
    o2, o3 = torch.empty_like(out), torch.empty_like(out)
    Xo = X
    if ln is not None:
        Xo = torch.empty_like(X); ops.layernorm_fwd(X, gam, bet, Xo, mean, rstd)
    if tr:
        Wt = W.t().contiguous()
        old = lambda: ops.linear_bwd_input_t(Xo, Wt, o2, gelu_grad=kw.get("R") if epi == ops.EPI_GELU_BWD else None)
    elif epi == ops.EPI_BIAS: old = lambda: ops.linear_fwd(Xo, W, bias, o2)
    elif epi == ops.EPI_BIAS_RES: old = lambda: ops.linear_residual_fwd(Xo, W, bias, kw["R"], T, o2)
    else: old = lambda: ops.linear_gelu_fwd(Xo, W, bias, o2, o3)
    old(); torch.cuda.synchronize()
    e_old = rel(o3 if epi == ops.EPI_BIAS_GELU else o2, ref)
    us_new, us_old = t_us(fn), t_us(old)
    if ln is not None:
        us_old += t_us(lambda: ops.layernorm_fwd(X, gam, bet, Xo, mean, rstd))
    tot_new += us_new if "LN" not in name else 0; tot_old += us_old if "LN" not in name else 0
    fl = 2.0 * T * N * W.shape[1 if not tr else 0]
    print(f"{name:22s} rp {us_new:7.1f} us ({fl/us_new/1e6:6.1f} TF)  err {e_new:.2e} | tile engine {us_old:7.1f} us  err {e_old:.2e}", flush=True)
print(f"sum (without the +LN rows) rp {tot_new:.0f} us, tile engine {tot_old:.0f} us")
