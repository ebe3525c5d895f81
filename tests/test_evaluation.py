"""evaluation on device.  CPU: the table -> metric arithmetic against sklearn (the
library the reference calls, tools/evaluation.py:16,46,122) and against a restatement of the
reference's own calculate_purity loop.  GPU: the contingency / argmax kernels and the whole
evaluate_* passes against the CPU oracle's BMUs / logits."""
import copy

import numpy as np
import pytest
import torch

from helpers import golden_params, load_golden


def purity_reference_loop(y_trues, y_preds):
    """tools/evaluation.py:130-151 restated (the O(N) loops, then accuracy of the voted labels)."""
    y_trues = y_trues.astype(np.int64)
    D = max(y_preds.max(), y_trues.max()) + 1
    w = np.zeros((D, D), dtype=np.int64)
    for i in range(y_preds.size):
        w[y_preds[i], y_trues[i]] += 1
    mapping = w.argmax(axis=1)
    voted = np.array([mapping[p] for p in y_preds])
    return float((voted == y_trues).mean()), w


@pytest.mark.parametrize("seed,npred,ntrue,n", [(0, 16, 10, 500), (1, 576, 10, 2000), (2, 3, 7, 50), (3, 1, 1, 10), (4, 40, 5, 37)])
def test_table_metrics_match_sklearn(seed, npred, ntrue, n):
    from sklearn.metrics import normalized_mutual_info_score
    from vit_som_amd.evaluation import nmi_from_table, purity_from_table
    rng = np.random.default_rng(seed)
    yt = rng.integers(0, ntrue, n)
    yp = (yt * 3 + rng.integers(0, 3, n)) % npred if npred > 1 else np.zeros(n, dtype=np.int64)
    pur, w = purity_reference_loop(yt, yp)
    assert abs(purity_from_table(w) - pur) < 1e-12
    assert abs(nmi_from_table(w) - normalized_mutual_info_score(yt, yp)) < 1e-10


@pytest.mark.parametrize("seed,ncls,n", [(0, 10, 400), (1, 100, 300), (2, 5, 20)])
def test_classification_metrics_match_sklearn(seed, ncls, n):
    from sklearn.metrics import accuracy_score, precision_recall_fscore_support
    from vit_som_amd.evaluation import classification_from_table
    rng = np.random.default_rng(seed)
    yt = rng.integers(0, ncls, n)
    yp = np.where(rng.random(n) < 0.6, yt, rng.integers(0, max(ncls // 2, 1), n))     # some classes never predicted
    cm = np.zeros((ncls, ncls), dtype=np.int64)
    np.add.at(cm, (yt, yp), 1)
    acc, p, r, f = classification_from_table(cm)
    P, R, F1, _ = precision_recall_fscore_support(yt, yp, average="macro", zero_division=np.nan)
    assert abs(acc - accuracy_score(yt, yp)) < 1e-12
    assert abs(p - P) < 1e-12 and abs(r - R) < 1e-12 and abs(f - F1) < 1e-12


@pytest.mark.gpu
def test_contingency_and_argmax_kernels():
    from vit_som_amd import ops
    g = torch.Generator().manual_seed(0)
    a = torch.randint(0, 37, (10007,), generator=g)
    b = torch.randint(0, 11, (10007,), generator=g)
    t = torch.zeros(37, 11, dtype=torch.int64, device="cuda"); bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    ops.contingency(a.cuda(), b.cuda(), t, bad)
    ops.contingency(a.cuda(), b.cuda(), t, bad)                       # accumulates
    ref = np.zeros((37, 11), dtype=np.int64); np.add.at(ref, (a.numpy(), b.numpy()), 2)
    assert np.array_equal(t.cpu().numpy(), ref) and int(bad) == 0
    ops.contingency(torch.tensor([40, -1, 3]).cuda(), torch.tensor([0, 0, 11]).cuda(), t, bad)
    assert int(bad) == 3 and np.array_equal(t.cpu().numpy(), ref)
    x = torch.randn(1001, 100, generator=g); x[5, 7] = x[5, 50] = 99.0          # tie -> first index
    out = torch.empty(1001, dtype=torch.int64, device="cuda")
    ops.argmax_rows(x.cuda(), out)
    assert torch.equal(out.cpu(), x.argmax(1)) and int(out[5]) == 7


class _Loader(list):
    pass


@pytest.mark.gpu
def test_evaluate_clustering_and_classification_against_oracle():
    import vit_som_amd
    from oracle import vitsom_oracle as O
    from sklearn.metrics import accuracy_score, normalized_mutual_info_score, precision_recall_fscore_support
    from vit_som_amd.evaluation import calculate_purity, evaluate_classification, evaluate_clustering
    z, cfg = load_golden("ref_cls_tiny")
    P = golden_params(z)
    d = O.Dims(cfg)
    g = torch.Generator().manual_seed(3)
    batches = _Loader((torch.randn(7, d.C, d.img, d.img, generator=g), torch.randint(0, d.num_classes, (7,), generator=g)) for _ in range(6))
    m = vit_som_amd.ViTSOM(copy.deepcopy(cfg), device="cuda:0")
    m.load_state_dict(P)
    bm, lg, ys = [], [], []
    for x, y in batches:
        _, _, logits, _, bmu = O.forward(P, x, d)
        bm.append(bmu); lg.append(logits.argmax(1)); ys.append(y)
    bm, lg, ys = torch.cat(bm).numpy(), torch.cat(lg).numpy(), torch.cat(ys).numpy()
    purity, nmi, _ = evaluate_clustering(m, cfg, batches)
    assert abs(purity - purity_reference_loop(ys, bm)[0]) < 1e-12
    assert abs(nmi - normalized_mutual_info_score(ys, bm)) < 1e-10
    acc, p, r, f1, _ = evaluate_classification(m, cfg, batches)
    P_, R_, F_, _ = precision_recall_fscore_support(ys, lg, average="macro", zero_division=np.nan)
    assert abs(acc - accuracy_score(ys, lg)) < 1e-12 and abs(p - P_) < 1e-12 and abs(r - R_) < 1e-12 and abs(f1 - F_) < 1e-12
    assert abs(calculate_purity(ys, bm) - purity) < 1e-12
    # clustering configs carry num_classes == 0 and loaders may be label-sorted: the first batch then lacks the
    # largest label and the table has to grow (the reference accepts arbitrary labels)
    cfg0 = copy.deepcopy(cfg); cfg0["data"]["num_classes"] = 0
    order = np.argsort(ys, kind="stable")
    xs_all = torch.cat([x for x, _ in batches])[order]
    sorted_batches = _Loader((xs_all[i:i + 7], torch.from_numpy(ys[order][i:i + 7])) for i in range(0, len(ys), 7))
    p_sorted, n_sorted, _ = evaluate_clustering(m, cfg0, sorted_batches)
    assert abs(p_sorted - purity) < 1e-12 and abs(n_sorted - nmi) < 1e-10
    # fast path == full forward
    x0 = batches[0][0].cuda()
    bmu_fast, logits_fast = m.predict(x0)
    bmu_fast, logits_fast = bmu_fast.clone(), logits_fast.clone()
    full = m(x0)
    assert torch.equal(full[4], bmu_fast) and torch.equal(full[2], logits_fast)


@pytest.mark.gpu
def test_evaluation_drives_the_desom_client():
    """evaluate_clustering / evaluate_classification (evaluation.py:38-39,115-116) on DESOM: same
    decoder-free predict() contract as ViTSOM; BMUs / predictions against the DESOM oracle."""
    import vit_som_amd
    from oracle import desom_oracle as D
    from sklearn.metrics import accuracy_score, normalized_mutual_info_score
    from vit_som_amd.evaluation import evaluate_classification, evaluate_clustering
    z, cfg = load_golden("ref_desom_cls_tiny")
    P = golden_params(z)
    d = cfg["data"]
    g = torch.Generator().manual_seed(4)
    batches = _Loader((torch.rand(9, d["num_channels"], d["input_size"], d["input_size"], generator=g),
                       torch.randint(0, d["num_classes"], (9,), generator=g)) for _ in range(5))
    m = vit_som_amd.DESOM(copy.deepcopy(cfg), device="cuda:0")
    m.load_state_dict(P)
    bm, lg, ys = [], [], []
    for x, y in batches:
        logits, _, _, bmu = D.forward(P, cfg, x)
        bm.append(bmu); lg.append(logits.argmax(1)); ys.append(y)
    bm, lg, ys = torch.cat(bm).numpy(), torch.cat(lg).numpy(), torch.cat(ys).numpy()
    purity, nmi, _ = evaluate_clustering(m, cfg, batches)
    assert abs(purity - purity_reference_loop(ys, bm)[0]) < 1e-12
    assert abs(nmi - normalized_mutual_info_score(ys, bm)) < 1e-10
    acc, *_ = evaluate_classification(m, cfg, batches)
    assert abs(acc - accuracy_score(ys, lg)) < 1e-12
