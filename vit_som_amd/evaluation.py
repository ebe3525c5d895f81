"""On-device evaluation mirroring the reference's tools/evaluation.py entry points
(evaluate_clustering :18-52, evaluate_classification :93-128, calculate_purity :130-151).

The model forward runs on the HIP kernels (encoder + SOM only: `ViTSOM.predict`); BMU / label
pairs are folded into a contingency table ON DEVICE batch by batch (`vsom_contingency`, integer
atomics), so the only device->host traffic of a whole evaluation pass is that table.  Purity, NMI
(sklearn's arithmetic-mean normalisation) and the macro precision / recall / F1 are O(classes^2)
host arithmetic on the table.
"""
import time

import numpy as np
import torch

from . import ops


class _Table:
    def __init__(self, na, nb, device):
        self.table = torch.zeros(na, nb, dtype=torch.int64, device=device)
        self.bad = torch.zeros(1, dtype=torch.int32, device=device)

    def add(self, a, b):
        ops.contingency(a.contiguous().view(-1), b.contiguous().view(-1).long(), self.table, self.bad)

    def grow_columns(self, nb):
        """Widen the table to `nb` columns (labels are not known in advance for the clustering configs)."""
        if nb > self.table.shape[1]:
            t = torch.zeros(self.table.shape[0], nb, dtype=torch.int64, device=self.table.device)
            t[:, :self.table.shape[1]] = self.table
            self.table = t

    def numpy(self, world_size=1):
        """The table on the host -- summed over the ranks first when every rank folded its own shard of the data
        (one all-reduce of the integer table: every rank then reports the metrics of the WHOLE set)."""
        if world_size > 1:
            import torch.distributed as dist
            both = torch.cat([self.table.view(-1), self.bad.view(-1).long()])
            dist.all_reduce(both)
            self.table, self.bad = both[:-1].view_as(self.table), both[-1:].int()
        nbad = int(self.bad.item())
        if nbad:
            raise ValueError(f"{nbad} label/prediction values fell outside the contingency table "
                             f"({self.table.shape[0]} x {self.table.shape[1]}); pass num_labels= for larger label sets")
        return self.table.cpu().numpy()


def purity_from_table(w: np.ndarray) -> float:
    """evaluation.py:142-151: majority-vote label per predicted cluster, then accuracy."""
    n = w.sum()
    return float(w.max(axis=1).sum() / n) if n else float("nan")


def nmi_from_table(w: np.ndarray) -> float:
    """sklearn.metrics.normalized_mutual_info_score(average_method='arithmetic') from a contingency table."""
    w = w.astype(np.float64)
    n = w.sum()
    a, b = w.sum(axis=1), w.sum(axis=0)
    na, nb = int((a > 0).sum()), int((b > 0).sum())
    if (na == 1 and nb == 1) or (na == 0 and nb == 0):
        return 1.0
    nz = w > 0
    outer = np.outer(a, b)
    mi = float((w[nz] / n * (np.log(w[nz]) - np.log(n) - (np.log(outer[nz]) - 2 * np.log(n)))).sum())
    mi = max(mi, 0.0)

    def ent(c):
        c = c[c > 0]
        return float(-(c / n * (np.log(c) - np.log(n))).sum())
    norm = max(0.5 * (ent(a) + ent(b)), np.finfo(np.float64).eps)
    return mi / norm


def classification_from_table(cm: np.ndarray):
    """accuracy + sklearn precision_recall_fscore_support(average='macro', zero_division=nan) from the
    confusion matrix cm[true, pred] over labels = union(y_true, y_pred)."""
    cm = cm.astype(np.float64)
    present = (cm.sum(axis=0) + cm.sum(axis=1)) > 0
    cm = cm[present][:, present]
    tp = np.diag(cm)
    pred_sum, true_sum = cm.sum(axis=0), cm.sum(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = np.where(pred_sum > 0, tp / pred_sum, np.nan)      # zero_division=nan: excluded from the macro mean
        recall = np.where(true_sum > 0, tp / true_sum, np.nan)
        f1 = 2 * tp / (true_sum + pred_sum)                            # sklearn: (1+b^2) tp / (b^2 true + pred), never 0/0 here
    acc = float(tp.sum() / cm.sum()) if cm.sum() else float("nan")
    return acc, float(np.nanmean(precision)), float(np.nanmean(recall)), float(np.nanmean(f1))


def calculate_purity(y_trues, y_preds):
    """evaluation.py:130-151 (same signature); the contingency table is built on the device."""
    dev = y_preds.device if isinstance(y_preds, torch.Tensor) and y_preds.is_cuda else torch.device("cuda", torch.cuda.current_device())
    yt = torch.as_tensor(np.asarray(y_trues) if not isinstance(y_trues, torch.Tensor) else y_trues).to(dev).long().view(-1)
    yp = torch.as_tensor(np.asarray(y_preds) if not isinstance(y_preds, torch.Tensor) else y_preds).to(dev).long().view(-1)
    assert yp.numel() == yt.numel(), f"y_preds ({yp.numel()}) and y_trues ({yt.numel()}) must be the same size"
    D = int(max(int(yp.max()), int(yt.max())) + 1)
    t = _Table(D, D, dev)
    t.add(yp, yt)
    return purity_from_table(t.numpy())


def _world(model):
    return int(getattr(model, "world_size", 1))


def evaluate_clustering(model, config, dataloader, num_labels=None):
    """evaluation.py:18-52 -> (purity, nmi, inference_time) from the model's native BMU assignments.  No host
    synchronisation inside the loop: labels outside [0, num_labels) are counted on the device and reported once at the
    end (num_labels defaults to max(data.num_classes, 256)).  With model.world_size > 1 every rank folds its shard and
    the tables are summed, so all ranks return the metrics of the whole set."""
    model.eval()
    d = config["data"]
    C, S = d["num_channels"], d["input_size"]
    K = model.som_layer.n_prototypes
    dev = model.arena.device
    L = int(num_labels) if num_labels else max(int(d.get("num_classes", 0)), 256)
    table, start = _Table(K, L, dev), time.time()
    for x, y in dataloader:
        x = x.to(dev, non_blocking=True).reshape(-1, C, S, S)
        y = y.to(dev, non_blocking=True)
        bmu, _ = model.predict(x)
        table.add(bmu, y.view(-1))
    w = table.numpy(_world(model))
    purity, nmi = purity_from_table(w), nmi_from_table(w)
    inference_time = time.time() - start
    print(f"Purity: {purity:.3f}, NMI: {nmi:.3f}, Inference Time: {inference_time:.3f}")
    return purity, nmi, inference_time


def evaluate_classification(model, config, dataloader):
    """evaluation.py:93-128 -> (accuracy, precision, recall, f1, inference_time) (macro averages); summed over the
    ranks like evaluate_clustering."""
    model.eval()
    d = config["data"]
    C, S, ncls = d["num_channels"], d["input_size"], d["num_classes"]
    dev = model.arena.device
    table, pred, start = _Table(ncls, ncls, dev), None, time.time()
    for x, y in dataloader:
        x = x.to(dev, non_blocking=True).reshape(-1, C, S, S)
        y = y.to(dev, non_blocking=True)
        _, logits = model.predict(x)
        if pred is None or pred.numel() != logits.shape[0]:
            pred = torch.empty(logits.shape[0], dtype=torch.int64, device=dev)
        ops.argmax_rows(logits, pred)
        table.add(y.view(-1), pred)                       # cm[true, pred]
    acc, precision, recall, f1 = classification_from_table(table.numpy(_world(model)))
    inference_time = time.time() - start
    print(f"Accuracy: {acc:.3f}, Precision: {precision:.3f}, Recall: {recall:.3f}, F1-score: {f1:.3f}, Inference Time: {inference_time:.3f}")
    return acc, precision, recall, f1, inference_time
