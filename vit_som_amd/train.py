"""Training driver equivalent to the reference's experiments/benchmarking/train_vit_som.py:27-130
 without Lightning: n_runs independent runs, seed 0, per-epoch LambdaLR stepping,
validation loop + best-by-val/accuracy checkpoint in classification mode, last-epoch checkpoint in
clustering mode, final evaluation (classification metrics on the test loader / purity + NMI of the
reloaded last checkpoint on the train loader), mean (std) report.

Data comes from a `make_loaders(config, rank, world_size)` callable returning
(train_loader, val_loader, test_loader); the reference's data/data.py:get_dataloaders needs
torchvision/timm and downloads (absent offline), so the default is a synthetic, class-structured
in-memory set of the configured shape.  One process per GPU: under torchrun (WORLD_SIZE > 1) every
rank takes an interleaved shard of each loader and gradients are summed by one RCCL all-reduce.

    python -m vit_som_amd.train --config configs/vit_som/vit_som_cifar-10.yaml [--runs 5] [--epochs N]
"""
import argparse
import copy
import os
import shutil
import time

import numpy as np
import torch
import yaml

from .evaluation import evaluate_classification, evaluate_clustering
from .model import ViTSOM


def load_config(config_path="./configs/config.yaml"):
    """The YAML config as a nested dict; a non-empty DATASET_NAME in the environment replaces data.dataset
    (behaviour of tools/utils.py:14-26)."""
    with open(config_path) as fh:
        cfg = yaml.safe_load(fh)
    cfg["data"]["dataset"] = os.environ.get("DATASET_NAME") or cfg["data"]["dataset"]
    return cfg


def clear_directory(directory):
    """Start every run from an empty checkpoint directory (behaviour of train_vit_som.py:19-25)."""
    shutil.rmtree(directory, ignore_errors=True)
    os.makedirs(directory, exist_ok=True)


def _report(all_metrics, n_runs, dataset_name, log):
    """Mean (std) over the runs, in the reference's output format (train_vit_som.py:118-130)."""
    log(f"\n--- Aggregated Results Across {n_runs} Runs for {dataset_name} ---")
    timed = {"run_duration", "inference_time"}
    for key, scores in all_metrics.items():
        if not scores:
            continue
        mean, std = float(np.mean(scores)), float(np.std(scores))
        label = key.capitalize()
        log(f"Avg {label} (Std): {mean:.2f}s ({std:.2f}s)" if key in timed else f"{label} Mean (Std): {mean:.4f} ({std:.4f})")


class TensorLoader:
    """Minimal DataLoader stand-in over in-memory tensors: fixed batch size, optional per-epoch
    shuffle, rank-interleaved sharding; exposes .dataset with __len__ (som_layer.py:131)."""

    def __init__(self, x, y, batch_size, shuffle=False, rank=0, world_size=1, seed=0, drop_last=False):
        self.x, self.y, self.batch_size, self.shuffle = x, y, int(batch_size), shuffle
        self.rank, self.world, self.seed, self.drop_last, self.epoch = rank, world_size, seed, drop_last, 0
        self.dataset = torch.utils.data.TensorDataset(x, y)

    def __len__(self):
        n = len(self.dataset) // self.world
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        n = len(self.dataset)
        idx = torch.randperm(n, generator=torch.Generator().manual_seed(self.seed + self.epoch)) if self.shuffle else torch.arange(n)
        self.epoch += 1
        idx = idx[: (n // self.world) * self.world][self.rank::self.world]
        for i in range(0, len(idx), self.batch_size):
            j = idx[i:i + self.batch_size]
            if self.drop_last and len(j) < self.batch_size:
                break
            yield self.x[j], self.y[j]


def synthetic_loaders(config, rank=0, world_size=1, n_train=2048, n_val=256, n_test=256, seed=0):
    """Class-structured synthetic images of the configured shape (each class = a fixed random
    template + noise), so that accuracy / purity move during training."""
    hp, d = config["hyperparameters"], config["data"]
    C, S = d["num_channels"], d["input_size"]
    ncls = max(int(d["num_classes"]), 1) if d["num_classes"] > 0 else 10
    g = torch.Generator().manual_seed(seed)
    templates = torch.randn(ncls, C, S, S, generator=g)

    def make(n):
        y = torch.randint(0, ncls, (n,), generator=g)
        return templates[y] + 0.5 * torch.randn(n, C, S, S, generator=g), y
    bs = hp["batch_size"]
    (xt, yt), (xv, yv), (xs, ys) = make(n_train), make(n_val), make(n_test)
    return (TensorLoader(xt, yt, bs, shuffle=True, rank=rank, world_size=world_size, seed=seed, drop_last=True),
            TensorLoader(xv, yv, bs, rank=rank, world_size=world_size), TensorLoader(xs, ys, bs, rank=rank, world_size=world_size))


def fit(model, config, train_loader, val_loader, ckpt_dir, dataset_name, use_validation, max_epochs=None, log=print):
    """The Lightning fit loop the reference relies on (train_vit_som.py:86-93), written out."""
    hp = config["hyperparameters"]
    epochs = int(max_epochs if max_epochs is not None else hp["total_epochs"])
    dev = model.arena.device
    steps_per_epoch = len(train_loader)
    model.set_schedule(len(train_loader.dataset), steps_per_epoch * epochs)     # trainer.estimated_stepping_batches
    (opt,), (sched,) = model.configure_optimizers()
    best_acc, best_path, last_path, history = -1.0, None, None, []
    for epoch in range(epochs):
        model.train()
        tot, nb = 0.0, 0
        for x, y in train_loader:
            loss = model.train_step_fused(x.to(dev, non_blocking=True), y.to(dev, non_blocking=True))
            opt.step()
            tot, nb = tot + loss, nb + 1                                 # device-side accumulation, no per-step sync
        sched.step()                                                     # LambdaLR, interval = epoch (vit_som.py:159-163)
        rec = {"epoch": epoch, "train/total_loss": float(tot / max(nb, 1)), "lr": opt.param_groups[0]["lr"]}
        if use_validation:
            model.eval()
            correct, seen, vloss, vb = 0.0, 0, 0.0, 0
            for x, y in val_loader:
                x, y = x.to(dev), y.to(dev)
                vloss, vb = vloss + model.validation_step((x, y), vb), vb + 1
                correct, seen = correct + model._last["acc"] * x.shape[0], seen + x.shape[0]
            if model.world_size > 1:                                     # every rank saw 1 / world of the validation set
                agg = torch.stack([torch.as_tensor(correct, dtype=torch.float32, device=dev).reshape(()),
                                   torch.tensor(float(seen), device=dev), torch.as_tensor(vloss, dtype=torch.float32, device=dev).reshape(()),
                                   torch.tensor(float(vb), device=dev)])
                torch.distributed.all_reduce(agg)
                correct, seen, vloss, vb = agg[0], float(agg[1]), agg[2], float(agg[3])
            rec["val/accuracy"] = float(correct / max(seen, 1))
            rec["val/total_loss"] = float(vloss / max(vb, 1))
            if model.rank == 0 and rec["val/accuracy"] > best_acc:        # ModelCheckpoint(monitor='val/accuracy', mode='max')
                best_acc = rec["val/accuracy"]
                best_path = model.save_checkpoint(os.path.join(ckpt_dir, f"vit_som_{dataset_name}_best.ckpt"), opt, sched, epoch)
        history.append(rec)
        log(" ".join(f"{k}={v:.5g}" if isinstance(v, float) else f"{k}={v}" for k, v in rec.items()))
    if not use_validation and model.rank == 0:                            # ModelCheckpoint(save_last=True)
        last_path = model.save_checkpoint(os.path.join(ckpt_dir, "last.ckpt"), opt, sched, epochs - 1)
    return {"history": history, "best_model_path": best_path, "last_model_path": last_path, "optimizer": opt}


def main(config, n_runs=5, max_epochs=None, make_loaders=synthetic_loaders, model_states_dir="experiments/states/vit_som",
         log=print):
    """train_vit_som.py:27-130."""
    hp, data_hp = config["hyperparameters"], config["data"]
    use_validation = data_hp["num_classes"] > 0
    dataset_name = data_hp["dataset"]
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(os.environ.get("VSOM_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    torch.manual_seed(0)                                                  # pl.seed_everything(0)
    np.random.seed(0)
    all_metrics = {k: [] for k in ("accuracy", "precision", "recall", "f1", "purity", "nmi", "run_duration", "inference_time")}
    for run in range(n_runs):
        log(f"Starting run {run + 1} for {dataset_name}...")
        start = time.time()
        if rank == 0:
            clear_directory(model_states_dir)
        train_loader, val_loader, test_loader = make_loaders(config, rank, world)
        model = ViTSOM(copy.deepcopy(config))
        model.set_distributed(world, rank)
        model.broadcast_parameters()                                      # DDP broadcasts rank 0's weights at construction
        out = fit(model, config, train_loader, val_loader, model_states_dir, dataset_name, use_validation, max_epochs, log)
        torch.cuda.synchronize()
        run_duration = time.time() - start
        log(f"Run {run + 1} duration: {run_duration:.2f} seconds")
        # Final evaluation: every rank folds ITS shard of the loader into the contingency table and the tables are summed
        # over the ranks (evaluation.py), so all ranks hold the metrics of the whole set.
        if use_validation:
            acc, prec, rec, f1, inf_t = evaluate_classification(model, config, test_loader)
            for k, v in (("accuracy", acc), ("precision", prec), ("recall", rec), ("f1", f1)):
                all_metrics[k].append(v)
        else:
            # the reference reloads the last checkpoint (train_vit_som.py:111): rank 0 wrote it, every rank loads that file
            path = os.path.join(model_states_dir, "last.ckpt")
            if world > 1:
                torch.distributed.barrier()
            final_model = ViTSOM.load_from_checkpoint(path, config=config)
            final_model.set_distributed(world, rank, backend="torch")
            purity, nmi, inf_t = evaluate_clustering(final_model, config, train_loader)
            all_metrics["purity"].append(purity)
            all_metrics["nmi"].append(nmi)
        all_metrics["run_duration"].append(run_duration)
        all_metrics["inference_time"].append(inf_t)
    if n_runs > 1:
        _report(all_metrics, n_runs, dataset_name, log)
    return all_metrics


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="ViT-SOM training driver (MI355X)")
    ap.add_argument("--config", type=str, required=True)
    ap.add_argument("--runs", type=int, default=5)
    ap.add_argument("--epochs", type=int, default=None)
    a = ap.parse_args()
    main(load_config(a.config), n_runs=a.runs, max_epochs=a.epochs)
