"""Host-side train / eval loops of the MIL teacher (`01_train_mil_teacher.py:170-305`)
and of the patch-graph GNN (`05_train_gnns.py:273-358`) over the HIP path, with the two
things the reference does not have: several bags / graphs per optimizer step in one
launch, and bag-sharded data parallelism (one process per GPU, RCCL gradient all-reduce
through ``ddp.GradSync``).

Step semantics (SURVEY.md §7 "Batch-vs-per-bag"): with ``per_step=1, world=1`` these loops
ARE the reference loops (one optimizer step per bag / graph); with B bags per step the loss
is the mean of the per-bag reference losses and the gradient the mean of the per-bag
gradients.
"""
from __future__ import annotations

import copy
import os
from collections import Counter

import numpy as np
import torch
import torch.distributed as dist

from . import ddp, ops, optim
from .bags import BagOffsets
from .graph import GraphBatch
from .lib import call


def dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_distributed():
    """One process per GPU under torchrun / torch.distributed.run; no-op otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return dist_info()


def weighted_sample_indices(labels, generator):
    """`01_train_mil_teacher.py:189-193`: WeightedRandomSampler(1/class_count, n, replacement=True)."""
    labels = np.asarray(labels)
    counts = Counter(labels.tolist())
    w = torch.tensor([1.0 / counts[int(l)] for l in labels], dtype=torch.float64)
    return torch.multinomial(w, len(labels), replacement=True, generator=generator).tolist()


class BagStore:
    """Bags resident in HBM: uniform bags as one [n, K, D] tensor (gathered by index per step),
    ragged bags as a list of device tensors concatenated per step."""

    def __init__(self, bags, device):
        self.device = device
        lens = {int(b.shape[0]) for b in bags}
        self.uniform = len(lens) == 1
        self.lengths = [int(b.shape[0]) for b in bags]

        def dev(b):            # numpy latents (`01_train_mil_teacher.py:51-67`) or tensors already resident in HBM
            if isinstance(b, torch.Tensor):
                return b.to(device=device, dtype=torch.float32)
            return torch.as_tensor(np.asarray(b, dtype=np.float32)).to(device)
        if self.uniform and not any(isinstance(b, torch.Tensor) for b in bags):
            self.data = torch.as_tensor(np.stack([np.asarray(b, dtype=np.float32) for b in bags])).to(device)
        elif self.uniform:
            self.data = torch.stack([dev(b) for b in bags])
        else:
            self.data = [dev(b) for b in bags]

    def batch(self, idx):
        if self.uniform:
            x = self.data[torch.as_tensor(idx, device=self.device)]
            return x.reshape(-1, x.shape[-1]), BagOffsets.uniform(len(idx), x.shape[1], self.device)
        xs = [self.data[i] for i in idx]
        return torch.cat(xs), BagOffsets.from_lengths([self.lengths[i] for i in idx], self.device)


def _make_optimizer(model, name, lr, wd):
    cls = optim.AdamW if name.lower() == "adamw" else optim.Adam if name.lower() == "adam" else None
    if cls is None:
        raise ValueError(f"Unknown optimizer: {name}")
    return cls(model.parameters(), lr=lr, weight_decay=wd)


def _sync_step(opt, sync, world):
    sync.finish()
    opt.step(grad_scale=1.0 / world)


# ----------------------------------------------------------------------------- MIL teacher
@torch.no_grad()
def eval_teacher(model, store, labels, chunk=64):
    """bag_probs + mean per-bag CE over all bags (`01:247-260`)."""
    model.eval()
    probs, losses = [], []
    n = len(labels)
    for lo in range(0, n, chunk):
        idx = list(range(lo, min(n, lo + chunk)))
        x, offs = store.batch(idx)
        out = model(x, offs)
        y = torch.as_tensor(np.asarray(labels)[idx], device=x.device)
        losses.append(ops.CrossEntropyFn.apply(out["bag_logits"], y, 0)[1].cpu())
        probs.append(out["bag_probs"].cpu())
    return torch.cat(probs).numpy(), float(torch.cat(losses).mean()) if losses else float("nan")


def train_teacher_fold(model, train_bags, train_labels, val_bags, val_labels, *, optimizer="adamw", lr=2.2e-4,
                       weight_decay=8.6e-4, epochs=200, patience=8, bags_per_step=1, seed=42, device=None, log=print,
                       metric_fn=None):
    """`01_train_mil_teacher.py:203-290` for one fold.  Returns dict(best_state_bacc, best_state_loss,
    history).  Every rank draws the same sampler stream and takes its slice of each step's bags."""
    from sklearn.metrics import balanced_accuracy_score
    rank, world = dist_info()
    device = device or next(model.parameters()).device
    tr, va = BagStore(train_bags, device), BagStore(val_bags, device)
    opt = _make_optimizer(model, optimizer, lr, weight_decay)
    ddp.broadcast_parameters(opt.flat.data)
    sync = ddp.GradSync(opt.flat.grad, world_size=world)
    gen = torch.Generator().manual_seed(seed)
    best = {"bacc": -np.inf, "loss": float("inf"), "state_bacc": None, "state_loss": None, "no_improve": 0}
    history = []
    per_step = bags_per_step * world
    for epoch in range(1, epochs + 1):
        model.train()
        order = weighted_sample_indices(train_labels, gen)
        for s in range(0, len(order), per_step):
            glob = order[s:s + per_step]
            lo, hi = ddp.shard_range(len(glob), rank, world)
            mine = glob[lo:hi]
            opt.zero_grad()
            sync.reset()
            if mine:
                x, offs = tr.batch(mine)
                out = model(x, offs)
                y = torch.as_tensor(np.asarray(train_labels)[mine], device=device)
                # mean over the GLOBAL step: local mean * (local / global count), summed by the all-reduce
                loss = ops.cross_entropy(out["bag_logits"], y) * (len(mine) * world / len(glob))
                ops.backward(loss)
            _sync_step(opt, sync, world)
        probs, val_loss = eval_teacher(model, va, val_labels)
        if len(val_labels) == 0:
            break
        pred = probs.argmax(axis=1)
        bacc = balanced_accuracy_score(np.asarray(val_labels), pred)
        if bacc > best["bacc"] + 1e-6:
            best["bacc"], best["state_bacc"] = bacc, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        if val_loss < best["loss"] - 1e-6:
            best["loss"], best["no_improve"] = val_loss, 0
            best["state_loss"] = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        else:
            best["no_improve"] += 1
        extra = metric_fn(np.asarray(val_labels), probs) if metric_fn else {}
        history.append({"epoch": epoch, "val_bacc": bacc, "val_loss": val_loss, **extra})
        if rank == 0 and log:
            log(f"    Epoch {epoch:03d}: Val BAcc: {bacc:.4f} (best: {best['bacc']:.4f})  | Val Loss: {val_loss:.4f} "
                f"(best: {best['loss']:.4f} ) | Epochs no improve: {best['no_improve']}/{patience}")
        if best["no_improve"] >= patience:
            break
    return {"best_state_bacc": best["state_bacc"], "best_state_loss": best["state_loss"], "history": history}


@torch.no_grad()
def collect_teacher_outputs(model, bags, labels, image_ids, device, chunk=64):
    """`01_train_mil_teacher.py:69-87`: rows {image_id, label, patch_probs, attention, patch_embeddings}."""
    import pandas as pd
    model.eval()
    store = BagStore(bags, device)
    rows = []
    for lo in range(0, len(bags), chunk):
        idx = list(range(lo, min(len(bags), lo + chunk)))
        x, offs = store.batch(idx)
        out = model(x, offs)
        pp, att = out["patch_probs"].cpu().numpy(), out["attention"].cpu().numpy()
        for j, i in enumerate(idx):
            a, b = int(offs.host[j]), int(offs.host[j + 1])
            rows.append({"image_id": image_ids[i], "label": int(labels[i]), "patch_probs": pp[a:b], "attention": att[a:b],
                         "patch_embeddings": np.asarray(bags[i], dtype=np.float32)})
    return pd.DataFrame(rows)


# ----------------------------------------------------------------------------- configs[1]: bags of image patches
class ImageBagStore:
    """Bags of K image patches + one radiomic vector per bag, resident in HBM (images as bf16: what the encoder's
    first kernel rounds to anyway)."""

    def __init__(self, images, radiomics, labels, device):
        images = torch.as_tensor(images)
        if images.dim() != 5:
            raise ValueError(f"expected images[n, K, 3, H, W], got {tuple(images.shape)}")
        self.images = images.to(device=device, dtype=torch.bfloat16)
        self.radiomics = torch.as_tensor(radiomics, dtype=torch.float32).to(device)
        self.labels = np.asarray(labels)
        self.y_dev = torch.as_tensor(self.labels, device=device)
        self.device = device

    def __len__(self):
        return len(self.labels)

    def batch(self, idx):
        sel = torch.as_tensor(list(idx), device=self.device)
        return self.images[sel], self.radiomics[sel], self.y_dev[sel]


@torch.no_grad()
def eval_milnet(model, store, chunk=64):
    """Fused-logit class probabilities and mean per-bag CE over all bags, eval mode (running BatchNorm statistics,
    no dropout) -- the evaluation of `01_train_mil_teacher.py:247-260` for the composed model."""
    model.eval()
    probs, losses = [], []
    for lo in range(0, len(store), chunk):
        img, rad, y = store.batch(range(lo, min(len(store), lo + chunk)))
        out = model(img, rad)
        losses.append(ops.CrossEntropyFn.apply(out["logits"], y, 0)[1].cpu())
        probs.append(ops.softmax_rows(out["logits"]).cpu())
    if not probs:
        return np.zeros((0, 0), np.float32), float("nan")
    return torch.cat(probs).numpy(), float(torch.cat(losses).mean())


def train_milnet_fold(model, train_set, val_set, *, optimizer="adamw", lr=2.2e-4, weight_decay=8.6e-4, epochs=200,
                      patience=8, bags_per_step=8, seed=42, num_classes=7, device=None, log=print):
    """The 01 loop (`01_train_mil_teacher.py:203-290`) for BASELINE.json configs[1]: ``model.MultiModalMILNet`` (ResNet-18
    patch encoder -> attention-MIL head -> radiomic fusion) trained end to end on bags of image patches.  Class-balanced
    sampler stream (`01:189-193`), ``bags_per_step`` bags per optimizer step and rank (bag-sharded DDP: every rank draws
    the same stream and takes its slice; gradients all-reduced through ``ddp.GradSync`` overlapped with the encoder's
    backward), validation after every epoch -> macro one-vs-rest AUROC / balanced accuracy / loss of the fused logits,
    best states by balanced accuracy and by loss, early stop on the loss (`01:265-290`).
    ``train_set`` / ``val_set`` = (images[n, K, 3, H, W], radiomics[n, R], labels[n])."""
    from sklearn.metrics import balanced_accuracy_score
    rank, world = dist_info()
    device = device or next(model.parameters()).device
    tr, va = ImageBagStore(*train_set, device), ImageBagStore(*val_set, device)
    opt = _make_optimizer(model, optimizer, lr, weight_decay)
    ddp.broadcast_parameters(opt.flat.data)
    sync = ddp.GradSync(opt.flat.grad, world_size=world)
    ddp.attach(model.encoder, opt.flat, sync)
    gen = torch.Generator().manual_seed(seed)
    best = {"bacc": -np.inf, "loss": float("inf"), "state_bacc": None, "state_loss": None, "no_improve": 0}
    history = []
    per_step = bags_per_step * world
    snapshot = lambda: {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for epoch in range(1, epochs + 1):
        model.train()
        order = weighted_sample_indices(tr.labels, gen)
        step_losses = []                          # device scalars: read back once per epoch (no sync inside the loop)
        for s in range(0, len(order), per_step):
            glob = order[s:s + per_step]
            lo, hi = ddp.shard_range(len(glob), rank, world)
            mine = glob[lo:hi]
            opt.zero_grad()
            sync.reset()
            if mine:
                img, rad, y = tr.batch(mine)
                with ops.fused_grad_accumulation():
                    out = model(img, rad)
                    loss = model.loss(out, y)
                    step_losses.append(loss.detach())
                    (loss * (len(mine) * world / len(glob))).backward()
            _sync_step(opt, sync, world)
        ddp.average_buffers(model)               # BatchNorm running statistics are per rank during the epoch
        probs, val_loss = eval_milnet(model, va)
        if len(va) == 0:
            break
        m = gnn_metrics(va.labels, probs, num_classes)
        bacc = balanced_accuracy_score(va.labels, probs.argmax(axis=1))
        if bacc > best["bacc"] + 1e-6:
            best["bacc"], best["state_bacc"] = bacc, snapshot()
        if val_loss < best["loss"] - 1e-6:
            best["loss"], best["no_improve"], best["state_loss"] = val_loss, 0, snapshot()
        else:
            best["no_improve"] += 1
        history.append({"epoch": epoch, "val_auc": m["auc"], "val_bacc": bacc, "val_loss": val_loss, "probs": probs,
                        "train_losses": [float(v) for v in torch.stack(step_losses).cpu()] if step_losses else []})
        if rank == 0 and log:
            log(f"    Epoch {epoch:03d}: Val AUROC: {m['auc']:.4f} | Val BAcc: {bacc:.4f} (best: {best['bacc']:.4f})  | "
                f"Val Loss: {val_loss:.4f} (best: {best['loss']:.4f} ) | Epochs no improve: {best['no_improve']}/{patience}")
        if best["no_improve"] >= patience:
            break
    model.encoder.grad_ready_hook = None
    return {"best_state_bacc": best["state_bacc"], "best_state_loss": best["state_loss"], "history": history}


# ----------------------------------------------------------------------------- patch-graph GNN
class GraphStore:
    """Graph records resident in HBM with their normalised CSR built once (graphs are static across
    epochs; the reference re-uploads x and edge_index every step, `05:340-343`)."""

    def __init__(self, records, device, needs_graph=True, mode="gcn"):
        """``mode``: aggregation of the CSR the model's layers need (``gnn_models._GRAPH_MODE``: 'gcn' =
        self loops + symmetric normalisation, 'sum' = GINConv, 'mean' = SAGEConv)."""
        if mode not in GraphBatch.MODES:
            raise ValueError(f"unknown graph mode '{mode}'")
        self.device, self.records, self.mode = device, records, mode
        def dev_tensor(v, dtype, np_dtype):      # records hold numpy arrays (05:248-270) or tensors already in HBM
            if isinstance(v, torch.Tensor):
                return v.to(device=device, dtype=dtype)
            return torch.as_tensor(np.asarray(v, dtype=np_dtype)).to(device)

        self.x = [dev_tensor(r["x"], torch.float32, np.float32) for r in records]
        self.y = np.asarray([int(r["y"]) for r in records])
        self.ei = [dev_tensor(r["edge_index"], torch.int64, np.int64)
                   if needs_graph and r.get("edge_index") is not None else None for r in records]
        self.needs_graph = needs_graph
        # equal-sized graphs (the reference's case): one [G, n, D] tensor, a step's batch is ONE index_select
        self.y_dev = torch.as_tensor(self.y, device=device)
        self._xstack = torch.stack(self.x) if self.x and len({tuple(t.shape) for t in self.x}) == 1 else None
        self._single = {}
        self._chunks = {}          # evaluation chunks (sequential index ranges) repeat every epoch: cached whole
        self._stack = None
        self._build_stack()

    def _build_stack(self):
        """Graphs are static across epochs, so their normalised CSR is built ONCE: one ``GraphBatch`` launch over
        the whole record set.  When every graph has the same node and edge count (the reference's case: 196
        nodes, k-NN / grid edges) the per-graph pieces are kept as [G, ...] stacks with LOCAL ids, and a step's
        batch is six ``index_select``s plus offsets -- no per-step sort / scan / Python loop over graphs."""
        if not self.needs_graph or not self.x or any(e is None for e in self.ei):
            return
        ns = {int(t.shape[0]) for t in self.x}
        es = {int(e.shape[1]) for e in self.ei}
        if len(ns) != 1 or len(es) != 1:
            return
        n, G = ns.pop(), len(self.x)
        noff = torch.arange(G, device=self.device, dtype=torch.int64) * n
        ei = (torch.stack(self.ei) + noff.view(G, 1, 1)).permute(1, 0, 2).reshape(2, -1)
        full = GraphBatch(ei, G * n, mode=self.mode)
        rp = full.rowptr.to(torch.int64)
        rpt = full.rowptr_t.to(torch.int64)
        used = int(rp[-1])                     # stored entries (the arrays are allocated for E + n)
        nnz = used // G
        if used != G * nnz or int(rpt[-1]) != used:
            return
        # graph g owns rows [g*n, (g+1)*n) and, because every graph has the same number of stored entries,
        # slots [g*nnz, (g+1)*nnz) of both the CSR and its transpose
        if not (bool((rp[::n] == torch.arange(G + 1, device=self.device) * nnz).all())
                and bool((rpt[::n] == torch.arange(G + 1, device=self.device) * nnz).all())):
            return
        eoff = (torch.arange(G, device=self.device, dtype=torch.int64) * nnz).view(G, 1)
        i32 = torch.int32
        self._stack = {
            "n": n, "nnz": nnz,
            "rowptr": (rp[:-1].view(G, n) - eoff).to(i32), "rowptr_t": (rpt[:-1].view(G, n) - eoff).to(i32),
            "col": (full.col[:used].view(G, nnz).to(torch.int64) - noff.view(G, 1)).to(i32),
            "col_t": (full.col_t[:used].view(G, nnz).to(torch.int64) - noff.view(G, 1)).to(i32),
            "val": full.val[:used].view(G, nnz).clone(), "val_t": full.val_t[:used].view(G, nnz).clone(),
            "perm_t": (full.perm_t[:used].view(G, nnz).to(torch.int64) - eoff).to(i32),
        }

    def _stacked_graph(self, sel, rows_out=None):
        """``sel``: device int64 tensor of graph indices.  ``rows_out``: int32 [>= B*n] that receives, for every node row of the
        batch, its row in the stacked record store (``batch_rows``)."""
        st = self._stack
        n, nnz, B = st["n"], st["nnz"], int(sel.numel())
        i32, dev = torch.int32, self.device
        parts = {"rowptr": torch.empty(B * n + 1, device=dev, dtype=i32), "rowptr_t": torch.empty(B * n + 1, device=dev, dtype=i32),
                 "col": torch.empty(B * nnz, device=dev, dtype=i32), "col_t": torch.empty(B * nnz, device=dev, dtype=i32),
                 "val": torch.empty(B * nnz, device=dev, dtype=torch.float32),
                 "val_t": torch.empty(B * nnz, device=dev, dtype=torch.float32),
                 "perm_t": torch.empty(B * nnz, device=dev, dtype=i32)}
        # ONE launch (graph.hip: csr_batch_assemble_kernel) instead of seven index_selects + offset adds + concatenations
        call("isic_csr_batch_assemble", sel, B, n, nnz, st["rowptr"], st["rowptr_t"], st["col"], st["col_t"], st["val"],
             st["val_t"], st["perm_t"], parts["rowptr"], parts["rowptr_t"], parts["col"], parts["col_t"], parts["val"],
             parts["val_t"], parts["perm_t"], rows_out)
        return GraphBatch.from_parts(B * n, B * (nnz - (n if self.mode == "gcn" else 0)), self.mode, parts)

    def batch_rows(self, idx):
        """The batch WITHOUT gathering its node features: (x_store[G*n, D], rows int32, n_rows, offsets, GraphBatch) with
        batch node row i = x_store[rows[i]] -- for ``GraphMIL(x_store, x_rows=(rows, n_rows), ...)``, whose input projection
        reads through the index (``ops.linear_rows``).  Equal-sized graphs with a stacked CSR and a device index only;
        ``None`` otherwise (use ``batch``)."""
        if not (isinstance(idx, torch.Tensor) and idx.is_cuda) or self._xstack is None or self._stack is None or not self.needs_graph:
            return None
        sel = idx.to(torch.int64)
        B, n, D = int(sel.numel()), int(self._xstack.shape[1]), int(self._xstack.shape[2])
        rows = torch.empty((B * n + 7) // 8 * 8 + 8, device=self.device, dtype=torch.int32)
        graph = self._stacked_graph(sel, rows_out=rows)
        return self._xstack.view(-1, D), rows, B * n, BagOffsets.uniform(B, n, self.device), graph

    def batch(self, idx, cache=False):
        """(x[sum n, D], offsets, GraphBatch) of the graphs ``idx`` (a list, or an int64 tensor already on the device:
        no host -> device copy then); ``cache=True`` keeps the result (evaluation chunks, which repeat every epoch)."""
        on_dev = isinstance(idx, torch.Tensor) and idx.is_cuda
        key = None if on_dev else tuple(int(i) for i in idx)
        if cache and key is not None and key in self._chunks:
            return self._chunks[key]
        if not on_dev and len(key) == 1:
            i = key[0]
            n = int(self.x[i].shape[0])
            if self.needs_graph and i not in self._single:
                self._single[i] = GraphBatch(self.ei[i], n, mode=self.mode)
            return self.x[i], BagOffsets.single(n, self.device), self._single.get(i)
        uniform = self._xstack is not None and (not self.needs_graph or self._stack is not None)
        if uniform:
            sel = idx.to(torch.int64) if on_dev else torch.as_tensor(key, device=self.device)
            n, D = int(self._xstack.shape[1]), int(self._xstack.shape[2])
            x = self._xstack[sel].reshape(-1, D)
            offs = BagOffsets.uniform(int(sel.numel()), n, self.device)
            graph = self._stacked_graph(sel) if self.needs_graph else None
        else:
            if on_dev:
                key = tuple(int(i) for i in idx.tolist())
            xs = [self.x[i] for i in key]
            x = torch.cat(xs)
            offs = BagOffsets.from_lengths([int(t.shape[0]) for t in xs], self.device)
            graph = None
            if self.needs_graph:
                ei = torch.cat([self.ei[i] + int(o) for i, o in zip(key, offs.host[:-1])], dim=1)
                graph = GraphBatch(ei, offs.total, mode=self.mode)
        out = (x, offs, graph)
        if cache and key is not None:
            self._chunks[key] = out
        return out


def gnn_metrics(labels, scores, num_classes):
    """`05_train_gnns.py:284-302` metric set (sklearn, as the reference)."""
    from sklearn.metrics import accuracy_score, balanced_accuracy_score, precision_recall_fscore_support, roc_auc_score
    pred = scores.argmax(axis=1)
    try:
        auc = roc_auc_score(labels, scores, multi_class="ovr", labels=np.arange(num_classes))
    except ValueError:
        auc = float("nan")
    f1 = precision_recall_fscore_support(labels, pred, average="macro", zero_division=0)[2]
    return {"accuracy": accuracy_score(labels, pred), "bacc": balanced_accuracy_score(labels, pred), "auc": auc,
            "macro_f1": f1}


@torch.no_grad()
def evaluate_gnn(model, store, num_classes, chunk=32):
    model.eval()
    n = len(store.x)
    if n == 0:
        return {k: float("nan") for k in ("loss", "accuracy", "bacc", "auc", "macro_f1")}
    scores, losses = [], []
    for lo in range(0, n, chunk):
        idx = list(range(lo, min(n, lo + chunk)))
        x, offs, g = store.batch(idx, cache=True)
        probs, _ = model(x, offsets=offs, graph=g)
        y = torch.as_tensor(store.y[idx], device=x.device)
        losses.append(ops.CrossEntropyFn.apply(probs, y, 1)[1].cpu())      # CE(log(p + 1e-9)), 05:282
        scores.append(probs.cpu())
    scores = torch.cat(scores).numpy()
    return {"loss": float(torch.cat(losses).mean()), **gnn_metrics(store.y, scores, num_classes)}


def train_gnn_fold(model, train_records, val_records, test_records, *, lr=1e-4, weight_decay=1e-4, epochs=1,
                   patience=16, min_delta=1e-6, graphs_per_step=1, num_classes=7, device=None, rng=None):
    """`05_train_gnns.py:305-358`: AdamW, epoch order = np.random.permutation, best state by validation
    balanced accuracy.  The class weights of `05:328-331` cancel for single-sample CE and
    batched steps keep the unweighted per-graph mean.  Returns (val_metrics, test_metrics, best_epoch)."""
    rank, world = dist_info()
    device = device or next(model.parameters()).device
    needs = model.gnn_type != "mlp"
    mode = model.graph_mode or "gcn"
    tr, va, te = (GraphStore(r, device, needs, mode=mode) for r in (train_records, val_records, test_records))
    opt = optim.AdamW(model.parameters(), lr=lr, weight_decay=weight_decay)
    ddp.broadcast_parameters(opt.flat.data)
    sync = ddp.GradSync(opt.flat.grad, world_size=world)
    rng = rng or np.random
    best_state, best_bacc, no_imp, best_epoch = None, -np.inf, 0, 0
    per_step = graphs_per_step * world
    for epoch in range(1, epochs + 1):
        model.train()
        order = rng.permutation(len(train_records)).tolist()
        order_dev = torch.as_tensor(order, device=device)        # ONE upload per epoch; steps slice it on the device
        for s in range(0, len(order), per_step):
            glob = order[s:s + per_step]
            lo, hi = ddp.shard_range(len(glob), rank, world)
            mine = glob[lo:hi]
            opt.zero_grad()
            sync.reset()
            if mine:
                mine_dev = order_dev[s + lo:s + hi]
                br = tr.batch_rows(mine_dev) if len(mine) > 1 and hasattr(model, "classifier_light") else None
                if br is not None:                           # equal-sized graphs: no gather, the projection reads through the index
                    x, rows, n_rows, offs, g = br
                    xr = (rows, n_rows)
                else:
                    x, offs, g = tr.batch(mine_dev if len(mine) > 1 else mine)
                    xr = None
                with ops.fused_grad_accumulation():          # zero_grad -> backward -> step: gradients go straight into the flat buffer
                    y = tr.y_dev[mine_dev]
                    w = len(mine) * world / len(glob)
                    if hasattr(model, "classifier_light"):      # GraphMIL: head + loss as one autograd node (two launches)
                        _probs, _att, loss = model(x, offsets=offs, graph=g, labels=y, x_rows=xr)
                    else:
                        probs, _ = model(x, offsets=offs, graph=g)
                        loss = ops.cross_entropy_from_probs(probs, y)
                    ops.backward(loss if w == 1.0 else loss * w)
            _sync_step(opt, sync, world)
        vm = evaluate_gnn(model, va, num_classes)
        if vm["bacc"] > best_bacc + min_delta:
            best_bacc, no_imp, best_epoch = vm["bacc"], 0, epoch
            best_state = copy.deepcopy(model.state_dict())
        else:
            no_imp += 1
        if no_imp >= patience:
            break
    if best_state is not None:
        model.load_state_dict(best_state)
    return evaluate_gnn(model, va, num_classes), evaluate_gnn(model, te, num_classes), best_epoch
