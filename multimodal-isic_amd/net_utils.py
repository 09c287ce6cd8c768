"""Drop-in for the reference's ``net_utils.py`` (`train`, `validate`, `test`,
`EarlyStopping`; reference `net_utils.py:6-158`) for models whose forward runs on the
HIP path (``model.MultiModalFusionNet`` / ``MultiModalMILNet``).  Same signatures, same
dict-batch contract (`dataset.py:45-56`), same return values."""
import copy

import torch
from sklearn.metrics import balanced_accuracy_score, classification_report

_KEYS = ("image", "radiomics", "age", "sex", "loc", "artifacts")


def _to_device(batch, device):
    moved = [batch[k].to(device) if k in batch and batch[k] is not None else None for k in _KEYS]
    return moved, batch["target"].to(device)


def _log(run, key, value, assign=False):
    if run is None:
        return
    if assign:
        run[key] = value
    else:
        run[key].log(value)


def _epoch(model, dataloader, criterion, device, optimizer=None):
    loss_sum, hits, seen, preds_all, targets_all = 0.0, 0, 0, [], []
    for batch in dataloader:
        inputs, target = _to_device(batch, device)
        if optimizer is not None:
            optimizer.zero_grad(set_to_none=True)
        outputs = model(*inputs)
        if criterion is not None:
            loss = criterion(outputs, target)
            if optimizer is not None:
                loss.backward()
                optimizer.step()
            loss_sum += loss.item()
        pred = outputs.argmax(dim=1)
        hits += int((pred == target).sum())
        seen += target.size(0)
        preds_all.extend(pred.tolist())
        targets_all.extend(target.tolist())
    return loss_sum / max(len(dataloader), 1), hits / max(seen, 1), preds_all, targets_all


def train(model, dataloader, criterion, optimizer, device, neptune_run, epoch):
    model.train()
    loss, acc, _, _ = _epoch(model, dataloader, criterion, device, optimizer)
    _log(neptune_run, "train/epoch_loss", loss)
    _log(neptune_run, "train/epoch_acc", acc)
    if neptune_run is not None and hasattr(model, "weights"):
        for i, w in enumerate(model.weights.detach().cpu().tolist()):
            _log(neptune_run, f"model/fusion_weight_modality_{i}", w)
    print(f"Epoch {epoch} - Train Loss: {loss:.4f}, Accuracy: {acc:.4f}")


def validate(model, dataloader, criterion, device, neptune_run, epoch, fold_idx=None):
    model.eval()
    with torch.no_grad():
        loss, acc, _, _ = _epoch(model, dataloader, criterion, device)
    prefix = f"{fold_idx}/val" if fold_idx else "val"
    _log(neptune_run, f"{prefix}/epoch_loss", loss)
    _log(neptune_run, f"{prefix}/epoch_acc", acc)
    print(f"Epoch {epoch} - Val Loss: {loss:.4f}, Accuracy: {acc:.4f}")
    return loss


def test(model, dataloader, device, neptune_run, fold_idx=None):
    model.eval()
    with torch.no_grad():
        _, acc, preds, targets = _epoch(model, dataloader, None, device)
    bacc = balanced_accuracy_score(targets, preds)
    report = classification_report(targets, preds, digits=5)
    prefix = f"{fold_idx}/test" if fold_idx else "test"
    _log(neptune_run, f"{prefix}/accuracy", acc, assign=True)
    _log(neptune_run, f"{prefix}/balanced_accuracy", bacc, assign=True)
    _log(neptune_run, f"{prefix}/classification_report", report, assign=True)
    print(f"Test Accuracy: {acc:.4f}")
    print("Classification Report:\n", report)
    return acc, report


class EarlyStopping:
    """Reference `net_utils.py:130-158`: ``counter`` counts DOWN from ``patience``; returns True
    when it reaches zero; keeps a deep copy of the best (lowest validation loss) state."""

    def __init__(self, patience=5, neptune_run=None):
        self.patience = patience
        self.counter = patience
        self.best_loss = float('inf')
        self.best_model_state = None
        self.neptune_run = neptune_run

    def __call__(self, current_loss, model):
        improved = current_loss < self.best_loss
        if improved:
            self.best_loss, self.counter = current_loss, self.patience
            self.best_model_state = copy.deepcopy(model.state_dict())
        else:
            self.counter -= 1
        _log(self.neptune_run, "val/patience_counter", self.counter)
        return not self.counter

    def get_best_model_state(self):
        return self.best_model_state
