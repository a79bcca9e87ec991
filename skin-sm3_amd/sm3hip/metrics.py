"""Evaluation metrics of the linear-probe path, on device, no host round trips per batch.

`multiclass_auroc` restates torchmetrics.functional.classification.multiclass_auroc(preds, target, num_classes,
average=None) as the reference calls it (src/utils/misc.py:319-325): softmax over the logits, then one-vs-rest
area under the ROC curve per class (trapezoid over distinct thresholds = the Mann-Whitney statistic with ties
counted one half).  `auc_avg` is the reference's `AUC_AVG` (misc.py:312-316): the AUROC of ONE selected class per
label (CLS_WEIGHTS index), averaged over the 8 labels -- the "8 avg" column of linear_results.csv.
"""
import torch

CLASSES_NAME = ["DIAG", "PN", "BWV", "VS", "PIG", "STR", "DaG", "RS"]
NUM_CLASSES = [5, 3, 2, 3, 3, 3, 3, 2]
CLS_WEIGHTS = [2, 2, 1, 2, 2, 2, 2, 1]


def binary_auroc(score, positive):
    """AUROC of `score` (higher = more positive) against boolean `positive`; ties count 1/2; NaN-free: returns 0
    when a class is absent (torchmetrics returns 0 with a warning in that case)."""
    score = score.double().flatten()
    positive = positive.flatten().bool()
    n_pos = int(positive.sum())
    n_neg = positive.numel() - n_pos
    if n_pos == 0 or n_neg == 0:
        return score.new_zeros(())
    order = torch.argsort(score)
    s = score[order]
    # average ranks with ties: rank of a tie group = mean of its positions (1-based)
    uniq, inv, counts = torch.unique_consecutive(s, return_inverse=True, return_counts=True)
    ends = torch.cumsum(counts, 0).double()
    avg_rank = ends - (counts.double() - 1) / 2.0
    ranks = avg_rank[inv]
    rank_sum_pos = ranks[positive[order]].sum()
    return (rank_sum_pos - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def multiclass_auroc(logits, target, num_classes):
    """Per-class one-vs-rest AUROC [num_classes] of softmax(logits) -- torchmetrics `average=None`."""
    probs = torch.softmax(logits.double(), dim=1)
    return torch.stack([binary_auroc(probs[:, c], target == c) for c in range(num_classes)])


def auc_avg(preds, targets, num_classes=NUM_CLASSES, cls_weights=CLS_WEIGHTS):
    """preds: list of 8 logits tensors [N, n_i]; targets [N, 8].  Returns (per-label AUROC list, their mean)."""
    per = [multiclass_auroc(preds[i], targets[:, i], num_classes[i])[cls_weights[i]] for i in range(len(num_classes))]
    return per, torch.stack(per).mean()
