"""Localization metrics of the reference's EvaluateLocalization callback that sit on the selection site of the
path (/root/reference/gloria/lightning/callbacks.py:26-70): for each percentile p the threshold is the
(total - int(total p))-th smallest attention value - `torch.topk(preds, total - top_k, largest=False).values.max()`
(:56) - and precision / recall / F1 / IoU use the strict `preds > threshold` mask (:57-61).  Threshold and counts
come from the exact selection kernels (gloria.select): identical picks to the CPU code, no tolerance.

AUROC / average precision / ROC curves (torchmetrics) and the callback's CSV / W&B plumbing are out of scope."""

import torch

from .. import select


class Metrics:
    def __init__(self, percentile_thresholds=(.05, .1, .2, .3)):
        self.percentile_thresholds = list(percentile_thresholds)

    def __call__(self, attn_overlay, segmentation_label):
        """attn_overlay [..., H, W] float (upsampled attention), segmentation_label same shape (bool / 0-1).
        Returns {metric_at_p: tensor[...]}; entries are NaN where the label is empty (the reference stores None)."""
        pred = attn_overlay.reshape(-1, attn_overlay.shape[-2] * attn_overlay.shape[-1]).float()
        tgt = segmentation_label.reshape(pred.shape)
        total = pred.shape[1]
        out = {}
        for p in self.percentile_thresholds:
            top_k = int(total * p)
            thr = select.kth_value(pred, total - top_k)
            c = select.threshold_counts(pred, tgt, thr).double()
            tp, pp, tt, un = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
            prec, rec = tp / pp, tp / tt
            f1 = 2 * prec * rec / (prec + rec)
            nan = torch.full_like(tp, float("nan"))
            empty = tt == 0
            out["threshold_at_%f" % p] = thr
            out["precision_at_%f" % p] = torch.where(empty, nan, prec)
            out["recall_at_%f" % p] = torch.where(empty, nan, rec)
            out["f1_at_%f" % p] = torch.where(empty, nan, f1)
            out["iou_at_%f" % p] = torch.where(empty, nan, tp / un)
        return out
