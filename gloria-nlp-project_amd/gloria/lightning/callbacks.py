"""Localization metrics of the reference's EvaluateLocalization callback
(/root/reference/gloria/lightning/callbacks.py:16-70 `Metrics`, `discrete_entropy`, `get_no_attn_weight`), computed on
the GPU WITHOUT the upsampled overlay.

The callback evaluates every attention map after `nn.Upsample(size=image_shape)` (nearest, callbacks.py:319).  A
nearest-upsampled map holds at most ih * iw distinct values, each copied to a block of pixels, so every metric of the
overlay is a statistic of the ih * iw cell values weighted by two integer counts per cell - the pixels that copy it
and the label pixels among them (`glr_cell_counts`, one read of the label).  From the cells in exact descending
order (`glr_topk_desc`: the same order a stable CPU sort gives) follow, in int64 / float64:

  * the percentile threshold `torch.topk(preds, total - top_k, largest=False).values.max()` (:56): the value of the
    cell at which the pixel count from the top first exceeds top_k - the same float, bit for bit;
  * precision / recall / F1 at that threshold (:57-58, torchmetrics `precision_recall` / `f1`, which binarise with
    `preds >= threshold`) and IoU with the strict `preds > threshold` mask (:59-60);
  * AUROC (trapezoid over the ROC points of the distinct scores = rank-sum with mid-ranks for ties) and average
    precision (sum over distinct scores of the recall step times the precision), ROC / PR curve points;
  * attn_entropy (entropy of [1 - sum, attn] as a categorical, :16-19) and no_attn_weight (1 - sum, :22-23).

Parity: torchmetrics is imported by the reference but NOT pinned in its requirements.txt, so its tie / threshold
conventions are restated from the 2021 releases (binarisation `>=`, curves over distinct scores) - parity unpinned for
those four entries; the threshold and the IoU follow the reference's own lines exactly.  The callback's CSV / W&B /
plotting plumbing is out of scope."""

import torch

from .. import select


def get_no_attn_weight(dist):
    return 1 - dist.sum(-1)


def discrete_entropy(dist):
    """entropy of Categorical([1 - sum(dist), dist]) (callbacks.py:16-19); dist [..., S]"""
    p = torch.cat([get_no_attn_weight(dist).unsqueeze(-1), dist], -1).double()
    p = p / p.sum(-1, keepdim=True)
    eps = torch.finfo(torch.float32).eps                    # torch.distributions clamps probabilities to [eps, 1 - eps]
    return -(p * p.clamp(eps, 1 - eps).log()).sum(-1)


class Metrics:
    def __init__(self, percentile_thresholds=(.05, .1, .2, .3)):
        self.percentile_thresholds = list(percentile_thresholds)
        self.attn_entropy = discrete_entropy
        self.no_attn_weight = get_no_attn_weight

    def __call__(self, attn, segmentation_label, curves=False):
        """attn [n, ih, iw] attention maps (device), segmentation_label [n, Hl, Wl] bool / 0-1 (device): the label
        the callback rasterises from the sentence's boxes at the image size.  Returns {name: tensor[n]} (float64);
        label-dependent entries are NaN where the label is empty (the reference stores None).  With curves=True also
        'roc_curve' / 'pr_curve': per map the (fpr, tpr, thresholds) / (precision, recall, thresholds) point lists."""
        n, ih, iw = attn.shape
        S = ih * iw
        v = attn.reshape(n, S).float()
        out = {"attn_entropy": self.attn_entropy(v), "no_attn_weight": self.no_attn_weight(v.double())}
        cnt, npix = select.cell_counts(segmentation_label, ih, iw)          # [n, S] int64
        order, vs = select.topk_desc(v, S)                                   # exact descending order of the cells
        pos = cnt.gather(1, order)
        tot = npix.gather(1, order)
        ctp = pos.cumsum(1)                                                  # label pixels with score >= this cell's
        cpp = tot.cumsum(1)                                                  # pixels with score >= this cell's
        P, T = ctp[:, -1], cpp[:, -1]
        Nn = T - P
        empty = P == 0
        nan = torch.full((n,), float("nan"), dtype=torch.float64, device=v.device)
        # distinct scores: a cell closes a tie group when the next value differs
        last = torch.ones_like(vs, dtype=torch.bool)
        last[:, :-1] = vs[:, :-1] != vs[:, 1:]
        first = torch.ones_like(last)
        first[:, 1:] = last[:, :-1]
        tp, pp = ctp.double(), cpp.double()
        fp = pp - tp
        # AUROC: trapezoids between consecutive ROC points (group ends), starting at (0, 0)
        tp_e = torch.where(last, tp, torch.zeros_like(tp))
        fp_e = torch.where(last, fp, torch.zeros_like(fp))
        tp_prev = _prev_group_value(tp, last)
        fp_prev = _prev_group_value(fp, last)
        area = ((fp_e - fp_prev) * (tp_e + tp_prev) / 2 * last).sum(1)
        out["auroc"] = torch.where(empty | (Nn == 0), nan, area / (P.double() * Nn.double()).clamp(min=1))
        # average precision: sum over distinct scores of (recall step) x precision at that score
        prec = tp / pp.clamp(min=1)
        ap = (((tp_e - tp_prev) / P.double().clamp(min=1).unsqueeze(1)) * prec * last).sum(1)
        out["avg_precision"] = torch.where(empty, nan, ap)
        if curves:
            out["roc_curve"], out["pr_curve"] = [], []
            for i in range(n):
                m = last[i]
                t_i, f_i, th = tp[i][m], fp[i][m], vs[i][m]
                z = t_i.new_zeros(1)
                out["roc_curve"].append((torch.cat([z, f_i]) / max(float(Nn[i]), 1.0),
                                         torch.cat([z, t_i]) / max(float(P[i]), 1.0),
                                         torch.cat([th[:1] + 1, th])))
                out["pr_curve"].append((torch.cat([(t_i / (t_i + f_i)).flip(0), t_i.new_ones(1)]),
                                        torch.cat([(t_i / max(float(P[i]), 1.0)).flip(0), z]), th.flip(0)))
        total = T
        for p in self.percentile_thresholds:
            top_k = (total.double() * p).long()                              # int(total * p)
            # k-th smallest with k = total - top_k  ==  the (top_k + 1)-th largest pixel: first cell (descending) whose
            # cumulative pixel count exceeds top_k
            hit = cpp > top_k.unsqueeze(1)
            j = hit.float().argmax(1)
            thr = vs.gather(1, j.unsqueeze(1)).squeeze(1)
            ge = vs >= thr.unsqueeze(1)
            gt = vs > thr.unsqueeze(1)
            tp_ge, pp_ge = (pos * ge).sum(1).double(), (tot * ge).sum(1).double()
            tp_gt, pp_gt = (pos * gt).sum(1).double(), (tot * gt).sum(1).double()
            pr, re = tp_ge / pp_ge, tp_ge / P.double()
            f1 = 2 * pr * re / (pr + re)
            iou = tp_gt / (pp_gt + P.double() - tp_gt)
            out["threshold_at_%f" % p] = thr
            out["precision_at_%f" % p] = torch.where(empty, nan, pr)
            out["recall_at_%f" % p] = torch.where(empty, nan, re)
            out["f1_at_%f" % p] = torch.where(empty, nan, f1)
            out["iou_at_%f" % p] = torch.where(empty, nan, iou)
        return out


def _prev_group_value(x, last):
    """for every position, the value of x at the END of the previous tie group (0 before the first group end)"""
    ends = torch.where(last, x, torch.full_like(x, -1.0))
    # running maximum works because cumulative counts are non-decreasing
    run = torch.cummax(ends, 1).values.clamp(min=0)
    prev = torch.zeros_like(x)
    prev[:, 1:] = run[:, :-1]
    return prev
