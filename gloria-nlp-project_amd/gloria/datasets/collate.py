"""GloriaCollateFn with the reference's interface (/root/reference/gloria/datasets/mimic_for_gloria.py:58-263),
image half on the GPU.

  process_img(images, device)   :120-133  list of 2-D images -> float32 [B, 3, crop, crop] in [-1, 1].  The reference
        loops over images on the host (cv2.INTER_AREA resize, np.pad, PIL "L"->"RGB", RandomCrop/CenterCrop, ToTensor,
        Normalize) and uploads the result; here the raw images are uploaded once (one packed buffer) and
        glr_image_minmax + glr_collate_images (csrc/glr_collate.hip) produce the batch tensor directly.
        Raw (non-uint8) images take the min-max -> uint8 conversion of original_tensor_to_numpy_image (:36-42) on
        the GPU as well (`__call__` hands them over unconverted).
  _resize_plan                  :136-181  the size / padding arithmetic of `_resize_img` (host integers)
  process_text(text, device)    :184-263  report cleaning + tokenisation; nltk's RegexpTokenizer(r"\\w+") is the
        regular expression itself, the tokenizer is a local BERT vocabulary (no network: `bert_type` must be a
        local directory / vocab file or a tokenizer object is passed in)
  get_batch / __call__          :66-108   sort by caption length (descending), permute every field alike

  get_segmentation_labels       :110-118  boxes -> bool [B, crop, crop] labels; host integers / tap tables instead of
        one full-resolution mask image per box (see `resized_box_mask`)

  random transforms             builder.py:167-186  RandomHorizontalFlip / RandomAffine / ColorJitter(brightness,
        contrast) of torchvision 0.8.2 on the cropped PIL image (no config of the reference enables them): the
        PARAMETERS are drawn on the host in torchvision's order (`draw_augmentation`; torchvision is not installed here,
        the draw order and the matrix formula are restated from its source: parity unpinned), the pixels are transformed
        on the GPU with Pillow's arithmetic (glr_aug_geom / glr_aug_jitter / glr_u8_to_tensor, include/glr.h).
Images smaller than `imsize` are enlarged the way cv2.INTER_AREA does it (fixed-point bilinear emulation).
There is no CPU path: process_img needs the HIP library and a GPU.
"""

import math
import os
import random
import re

import numpy as np
import torch

from .. import _native as N

_SRC_CODE = {np.dtype(np.uint8): 0, np.dtype(np.int16): 1, np.dtype(np.float32): 2}
_SPLITTER = re.compile(r"[0-9]+\.")
_WORD = re.compile(r"\w+")


def resize_plan(h, w, scale):
    """(dst_h, dst_w, pad_top, pad_left) of `_resize_img` (mimic_for_gloria.py:143-176)."""
    if h >= w:                                    # size.index(max(size)) == 0, also for square images
        pct = scale / float(h)
        dh, dw = scale, int(float(w) * float(pct))
        return dh, dw, 0, int(math.floor((scale - dw) / 2))
    pct = scale / float(w)
    dh, dw = int(float(h) * float(pct)), scale
    return dh, dw, int(math.floor((scale - dh) / 2)), 0


def inverse_affine_matrix(center, angle, translate, scale, shear=(0.0, 0.0)):
    """torchvision 0.8.2 functional._get_inverse_affine_matrix (called by F.affine with center = (w * 0.5, h * 0.5)):
    the six coefficients PIL's Image.transform(AFFINE) takes - output pixel -> input position."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [v / scale for v in (d, -b, 0.0, -c, a, 0.0)]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def draw_augmentation(flip_p, affine, jitter, size):
    """One image's random transform parameters, drawn from torch's global generator in the order torchvision 0.8.2's
    Compose([RandomHorizontalFlip, RandomAffine, ColorJitter]) draws them (restated from its source, unpinned):
      flip     torch.rand(1) < p
      affine   RandomAffine.get_params: angle, then tx, ty (rounded to whole pixels), then scale - each
               torch.empty(1).uniform_(lo, hi)
      jitter   fn_idx = torch.randperm(4); brightness (0) / contrast (1) factor torch.tensor(1.0).uniform_(lo, hi) in
               that order (saturation / hue are not configured by builder.py:178-184)
    flip_p: float | None; affine: None | dict(degrees, translate, scale); jitter: None | dict(brightness, contrast) of
    [lo, hi] pairs.  Returns the dict collate_images / the oracle's `augment` take."""
    aug = {"flip": False, "affine": None, "jitter": []}
    if flip_p is not None:
        aug["flip"] = bool(torch.rand(1) < flip_p)
    if affine is not None:
        deg = affine["degrees"]
        deg = (-float(deg), float(deg)) if isinstance(deg, (int, float)) else (float(deg[0]), float(deg[1]))
        angle = float(torch.empty(1).uniform_(deg[0], deg[1]).item())
        tx = ty = 0
        if affine.get("translate") is not None:
            max_dx, max_dy = float(affine["translate"][0] * size), float(affine["translate"][1] * size)
            tx = int(round(torch.empty(1).uniform_(-max_dx, max_dx).item()))
            ty = int(round(torch.empty(1).uniform_(-max_dy, max_dy).item()))
        sc = 1.0
        if affine.get("scale") is not None:
            sc = float(torch.empty(1).uniform_(float(affine["scale"][0]), float(affine["scale"][1])).item())
        aug["affine"] = (angle, (tx, ty), sc)
    if jitter is not None:
        for fn_id in torch.randperm(4).tolist():
            for k, name in ((0, "brightness"), (1, "contrast")):
                rng_ = jitter.get(name)
                if fn_id == k and rng_ is not None:
                    aug["jitter"].append((name, float(torch.tensor(1.0).uniform_(float(rng_[0]), float(rng_[1])).item())))
    return aug


def _apply_augmentation(u8, augs, crop, device):
    """uint8 [B, crop, crop] on the GPU -> float32 [B, 3, crop, crop]: flip + affine, the colour steps, ToTensor + Normalize"""
    L = N.lib()
    B = u8.shape[0]
    flip = np.array([1 if a.get("flip") else 0 for a in augs], dtype=np.int32)
    mats = np.full((B, 6), np.nan, dtype=np.float64)
    for b, a in enumerate(augs):
        if a.get("affine") is not None:
            angle, translate, sc = a["affine"]
            mats[b] = inverse_affine_matrix((crop * 0.5, crop * 0.5), angle, translate, sc)
    if flip.any() or not np.isnan(mats[:, 0]).all():
        dst = torch.empty_like(u8)
        flip_d, mats_d = N.upload(flip, device), N.upload(mats, device)      # named: alive until the launch is queued
        N.check(L.glr_aug_geom(N.ptr(u8), N.ptr(dst), B, crop, N.ptr(flip_d), N.ptr(mats_d), N.stream()), "glr_aug_geom")
        u8 = dst
    steps = max((len(a.get("jitter", ())) for a in augs), default=0)
    if steps:
        sums = torch.empty(B, dtype=torch.int64, device=device)
        code = {"brightness": 1, "contrast": 2}
        for j in range(steps):
            kind = np.zeros(B, dtype=np.int32)
            alpha = np.ones(B, dtype=np.float32)
            for b, a in enumerate(augs):
                jit = a.get("jitter", ())
                if j < len(jit):
                    kind[b], alpha[b] = code[jit[j][0]], jit[j][1]
            kind_d, alpha_d = N.upload(kind, device), N.upload(alpha, device)
            N.check(L.glr_aug_jitter(N.ptr(u8), B, crop, N.ptr(kind_d), N.ptr(alpha_d), N.ptr(sums), N.stream()),
                    "glr_aug_jitter")
    out = torch.empty(B, 3, crop, crop, dtype=torch.float32, device=device)
    N.check(L.glr_u8_to_tensor(N.ptr(u8), B, crop, N.ptr(out), N.stream()), "glr_u8_to_tensor")
    return out


def collate_images(images, crop_offsets, scale=256, crop=224, device="cuda", minmax=None, augs=None):
    """images: list of 2-D numpy arrays / torch tensors.  uint8 images are taken as they are (minmax=False) unless
    minmax=True; other dtypes are min-max normalised to uint8 on the GPU.  augs: None, or one `draw_augmentation` dict
    per image (flip / affine / colour steps between the crop and ToTensor).  Returns float32 [B, 3, crop, crop]."""
    arrs = []
    for im in images:
        a = im.detach().cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
        if a.ndim != 2:
            raise ValueError(f"expected single-channel 2-D images, got shape {a.shape}")
        arrs.append(a)
    if not arrs:
        raise ValueError("empty image batch")
    dt = arrs[0].dtype if all(a.dtype == arrs[0].dtype for a in arrs) else np.dtype(np.float32)
    if dt not in _SRC_CODE:
        dt = np.dtype(np.float32)                 # image.float() of the reference (:37)
    if minmax is None:
        minmax = dt != np.dtype(np.uint8)
    if not minmax and dt != np.dtype(np.uint8):
        raise ValueError("images that are not uint8 need the min-max conversion")
    desc = np.zeros((len(arrs), 8), dtype=np.int32)
    offsets = np.zeros(len(arrs), dtype=np.int64)
    total = 0
    for b, (a, (cy, cx)) in enumerate(zip(arrs, crop_offsets)):
        h, w = a.shape
        dh, dw, top, left = resize_plan(h, w, scale)
        if dh <= 0 or dw <= 0:
            raise ValueError(f"degenerate image {h}x{w}")
        if not (0 <= cy <= scale - crop and 0 <= cx <= scale - crop):
            raise ValueError("crop window outside the frame")
        desc[b] = (h, w, dh, dw, top, left, cy, cx)
        offsets[b] = total
        total += (h * w * dt.itemsize + 15) // 16 * 16
    host = torch.empty(total, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else torch.empty(total, dtype=torch.uint8)
    hv = host.numpy()
    for a, o in zip(arrs, offsets):
        n = a.size * dt.itemsize
        hv[o:o + n] = np.ascontiguousarray(a, dtype=dt).reshape(-1).view(np.uint8)
    src = host.to(device, non_blocking=True)
    N.require_cuda(src)
    meta = torch.from_numpy(np.concatenate([offsets.view(np.int32), desc.reshape(-1)])).to(device, non_blocking=True)
    off_d, desc_d = meta[:2 * len(arrs)], meta[2 * len(arrs):]
    if augs is not None and len(augs) != len(arrs):
        raise ValueError("one augmentation record per image")
    out = torch.empty(len(arrs), 3, crop, crop, dtype=torch.float32, device=device) if augs is None else None
    u8 = torch.empty(len(arrs), crop, crop, dtype=torch.uint8, device=device) if augs is not None else None
    L = N.lib()
    code = _SRC_CODE[dt]
    state = None
    if minmax:
        state = torch.empty(len(arrs), 2, dtype=torch.int32, device=device)
        N.check(L.glr_image_minmax(N.ptr(src), N.ptr(off_d), N.ptr(desc_d), len(arrs), code, N.ptr(state), N.stream()),
                "glr_image_minmax")
    N.check(L.glr_collate_images(N.ptr(src), N.ptr(off_d), N.ptr(desc_d), N.ptr(state), len(arrs), code, crop,
                                 N.ptr(out), N.ptr(u8), N.stream()), "glr_collate_images")
    if augs is not None:
        return _apply_augmentation(u8, augs, crop, device)
    return out


# ---------------------------------------------------------------- segmentation labels from bounding boxes (host)
# The reference renders every box as a full-size 0/255 mask, sends it through process_img (cv2 resize, pad, crop,
# normalise), thresholds `> 0` on the NORMALISED tensor (i.e. grey level >= 128), takes the bounding box of what is
# left and ORs the boxes into a crop x crop label (mimic_for_gloria.py:13-55, 110-118).  The resized mask of a
# rectangle is separable, so the same numbers come out of the tap tables directly: O(dst_h * dst_w) host work per box
# instead of a full-resolution image, with OpenCV's fp32 summation order kept (zero terms add +0.0 exactly).
_DBL_EPSILON = 2.220446049250313e-16


def _area_taps(ssize, dsize, scale):
    """OpenCV computeResizeAreaTab as dense tables: source index and fp32 weight of every ordered tap of every
    destination index (absent taps: weight 0 on index 0)."""
    rows = []
    for d in range(dsize):
        fs1 = d * scale
        fs2 = fs1 + scale
        cell = min(scale, ssize - fs1)
        s2 = min(math.floor(fs2), ssize - 1)
        s1 = min(math.ceil(fs1), s2)
        t = []
        if s1 - fs1 > 1e-3:
            t.append((s1 - 1, (s1 - fs1) / cell))
        t.extend((sx, 1.0 / cell) for sx in range(s1, s2))
        if fs2 - s2 > 1e-3:
            t.append((s2, min(min(fs2 - s2, 1.0), cell) / cell))
        rows.append(t)
    n = max(len(t) for t in rows)
    idx = np.zeros((dsize, n), dtype=np.int64)
    alpha = np.zeros((dsize, n), dtype=np.float32)
    for d, t in enumerate(rows):
        for k, (si, a) in enumerate(t):
            idx[d, k], alpha[d, k] = si, np.float32(a)
    return idx, alpha


def resized_box_mask(h, w, dh, dw, box):
    """cv2.resize(mask, (dw, dh), INTER_AREA) of the 0/255 mask of `box` = [x0, y0, x1, y1] (inclusive), uint8 [dh, dw]."""
    x0, y0, x1, y1 = (int(c) for c in box)
    cols, rows = np.arange(w), np.arange(h)
    in_x, in_y = (cols >= x0) & (cols <= x1), (rows >= y0) & (rows <= y1)
    if (dh, dw) == (h, w):
        return (np.outer(in_y, in_x) * 255).astype(np.uint8)
    scale_x, scale_y = 1.0 / (float(dw) / w), 1.0 / (float(dh) / h)
    ix, iy = int(np.rint(scale_x)), int(np.rint(scale_y))
    if abs(scale_x - ix) < _DBL_EPSILON and abs(scale_y - iy) < _DBL_EPSILON:
        cnt = np.outer(in_y[:dh * iy].reshape(dh, iy).sum(1), in_x[:dw * ix].reshape(dw, ix).sum(1)).astype(np.int32) * 255
        if ix == 2 and iy == 2:
            return ((cnt + 2) >> 2).astype(np.uint8)
        return np.clip(np.rint(cnt.astype(np.float32) * np.float32(1.0 / (ix * iy))), 0, 255).astype(np.uint8)
    xi, xa = _area_taps(w, dw, scale_x)
    yi, ya = _area_taps(h, dh, scale_y)
    bx = np.zeros(dw, dtype=np.float32)                       # one in-box source row, resampled along x
    for k in range(xi.shape[1]):
        bx = bx + np.where(in_x[xi[:, k]], np.float32(255), np.float32(0)) * xa[:, k]
    total = None
    for j in range(yi.shape[1]):
        buf = np.where(in_y[yi[:, j]][:, None], bx[None, :], np.float32(0))
        term = ya[:, j, None] * buf
        total = term if total is None else total + term
    return np.clip(np.rint(total), 0, 255).astype(np.uint8)


def resized_box(h, w, box, scale, crop, crop_offset):
    """process_bboxes (mimic_for_gloria.py:45-55) for one box: [xmin, ymin, xmax, ymax] in the crop window, or
    [-1, -1, -1, -1] when nothing reaches grey level 128.  Boxes that are empty or cover the whole image make the
    reference divide 0 / 0 in `normalize`; they count as empty here."""
    x0, y0, x1, y1 = (int(c) for c in box)
    x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, w - 1), min(y1, h - 1)
    if x1 < x0 or y1 < y0 or (x0 == 0 and y0 == 0 and x1 == w - 1 and y1 == h - 1):
        return [-1, -1, -1, -1]
    dh, dw, top, left = resize_plan(h, w, scale)
    frame = np.zeros((scale, scale), dtype=bool)
    frame[top:top + dh, left:left + dw] = resized_box_mask(h, w, dh, dw, (x0, y0, x1, y1)) >= 128   # (v/255 - .5)/.5 > 0
    cy, cx = crop_offset
    win = frame[cy:cy + crop, cx:cx + crop]
    if not win.any():
        return [-1, -1, -1, -1]
    ys, xs = np.nonzero(win.any(1))[0], np.nonzero(win.any(0))[0]
    return [int(xs[0]), int(ys[0]), int(xs[-1]), int(ys[-1])]


def clean_report(text, full_report=True, rng=random):
    """mimic_for_gloria.py:190-223: sentence split, \\w+ tokens, ascii-only, lower case."""
    text = text.replace("\n", " ")
    captions = [sent for point in _SPLITTER.split(text) for sent in point.split(".")]
    all_sents = []
    for t in captions:
        t = t.replace("\ufffd\ufffd", " ")
        tokens = _WORD.findall(t.lower())
        if len(tokens) <= 1:
            continue
        kept = [w for w in (tok.encode("ascii", "ignore").decode("ascii") for tok in tokens) if len(w) > 0]
        all_sents.append(" ".join(kept))
    if full_report is True:
        return " ".join(all_sents)
    return all_sents[rng.randint(0, len(all_sents) - 1)]


def load_tokenizer(bert_type):
    """A BERT word-piece tokenizer from a LOCAL directory or vocab.txt (the reference downloads
    `AutoTokenizer.from_pretrained(bert_type)`, mimic_for_gloria.py:61; there is no network here)."""
    from transformers import BertTokenizerFast
    if os.path.isdir(bert_type):
        return BertTokenizerFast.from_pretrained(bert_type, local_files_only=True)
    if os.path.isfile(bert_type):
        with open(bert_type, encoding="utf-8") as f:
            vocab = {line.rstrip("\n"): i for i, line in enumerate(f)}
        try:
            return BertTokenizerFast(vocab=vocab, do_lower_case=True)          # transformers >= 5
        except TypeError:
            return BertTokenizerFast(vocab_file=bert_type, do_lower_case=True)   # transformers 4.x
    raise FileNotFoundError(f"tokenizer '{bert_type}' is not a local directory or vocab file (no network access)")


class GloriaCollateFn:
    def __init__(self, cfg, split, device="cuda", include_instances=True, tokenizer=None):
        self.cfg = cfg
        self.tokenizer = tokenizer if tokenizer is not None else load_tokenizer(cfg.model.text.bert_type)
        self.split = split
        self.device = device
        self.ixtoword = {v: k for k, v in self.tokenizer.get_vocab().items()}
        self.include_instances = include_instances
        t = cfg.transforms
        # builder.py:167-186: the random transforms exist for the train split only
        self.flip_p = self.affine = self.jitter = None
        if t is not None and split == "train":
            if getattr(t, "random_horizontal_flip", None) is not None:
                self.flip_p = float(t.random_horizontal_flip)
            ra = getattr(t, "random_affine", None)
            if ra is not None:
                self.affine = {"degrees": ra.degrees, "translate": list(ra.translate), "scale": list(ra.scale)}
            cj = getattr(t, "color_jitter", None)
            if cj is not None:          # `bightness` is the reference's spelling of the key (builder.py:181)
                rng_b, rng_c = [float(v) for v in cj.bightness], [float(v) for v in cj.contrast]
                self.jitter = {"brightness": None if rng_b[0] == rng_b[1] == 1.0 else rng_b,
                               "contrast": None if rng_c[0] == rng_c[1] == 1.0 else rng_c}
        if t is None or t.norm != "half":
            raise NotImplementedError("only Normalize(0.5, 0.5) ('half') is built (builder.py:196-197)")
        self.scale = cfg.data.image.imsize
        self.crop = t.random_crop.crop_size if (t is not None and t.random_crop is not None) else self.scale

    # ---- images
    def crop_offsets(self, n):
        """RandomCrop.get_params for the train split (torchvision 0.8.2: i then j from torch.randint per image,
        nothing drawn when the sizes are equal); CenterCrop otherwise (builder.py:162-190)."""
        room = self.scale - self.crop
        if room == 0:
            return [(0, 0)] * n
        if self.split == "train":
            return [(int(torch.randint(0, room + 1, size=(1,)).item()), int(torch.randint(0, room + 1, size=(1,)).item()))
                    for _ in range(n)]
        o = int(round(room / 2.0))
        return [(o, o)] * n

    def augmented(self):
        return self.flip_p is not None or self.affine is not None or self.jitter is not None

    def draw_params(self, n):
        """crop offsets and transform parameters of n images in the reference's order: the Compose runs per image, so
        image b's crop, flip, affine and colour draws all come before image b + 1's"""
        room = self.scale - self.crop
        crops, augs = [], []
        for _ in range(n):
            if room == 0:
                crops.append((0, 0))
            else:
                crops.append((int(torch.randint(0, room + 1, size=(1,)).item()), int(torch.randint(0, room + 1, size=(1,)).item())))
            augs.append(draw_augmentation(self.flip_p, self.affine, self.jitter, self.crop))
        return crops, augs

    def process_img(self, images, device, minmax=None):
        if self.augmented():
            crops, augs = self.draw_params(len(images))
            return collate_images(images, crops, self.scale, self.crop, device, minmax, augs)
        return collate_images(images, self.crop_offsets(len(images)), self.scale, self.crop, device, minmax)

    # ---- text
    def process_text(self, text, device, objects=None):
        if objects is not None:
            raise NotImplementedError
        ids, att, typ, sents = [], [], [], []
        for t in text:
            t = clean_report(t, self.cfg.data.text.full_report)
            enc = self.tokenizer(t, return_tensors="pt", truncation=True, padding="max_length",
                                 max_length=self.cfg.data.text.word_num)
            ids.append(enc["input_ids"])
            att.append(enc["attention_mask"])
            typ.append(enc["token_type_ids"])
            sents.append([self.ixtoword[ix] for ix in enc["input_ids"][0].tolist()])
        out = {}
        for key, parts in (("caption_ids", ids), ("attention_mask", att), ("token_type_ids", typ)):
            stacked = torch.stack(parts)
            out[key] = (stacked.squeeze(0) if len(text) == 1 else stacked.squeeze()).to(device)
        out["cap_lens"] = [len([w for w in s if not w.startswith("[")]) + 1 for s in sents]
        return out

    # ---- batch
    def get_segmentation_labels(self, bboxes, original_shapes, new_shape, device):
        """mimic_for_gloria.py:110-118.  Every box draws its own crop window, in the reference's order (its masks
        go through process_img, whose transform contains the RandomCrop)."""
        labels = torch.zeros((len(bboxes),) + tuple(new_shape), dtype=torch.bool)
        for b, (bbs, (h, w)) in enumerate(zip(bboxes, original_shapes)):
            for box, off in zip(bbs, self.crop_offsets(len(bbs))):
                xa, ya, xb, yb = resized_box(h, w, box, self.scale, self.crop, off)
                if xa >= 0:
                    labels[b, ya:yb + 1, xa:xb + 1] = True
        return labels.to(device)

    def get_batch(self, images, captions, instances=None, sort=True, bboxes=None, minmax=None):
        imgs = self.process_img(images, self.device, minmax)
        seg_labels = None
        if bboxes is not None:
            shapes = [tuple(im.shape) for im in images]
            seg_labels = self.get_segmentation_labels(bboxes, shapes, (self.crop, self.crop), self.device)
        cap = self.process_text(captions, self.device)
        lens = torch.tensor(cap["cap_lens"])
        if sort:
            lens, order = torch.sort(lens, 0, True)
        else:
            order = torch.arange(len(lens))
        lens, order = lens.to(self.device), order.to(self.device)
        batch = {k: v[order] for k, v in cap.items() if k != "cap_lens"}
        batch["cap_lens"] = lens
        batch["imgs"] = imgs[order]
        if seg_labels is not None:
            batch["segmentation_labels"] = seg_labels[order]
        if instances is not None:
            batch["instances"] = [instances[i] for i in order.tolist()]
        return batch

    def __call__(self, instances):
        images, captions, bboxes = [], [], []
        for instance in instances:
            patient_id = next(iter(instance.keys()))
            study_id = next(iter(instance[patient_id].keys()))
            inst = instance[patient_id][study_id]
            dicom_id = next(iter(inst["images"].keys()))
            images.append(inst["images"][dicom_id])          # raw: the min-max -> uint8 step runs on the GPU
            if "sentence" in inst.keys():
                captions.append(inst["sentence"])
                bboxes.append(inst["objects"][dicom_id]["sent_to_bboxes"][inst["sent_id"]]["coords_original"])
            else:
                captions.append(inst["report"])
        return self.get_batch(images, captions, instances=instances if self.include_instances else None,
                              bboxes=bboxes if len(bboxes) > 0 else None, minmax=True)


def multimodal_collate_fn(batch):
    """The CheXpert-path collate of the reference (gloria/datasets/pretraining_dataset.py:250-282): samples are
    (img [3, H, W], tokenizer output, cap_len, path); every tensor field is stacked and permuted by descending
    caption length, `path` keeps the sample order (as in the reference)."""
    imgs, caps, lens, paths = zip(*batch)
    order = torch.sort(torch.tensor(lens), 0, True)
    fields = {name: torch.stack([c[key] for c in caps]).squeeze()[order.indices]
              for name, key in (("caption_ids", "input_ids"), ("token_type_ids", "token_type_ids"),
                                ("attention_mask", "attention_mask"))}
    fields["imgs"] = torch.stack(imgs)[order.indices]
    fields["cap_lens"] = order.values
    fields["path"] = list(paths)
    return fields
