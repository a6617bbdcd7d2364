"""TEST INFRASTRUCTURE ONLY - CPU restatement (numpy) of the image half of the reference's collate function,
SURVEY.md 8f-4.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Follows /root/reference/gloria/datasets/mimic_for_gloria.py:
  :36-42   normalize / original_tensor_to_numpy_image   (min-max to [0,1] in fp32, *255, truncate to uint8)
  :120-133 process_img       (_resize_img -> PIL "L"->"RGB" -> transform -> stack)
  :136-181 _resize_img       (long side -> `scale` with cv2.INTER_AREA, short side zero padded)
and /root/reference/gloria/builder.py:159-201 build_transformation for the imagenome_pretrain config
(configs/imagenome_pretrain_config.yaml:101-104: RandomCrop(224) for train / CenterCrop(224) otherwise, ToTensor,
Normalize(0.5, 0.5)).

PARITY UNPINNED for the cv2 stage: cv2.resize lives in opencv-python==4.5.1.48 (requirements.txt:38), which is not
installed here and cannot be fetched, and the reference holds no image fixtures.  `resize_area_u8` restates
OpenCV 4.5's published INTER_AREA algorithm for 8-bit single-channel DOWNSCALING (modules/imgproc/src/resize.cpp:
computeResizeAreaTab, ResizeArea_Invoker, resizeAreaFast_Invoker and its 2x2 8u SIMD rounding); the remaining stages
(min-max, truncation, padding, crop, ToTensor, Normalize) are plain IEEE fp32 / integer operations restated exactly.
The upscaling branch of cv2.INTER_AREA (fixed-point bilinear) is not restated: images whose long side is below
`scale` are rejected.
"""

import math

import numpy as np

DBL_EPSILON = 2.220446049250313e-16


def to_u8(image):
    """mimic_for_gloria.py:36-42: ((x - min) / (max - min) * 255) in fp32, C cast (truncation) to uint8."""
    x = np.asarray(image).astype(np.float32)
    mn, mx = x.min(), x.max()
    y = ((x - mn) / (mx - mn)) * np.float32(255)
    return y.astype(np.uint8)


def resize_plan(h, w, scale):
    """mimic_for_gloria.py:143-176: (dst_h, dst_w, top, left) of the aspect-preserving resize + zero padding."""
    if h >= w:                                   # size.index(max(size)) == 0 also for square images
        pct = scale / float(h)
        dh, dw = scale, int(float(w) * float(pct))
        pad = scale - dw
        return dh, dw, 0, int(math.floor(pad / 2))
    pct = scale / float(w)
    dh, dw = int(float(h) * float(pct)), scale
    pad = scale - dh
    return dh, dw, int(math.floor(pad / 2)), 0


def area_tab(ssize, dsize, scale):
    """OpenCV computeResizeAreaTab: per destination index the ordered (source index, fp32 weight) taps."""
    taps = []
    for d in range(dsize):
        fs1 = d * scale
        fs2 = fs1 + scale
        cell = min(scale, ssize - fs1)
        s1, s2 = math.ceil(fs1), math.floor(fs2)
        s2 = min(s2, ssize - 1)
        s1 = min(s1, s2)
        t = []
        if s1 - fs1 > 1e-3:
            t.append((s1 - 1, np.float32((s1 - fs1) / cell)))
        for s in range(s1, s2):
            t.append((s, np.float32(1.0 / cell)))
        if fs2 - s2 > 1e-3:
            t.append((s2, np.float32(min(min(fs2 - s2, 1.0), cell) / cell)))
        taps.append(t)
    return taps


def _dense(taps):
    n = max(len(t) for t in taps)
    si = np.zeros((len(taps), n), dtype=np.int64)
    al = np.zeros((len(taps), n), dtype=np.float32)          # absent taps: weight 0 on pixel 0 adds +0.0 exactly
    for d, t in enumerate(taps):
        for j, (s, a) in enumerate(t):
            si[d, j], al[d, j] = s, a
    return si, al


def _rint_u8(x):
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)       # saturate_cast<uchar>(float): round half to even


def resize_area_u8(src, dh, dw):
    """cv2.resize(src, (dw, dh), interpolation=cv2.INTER_AREA) for 2-D uint8, downscaling in both directions."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape
    if (dh, dw) == (sh, sw):
        return src.copy()
    if dh > sh or dw > sw or dh <= 0 or dw <= 0:
        raise ValueError("only the downscaling branch of INTER_AREA is restated")
    scale_x = 1.0 / (float(dw) / sw)
    scale_y = 1.0 / (float(dh) / sh)
    ix, iy = int(np.rint(scale_x)), int(np.rint(scale_y))
    if abs(scale_x - ix) < DBL_EPSILON and abs(scale_y - iy) < DBL_EPSILON:
        blocks = src[:dh * iy, :dw * ix].reshape(dh, iy, dw, ix).astype(np.int32).sum(axis=(1, 3))
        if ix == 2 and iy == 2:
            return ((blocks + 2) >> 2).astype(np.uint8)       # 8u 2x2 SIMD path
        return _rint_u8(blocks.astype(np.float32) * np.float32(1.0 / (ix * iy)))
    xs, xa = _dense(area_tab(sw, dw, scale_x))
    ys, ya = _dense(area_tab(sh, dh, scale_y))
    f = src.astype(np.float32)
    total = None
    for j in range(ys.shape[1]):                               # source rows of a destination row, in order
        rows = f[ys[:, j]]                                      # [dh, sw]
        buf = np.zeros((dh, dw), dtype=np.float32)
        for k in range(xs.shape[1]):                            # taps of a destination column, in order
            buf = buf + rows[:, xs[:, k]] * xa[None, :, k]
        term = ya[:, j, None] * buf
        total = term if total is None else total + term
    return _rint_u8(total)


def resize_img(img_u8, scale):
    """mimic_for_gloria.py:136-181 `_resize_img`."""
    h, w = img_u8.shape
    if max(h, w) < scale:
        raise ValueError("long side below the target: cv2 would upscale (bilinear branch), not restated")
    dh, dw, top, left = resize_plan(h, w, scale)
    out = np.zeros((scale, scale), dtype=np.uint8)
    out[top:top + dh, left:left + dw] = resize_area_u8(img_u8, dh, dw)
    return out


def process_img(images_u8, crop_offsets, scale=256, crop=224):
    """process_img (:120-133) with the imagenome transform: float32 [B, 3, crop, crop] in [-1, 1].
    crop_offsets[b] = (top, left) of the crop window inside the scale x scale frame."""
    out = np.empty((len(images_u8), 3, crop, crop), dtype=np.float32)
    for b, (img, (cy, cx)) in enumerate(zip(images_u8, crop_offsets)):
        frame = resize_img(img, scale)[cy:cy + crop, cx:cx + crop]
        t = frame.astype(np.float32) / np.float32(255)          # ToTensor
        out[b, :] = ((t - np.float32(0.5)) / np.float32(0.5))[None]   # Normalize(0.5, 0.5), 3 equal channels
    return out


def center_crop_offset(scale=256, crop=224):
    o = int(round((scale - crop) / 2.0))                        # torchvision CenterCrop
    return o, o
