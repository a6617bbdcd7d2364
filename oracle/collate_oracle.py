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
OpenCV 4.5's published INTER_AREA algorithm for 8-bit single-channel images (modules/imgproc/src/resize.cpp):
  * both directions shrinking (scale_x >= 1 and scale_y >= 1): computeResizeAreaTab, ResizeArea_Invoker,
    resizeAreaFast_Invoker and its 2x2 8u SIMD rounding;
  * otherwise (an image whose long side is below `scale` is ENLARGED): cv::resize "emulates" INTER_AREA with its
    bilinear machinery and area-style coordinates (`area_mode`): sx = floor(dx * scale_x),
    fx = (dx + 1) - (sx + 1) * inv_scale_x clipped to [0, 1), 11-bit fixed-point coefficients
    (saturate_cast<short>(c * 2048)), HResizeLinear into int32 and the 8u VResizeLinear rounding
    ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2  (`resize_area_up_u8`).
The remaining stages (min-max, truncation, padding, crop, ToTensor, Normalize) are plain IEEE fp32 / integer operations
restated exactly.

The random transforms of builder.py:167-186 (RandomHorizontalFlip, RandomAffine, ColorJitter on the PIL "RGB" image; no
config of the reference enables them) are restated from torchvision 0.8.2 (requirements.txt:81, NOT installed here:
functional._get_inverse_affine_matrix, F.affine, adjust_brightness / adjust_contrast) and from Pillow's C code they
call (Geometry.c ImagingScaleAffine / affine_fixed with nearest resampling and zero fill, Blend.c ImagingBlend,
ImageEnhance / ImageStat).  The Pillow half IS pinned: tests/test_collate_host.py checks every function below against
the Pillow importable here (12.x; the reference pins 8.1.0, same algorithms).  The torchvision half (matrix formula,
order of the random draws) is PARITY UNPINNED.  All transforms act on three equal channels alike (PIL "L" -> "RGB",
convert("L") of equal channels is the identity), so one channel is processed and replicated at the end.
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
    if dh <= 0 or dw <= 0:
        raise ValueError("empty destination")
    if dh > sh or dw > sw:                                     # not (scale_x >= 1 and scale_y >= 1): bilinear emulation
        return resize_area_up_u8(src, dh, dw)
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


def _sat_short(x):
    return np.clip(np.rint(x), -32768, 32767).astype(np.int64)        # saturate_cast<short>(float): cvRound, clamp


def linear_area_tab(ssize, dsize, clamp_last):
    """cv::resize coefficient loop for INTER_AREA outside the true-area case (area_mode, ksize 2, fixed point):
    (source index, [c0, c1] as 11-bit fixed point) per destination index and `dmax`, the first destination index whose
    second tap would fall outside (x only: HResizeLinear copies S[sx] * 2048 from there on)."""
    inv = float(dsize) / ssize
    scale = 1.0 / inv
    ofs = np.zeros(dsize, dtype=np.int64)
    coef = np.zeros((dsize, 2), dtype=np.int64)
    dmax = dsize
    for d in range(dsize):
        s = int(math.floor(d * scale))
        f = np.float32((d + 1) - (s + 1) * inv)
        f = np.float32(0) if f <= 0 else np.float32(f - np.float32(math.floor(f)))
        if clamp_last:
            if s < 0:
                f, s = np.float32(0), 0
            if s + 1 >= ssize:
                dmax = min(dmax, d)
                if s >= ssize - 1:
                    f, s = np.float32(0), ssize - 1
        ofs[d] = s
        coef[d, 0] = _sat_short((np.float32(1) - f) * np.float32(2048))
        coef[d, 1] = _sat_short(f * np.float32(2048))
    return ofs, coef, dmax


def resize_area_up_u8(src, dh, dw):
    """cv2.resize(..., INTER_AREA) when at least one direction enlarges: fixed-point bilinear with area coordinates."""
    src = np.ascontiguousarray(src, dtype=np.uint8).astype(np.int64)
    sh, sw = src.shape
    xo, xc, xmax = linear_area_tab(sw, dw, True)
    yo, yc, _ = linear_area_tab(sh, dh, False)
    x1 = np.minimum(xo + 1, sw - 1)
    hbuf = src[:, xo] * xc[None, :, 0] + src[:, x1] * xc[None, :, 1]            # HResizeLinear, dx < xmax
    hbuf[:, xmax:] = src[:, xo[xmax:]] * 2048                                    # dx >= xmax: S[sx] * ONE
    r0 = np.clip(yo, 0, sh - 1)
    r1 = np.clip(yo + 1, 0, sh - 1)
    out = (((yc[:, 0, None] * (hbuf[r0] >> 4)) >> 16) + ((yc[:, 1, None] * (hbuf[r1] >> 4)) >> 16) + 2) >> 2
    return (out & 255).astype(np.uint8)                                          # uchar(...) cast


def resize_img(img_u8, scale):
    """mimic_for_gloria.py:136-181 `_resize_img`."""
    h, w = img_u8.shape
    dh, dw, top, left = resize_plan(h, w, scale)
    out = np.zeros((scale, scale), dtype=np.uint8)
    out[top:top + dh, left:left + dw] = resize_area_u8(img_u8, dh, dw)
    return out


# ---------------------------------------------------------------- random transforms (builder.py:167-186)
def hflip(img):
    """PIL transpose(FLIP_LEFT_RIGHT) (torchvision F.hflip)"""
    return np.ascontiguousarray(img[:, ::-1])


def inverse_affine_matrix(center, angle, translate, scale, shear=(0.0, 0.0)):
    """torchvision 0.8.2 functional._get_inverse_affine_matrix: the six coefficients PIL's AFFINE transform takes
    (output pixel -> input position).  F.affine calls it with center = (width * 0.5, height * 0.5)."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def _coord(v):
    return -1 if v < 0.0 else int(v)                           # Geometry.c COORD


def affine_nearest_u8(img, m):
    """PIL Image.transform(img.size, AFFINE, m, resample=NEAREST) with zero fill (Geometry.c ImagingTransformAffine):
    the scaling special case (m[1] == m[3] == 0: ImagingScaleAffine, positions accumulated in double) or 16.16 fixed
    point (affine_fixed)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    a = [float(v) for v in m]
    out = np.zeros_like(img)
    if a[1] == 0 and a[3] == 0:
        xo = a[2] + a[0] * 0.5
        yo = a[5] + a[4] * 0.5
        xin = []
        for _ in range(w):
            xin.append(_coord(xo))
            xo += a[0]
        xin = np.array(xin)
        okx = (xin >= 0) & (xin < w)
        for y in range(h):
            yi = _coord(yo)
            if 0 <= yi < h:
                out[y, okx] = img[yi, xin[okx]]
            yo += a[4]
        return out

    def fix(v):
        v = v * 65536.0 + 0.5
        return int(math.floor(v)) if v < 0.0 else int(v)       # FLOOR macro
    a0, a1, a3, a4 = fix(a[0]), fix(a[1]), fix(a[3]), fix(a[4])
    a2 = fix(a[2] + a[0] * 0.5 + a[1] * 0.5)
    a5 = fix(a[5] + a[3] * 0.5 + a[4] * 0.5)
    ys, xs = np.mgrid[0:h, 0:w]
    xin = (a2 + ys * a1 + xs * a0) >> 16
    yin = (a5 + ys * a4 + xs * a3) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def blend_u8(degenerate, img, alpha):
    """PIL Image.blend(degenerate, img, alpha) for 8-bit images (Blend.c ImagingBlend; alpha is a C float)."""
    a = np.float32(alpha)
    if a == 0:
        return degenerate.copy()
    if a == 1:
        return img.copy()
    d = (img.astype(np.int32) - degenerate.astype(np.int32)).astype(np.float32)
    t = degenerate.astype(np.float32) + a * d                  # two fp32 operations, each rounded
    if 0 <= a <= 1:
        return t.astype(np.uint8)                              # (UINT8) cast: truncation
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def adjust_brightness(img, factor):
    """torchvision F.adjust_brightness -> ImageEnhance.Brightness(img).enhance(factor): blend with black"""
    return blend_u8(np.zeros_like(img), img, factor)


def contrast_mean(img):
    """ImageEnhance.Contrast: int(ImageStat.Stat(img.convert("L")).mean[0] + 0.5)"""
    return int(float(int(img.astype(np.int64).sum())) / img.size + 0.5)


def adjust_contrast(img, factor):
    """torchvision F.adjust_contrast -> ImageEnhance.Contrast(img).enhance(factor): blend with the mean grey level"""
    return blend_u8(np.full_like(img, contrast_mean(img)), img, factor)


def augment(img, aug):
    """builder.py:167-186 in Compose order on the cropped 8-bit image: aug = dict(flip=bool, affine=None | (angle,
    (tx, ty), scale), jitter=[("brightness" | "contrast", factor), ...] in the order drawn)."""
    if aug.get("flip"):
        img = hflip(img)
    if aug.get("affine") is not None:
        angle, translate, scale = aug["affine"]
        h, w = img.shape
        img = affine_nearest_u8(img, inverse_affine_matrix((w * 0.5, h * 0.5), angle, translate, scale))
    for kind, factor in aug.get("jitter", ()):
        img = adjust_brightness(img, factor) if kind == "brightness" else adjust_contrast(img, factor)
    return img


def process_img(images_u8, crop_offsets, scale=256, crop=224, augs=None):
    """process_img (:120-133) with the imagenome transform: float32 [B, 3, crop, crop] in [-1, 1].
    crop_offsets[b] = (top, left) of the crop window inside the scale x scale frame; augs[b]: see `augment`."""
    out = np.empty((len(images_u8), 3, crop, crop), dtype=np.float32)
    for b, (img, (cy, cx)) in enumerate(zip(images_u8, crop_offsets)):
        frame = resize_img(img, scale)[cy:cy + crop, cx:cx + crop]
        if augs is not None:
            frame = augment(frame, augs[b])
        t = frame.astype(np.float32) / np.float32(255)          # ToTensor
        out[b, :] = ((t - np.float32(0.5)) / np.float32(0.5))[None]   # Normalize(0.5, 0.5), 3 equal channels
    return out


def center_crop_offset(scale=256, crop=224):
    o = int(round((scale - crop) / 2.0))                        # torchvision CenterCrop
    return o, o
