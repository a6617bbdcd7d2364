"""GPU parity of the collate kernels (glr_image_minmax / glr_collate_images) against the CPU
restatement oracle/collate_oracle.py: BIT-EXACT (every output is one of 256 fp32 values chosen by integer /
correctly-rounded arithmetic in a fixed order).  The oracle's cv2 stage is "parity unpinned" (see its header)."""

import numpy as np
import pytest
import torch

from oracle import collate_oracle as co

pytestmark = pytest.mark.gpu

# (H, W): 2x2 integer fast path, 4x4 fast path, 3x fast, general both axes, one axis integer, equal size (copy),
# equal size with padding, tall / wide, full-resolution chest film
SHAPES = [(512, 512), (1024, 1000), (768, 768), (300, 256), (777, 1033), (512, 501), (256, 256), (256, 200), (3056, 2544),
          (2544, 3056), (257, 1999)]


def images(dtype, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for h, w in SHAPES:
        if dtype == np.uint8:
            a = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        elif dtype == np.int16:
            a = rng.integers(-1024, 3072, size=(h, w)).astype(np.int16)
        else:
            a = (rng.standard_normal((h, w)) * 417.3 + 1000).astype(np.float32)
        # smooth structure on top of the noise so that neighbouring cells differ systematically
        a = (a * 0.5 + (np.add.outer(np.arange(h), np.arange(w)) % 251) * (0.5 if dtype == np.uint8 else 7)).astype(dtype)
        out.append(a)
    return out


def offsets(n, seed=3):
    rng = np.random.default_rng(seed)
    o = [(int(a), int(b)) for a, b in rng.integers(0, 33, size=(n, 2))]
    o[0], o[1] = (0, 0), (32, 32)
    return o


@pytest.mark.parametrize("dtype,minmax", [(np.uint8, False), (np.uint8, True), (np.int16, True), (np.float32, True)])
def test_collate_bit_exact(dtype, minmax):
    from gloria.datasets.collate import collate_images
    imgs = images(dtype)
    offs = offsets(len(imgs))
    got = collate_images(imgs, offs, 256, 224, "cuda", minmax=minmax).cpu().numpy()
    u8 = [co.to_u8(a) if minmax else a for a in imgs]
    want = co.process_img(u8, offs, 256, 224)
    bad = [i for i in range(len(imgs)) if not np.array_equal(got[i], want[i])]
    assert not bad, [(SHAPES[i], float(np.abs(got[i] - want[i]).max()) * 127.5) for i in bad]


SMALL = [(100, 80), (255, 255), (64, 250), (250, 64), (17, 23), (200, 256), (256, 130), (1, 1), (2, 255)]


@pytest.mark.parametrize("dtype,minmax", [(np.uint8, False), (np.int16, True), (np.float32, True)])
def test_enlarged_images_bit_exact(dtype, minmax):
    """long side below imsize: cv2.INTER_AREA's bilinear emulation (mode 3 of k_collate) against resize_area_up_u8"""
    from gloria.datasets.collate import collate_images
    rng = np.random.default_rng(9)
    imgs = []
    for h, w in SMALL:
        if dtype == np.uint8:
            imgs.append(rng.integers(0, 256, size=(h, w), dtype=np.uint8))
        elif dtype == np.int16:
            imgs.append(rng.integers(-1024, 3072, size=(h, w)).astype(np.int16))
        else:
            imgs.append((rng.standard_normal((h, w)) * 417.3 + 1000).astype(np.float32))
    if minmax:
        imgs[7] = np.array([[3]], dtype=dtype)                 # constant image: defined as grey level 0
    offs = offsets(len(imgs))
    got = collate_images(imgs, offs, 256, 224, "cuda", minmax=minmax).cpu().numpy()
    u8 = [co.to_u8(a) if minmax else a for a in imgs]
    if minmax:
        u8[7] = np.zeros((1, 1), np.uint8)
    want = co.process_img(u8, offs, 256, 224)
    bad = [i for i in range(len(imgs)) if not np.array_equal(got[i], want[i])]
    assert not bad, [(SMALL[i], float(np.abs(got[i] - want[i]).max()) * 127.5) for i in bad]


def test_random_transforms_bit_exact():
    """flip / affine (PIL's scaling special case and its 16.16 fixed-point path) / brightness / contrast in both orders,
    interpolating and extrapolating factors: the device passes against the oracle (itself pinned against Pillow)"""
    from gloria.datasets.collate import collate_images
    imgs = images(np.uint8)[:8] + [np.random.default_rng(2).integers(0, 256, size=(120, 90), dtype=np.uint8)]
    offs = offsets(len(imgs))
    augs = [
        {"flip": True, "affine": None, "jitter": []},
        {"flip": False, "affine": (0.0, (7, -5), 1.1), "jitter": []},                       # scaling path
        {"flip": True, "affine": (12.5, (3, 4), 0.95), "jitter": [("brightness", 0.8)]},     # fixed-point path
        {"flip": False, "affine": (-30.0, (-10, 8), 1.05), "jitter": [("contrast", 1.3), ("brightness", 0.7)]},
        {"flip": False, "affine": None, "jitter": [("brightness", 1.4), ("contrast", 0.6)]},
        {"flip": True, "affine": (90.0, (0, 0), 1.0), "jitter": [("contrast", 0.0)]},
        {"flip": False, "affine": (0.0, (0, 0), 0.9), "jitter": [("contrast", 1.0), ("brightness", 0.0)]},
        {"flip": False, "affine": None, "jitter": []},
        {"flip": True, "affine": (5.0, (20, -20), 1.2), "jitter": [("contrast", 2.0)]},
    ]
    got = collate_images(imgs, offs, 256, 224, "cuda", minmax=False, augs=augs).cpu().numpy()
    want = co.process_img(imgs, offs, 256, 224, augs=augs)
    bad = [i for i in range(len(imgs)) if not np.array_equal(got[i], want[i])]
    assert not bad, [(i, int((got[i] != want[i]).sum())) for i in bad]
    plain = collate_images(imgs, offs, 256, 224, "cuda", minmax=False).cpu().numpy()
    assert np.array_equal(plain[7], got[7])                                                  # no transform = the plain path


def test_minmax_state_and_constant_image():
    from gloria import _native as N
    from gloria.datasets.collate import collate_images
    # a constant image has max == min: the reference divides 0/0 -> NaN -> uint8 cast; defined here as 0 -> -1.0
    out = collate_images([np.full((300, 300), 7, dtype=np.int16)], [(16, 16)], minmax=True)
    assert (out == -1).all()
    # constant uint8 image without min-max: every resized pixel keeps the value, padding is -1
    out = collate_images([np.full((400, 300), 200, dtype=np.uint8)], [(16, 16)]).cpu().numpy()
    dh, dw, top, left = co.resize_plan(400, 300, 256)
    val = np.float32((np.float32(200) / np.float32(255) - np.float32(0.5)) / np.float32(0.5))
    cols = np.arange(16, 240)
    inside = (cols >= left) & (cols < left + dw)
    assert (out[0, :, :, inside] == val).all() and (out[0, :, :, ~inside] == -1).all()
    assert N.lib().glr_collate_images(None, None, None, None, 1, 0, 224, None, None, None) == -1


def test_full_size_batch_properties():
    """BASELINE-sized batch (256 full-resolution 16-bit films would be 4 GB; 64 here): per-image results do not depend
    on the batch they are in, and channels are equal."""
    from gloria.datasets.collate import collate_images
    rng = np.random.default_rng(11)
    base = [rng.integers(0, 4096, size=s).astype(np.int16) for s in ((3056, 2544), (2544, 3056), (2022, 2022), (1760, 2140))]
    imgs = [base[i % 4] for i in range(64)]
    offs = [(i % 33, (7 * i) % 33) for i in range(64)]
    out = collate_images(imgs, offs, minmax=True)
    assert torch.equal(out[:, 0], out[:, 1]) and torch.equal(out[:, 0], out[:, 2])
    for i in (0, 1, 2, 3, 37, 63):
        single = collate_images([imgs[i]], [offs[i]], minmax=True)
        assert torch.equal(single[0], out[i])
    want = co.process_img([co.to_u8(base[2])], [offs[2]])
    assert np.array_equal(out[2].cpu().numpy(), want[0])


def test_get_batch_end_to_end(tmp_path):
    from gloria.config import pretrain_config
    from gloria.datasets.collate import GloriaCollateFn
    vocab = tmp_path / "vocab.txt"
    words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "the", "heart", "is", "normal", "lungs", "are", "clear", "no", "effusion"]
    vocab.write_text("\n".join(words) + "\n")
    cfg = pretrain_config("imagenome", batch_size=3)
    cfg.set_path("model.text.bert_type", str(vocab))
    fn = GloriaCollateFn(cfg, "valid", device="cuda")
    rng = np.random.default_rng(2)
    raw = [torch.from_numpy(rng.integers(0, 4096, size=s).astype(np.int16)) for s in ((600, 500), (512, 512), (300, 700))]
    caps = ["The heart is normal.", "The heart is normal. Lungs are clear. No effusion.", "Lungs are clear. No effusion"]
    instances = [{f"p{i}": {f"s{i}": {"images": {f"d{i}": raw[i]}, "report": caps[i]}}} for i in range(3)]
    batch = fn(instances)
    assert batch["cap_lens"].tolist() == [10, 6, 5]                          # sorted descending (mimic_for_gloria.py:95)
    assert batch["imgs"].shape == (3, 3, 224, 224) and batch["caption_ids"].shape == (3, 97)
    order = [1, 2, 0]
    assert [next(iter(i.keys())) for i in batch["instances"]] == [f"p{j}" for j in order]
    want = co.process_img([co.to_u8(raw[j].numpy()) for j in order], [co.center_crop_offset()] * 3)
    assert np.array_equal(batch["imgs"].cpu().numpy(), want)


def test_rows_staged_in_chunks_and_unaligned_fallback():
    """scale 20 (5120 x 4400): the ~22 source rows of an output row do not fit the 40 KB LDS stage at once -> chunked
    staging; a packed offset that is not 16-byte aligned takes the same arithmetic straight from global memory."""
    from gloria import _native as N
    from gloria.datasets.collate import collate_images, resize_plan
    rng = np.random.default_rng(4)
    big = rng.integers(0, 256, size=(5120, 4400), dtype=np.uint8)
    offs = [(5, 9)]
    got = collate_images([big], offs).cpu().numpy()
    assert np.array_equal(got, co.process_img([big], offs))
    # direct C-ABI call with a 2-byte shifted int16 image
    img = rng.integers(-500, 3000, size=(700, 900)).astype(np.int16)
    buf = torch.zeros(img.size * 2 + 64, dtype=torch.uint8, device="cuda")
    buf[2:2 + img.size * 2] = torch.from_numpy(img.reshape(-1).view(np.uint8)).cuda()
    dh, dw, top, left = resize_plan(700, 900, 256)
    desc = torch.tensor([700, 900, dh, dw, top, left, 11, 3], dtype=torch.int32, device="cuda")
    off = torch.tensor([2], dtype=torch.int64, device="cuda")
    state = torch.empty(2, dtype=torch.int32, device="cuda")
    out = torch.empty(1, 3, 224, 224, device="cuda")
    L = N.lib()
    N.check(L.glr_image_minmax(N.ptr(buf), N.ptr(off), N.ptr(desc), 1, 1, N.ptr(state), N.stream()), "minmax")
    N.check(L.glr_collate_images(N.ptr(buf), N.ptr(off), N.ptr(desc), N.ptr(state), 1, 1, 224, N.ptr(out), None, N.stream()), "collate")
    assert np.array_equal(out.cpu().numpy(), co.process_img([co.to_u8(img)], [(11, 3)]))


def test_wide_range_int16_takes_the_arithmetic_quantiser():
    """more than 8192 grey levels: no per-workgroup lookup table, the exact division runs per staged pixel"""
    from gloria.datasets.collate import collate_images
    rng = np.random.default_rng(9)
    imgs = [rng.integers(-30000, 30000, size=s).astype(np.int16) for s in ((777, 1033), (512, 512), (1024, 1000), (300, 256))]
    offs = [(1, 2), (30, 31), (16, 16), (0, 32)]
    got = collate_images(imgs, offs, minmax=True).cpu().numpy()
    assert np.array_equal(got, co.process_img([co.to_u8(a) for a in imgs], offs))


def test_shortened_division_is_exact_for_every_integer_pair():
    """k_collate quantises 8/16-bit sources with a 3-instruction division (reciprocal hoisted, ONE residual correction).
    Its claim - bit-identical to fp32 `/` for all integers 0 <= n <= d < 2^17 (every (pixel - min, max - min) such a source
    can produce) - is checked exhaustively on the device: 8.6e9 pairs."""
    from gloria import _native as N
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    N.check(N.lib().glr_selftest_quotient(1, (1 << 17) + 1, N.ptr(bad), N.stream()), "glr_selftest_quotient")
    assert int(bad.item()) == 0
