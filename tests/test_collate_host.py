"""CPU tests of the collate function: the numpy restatement of the image path (oracle/collate_oracle.py,
PARITY UNPINNED for the cv2.INTER_AREA stage: opencv-python==4.5.1.48 is absent and the reference holds no image
fixtures) checked through known answers and an independent area average, plus the host-side text half."""

import numpy as np
import pytest
import torch

from oracle import collate_oracle as co


def test_resize_plan_matches_reference_arithmetic():
    from gloria.datasets import collate
    # mimic_for_gloria.py:143-176 by hand: 2544x3056 (wider): dst_h = int(2544 * 256/3056) = 213, pad 43 -> top 21
    assert co.resize_plan(2544, 3056, 256) == (213, 256, 21, 0)
    assert co.resize_plan(3056, 2544, 256) == (256, 213, 0, 21)
    assert co.resize_plan(512, 512, 256) == (256, 256, 0, 0)          # square counts as "taller"
    rng = np.random.default_rng(0)
    for h, w in rng.integers(256, 4000, size=(200, 2)):
        assert collate.resize_plan(int(h), int(w), 256) == co.resize_plan(int(h), int(w), 256)


def test_min_max_to_uint8_truncates():
    x = np.array([[0.0, 1.0], [2.0, 3.0]], dtype=np.float32)
    assert co.to_u8(x).tolist() == [[0, 85], [170, 255]]                # 1/3*255 = 85.0, 2/3*255 = 170.0
    x = np.array([[-5, 0, 7, 11]], dtype=np.int16)
    want = [int(np.float32(np.float32(v + 5) / np.float32(16)) * np.float32(255)) for v in (-5, 0, 7, 11)]
    assert co.to_u8(x).tolist() == [want]


def test_area_fast_paths_known_answers():
    src = np.array([[1, 2, 5, 5], [3, 4, 5, 6]], dtype=np.uint8)
    assert co.resize_area_u8(src, 1, 2).tolist() == [[3, 5]]            # (10+2)>>2 = 3 (2.5 rounds UP), (21+2)>>2 = 5
    src = np.arange(32, dtype=np.uint8).reshape(4, 8)
    got = co.resize_area_u8(src, 1, 2)                                   # 4x4 boxes: means 13.5 / 17.5 -> half to even
    assert got.tolist() == [[14, 18]]
    src = np.full((6, 9), 10, dtype=np.uint8); src[0, 0] = 15            # 3x3 box sum 95 -> 95/9 = 10.56 -> 11
    assert co.resize_area_u8(src, 2, 3).tolist() == [[11, 10, 10], [10, 10, 10]]
    assert co.resize_area_u8(src, 6, 9) is not src and (co.resize_area_u8(src, 6, 9) == src).all()


def test_area_tab_weights():
    taps = co.area_tab(5, 2, 2.5)                                        # cells [0, 2.5) and [2.5, 5)
    f = lambda v: float(np.float32(v))
    assert [(s, float(a)) for s, a in taps[0]] == [(0, f(0.4)), (1, f(0.4)), (2, f(0.2))]
    assert [(s, float(a)) for s, a in taps[1]] == [(2, f(0.2)), (3, f(0.4)), (4, f(0.4))]
    for ssize, dsize in ((3056, 256), (2544, 213), (300, 256), (1033, 256)):
        scale = 1.0 / (float(dsize) / ssize)
        for t in co.area_tab(ssize, dsize, scale):
            assert abs(sum(float(a) for _, a in t) - 1.0) < 1e-5
            idx = [s for s, _ in t]
            assert idx == list(range(idx[0], idx[0] + len(idx))) and 0 <= idx[0] and idx[-1] < ssize


def exact_area_mean(src, dh, dw):
    """independent float64 area average over the real-valued cells (no rounding order games)"""
    sh, sw = src.shape

    def weights(ssize, dsize):
        m = np.zeros((dsize, ssize))
        sc = ssize / dsize
        for d in range(dsize):
            lo, hi = d * sc, (d + 1) * sc
            for s in range(int(np.floor(lo)), min(int(np.ceil(hi)), ssize)):
                m[d, s] = max(0.0, min(hi, s + 1) - max(lo, s)) / sc
        return m
    return weights(sh, dh) @ src.astype(np.float64) @ weights(sw, dw).T


@pytest.mark.parametrize("shape", [(300, 256), (777, 1033), (512, 501), (1000, 1024)])
def test_general_area_path_is_an_area_average(shape):
    rng = np.random.default_rng(sum(shape))
    src = rng.integers(0, 256, size=shape, dtype=np.uint8)
    dh, dw, _, _ = co.resize_plan(*shape, 256)
    got = co.resize_area_u8(src, dh, dw).astype(np.float64)
    want = exact_area_mean(src, dh, dw)
    assert np.abs(got - want).max() <= 0.5 + 1e-3                        # correct rounding of the exact mean up to fp32 noise
    const = np.full(shape, 201, dtype=np.uint8)
    assert (co.resize_area_u8(const, dh, dw) == 201).all()


def test_process_img_layout_and_range():
    rng = np.random.default_rng(1)
    imgs = [rng.integers(0, 256, size=s, dtype=np.uint8) for s in ((300, 400), (512, 512))]
    out = co.process_img(imgs, [(3, 30), (16, 16)])
    assert out.shape == (2, 3, 224, 224) and out.dtype == np.float32
    assert (out[:, 0] == out[:, 1]).all() and (out[:, 0] == out[:, 2]).all()
    assert out.min() >= -1 and out.max() <= 1
    top = co.resize_plan(300, 400, 256)[2]                               # zero padding rows -> (0/255 - .5)/.5 = -1
    assert (out[0, 0, : top - 3] == -1).all()
    frame = co.resize_img(imgs[1], 256)
    np.testing.assert_array_equal(out[1, 0], (frame[16:240, 16:240].astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5))
    small = co.resize_img(np.full((100, 200), 9, dtype=np.uint8), 256)   # long side below imsize: enlarged, then padded
    assert small.shape == (256, 256) and (small[64:192] == 9).all() and (small[:64] == 0).all() and (small[192:] == 0).all()


# ---------------------------------------------------------------- text half (host logic of the product)
VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "no", "acute", "cardio", "##pulmonary", "process", "the", "heart",
         "is", "normal", "in", "size", "lungs", "are", "clear", "pleural", "effusion", "##s", "2", "mm"]


@pytest.fixture()
def collate_fn(tmp_path):
    from gloria.config import pretrain_config
    from gloria.datasets.collate import GloriaCollateFn
    vocab = tmp_path / "vocab.txt"
    vocab.write_text("\n".join(VOCAB) + "\n")
    cfg = pretrain_config("imagenome", batch_size=4)
    cfg.set_path("model.text.bert_type", str(vocab))
    cfg.set_path("data.text.word_num", 16)
    return GloriaCollateFn(cfg, "train", device="cpu")


def test_clean_report():
    from gloria.datasets.collate import clean_report
    text = "FINDINGS:\n1. The heart is normal in size. 2. No pleural effusions\nLungs are clear. X. café 2mm."
    # "FINDINGS:" and "X" are one-token sentences: dropped (mimic_for_gloria.py:209-210)
    assert clean_report(text) == "the heart is normal in size no pleural effusions lungs are clear caf 2mm"

    class First:
        @staticmethod
        def randint(a, b):
            return b
    assert clean_report(text, full_report=False, rng=First) == "caf 2mm"


def test_process_text_ids_and_cap_lens(collate_fn):
    out = collate_fn.process_text(["No acute cardiopulmonary process.", "1. The heart is normal. 2. Lungs are clear, zzz."], "cpu")
    ids = out["caption_ids"]
    assert ids.shape == (2, 16) and out["attention_mask"].shape == (2, 16) and ids.dtype == torch.int64
    w = {t: i for i, t in enumerate(VOCAB)}
    assert ids[0, :7].tolist() == [w["[CLS]"], w["no"], w["acute"], w["cardio"], w["##pulmonary"], w["process"], w["[SEP]"]]
    assert ids[0, 7:].eq(0).all() and out["attention_mask"][0].sum() == 7
    assert ids[1, :10].tolist() == [w["[CLS]"], w["the"], w["heart"], w["is"], w["normal"], w["lungs"], w["are"], w["clear"],
                                    w["[UNK]"], w["[SEP]"]]
    # cap_lens counts tokens not starting with "[" (word pieces included) + 1   (mimic_for_gloria.py:252-254)
    assert out["cap_lens"] == [5 + 1, 7 + 1]
    single = collate_fn.process_text(["Lungs are clear."], "cpu")
    assert single["caption_ids"].shape == (1, 16)


def test_crop_offsets_follow_torchvision_rng_order(collate_fn):
    torch.manual_seed(7)
    got = collate_fn.crop_offsets(3)
    torch.manual_seed(7)
    want = [(int(torch.randint(0, 33, (1,))), int(torch.randint(0, 33, (1,)))) for _ in range(3)]
    assert got == want
    collate_fn.split = "valid"
    assert collate_fn.crop_offsets(2) == [(16, 16)] * 2 == [co.center_crop_offset()] * 2


def test_collate_images_needs_the_gpu(collate_fn):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        collate_fn.process_img([np.zeros((300, 300), dtype=np.uint8)], "cpu")
    with pytest.raises(RuntimeError):                      # small images are enlarged - on the GPU like the others
        collate_fn.process_img([np.zeros((100, 100), dtype=np.uint8)], "cpu")


# ---------------------------------------------------------------- enlarging branch of cv2.INTER_AREA (oracle; unpinned)
def test_enlarging_resize_known_answers():
    # worked by hand from cv::resize's coefficient loop (area_mode, 11-bit fixed point), see the oracle's docstring
    assert co.resize_area_u8(np.array([[0, 255]], np.uint8), 1, 4).tolist() == [[0, 0, 255, 255]]
    assert co.resize_area_u8(np.array([[0, 255]], np.uint8), 1, 3).tolist() == [[0, 128, 255]]
    assert co.resize_area_u8(np.array([[0], [255]], np.uint8), 3, 1).tolist() == [[0], [128], [255]]
    rng = np.random.default_rng(5)
    const = np.full((50, 60), 77, np.uint8)
    assert (co.resize_area_u8(const, 213, 256) == 77).all()
    img = rng.integers(0, 256, (90, 120), dtype=np.uint8)
    up = co.resize_area_u8(img, 192, 256)
    assert up.shape == (192, 256) and up.min() >= img.min() and up.max() <= img.max()       # convex combinations
    assert abs(float(up.mean()) - float(img.mean())) < 1.0
    frame = co.resize_img(img, 256)                                                          # 90 x 120 -> 192 x 256, padded
    assert frame.shape == (256, 256) and (frame[:32] == 0).all() and (frame[224:] == 0).all()
    assert np.array_equal(frame[32:224], up)
    # one direction enlarged, the other kept: the x coefficients degenerate to a copy
    tall = rng.integers(0, 256, (40, 256), dtype=np.uint8)
    assert np.array_equal(co.resize_area_u8(tall, 80, 256)[::2], tall)


# ---------------------------------------------------------------- random transforms: the oracle against Pillow itself
def test_transform_oracle_matches_pillow():
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageEnhance
    rng = np.random.default_rng(0)
    for size in (224, 97):
        img = rng.integers(0, 256, (size, size), dtype=np.uint8)
        pil = Image.fromarray(img, "L").convert("RGB")                # process_img: PIL "L" -> "RGB" (:127-128)
        assert np.array_equal(np.asarray(pil.transpose(Image.FLIP_LEFT_RIGHT))[..., 0], co.hflip(img))
        c = size * 0.5
        for angle, tr, sc in [(0.0, (5, -7), 1.1), (0.0, (0, 0), 0.9), (12.5, (3, 4), 1.05), (-30.0, (-10, 8), 0.95),
                              (90.0, (0, 0), 1.0), (0.0, (0, 0), 1.0), (7.3, (22, -22), 1.2), (179.9, (1, 1), 0.8)]:
            m = co.inverse_affine_matrix((c, c), angle, tr, sc)
            ref = np.asarray(pil.transform(pil.size, Image.AFFINE, m, Image.NEAREST))
            got = co.affine_nearest_u8(img, m)
            assert np.array_equal(ref[..., 0], got) and np.array_equal(ref[..., 2], got), (size, angle, tr, sc)
        for f in [0.0, 1.0, 0.3, 0.8, 0.9999, 1.2, 1.7, 0.123456789, 2.5]:
            assert np.array_equal(np.asarray(ImageEnhance.Brightness(pil).enhance(f))[..., 0], co.adjust_brightness(img, f)), f
            assert np.array_equal(np.asarray(ImageEnhance.Contrast(pil).enhance(f))[..., 0], co.adjust_contrast(img, f)), f
    # Compose order: flip -> affine -> colour steps as drawn
    aug = {"flip": True, "affine": (10.0, (4, -3), 1.1), "jitter": [("contrast", 1.3), ("brightness", 0.7)]}
    ref = pil.transpose(Image.FLIP_LEFT_RIGHT)
    ref = ref.transform(ref.size, Image.AFFINE, co.inverse_affine_matrix((c, c), 10.0, (4, -3), 1.1), Image.NEAREST)
    ref = ImageEnhance.Brightness(ImageEnhance.Contrast(ref).enhance(1.3)).enhance(0.7)
    assert np.array_equal(np.asarray(ref)[..., 1], co.augment(img, aug))


def test_augmentation_draws_follow_torchvision_order(collate_fn):
    from gloria.datasets import collate as C
    torch.manual_seed(11)
    got = C.draw_augmentation(0.5, {"degrees": 10, "translate": [0.1, 0.05], "scale": [0.9, 1.1]},
                              {"brightness": [0.8, 1.2], "contrast": [0.7, 1.3]}, 224)
    torch.manual_seed(11)
    flip = bool(torch.rand(1) < 0.5)
    angle = float(torch.empty(1).uniform_(-10.0, 10.0).item())
    tx = int(round(torch.empty(1).uniform_(-22.4, 22.4).item()))
    ty = int(round(torch.empty(1).uniform_(-11.2, 11.2).item()))
    sc = float(torch.empty(1).uniform_(0.9, 1.1).item())
    jit = []
    for fn_id in torch.randperm(4).tolist():
        if fn_id == 0:
            jit.append(("brightness", float(torch.tensor(1.0).uniform_(0.8, 1.2).item())))
        if fn_id == 1:
            jit.append(("contrast", float(torch.tensor(1.0).uniform_(0.7, 1.3).item())))
    assert got == {"flip": flip, "affine": (angle, (tx, ty), sc), "jitter": jit}
    assert len(jit) == 2 and -10 <= angle <= 10 and abs(tx) <= 23 and abs(ty) <= 12
    # the collate function takes the reference's transform keys (builder.py:167-186, `bightness` spelled as there)
    cfg = collate_fn.cfg
    cfg.set_path("transforms.random_horizontal_flip", 0.5)
    cfg.set_path("transforms.random_affine", {"degrees": 10, "translate": [0.1, 0.1], "scale": [0.9, 1.1]})
    cfg.set_path("transforms.color_jitter", {"bightness": [0.8, 1.2], "contrast": [1.0, 1.0]})
    fn = C.GloriaCollateFn(cfg, "train", device="cpu", tokenizer=collate_fn.tokenizer)
    assert fn.augmented() and fn.jitter == {"brightness": [0.8, 1.2], "contrast": None}
    crops, augs = fn.draw_params(3)
    assert len(crops) == len(augs) == 3 and all(len(a["jitter"]) == 1 for a in augs)
    assert not C.GloriaCollateFn(cfg, "valid", device="cpu", tokenizer=collate_fn.tokenizer).augmented()
    for k in ("random_horizontal_flip", "random_affine", "color_jitter"):
        cfg.set_path("transforms." + k, None)


# ---------------------------------------------------------------- segmentation labels from boxes (host logic)
def _label_by_rendering(shape, boxes, offs):
    """the reference's route (mimic_for_gloria.py:45-55, 110-118) through the oracle: render each box as a full-size
    0/255 mask, process_img it, threshold the normalised tensor at > 0, take the bounding box, OR the boxes"""
    label = np.zeros((224, 224), dtype=bool)
    found = []
    for box, off in zip(boxes, offs):
        mask = np.zeros(shape, dtype=np.uint8)
        mask[box[1]:box[3] + 1, box[0]:box[2] + 1] = 255
        m = co.process_img([mask], [off])[0, 0] > 0
        if m.any():
            ys, xs = np.nonzero(m.any(1))[0], np.nonzero(m.any(0))[0]
            found.append([int(xs[0]), int(ys[0]), int(xs[-1]), int(ys[-1])])
            label[ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] = True
        else:
            found.append([-1, -1, -1, -1])
    return label, found


@pytest.mark.parametrize("shape", [(777, 1033), (512, 512), (1024, 1000), (256, 200), (300, 256), (1900, 1500)])
def test_boxes_to_labels_match_rendered_masks(shape, collate_fn):
    from gloria.datasets.collate import resized_box
    h, w = shape
    rng = np.random.default_rng(h + w)
    boxes = []
    for _ in range(6):
        x0, y0 = int(rng.integers(0, w - 2)), int(rng.integers(0, h - 2))
        boxes.append([x0, y0, int(rng.integers(x0, w)), int(rng.integers(y0, h))])
    boxes += [[0, 0, 3, 3], [w - 2, h - 2, w - 1, h - 1], [w // 3, h // 3, w // 3, h // 3]]      # corners, a single pixel
    offs = [(int(a), int(b)) for a, b in rng.integers(0, 33, size=(len(boxes), 2))]
    want_label, want_boxes = _label_by_rendering(shape, boxes, offs)
    got_boxes = [resized_box(h, w, b, 256, 224, o) for b, o in zip(boxes, offs)]
    assert got_boxes == want_boxes
    # through the collate object: the crop windows come from torch's RNG in the reference's order
    collate_fn.split = "train"
    torch.manual_seed(3)
    offs2 = collate_fn.crop_offsets(len(boxes))
    torch.manual_seed(3)
    got = collate_fn.get_segmentation_labels([boxes], [shape], (224, 224), "cpu")
    assert got.dtype == torch.bool and got.shape == (1, 224, 224)
    assert np.array_equal(got[0].numpy(), _label_by_rendering(shape, boxes, offs2)[0])


def test_degenerate_boxes_are_empty():
    from gloria.datasets.collate import resized_box
    assert resized_box(600, 500, [0, 0, 499, 599], 256, 224, (16, 16)) == [-1, -1, -1, -1]      # whole image: 0/0 in the reference
    assert resized_box(600, 500, [10, 10, 5, 5], 256, 224, (16, 16)) == [-1, -1, -1, -1]
    assert resized_box(3000, 2500, [100, 100, 101, 101], 256, 224, (16, 16)) == [-1, -1, -1, -1]  # < half a pixel after resizing


def test_call_with_sentence_instances_builds_sorted_labels(collate_fn, monkeypatch):
    """__call__ on ImaGenome-style sentence instances (mimic_for_gloria.py:66-84): labels follow the caption-length
    sort like every other field.  The image kernels need a GPU, so process_img is stubbed here."""
    collate_fn.split = "valid"
    monkeypatch.setattr(collate_fn, "process_img", lambda images, device, minmax=None: torch.zeros(len(images), 3, 224, 224))
    raw = [torch.zeros(600, 500, dtype=torch.int16), torch.zeros(512, 512, dtype=torch.int16)]
    boxes = [[[50, 60, 300, 400]], [[0, 0, 255, 255], [256, 256, 511, 511]]]
    sents = ["The heart is normal.", "Lungs are clear. No pleural effusions."]
    instances = [{f"p{i}": {f"s{i}": {"images": {f"d{i}": raw[i]}, "sentence": sents[i], "sent_id": 0,
                                     "objects": {f"d{i}": {"sent_to_bboxes": [{"coords_original": boxes[i]}]}}}}}
                 for i in range(2)]
    batch = collate_fn(instances)
    assert batch["cap_lens"].tolist() == [8, 5]          # "effusions" is two word pieces
    lab = batch["segmentation_labels"]
    assert lab.shape == (2, 224, 224) and lab.dtype == torch.bool
    want0, _ = _label_by_rendering((512, 512), boxes[1], [(16, 16)] * 2)          # longer caption first
    want1, _ = _label_by_rendering((600, 500), boxes[0], [(16, 16)])
    assert np.array_equal(lab[0].numpy(), want0) and np.array_equal(lab[1].numpy(), want1)
    assert [next(iter(i.keys())) for i in batch["instances"]] == ["p1", "p0"]


def test_multimodal_collate_sorts_by_caption_length():
    from gloria.datasets.collate import multimodal_collate_fn
    def sample(i, n):
        ids = torch.zeros(1, 8, dtype=torch.int64); ids[0, :n] = i + 1
        cap = {"input_ids": ids, "token_type_ids": torch.zeros(1, 8, dtype=torch.int64), "attention_mask": (ids > 0).long()}
        return torch.full((3, 4, 4), float(i)), cap, n, f"path{i}"
    out = multimodal_collate_fn([sample(0, 3), sample(1, 7), sample(2, 5)])
    assert out["cap_lens"].tolist() == [7, 5, 3]
    assert out["caption_ids"].shape == (3, 8) and out["caption_ids"][:, 0].tolist() == [2, 3, 1]
    assert out["imgs"][:, 0, 0, 0].tolist() == [1.0, 2.0, 0.0]
    assert out["attention_mask"].sum(1).tolist() == [7, 5, 3]
    assert out["path"] == ["path0", "path1", "path2"]                      # not permuted (pretraining_dataset.py:279)


def test_product_tap_tables_equal_the_oracle_over_many_sizes():
    """the host copy of OpenCV's tap table (used for the box labels) and the oracle's agree entry by entry"""
    from gloria.datasets.collate import _area_taps
    rng = np.random.default_rng(21)
    for _ in range(60):
        ssize = int(rng.integers(257, 4200))
        dsize = int(rng.integers(max(2, ssize // 40), min(ssize, 256) + 1))
        scale = 1.0 / (float(dsize) / ssize)
        idx, alpha = _area_taps(ssize, dsize, scale)
        for d, taps in enumerate(co.area_tab(ssize, dsize, scale)):
            assert [(int(idx[d, k]), alpha[d, k]) for k in range(len(taps))] == [(s, a) for s, a in taps]
            assert (alpha[d, len(taps):] == 0).all()
