"""Pins of the two encoders that FEED the hot path.

Their arithmetic lives in third-party packages the reference pins (transformers==4.2.1 BertModel,
torchvision==0.8.2 resnet50; /root/reference/requirements.txt:81,83) and the reference holds no fixtures for
them, so what can be pinned is the ARCHITECTURE:

  * BERT: `transformers.BertModel(BertConfig)` is constructible offline in this image (random init, nothing is
    fetched).  Its state_dict must load into `gloria.models.bert.BertModel` key for key, and every hidden state
    and the pooler output must agree (the reference consumes outputs[2][-4:], text_model.py:94-103).
  * ResNet-50: torchvision is not importable, so the pin is an INDEPENDENT functional restatement of
    torchvision's resnet50 (v1.5: stride on the 3x3 convolution) written here with F.conv2d / F.batch_norm /
    F.max_pool2d over a flat state_dict, compared with `ImageEncoder.resnet_forward` (vision_model.py:67-86)
    in train mode (batch statistics) and in eval mode.  Parity with torchvision itself stays unpinned.
"""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gloria.config import load_config  # noqa: F401  (import check of the package)
from gloria.models import bert as gbert


def _hf_pair(hidden=64, layers=3, heads=4, inter=128, vocab=400, seed=0):
    tr = pytest.importorskip("transformers")
    torch.manual_seed(seed)
    hc = tr.BertConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                       intermediate_size=inter, max_position_embeddings=128, type_vocab_size=2)
    hf = tr.BertModel(hc).eval()
    ours = gbert.BertModel(gbert.BertConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers,
                                            num_attention_heads=heads, intermediate_size=inter,
                                            max_position_embeddings=128)).eval()
    sd = hf.state_dict()
    missing, unexpected = ours.load_state_dict(sd, strict=False)
    assert not missing, missing
    assert all(k.endswith("position_ids") for k in unexpected), unexpected
    return hf, ours


def _bert_inputs(B=5, L=97, vocab=400, seed=1):
    rng = np.random.default_rng(seed)
    lens = [L, 40, 17, 3, 1][:B]
    ids = np.zeros((B, L), dtype=np.int64)
    mask = np.zeros((B, L), dtype=np.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(1, vocab, size=n)
        mask[b, :n] = 1
    tt = np.zeros((B, L), dtype=np.int64)
    tt[0, 50:] = 1
    return torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(tt)


def test_bert_matches_transformers_bertmodel():
    hf, ours = _hf_pair()
    ids, mask, tt = _bert_inputs()
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=mask, token_type_ids=tt, output_hidden_states=True)
        last, pooled, hidden = ours(ids, mask, tt)
    assert len(hidden) == len(ref.hidden_states) == 4
    keep = mask.bool()
    for a, b in zip(hidden, ref.hidden_states):
        # rows of padded QUERY positions are unspecified in both implementations; real tokens must agree
        np.testing.assert_allclose(a[keep].numpy(), b[keep].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(last[keep].numpy(), ref.last_hidden_state[keep].numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pooled.numpy(), ref.pooler_output.numpy(), rtol=1e-5, atol=1e-6)


def test_bert_base_geometry_keys_match_transformers():
    """BERT-base (the geometry of Bio_ClinicalBERT, text_model.py:18): identical parameter names and shapes"""
    tr = pytest.importorskip("transformers")
    with torch.device("meta"):
        hf = tr.BertModel(tr.BertConfig(vocab_size=28996))
        ours = gbert.BertModel(gbert.BertConfig())
    a = {k: tuple(v.shape) for k, v in hf.state_dict().items() if not k.endswith("position_ids")}
    b = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    assert a == b


@pytest.mark.gpu
def test_bert_gpu_matches_transformers_cpu():
    hf, ours = _hf_pair(hidden=128, layers=4, heads=4, inter=256, seed=3)
    ids, mask, tt = _bert_inputs(seed=4)
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=mask, token_type_ids=tt, output_hidden_states=True)
        ours = ours.to("cuda:0")
        last, pooled, hidden = ours(ids.cuda(), mask.cuda(), tt.cuda())
    keep = mask.bool()
    for a, b in zip(hidden, ref.hidden_states):
        np.testing.assert_allclose(a.cpu()[keep].numpy(), b[keep].numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(pooled.cpu().numpy(), ref.pooler_output.numpy(), rtol=2e-4, atol=2e-5)


# ------------------------------------------------------------------------------------------ ResNet-50

def _bn(sd, p, x, train):
    return F.batch_norm(x, None if train else sd[p + ".running_mean"], None if train else sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], training=train, momentum=0.0, eps=1e-5)


def _bottleneck(sd, p, x, stride, train):
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"]), train))
    out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=1), train))
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]), train)
    if (p + ".downsample.0.weight") in sd:
        x = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), train)
    return F.relu(out + x)


def resnet50_functional(sd, x, train):
    """torchvision resnet50 (v1.5) feature path as the reference drives it (vision_model.py:67-86)."""
    x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=True)
    x = F.relu(_bn(sd, "bn1", F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), train))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    local = None
    for li, blocks in enumerate((3, 4, 6, 3), start=1):
        for bi in range(blocks):
            x = _bottleneck(sd, f"layer{li}.{bi}", x, 2 if (bi == 0 and li > 1) else 1, train)
        if li == 3:
            local = x
    return x.mean((2, 3)), local


def _image_encoder(seed=0):
    from gloria.models import cnn_backbones
    torch.manual_seed(seed)
    model, fd, idim = cnn_backbones.resnet_50(pretrained=False)
    assert (fd, idim) == (2048, 1024)
    # non-trivial BatchNorm state so that eval mode is a real check
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    from gloria.models.vision_model import ImageEncoder
    enc = ImageEncoder.__new__(ImageEncoder)
    torch.nn.Module.__init__(enc)
    enc.model = model
    enc.pool = torch.nn.AdaptiveAvgPool2d((1, 1))
    return enc


@pytest.mark.parametrize("train", [False, True])
def test_resnet50_matches_functional_restatement(train):
    enc = _image_encoder()
    enc.train(train)
    sd = {k: v.clone() for k, v in enc.model.state_dict().items()}
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        g_ref, l_ref = resnet50_functional(sd, x, train)
        g, l = enc.resnet_forward(x, extract_features=True)
    assert l.shape == (2, 1024, 19, 19) and g.shape == (2, 2048)
    np.testing.assert_allclose(l.numpy(), l_ref.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g.numpy(), g_ref.numpy(), rtol=1e-4, atol=1e-5)


def test_resnet50_parameter_names_are_torchvisions():
    """the names a reference checkpoint carries under gloria.img_encoder.model.* (torchvision resnet50)"""
    from gloria.models import cnn_backbones
    model, _, _ = cnn_backbones.resnet_50(pretrained=False)
    keys = set(model.state_dict().keys())
    for k in ("conv1.weight", "bn1.running_var", "bn1.num_batches_tracked", "layer1.0.downsample.0.weight",
              "layer1.0.downsample.1.bias", "layer2.3.conv3.weight", "layer3.5.bn2.weight", "layer4.2.bn3.running_mean"):
        assert k in keys, k
    assert sum(p.numel() for p in model.parameters()) == 23508032       # resnet50 minus the 1000-way classifier


@pytest.mark.gpu
def test_resnet50_gpu_matches_functional_cpu():
    """the same module on the GPU (MIOpen convolutions, fp32) against the CPU functional restatement, eval mode.
    MIOpen's fp32 solvers (Winograd / implicit GEMM on the matrix cores) are not bit-compatible with the CPU's
    direct sums: the error is printed stage by stage and bounded relative to the feature scale; the architecture
    itself is pinned at 1e-4 by the CPU test above."""
    enc = _image_encoder(seed=2).train(False)
    sd = {k: v.clone() for k, v in enc.model.state_dict().items()}
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        g_ref, l_ref = resnet50_functional(sd, x, False)
        enc = enc.to("cuda:0")
        g, l = enc.resnet_forward(x.cuda(), extract_features=True)
        # stage-wise view (stem output) for the log
        m = enc.model
        xs = F.interpolate(x.cuda(), size=(299, 299), mode="bilinear", align_corners=True)
        stem = m.maxpool(torch.relu(m.bn1(m.conv1(xs)))).cpu()
        xc = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=True)
        stem_ref = F.max_pool2d(F.relu(_bn(sd, "bn1", F.conv2d(xc, sd["conv1.weight"], stride=2, padding=3), False)), 3, 2, 1)
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))   # noqa: E731
    print(f"[resnet gpu-vs-functional] stem: {rel(stem.numpy(), stem_ref.numpy()):.2e}")
    for name, a, b in (("local", l.cpu().numpy(), l_ref.numpy()), ("global", g.cpu().numpy(), g_ref.numpy())):
        r = rel(a, b)
        print(f"[resnet gpu-vs-functional] {name}: relative Frobenius error {r:.2e}")
        assert r < 5e-2, (name, r)
