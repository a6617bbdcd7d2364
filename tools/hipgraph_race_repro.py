import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import torch
from gloria import builder
from gloria.config import pretrain_config
from gloria.datasets.synthetic import make_batch
from gloria.models import gloria_model as GM
from gloria.trainer import Trainer
B = 64
cfg = pretrain_config("imagenome", batch_size=B)
torch.manual_seed(31)
model = builder.build_lightning_model(cfg, builder.build_data_module(cfg))
tr = Trainer(cfg, device="cuda:0", precision="bf16", graph_text_encoder=False)
tr.setup(model)
model.train()
batch = tr.to_device(make_batch(B, seed=8, lengths="words"))
G = model.gloria
if os.environ.get("NO_GRAPH") != "1":
    print("graph enabled:", G.enable_image_graph(batch["imgs"], torch.bfloat16), "env", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), flush=True)
watch = {n: p for n, p in model.named_parameters() if n.endswith("img_encoder.model.conv1.weight") or n.endswith("layer1.2.conv3.weight") or n.endswith("layer1.0.conv1.weight") or n.endswith("layer3.0.conv1.weight")}
mode = os.environ.get("MODE", "img")
side = torch.cuda.Stream()
for it in range(4):
    for p in model.parameters():
        if os.environ.get("KEEP") == "1":
            if p.grad is not None:
                p.grad.zero_()
        else:
            p.grad = None
    cm = torch.cuda.stream(side) if os.environ.get("SIDE") == "1" else torch.autocast("cuda", enabled=False)
    if os.environ.get("SIDE") == "1":
        side.wait_stream(torch.cuda.current_stream())
    with cm, torch.autocast("cuda", dtype=torch.bfloat16):
        l, g = G.image_encoder_forward(batch["imgs"])
        loss = l.float().pow(2).mean() + g.float().pow(2).mean()
        if mode in ("text", "textbwd"):
            tl, tg, _ = G.text_encoder_forward(batch["caption_ids"], batch["attention_mask"], batch["token_type_ids"])
            if mode == "textbwd":
                loss = loss + tl.float().pow(2).mean() + tg.float().pow(2).mean()
        if mode == "junk":      # eager allocations + kernels on the same stream between the replays
            junk = [torch.randn(64, 97, 768, device="cuda:0") * 2 for _ in range(40)]
    loss.backward()
    torch.cuda.synchronize()
    print("iter", it, {n.split("img_encoder.model.")[1]: round(float(p.grad.float().norm()), 4) for n, p in watch.items()}, flush=True)
