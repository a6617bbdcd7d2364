import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gloria-nlp-project_amd"))
import torch
from gloria.config import pretrain_config
from gloria.models import text_model as TM
from gloria import hipgraph
warnings.simplefilter("always")
cfg = pretrain_config("imagenome", batch_size=8)
if os.environ.get("P0") == "1":
    cfg.model.text.bert_config = dict(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(3)
enc = TM.BertEncoder(cfg).to("cuda:0").train()
for m in enc.modules():
    if isinstance(m, torch.nn.Linear):
        m.to(torch.bfloat16)
B = 8
ids = torch.randint(5, 1000, (B, 97), device="cuda:0"); ids._glr_host = ids.cpu().numpy()
am = torch.ones_like(ids); tt = torch.zeros_like(ids)
orig = hipgraph.consistent
def verbose(name, ref, reps, rel_tol=5e-2):
    den = sum(float(t.float().pow(2).sum()) for t in ref) ** 0.5
    for k, rep in enumerate(reps):
        worst = sorted(((float((a - b).norm()) / (float(a.norm()) + 1e-12), i, tuple(a.shape)) for i, (a, b) in enumerate(zip(ref, rep))), reverse=True)[:5]
        tot = sum(float((a - b).pow(2).sum()) for a, b in zip(ref, rep)) ** 0.5 / den
        print("replay", k, "total rel", tot, "worst per-tensor", worst, flush=True)
    return orig(name, ref, reps, rel_tol)
hipgraph.consistent = verbose
with torch.autocast("cuda", dtype=torch.bfloat16):
    print("enabled:", enc.enable_graph(ids, am, tt, torch.bfloat16))
# trace of the keys: eager sites vs the cell
from gloria.models import rng as R
import gloria.models.fused_ln as FL, gloria.models.fused_attn as FA
log = []
orig_pa = R.philox_args
def traced(dev):
    r = orig_pa(dev)
    log.append(("site", r[0], r[1], r[2] is not None))
    return r
FL.philox_args = traced; FA.philox_args = traced
orig_refresh = R.GraphRng.refresh
def refresh(self):
    orig_refresh(self)
    torch.cuda.synchronize()
    log.append(("cell", [int(v) for v in self.cell.cpu()], self.slots))
R.GraphRng.refresh = refresh
if os.environ.get("P0") != "1":
    enc2 = TM.BertEncoder(cfg).to("cuda:0").train()
    for m in enc2.modules():
        if isinstance(m, torch.nn.Linear):
            m.to(torch.bfloat16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        enc2.enable_graph(ids, am, tt, torch.bfloat16)
    sites = [l for l in log if l[0] == "site"]
    print("n site draws", len(sites), "capturing draws", sum(1 for l in sites if l[3]))
    print("last 40 eager (non-capture) draws:", [(l[1] % 1000, l[2]) for l in sites if not l[3]][-40:][:6], "...")
    print("cells:", [l for l in log if l[0] == "cell"])
    print("capture draws:", [(l[2]) for l in sites if l[3]][:8])
