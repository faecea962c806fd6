"""Which Python call sites issue the ~116 small device copies (__amd_rocclr_copyBuffer) of a training step?
Wraps Tensor.copy_ / clone / contiguous (when it copies) / to and counts call sites over one step of config A."""
import collections, os, sys, traceback, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (for synthetic_batch)
from bifold.losses import Losses
from bifold.models import Models
from bifold.optim import Optimizers
dev = torch.device("cuda:0")
torch.manual_seed(3407)
cfg = dict(name="rgb_clip", image_size=224, is_bimanual=False, requires_graph=False, text_encoder="ViT-B/16", patch_size=16,
           depth=8, heads=16, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
           pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype="fp16")
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model = Models.get_by_name(cfg, device=dev).to(dev)
model.train()
loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=False, mask_pick_heatmap=False))
opt = Optimizers.get_by_name(cfg=dict(name="adam", lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0),
                             params=[p for p in model.parameters() if p.requires_grad])
sample = bench.synthetic_batch(64, dev, 3407)
def step():
    out = model(sample); loss, _ = loss_fn(out, sample); opt.zero_grad(); loss.backward(); opt.step(); return loss
for _ in range(2): step()
torch.cuda.synchronize()
sites = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        r = orig(self, *a, **k)
        if self.is_cuda and (name != "contiguous" or r.data_ptr() != self.data_ptr()):
            fr = [x for x in traceback.extract_stack()[:-1] if "/repo/" in x.filename and "find_copies" not in x.filename]
            sites[(name, " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno}" for x in fr[-3:]))] += 1
        return r
    setattr(torch.Tensor, name, f)
for n in ("copy_", "clone", "contiguous", "to", "float", "zero_", "fill_"):
    wrap(n)
step()
torch.cuda.synchronize()
for (name, where), c in sites.most_common(40):
    print(f"{c:4d}  {name:10s} {where}")
