"""configs[2] with the LoRA adapters trained (siglip_sequential @384, T = 4, B = 128): throughput of five consecutive
runs in ONE process and the peak of torch's allocator, per recompute policy (`lora_recompute`, siglip_encoder.py).
    python tools/lora_recompute_bench.py [policy ...]     (default: mlp full none)"""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bifold.losses import Losses
from bifold.models import Models
from bifold.optim import Optimizers
dev = torch.device("cuda:0")
B = int(os.environ.get("BATCH", "128"))
policies = sys.argv[1:] or ["mlp", "full", "none"]
sample = bench.synthetic_sequential_batch(B, dev, seed=3407)
for policy in policies:
    torch.manual_seed(3407)
    cfg = dict(name="siglip_sequential", image_size=384, is_bimanual=True, requires_graph=False, patch_size=16,
               automodel_name="google/siglip-base-patch16-384", dim=768, emb_dropout=0.0, lora=True, r=8, lora_alpha=32,
               lora_dropout=0.0, target_modules=["q_proj", "v_proj"], threshold=0.5, text_encoder=None,
               pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", depth=8, heads=16,
               mlp_ratio=4, dropout=0.0, dtype="fp16", context_length=4, lora_recompute=policy)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = Models.get_by_name(cfg, device=dev).to(dev)
    model.train()
    loss_fn = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=True, mask_pick_heatmap=False))
    opt = Optimizers.get_by_name(cfg=dict(name="adam", lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0),
                                 params=[p for p in model.parameters() if p.requires_grad])
    def step():
        out = model(sample); loss, _ = loss_fn(out, sample); opt.zero_grad(); loss.backward(); opt.step(); return loss
    torch.cuda.reset_peak_memory_stats()
    runs = []
    for run in range(5):
        for _ in range(2): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): loss = step()
        torch.cuda.synchronize(); runs.append(B * 4 / (time.perf_counter() - t0))
    print(f"lora_recompute={policy:5s} B={B}: samples/s per run {' '.join(f'{r:6.1f}' for r in runs)} | peak allocated "
          f"{torch.cuda.max_memory_allocated() / 2**30:6.1f} GiB, reserved {torch.cuda.max_memory_reserved() / 2**30:6.1f} GiB | "
          f"loss {float(loss):.6f}", flush=True)
    del model, opt, loss
    torch.cuda.empty_cache()
