#!/usr/bin/env python3
"""The frozen CLIP vision tower (12 blocks, B = 256) as ONE batch on one stream against TWO half-batches on two HIP
streams: while one half's persistent GEMM holds its workgroups' CUs, can the other half's LayerNorm / attention passes
(HBM- and latency-bound, far below the power cap) run on the CUs a capped GEMM grid leaves free?

    BIFOLD_P8_GRID=<0|224|208|192> python tools/tower_split_probe.py [--batch 256] [--dtype fp16] [--parts 2]

The step is at the socket's power cap in its GEMMs (profiles/r06_step_gaps.txt), and a GEMM on 192 of 256 CUs is 11 % slower,
not 33 % (profiles/r06_p8_grid_sweep.txt): the question is whether the passes between the GEMMs can be had for free.
Diagnostic only (tools/README.md).
"""
import argparse
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bifold.models import Models  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = dict(name="rgb_clip", image_size=224, is_bimanual=False, requires_graph=False, text_encoder="ViT-B/16",
               patch_size=16, depth=8, heads=16, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
               pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype=args.dtype)
    torch.manual_seed(3407)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = Models.get_by_name(cfg, device=dev).to(dev).eval()
    enc, cdt = m.clip_encoder, m.compute_dtype
    enc._runtime(cdt)
    B = args.batch
    rgb = torch.randn(B, 3, 224, 224, device=dev)
    parts = [rgb[i * (B // args.parts):(i + 1) * (B // args.parts)].contiguous() for i in range(args.parts)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.parts)]

    def whole():
        return enc.encode_image_with_embeddings(rgb, cdt)

    def split():
        main_s = torch.cuda.current_stream()
        outs = []
        for s, p in zip(streams, parts):
            s.wait_stream(main_s)
            with torch.cuda.stream(s):
                outs.append(enc.encode_image_with_embeddings(p, cdt))
        for s in streams:
            main_s.wait_stream(s)
        return outs

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return 1e3 * ts[len(ts) // 2], 1e3 * ts[0]

    with torch.no_grad():
        ref = whole()
        got = torch.cat(split(), dim=0)
        err = ((got.float() - ref.float()).abs().max() / ref.float().abs().max()).item()
        a = timed(whole)
        b = timed(split)
        a2 = timed(whole)
    grid = os.environ.get("BIFOLD_P8_GRID", "0")
    print(f"P8_GRID={grid:>3}  one batch of {B}: {a[0]:.3f} ms (min {a[1]:.3f}; again {a2[0]:.3f})   {args.parts} x {B // args.parts} on "
          f"{args.parts} streams: {b[0]:.3f} ms (min {b[1]:.3f})   ratio {b[0] / a[0]:.3f}   max rel diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
