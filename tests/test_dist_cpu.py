"""Data-parallel gradient averaging with world_size 2 over gloo (CPU): the bucket schedule of
GradSync must leave every rank with the mean of the per-rank gradient arenas, which is the
single-process gradient of the concatenated batch when each rank's loss is a local mean."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from bifold import dist as bdist
    r, lr, w = bdist.init_process_group_from_env("gloo")
    assert (r, w) == (rank, world)
    depth = 3
    names, offsets, off = [], {}, 0
    for n, cnt in [("text_token", 64), ("project.weight", 300)] + \
                  [(f"pick_place.fusion.transformer_encoder.layers.{i}.{j}.w", 200 + i) for i in range(depth) for j in (0, 1)] + \
                  [("pick_place.pick_decoder.decoder_net.0.weight", 130)]:
        names.append(n); offsets[n] = (off, cnt); off += (cnt + 63) // 64 * 64
    grad = torch.full((off,), float(rank + 1)) + torch.arange(off) * (rank + 1) * 1e-3
    expect = (torch.full((off,), 1.5) + torch.arange(off) * 1.5e-3)
    ok = True
    for coalesce in (1, 2, 3, None):              # one all-reduce per layer ... everything in the final one
        g = grad.clone()
        s2 = bdist.GradSync(g, bdist.layer_buckets(names, offsets, depth, off), depth, coalesce=coalesce)
        sent = sorted(sum(list(s2.plan.values()), []) + s2.final)
        ok = ok and sent[0][0] == 0 and sent[-1][1] == off and all(sent[i][1] == sent[i + 1][0]
                                                                  for i in range(len(sent) - 1))
        for layer in reversed(range(depth)):
            s2.on_ready(layer)
        s2.on_ready(-1)
        s2.finish()
        ok = ok and torch.allclose(g, expect, rtol=1e-6)
    sync = bdist.GradSync(grad, bdist.layer_buckets(names, offsets, depth, off), depth)
    for layer in reversed(range(depth)):          # the order the fused backward reports completion
        sync.on_ready(layer)
    sync.on_ready(-1)
    sync.finish()
    ok = ok and torch.allclose(grad, expect, rtol=1e-6)
    # a second step must work identically (state reset by finish())
    grad.copy_(torch.full((off,), float(10 * (rank + 1))))
    for layer in reversed(range(depth)):
        sync.on_ready(layer)
    sync.on_ready(-1)
    sync.finish()
    ok = ok and torch.allclose(grad, torch.full((off,), 15.0))
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert dict(out) == {0: True, 1: True}
