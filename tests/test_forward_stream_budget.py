"""Why the FORWARD residual streams stay fp32: what rounding them to fp16 after every add would cost
against north_star's 1e-3 logit gate, measured on the CPU oracle alone (config A's sizes, random-init weights, B = 2).

The gradient of the residual stream travels in fp16 in the fp16 mode (bifold/models/pickplace.py, _GRAD_STREAM16): the
gradient gates are 1e-2.  The same step on the forward would take a third off every LayerNorm pass -- and, as this test
records, spend half of the logit gate by itself, on top of the 3.9e-4 the 16-bit GEMM operands already use."""
import torch
import torch.nn.functional as F

from oracle import bifold_oracle as O


def _round16(t):
    return t.half().float()


def _fusion_layer_16(state, pfx, x, heads, query_row_mask=None, drop=None):      # O.fusion_layer with the stream rounded after each add
    w = x.shape[-1]
    h = O.layer_norm(x, state[pfx + "0.norm.weight"], state[pfx + "0.norm.bias"])
    qkv = F.linear(h, state[pfx + "0.fn.to_qkv.weight"])
    q, k, v = (O.split_heads(t, heads) for t in qkv.chunk(3, dim=-1))
    o = O.attention_core(q, k, v, scale=(w // heads) ** -0.5, query_row_mask=query_row_mask)
    o = F.linear(O.merge_heads(o), state[pfx + "0.fn.to_out.0.weight"], state[pfx + "0.fn.to_out.0.bias"])
    x = _round16(o + x)
    h = O.layer_norm(x, state[pfx + "1.norm.weight"], state[pfx + "1.norm.bias"])
    h = O.erf_gelu(F.linear(h, state[pfx + "1.fn.net.0.weight"], state[pfx + "1.fn.net.0.bias"]))
    h = F.linear(h, state[pfx + "1.fn.net.3.weight"], state[pfx + "1.fn.net.3.bias"])
    return _round16(h + x)


def _clip_resblock_16(state, pfx, x, heads, causal):                  # O.clip_resblock, likewise
    w = x.shape[-1]
    h = O.layer_norm(x, state[pfx + "ln_1.weight"], state[pfx + "ln_1.bias"])
    qkv = F.linear(h, state[pfx + "attn.in_proj_weight"], state[pfx + "attn.in_proj_bias"])
    q, k, v = (O.split_heads(t, heads) for t in qkv.split(w, dim=-1))
    o = O.attention_core(q, k, v, scale=(w // heads) ** -0.5, causal=causal)
    o = F.linear(O.merge_heads(o), state[pfx + "attn.out_proj.weight"], state[pfx + "attn.out_proj.bias"])
    x = _round16(x + o)
    h = O.layer_norm(x, state[pfx + "ln_2.weight"], state[pfx + "ln_2.bias"])
    h = O.quick_gelu(F.linear(h, state[pfx + "mlp.c_fc.weight"], state[pfx + "mlp.c_fc.bias"]))
    h = F.linear(h, state[pfx + "mlp.c_proj.weight"], state[pfx + "mlp.c_proj.bias"])
    return _round16(x + h)


def test_fp16_forward_residual_streams_would_spend_the_logit_gate(monkeypatch, observed):
    g = torch.Generator().manual_seed(3407)
    state = {**O.make_clip_state(g), **O.make_tail_state(g)}
    sample = O.synthetic_sample(2, g)

    def logits():
        with torch.no_grad():
            out = O.rgb_clip_forward(state, sample, heads=16)
        return {k: torch.logit(v.double()) for k, v in out.items() if k.endswith("heatmap")}

    def rel(a, b):
        return max(((a[k] - b[k]).abs().max() / b[k].abs().max()).item() for k in b)

    ref = logits()
    monkeypatch.setattr(O, "fusion_layer", _fusion_layer_16)
    fusion = rel(logits(), ref)
    monkeypatch.undo()
    monkeypatch.setattr(O, "clip_resblock", _clip_resblock_16)
    towers = rel(logits(), ref)
    monkeypatch.setattr(O, "fusion_layer", _fusion_layer_16)
    both = rel(logits(), ref)
    observed("fp16_forward_residual_stream_cost", fusion=fusion, towers=towers, both=both)
    # measured: 5.3e-4 / 4.0e-4 / 7.3e-4 of a 1e-3 gate -- each of them more than the whole fp16 step's 3.9e-4 today
    assert 2e-4 < fusion < 1e-3 and 2e-4 < towers < 1e-3 and both > fusion
