"""configs[4] (e4m3 towers) against north_star's 1e-3 gate: what 8-bit operands cost by themselves, measured on the CPU oracle
alone (config A's sizes, random-init weights, B = 2), for the two scale layouts gfx950's MFMA offers.

The GPU path quantises the towers' LayerNorm outputs with ONE scale per token and the QKV / fc1 weights with one scale per
output channel, and feeds `v_mfma_scale_f32_16x16x128_f8f6f4` block scales of 2^0 (csrc/gemm_fp8.hip).  The round-4 verdict
asked for the hardware's own block scales instead -- MXFP8: an E8M0 (power of two) scale per 32 elements along K for both
operands -- or a statement that the gate cannot be met.  This test is that statement's evidence: in an otherwise fp32 model
(no 16-bit rounding anywhere), e4m3 operands on exactly the GEMMs the GPU path quantises put the logits at ~1e-2 of their
range under EITHER layout.  An e4m3 element has 3 mantissa bits (relative step 2^-4 at the top of a binade, 2^-3 at the
bottom); a scale only decides where the element range sits.  Per-token / per-channel scales already use the full range per row;
block scales shrink the dynamic range a scale has to cover (32 values instead of 768) but cannot add mantissa bits, and
power-of-two scales waste up to one binade of the element range that a real-valued per-row scale does not.  The logit error
is therefore a property of the 8-bit operand format, ten times the gate, and configs[4] stays a throughput diagnostic."""
import torch
import torch.nn.functional as F

from oracle import bifold_oracle as O

E4M3_MAX = 448.0


def _e4m3(t):
    return t.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn).float()


def _rowwise(x, w):
    """x [.., K] one real scale per token (row), w [N, K] one per output channel: what csrc/gemm_fp8.hip computes."""
    sx = x.abs().amax(-1, keepdim=True).clamp_min(1e-30) / E4M3_MAX
    sw = w.abs().amax(-1, keepdim=True).clamp_min(1e-30) / E4M3_MAX
    return F.linear(_e4m3(x / sx), _e4m3(w / sw)) * sx * sw.t().reshape(*([1] * (x.dim() - 1)), -1)


def _mx(t):
    """OCP MXFP8 (e4m3 elements): an E8M0 scale per 32 elements along the last axis, 2^(floor(log2(amax)) - 8)."""
    k = t.shape[-1]
    b = t.reshape(*t.shape[:-1], k // 32, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(2.0 ** -120)
    s = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
    return (_e4m3(b / s) * s).reshape(t.shape)


def _blockwise(x, w):
    return F.linear(_mx(x), _mx(w))


def test_e4m3_operands_alone_put_the_logits_ten_gates_out_under_either_scale_layout(monkeypatch, observed):
    g = torch.Generator().manual_seed(3407)
    state = {**O.make_clip_state(g), **O.make_tail_state(g)}
    sample = O.synthetic_sample(2, g)
    quantised = {id(v) for k, v in state.items()
                 if k.startswith("clip_encoder.") and (k.endswith("attn.in_proj_weight") or k.endswith("mlp.c_fc.weight"))}
    assert len(quantised) == 48                       # QKV and fc1 of 12 + 12 tower blocks: the GEMMs behind a LayerNorm

    def logits():
        with torch.no_grad():
            out = O.rgb_clip_forward(state, sample, heads=16)
        return {k: torch.logit(v.double()) for k, v in out.items() if k.endswith("heatmap")}

    def rel(a, b):
        return max(((a[k] - b[k]).abs().max() / b[k].abs().max()).item() for k in b)

    ref = logits()
    plain = F.linear

    class _F:                                          # the oracle's `F`, with the tower GEMMs' operands through e4m3
        def __init__(self, product):
            self.product = product

        def __getattr__(self, name):
            return getattr(F, name)

        def linear(self, x, w, b=None):
            if id(w) not in quantised:
                return plain(x, w, b)
            y = self.product(x, w)
            return y if b is None else y + b

    monkeypatch.setattr(O, "F", _F(_rowwise))
    row = rel(logits(), ref)
    monkeypatch.setattr(O, "F", _F(_blockwise))
    block = rel(logits(), ref)
    observed("fp8_operand_floor_logits", per_token_per_channel=row, mx_block32=block)
    # measured (this container, seed 3407): see DESIGN.md section 5 / 6 -- both an order of magnitude beyond the 1e-3 gate
    assert row > 3e-3 and block > 3e-3
    assert block > 0.5 * row                           # block scales do not buy the missing mantissa bits
