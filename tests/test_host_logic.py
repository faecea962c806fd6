"""Host-side logic that needs no GPU: plugin factories and their error behaviour, the
state-dict schema, the LR schedule, config composition, the C ABI's exported symbols and its
argument validation (which returns before any kernel launch)."""
import ctypes as C
import os
import re
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY_CLIP = dict(embed_dim=32, image_resolution=64, vision_layers=2, vision_width=128, vision_patch_size=16,
                 context_length=77, vocab_size=512, transformer_width=64, transformer_heads=1,
                 transformer_layers=2)


def tiny_cfg(**over):
    from bifold.models.clip_encoder import CLIP_CONFIGS
    CLIP_CONFIGS["tiny-test"] = TINY_CLIP
    cfg = dict(name="rgb_clip", image_size=64, is_bimanual=False, requires_graph=False, text_encoder="tiny-test",
               patch_size=16, depth=2, heads=2, mlp_ratio=4, dropout=0.0, text_dropout=0.0, rgb_dropout=0.0,
               pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", dtype="fp32")
    cfg.update(over)
    return cfg


def test_header_symbols_are_exported_and_bound():
    from bifold_amd.hip import lib as L
    hdr = open(os.path.join(ROOT, "include", "bifold_hip.h")).read()
    declared = set(re.findall(r"\b(bf_[a-z0-9_]+)\s*\(", hdr)) - {"bf_gemm_args"}
    assert declared, "no declarations found"
    assert declared == set(L.PROTOTYPES), declared ^ set(L.PROTOTYPES)
    for name in declared:
        assert hasattr(L.lib, name)
    assert L.lib.bf_abi_version() == L.ABI_VERSION == 12


def test_c_abi_rejects_bad_arguments_before_launching():
    from bifold_amd.hip import lib as L
    g = L.GemmArgs()
    assert L.lib.bf_gemm(C.byref(g), None) == -1                       # m = n = k = 0
    assert b"m, n, k" in L.lib.bf_last_error()
    assert L.lib.bf_layernorm_fwd(None, 0, None, None, None, 0, None, None, 4, 30, 1e-5, None) == -1
    assert L.lib.bf_attention_fwd(None, None, None, None, 1, 8, 1, 40, 1.0, 0, 1, None) == -1   # head_dim 40
    assert b"head_dim" in L.lib.bf_last_error()
    assert L.lib.bf_adam_step(None, None, None, None, None, 1, 6, 1, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, None, None, None, 0.0, None) == -1
    assert L.lib.bf_scale_check(None, 0, 1.0, None, None) == -1
    g.m = g.n = g.k = 64; g.lda = g.ldb = g.ldc = 64; g.cdt, g.c_dtype = L.F16, L.BF16      # 16-bit output of another type
    assert L.lib.bf_gemm(C.byref(g), None) == -1 and b"c_dtype" in L.lib.bf_last_error()
    assert L.lib.bf_bce_parts(256) == 1024 and L.lib.bf_layernorm_bwd_parts(70400) > 0
    # ABI 11: the LoRA entry points and the GEMM's low-rank addend
    sl = L.LoraSlots()
    sl.n_slots, sl.rank = 2, 4
    assert L.lib.bf_lora_down(None, 64, None, None, None, 8, 64, C.byref(sl), 1.0, L.F16, None) == -1
    assert b"rank 4" in L.lib.bf_last_error()
    sl.rank = 8
    assert L.lib.bf_lora_up(None, None, None, 64, 8, 64, C.byref(sl), 1.0, L.F32, None) == -1 and b"16-bit" in L.lib.bf_last_error()
    assert L.lib.bf_lora_tn_workspace_elems(1000, 768, C.byref(sl)) == 32 * 16 * 768      # 32 chunks of 32 rows
    # ABI 12: the mask head's product and the mask losses
    assert L.lib.bf_mask_mul_fwd(None, None, None, 2, 6, None) == -1 and b"multiple of 4" in L.lib.bf_last_error()
    assert L.lib.bf_mask_mul_bwd(None, None, None, None, None, None, 5, 8, 0, None) == -1 and b"heads" in L.lib.bf_last_error()
    assert L.lib.bf_rows_accumulate(None, 0, None, L.BF16, 0, 1, 8, None) == -1 and b"fp32 or fp16" in L.lib.bf_last_error()
    assert L.lib.bf_dice_fwd(None, None, None, None, 0, 64, None) == -1
    assert L.lib.bf_dice_bwd(None, None, None, None, 2, 6, None) == -1
    assert L.lib.bf_focal_fwd(None, None, None, None, 64, 2000, 1.0, 0.25, 2.0, None) == -1
    assert L.lib.bf_focal_bwd(None, None, None, None, 6, 1.0, 0.25, 2.0, None) == -1 and b"multiple of 4" in L.lib.bf_last_error()
    g2 = L.GemmArgs()
    g2.m, g2.n, g2.k, g2.lda, g2.ldb, g2.ldc, g2.cdt, g2.c_dtype, g2.lr_k = 65536, 2304, 768, 768, 768, 2304, L.F16, L.F16, 8
    assert L.lib.bf_gemm_lowrank_ok(C.byref(g2)) == 0 and L.lib.bf_gemm_lowrank_ok(None) == 0      # lr_k must be 16 or 32


def test_ctypes_structs_measure_what_the_header_declares():
    from bifold_amd.hip import lib as L
    assert L.lib.bf_struct_size(0) == C.sizeof(L.GemmArgs)
    assert L.lib.bf_struct_size(1) == C.sizeof(L.GemmTnProblem)
    assert L.lib.bf_struct_size(2) == C.sizeof(L.LoraSlots)
    assert L.lib.bf_struct_size(3) == 0


def test_missing_library_fails_loudly(tmp_path):
    import subprocess, sys
    code = "import os; os.environ['BIFOLD_HIP_LIB']='/nonexistent/lib.so'; import bifold_amd.hip.lib"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_factories_and_error_behaviour():
    from bifold.losses import Losses
    from bifold.models import Components, Models
    from bifold.optim import Optimizers, Schedulers
    with pytest.raises(ValueError, match="not recognized"):
        Models.get_by_name(dict(name="nope"))
    with pytest.raises(ValueError, match="not recognized"):
        Components.get_by_name("nope")
    with pytest.raises(ValueError, match="not recognized"):
        Losses.get_by_name(cfg=dict(name="nope"))
    with pytest.raises(ValueError, match="not recognized"):
        Optimizers.get_by_name(dict(name="sgd"), [])
    with pytest.raises(ValueError, match="not recognized"):
        Schedulers.get_by_name(dict(name="step"))
    assert Schedulers.get_by_name(dict(name=None)) is None
    with pytest.raises(NotImplementedError):
        Models.get_by_name(dict(name="text_unet"))
    cfg = tiny_cfg()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = Models.get_by_name(cfg, device=torch.device("cpu"))
    assert "name" not in cfg                                   # like the reference: `del cfg.name`
    assert m.threshold == 0.01 and m.image_size == 64
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert frozen and all("clip_encoder" in n for n in frozen)
    assert all("clip_encoder" not in n for n, p in m.named_parameters() if p.requires_grad)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m({"rgb": torch.zeros(1, 3, 64, 64), "instruction": torch.zeros(1, 77, dtype=torch.int32)})
    with pytest.raises(AssertionError):
        Models.get_by_name(tiny_cfg(image_size=70), device=torch.device("cpu"))
    md = Models.get_by_name(tiny_cfg(dropout=0.1, text_dropout=0.2), device=torch.device("cpu"))   # round 6: dropout > 0 is built
    assert md.pick_place.fusion.dropout == 0.1 and md.text_dropout == 0.2 and md.rgb_dropout == 0.0
    with pytest.raises(ValueError, match="dropout"):
        Models.get_by_name(tiny_cfg(dropout=1.0), device=torch.device("cpu"))
    loss = Losses.get_by_name(cfg=dict(name="bce_gaussmap", is_bimanual=True, mask_pick_heatmap=False))
    assert loss.heads() == ["left_pick", "left_place", "right_pick", "right_place"]
    # round 6: the mask head and its losses (conf/loss/bce_mask.yaml, composed.yaml, composed_dice_focal.yaml)
    comp = Losses.get_by_name(cfg=dict(name="composed", loss_names=["bce_gaussmap", "focal", "dice"], weights=[40, 20, 1],
                                       is_bimanual=False, mask_pick_heatmap=True))
    assert list(comp.loss_dict) == ["bce_gaussmap", "focal", "dice"] and comp.weights["focal"] == 20
    assert comp.loss_dict["focal"].alpha == 0.25 and comp.loss_dict["focal"].gamma == 2
    assert comp.loss_dict["bce_gaussmap"].mask_pick_heatmap is True
    mm = Models.get_by_name(tiny_cfg(compute_mask=True, detach_mask=True), device=torch.device("cpu"))
    keys = [n for n, _ in mm.named_parameters() if n.startswith("pick_place.") and "fusion" not in n]
    assert keys[0] == "pick_place.mask_head.decoder_net.0.weight" and keys[10] == "pick_place.pick_decoder.decoder_net.0.weight"
    assert mm.pick_place.has_mask and mm.pick_place.detach_mask and mm.pick_place.decoders_all == ["pick", "place", "mask"]


@pytest.mark.parametrize("fixture,bim", [("tiny_unimanual.npz", False), ("tiny_bimanual.npz", True)])
def test_state_dict_schema_matches_reference(golden_dir, fixture, bim):
    from bifold.models import Models
    z = np.load(os.path.join(golden_dir, fixture))
    ref = {k[6:]: tuple(z[k].shape) for k in z.files if k.startswith("state/")}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = Models.get_by_name(tiny_cfg(is_bimanual=bim), device=torch.device("cpu"))
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == ref
    m.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state/")}, strict=True)


def test_linear_warmup_reproduces_reference_trace(golden_dir):
    from bifold.optim import Schedulers

    class FakeOpt:
        param_groups = [{"lr": 1e-4}]
    opt = FakeOpt()
    sch = Schedulers.get_by_name(dict(name="linear_warmup", warmup_portion=0.1, warmup_start_lr=0,
                                      use_cosine_decay=True), optimizer=opt, max_iters=100)
    trace = []
    for _ in range(30):
        trace.append(opt.param_groups[0]["lr"])
        sch.step()
    ref = np.load(os.path.join(golden_dir, "per_op.npz"))["warmup/lrs"]
    assert np.allclose(trace, ref, rtol=1e-9, atol=1e-15)
    assert trace[0] < 0 and trace[1] == 0.0                    # quirk: -1e-5, 0, 1e-5, ...
    sd = sch.state_dict()
    assert "optimizer" not in sd and sd["last_step"] == sch.last_step


def test_upsample_operator_matches_interpolate():
    import torch.nn.functional as F
    from bifold.models.decoder_math import upsample_operator
    for g in (3, 14, 24):
        U = upsample_operator(g).double()
        x = torch.randn(2, 1, g, g, dtype=torch.float64)
        ref = x
        for _ in range(4):
            ref = F.interpolate(ref, scale_factor=2, mode="bilinear", align_corners=False)
        got = torch.einsum("yi,bcij,xj->bcyx", U, x, U)
        assert torch.allclose(got, ref, atol=1e-6)
        assert torch.allclose(U.sum(1), torch.ones(16 * g, dtype=torch.float64), atol=1e-6)


CONF = os.path.join(ROOT, "bifold", "conf")


def test_config_composition_and_overrides():
    from bifold.config import compose
    cfg = compose(CONF, [])
    # the reference's defaults, unchanged (bifold/conf/config.yaml:2-33)
    assert cfg.model.name == "siglip_sequential" and cfg.train_dataset.name == "bimanual_sequential"
    assert cfg.model.lora is True and cfg.model.lora_dropout == 0.01 and cfg.model.context_length == 3
    assert cfg.use_wandb is True and cfg.simulator == "softgym" and cfg.batch_size == 2 and "hydra" not in cfg
    assert cfg.optim.lr == 1e-4 and isinstance(cfg.optim.lr, float)
    assert cfg.loss.is_bimanual is True and cfg.scheduler.name is None
    cfg = compose(CONF, ["model=rgb_clip", "dataset@train_dataset=synthetic"])
    assert cfg.model.name == "rgb_clip" and cfg.model.is_bimanual is False
    assert cfg.processor.model_image_size == 224 and cfg.processor.text_encoder == "ViT-B/16"
    cfg = compose(CONF, ["model=rgb_clip", "optim=adamw", "scheduler=linear_warmup", "batch_size=8", "model.depth=4",
                         "dataset@train_dataset=single", "train_dataset.n_samples=1000"])
    assert cfg.optim.name == "adamw" and cfg.optim.weight_decay == 0.01
    assert cfg.scheduler.warmup_portion == 0.1 and cfg.batch_size == 8 and cfg.model.depth == 4
    assert cfg.train_dataset.dataset_path.endswith("single_data/All_1000.pkl")
    node = cfg.optim
    del node.name
    assert "name" not in node and dict(**node)["lr"] == 1e-4
    with pytest.raises(FileNotFoundError):
        compose(CONF, ["model=does_not_exist"])


def test_conf_tree_carries_every_reference_file_and_value():
    """bifold/conf holds the reference's 20 YAML files; every key the reference sets keeps its value (new keys
    are allowed, changed values are not).  The expected values below are the reference's (conf/**), restated."""
    import yaml
    want_files = {"config.yaml", "dataset/bimanual.yaml", "dataset/bimanual_sequential.yaml", "dataset/none.yaml",
                  "dataset/real.yaml", "dataset/single.yaml", "dataset/single_sequential.yaml",
                  "loss/bce_gaussmap.yaml", "loss/bce_mask.yaml", "loss/composed.yaml", "loss/composed_dice_focal.yaml",
                  "model/rgb_clip.yaml", "model/siglip.yaml", "model/siglip_sequential.yaml", "model/text_unet.yaml",
                  "optim/adam.yaml", "optim/adamw.yaml", "processor/default.yaml", "scheduler/linear_warmup.yaml",
                  "scheduler/none.yaml"}
    have = {os.path.relpath(os.path.join(d, f), CONF) for d, _, fs in os.walk(CONF) for f in fs}
    assert want_files <= have, want_files - have
    top = yaml.safe_load(open(os.path.join(CONF, "config.yaml")))
    assert top["defaults"][:8] == ["_self_", {"model": "siglip_sequential"},
                                   {"dataset@train_dataset": "bimanual_sequential"}, {"dataset@test_dataset": "none"},
                                   {"processor": "default"}, {"loss": "bce_gaussmap"}, {"optim": "adam"},
                                   {"scheduler": "none"}]
    for k, v in dict(seed=3407, eval_only=False, eval_epochs=20, save_epochs=None, load_best=False, use_wandb=True,
                     num_workers=6, batch_size=2, gradient_clip=None, test_batch_size=2, epochs=100, debug=False,
                     num_evals=50, use_cpu=False, simulator="softgym").items():
        assert top[k] == v, k
    assert top["hydra"]["run"]["dir"] == "${oc.env:HOME}/bifold/outputs/${train_dataset.name}/${hydra.job.override_dirname}"
    assert top["hydra"]["job"]["chdir"] is True and "seed" in top["hydra"]["job"]["config"]["override_dirname"]["exclude_keys"]
    sig = yaml.safe_load(open(os.path.join(CONF, "model", "siglip_sequential.yaml")))
    assert sig["lora"] is True and sig["lora_dropout"] == 0.01 and sig["r"] == 8 and sig["lora_alpha"] == 32
    assert sig["target_modules"] == ["q_proj", "v_proj"] and sig["context_length"] == "${train_dataset.max_context_length}"


def test_run_directory_follows_hydra_override_dirname(monkeypatch):
    """hydra.run.dir = ~/bifold/outputs/<dataset>/<sorted, non-excluded overrides joined by ','>
    (conf/config.yaml:34-59): two runs with different models / learning rates land in different directories,
    the same command line lands where the reference's Hydra would have put it."""
    from bifold.config import compose, override_dirname
    monkeypatch.setenv("HOME", "/home/u")
    ov = ["model=rgb_clip", "dataset@train_dataset=synthetic", "seed=7", "optim.lr=0.001", "batch_size=64", "debug=true"]
    cfg, hydra = compose(CONF, ov, return_hydra=True)
    assert hydra.job.override_dirname == "batch_size=64,model=rgb_clip,optim.lr=0.001"
    assert hydra.run.dir == "/home/u/bifold/outputs/synthetic/batch_size=64,model=rgb_clip,optim.lr=0.001"
    assert hydra.job.chdir is True and hydra.overrides.task == ov
    _, h2 = compose(CONF, ["model=siglip", "dataset@train_dataset=synthetic"], return_hydra=True)
    assert h2.run.dir != hydra.run.dir
    assert override_dirname(["+a.b=1", "z=2", "seed=3"], ("seed",)) == "+a.b=1,z=2"


def test_layer_buckets_tile_the_arena():
    from bifold.dist import layer_buckets
    names, offsets, off = [], {}, 0
    def add(n, cnt):
        nonlocal off
        names.append(n); offsets[n] = (off, cnt); off += (cnt + 63) // 64 * 64
    add("text_token", 64); add("project.weight", 1000)
    for i in range(3):
        add(f"pick_place.fusion.transformer_encoder.layers.{i}.0.norm.weight", 64)
        add(f"pick_place.fusion.transformer_encoder.layers.{i}.1.fn.net.3.bias", 70)
    add("pick_place.pick_decoder.decoder_net.0.weight", 500)
    b = layer_buckets(names, offsets, 3, off)
    assert len(b) == 5 and b[0][1] == off and b[-1][0] == 0
    # a parameter registered BEHIND the decoders (not a decoder) must ride in the final bucket, never in the
    # decoders' bucket that leaves first (stale gradients would be averaged)
    add("late_head.weight", 100)
    b2 = layer_buckets(names, offsets, 3, off)
    assert len(b2) == 6 and b2[0] == (b[0][0], offsets["late_head.weight"][0]) and b2[-1] == (offsets["late_head.weight"][0], off)
    names.pop(); off = offsets.pop("late_head.weight")[0]
    covered = sorted(b)
    assert covered[0][0] == 0 and all(covered[i][1] == covered[i + 1][0] for i in range(4)) and covered[-1][1] == off
    assert b[1] == (offsets[names[6]][0], b[0][0])             # layer 2 comes right after the decoders
    # the mask head (compute_mask=True) is registered BETWEEN the fusion layers and the heads, like the reference's, and its
    # gradient is written with the heads': it rides in the decoders' bucket
    names3, offsets3, off3 = names[:-1], {k: v for k, v in offsets.items() if k != names[-1]}, offsets[names[-1]][0]
    for n, cnt in (("pick_place.mask_head.decoder_net.0.weight", 500), ("pick_place.pick_decoder.decoder_net.0.weight", 500)):
        names3.append(n); offsets3[n] = (off3, cnt); off3 += (cnt + 63) // 64 * 64
    b3 = layer_buckets(names3, offsets3, 3, off3)
    assert len(b3) == 5 and b3[0] == (offsets3["pick_place.mask_head.decoder_net.0.weight"][0], off3)


@pytest.mark.parametrize("name,fixture,extra", [("siglip", "tiny_siglip.npz", {}),
                                                ("siglip_sequential", "tiny_siglip_sequential.npz", {"context_length": 2})])
def test_siglip_state_dict_schema_matches_reference(golden_dir, name, fixture, extra):
    from bifold.models import Models
    from bifold.models.siglip_encoder import SIGLIP_CONFIGS
    SIGLIP_CONFIGS["tiny"] = dict(
        vision=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=32, patch_size=16),
        text=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, vocab_size=100,
                  max_position_embeddings=64))
    z = np.load(os.path.join(golden_dir, fixture))
    ref = {k[6:]: tuple(z[k].shape) for k in z.files if k.startswith("state/")}
    cfg = dict(name=name, image_size=32, is_bimanual=bool(int(z["meta/is_bimanual"])), requires_graph=False, patch_size=16,
               automodel_name="tiny", dim=64, emb_dropout=0.0, lora=False, r=8, lora_alpha=32, lora_dropout=0.01,
               target_modules=["q_proj", "v_proj"], threshold=0.5, text_encoder=None,
               pick_place_model="pick_place_convdecoder", fusion_model="concat_transformer", depth=2, heads=2,
               mlp_ratio=4, dropout=0.0, dtype="fp32", **extra)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = Models.get_by_name(dict(cfg), device=torch.device("cpu"))
        ml = Models.get_by_name(dict(cfg, lora=True), device=torch.device("cpu"))      # the shipped lora_dropout = 0.01
        assert ml.siglip_model.lora["dropout"] == 0.01
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == ref
    assert all("siglip_model" in n for n, p in m.named_parameters() if not p.requires_grad)
    # lora=True: PEFT's key layout after LoraModel(model, config, "siglip_adapter") (models/siglip.py:33-40) --
    # base_model.model. prefix, adapted Linears under .base_layer, adapters under lora_{A,B}.<adapter>
    def peft_name(k):
        if not k.startswith("siglip_model."):
            return k
        k = "siglip_model.base_model.model." + k[len("siglip_model."):]
        for t in ("q_proj", "v_proj"):
            k = k.replace(f".self_attn.{t}.", f".self_attn.{t}.base_layer.")
        return k
    want = {peft_name(k): v for k, v in ref.items()}
    for tower in ("vision_model", "text_model"):
        for i in range(2):
            for t in ("q_proj", "v_proj"):
                base = f"siglip_model.base_model.model.{tower}.encoder.layers.{i}.self_attn.{t}."
                want[base + "lora_A.siglip_adapter.weight"] = (8, 64)
                want[base + "lora_B.siglip_adapter.weight"] = (64, 8)
    assert {k: tuple(v.shape) for k, v in ml.state_dict().items()} == want
    train = {n for n, p in ml.named_parameters() if p.requires_grad}
    assert {n for n in train if "siglip_model" in n} == {k for k in want if ".lora_" in k}
    assert all(float(p.abs().max()) == 0 for n, p in ml.named_parameters() if ".lora_B." in n)


def test_siglip_config_resolves():
    from bifold.config import compose
    cfg = compose(CONF, ["model=siglip_sequential", "dataset@train_dataset=synthetic_bimanual_sequential"])
    assert cfg.model.automodel_name == "google/siglip-base-patch16-384"
    assert cfg.model.context_length == 3 and cfg.model.is_bimanual is True and cfg.loss.is_bimanual is True


def test_select_arms_matches_reference_action(golden_dir):
    """The arm-selection rule of get_action (models/__init__.py:118-131) on the golden
    confidences: which arms the reference kept is visible as non-dummy picks."""
    from bifold.models import select_arms
    from oracle import bifold_oracle as O
    z = np.load(os.path.join(golden_dir, "postproc.npz"))
    mask = z["mask"]
    conf = np.stack([O.sample_from_heatmap(z[f"bi.{a}_pick_heatmap"], mask)[1] for a in ("left", "right")])
    keep = select_arms(conf, 0.01)
    ref_keep = np.stack([~(z[f"bi.expect.{a}_pick"] == -1).all(axis=1) for a in ("left", "right")])
    assert np.array_equal(keep, ref_keep)
    assert keep.any(axis=0).all()           # at least one arm always acts
    assert (~keep).any()


def test_bench_contract_helpers():
    """bench.py is the driver's measurement contract: it must import without a GPU, price the step with
    SURVEY 8d's FLOP count against the dense bf16 peak, and find the committed PMC summary it quotes."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.GF_PER_SAMPLE == pytest.approx(91.49) and bench.PEAK_BF16_TFLOPS == 2500.0
    traffic = bench.profiled_traffic("gemm_bf16_nt[p8]")
    assert traffic is not None and 1e8 < traffic[0] < 2e9 and "profiles/" in traffic[1]
    assert bench.profiled_traffic("gemm_bf16_nt[k32]") is None
    src = open(os.path.join(root, "bench.py")).read()
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"',
                '"vs_baseline"', '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"'):
        assert key in src, key


@pytest.mark.parametrize("fixture", ["tiny_siglip_lora.npz", "tiny_siglip_sequential_lora.npz"])
def test_lora_oracle_is_pinned_by_the_reference_at_dropout_0(golden_dir, fixture):
    """oracle/lora_oracle.py (adapters applied UNMERGED, gradients from torch autograd) against what the reference
    itself computed with the merged weights W' = W + (alpha / r) B A in its `lora=False` towers
    (oracle/gen_golden.py::siglip_lora_fixture): heatmaps, loss, every tail gradient, and the adapter gradients that
    the chain rule gives from the reference's dL/dW' -- dA = (alpha / r) B^T G, dB = (alpha / r) G A^T."""
    from oracle import bifold_oracle as O
    from oracle import lora_oracle as LO
    z = np.load(os.path.join(golden_dir, fixture))
    state = {k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state/")}
    sample = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sample/")}
    bim, T = bool(int(z["meta/is_bimanual"])), int(z["meta/context_length"])
    r, alpha = int(z["meta/r"]), float(z["meta/lora_alpha"])
    keys = list(LO.trainable_keys(state))
    for k in keys:
        state[k].requires_grad_(True)
    fwd = LO.siglip_sequential_forward if T else LO.siglip_forward
    out = fwd(state, sample, 2, 2, bim, r=r, lora_alpha=alpha)
    loss, _ = O.bce_gaussmap(out, sample, bim)
    grads = dict(zip(keys, torch.autograd.grad(loss, [state[k] for k in keys])))

    def close(a, b, tol=2e-5):
        a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
        return (a - b).abs().max().item() <= tol * b.abs().max().item() + 1e-9
    for k in z.files:
        if k.startswith("out/"):
            assert close(out[k[4:]].detach(), z[k]), k
        elif k.startswith("grad/"):
            assert close(grads[k[5:]], z[k], 1e-4), k
    assert abs(loss.item() - float(z["loss"])) < 1e-6 * abs(float(z["loss"]))
    n = 0
    for k in z.files:
        if not k.startswith("gradw/"):
            continue
        stem, g = LO.PEFT_PREFIX + k[6:].rsplit(".", 1)[0], torch.from_numpy(z[k])
        a, b = state[stem + ".lora_A.siglip_adapter.weight"].detach(), state[stem + ".lora_B.siglip_adapter.weight"].detach()
        assert close(grads[stem + ".lora_A.siglip_adapter.weight"], (alpha / r) * b.t() @ g, 1e-4), k
        assert close(grads[stem + ".lora_B.siglip_adapter.weight"], (alpha / r) * g @ a.t(), 1e-4), k
        n += 1
    assert n == 8


def test_the_build_checked_the_hand_counted_asynchronous_loads():
    """attn_bwd_fused32p_kernel (config A's attention backward) issues its global loads from inline assembly and waits for
    them by hand; the compiler may copy or spill a destination register before that wait.  The Makefile therefore runs
    tools/check_async_loads.py over the ISA of every build of attention.hip and fails the build on a violation; the verdict
    of the build that produced the library under test must be on file and clean (advisor finding, round 5)."""
    build = os.path.join(ROOT, "bifold_amd", "csrc", "build")
    report = os.path.join(build, "async_loads_check.txt")
    obj = os.path.join(build, "attention.o")
    if not os.path.isfile(obj):
        pytest.skip("no object directory beside the library (a prebuilt library travelled alone)")
    assert os.path.isfile(report), "attention.o was built without the asynchronous-load check"
    text = open(report).read()
    assert "violations: 0" in text and "attn_bwd_fused32p" in text, text[-500:]
    assert os.path.getmtime(report) >= os.path.getmtime(obj) - 1.0
