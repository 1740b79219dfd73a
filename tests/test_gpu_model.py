"""GPU: builder.MODEL end to end (BASELINE config 1 shapes: ResNet-18, 64x64 crops, queue 1024)
against the CPU oracle, eager vs hipGraph step equivalence, and the DenseCL path."""
import copy
import os

import numpy as np
import pytest
import torch

from cp2_amd import builder, synthetic
from cp2_amd.config import Config
from cp2_amd.engine import TrainStep
from cp2_amd.main import make_optimizer
from cp2_amd.pretrain_types import PretrainType
from oracle import cp2_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def small_model(K=1024, pretrain_type=PretrainType.CP2, cfg_name="config_pretrain_r18.py", **kw):
    torch.manual_seed(0)
    cfg = Config.fromfile(os.path.join(ROOT, "configs", cfg_name))
    extra = dict(lmbd_cp2_dense_loss=0.5, dense_logits_temp=0.2) if pretrain_type == PretrainType.DENSECL else {}
    m = builder.MODEL(cfg, rank=0, K=K, pretrain_from_scratch=True, pretrain_type=pretrain_type, device=DEV, **extra, **kw)
    return m.to(DEV).train()


def test_forward_cp2_matches_oracle_on_same_features():
    model = small_model()
    assert model.output_stride == 16
    batch = synthetic.make_batch(8, 64, 64, DEV, seed=3)
    feats = {}
    hq = model.encoder_q.register_forward_hook(lambda m, i, o: feats.__setitem__("q", o))
    hk = model.encoder_k.register_forward_hook(lambda m, i, o: feats.__setitem__("k", (i[0], o)))
    queue0, ptr0 = model.queue.clone(), int(model.queue_ptr)
    pk0 = [p.detach().clone() for p in model.encoder_k.parameters()]
    pq0 = [p.detach().clone() for p in model.encoder_q.parameters()]
    perm = torch.randperm(8)
    loss = model(visualize=False, step=0, new_epoch=False, idx_shuffle=perm.to(DEV), **batch)
    loss.backward()
    hq.remove(), hk.remove()
    # EMA happened before the key forward, bit-exact
    want_k = O.momentum_update([p.cpu() for p in pk0], [p.cpu() for p in pq0], 0.999)
    for p, w in zip(model.encoder_k.parameters(), want_k):
        assert torch.equal(p.detach().cpu(), w)
    # the key encoder saw the composed + shuffled batch (bit-exact composition and row gather)
    want_img_b, _ = O.compose_mask(batch["img_b"].cpu(), batch["bg1"].cpu())
    assert torch.equal(feats["k"][0].cpu(), O.shuffle_take(want_img_b, perm, 0, 1))
    # loss section from the very same encoder outputs
    q_feat = feats["q"].detach().cpu().requires_grad_(True)
    k_feat = O.unshuffle_take(feats["k"][1].detach().cpu(), perm, 0, 1)
    r = O.cp2_loss_section(q_feat, k_feat, batch["bg0"].cpu(), batch["bg1"].cpu(), batch["pixel_ids_a"].cpu(),
                           batch["pixel_ids_b"].cpu(), batch["region_ids_a"].cpu(), batch["region_ids_b"].cpu(),
                           queue0.cpu(), output_stride=16)
    assert abs(float(loss) - float(r["loss"])) <= 2e-5
    # queue update with k_pos, pointer advanced on the device
    qa, ptr = O.dequeue_and_enqueue(queue0.cpu(), ptr0, r["k_pos"].detach())
    assert int(model.queue_ptr) == ptr
    assert (model.queue.cpu() - qa).abs().max() <= 2e-6
    ious, ious_m = model.epoch_ious()
    assert np.array_equal(np.float32(ious), r["iou"].numpy()) and np.array_equal(np.float32(ious_m), r["iou_masked"].numpy())
    logs = model.flush_logs()
    assert abs(logs[0][1]["train/loss_ins_step"] - float(r["loss_instance"])) <= 2e-5
    assert abs(logs[0][1]["train/loss_dense_step"] - float(r["loss_dense"])) <= 2e-5


def test_state_dict_contract():
    model = small_model()
    keys = set(model.state_dict().keys())
    for k in ("queue", "queue_ptr", "queue2", "queue2_ptr", "encoder_q.backbone.conv1.weight", "encoder_q.backbone.bn1.weight",
              "encoder_k.backbone.layer4.1.conv2.weight", "encoder_q.decode_head.contrast_conv.0.weight",
              "encoder_q.decode_head.contrast_conv.2.bias", "encoder_q.decode_head.conv_seg.weight"):
        assert k in keys, k
    assert model.queue.shape == (128, 1024) and model.queue_ptr.dtype == torch.long
    model._momentum_update_key_encoder()                 # flattening must not change the state dict
    sd = model.state_dict()
    assert set(sd.keys()) == keys
    for n, p in model.named_parameters():
        assert torch.equal(sd[n], p.detach())


def test_densecl_forward_backward_runs_and_advances_both_queues():
    """Smoke level: finite loss and gradients, both pointers advance, logged scalars are self-consistent.  The parity
    check of the same path is test_forward_densecl_matches_oracle_on_same_features below."""
    model = small_model(K=512, pretrain_type=PretrainType.DENSECL, cfg_name="config_moco.py")
    batch = synthetic.make_batch(4, 64, 64, DEV, seed=5)
    q2_before = model.queue2.clone()
    loss = model(visualize=False, step=0, new_epoch=False, **batch)
    loss.backward()
    assert torch.isfinite(loss)
    g = [p.grad for p in model.encoder_q.parameters() if p.grad is not None]
    assert len(g) > 100 and all(torch.isfinite(x).all() for x in g)
    assert int(model.queue_ptr) == 4 and int(model.queue2_ptr) == 4
    assert not torch.equal(model.queue2, q2_before)
    logs = model.flush_logs()[0][1]
    assert abs(0.5 * logs["train/loss_ins_step"] + 0.5 * logs["train/loss_dense_step"] - logs["train/loss_step"]) < 1e-5


def test_densecl_neck_under_autocast_stays_close_to_the_fp32_neck():
    """Round 4: the DenseCL neck (builder.py:179-274) runs in the encoders' autocast precision (`MODEL.neck_autocast`), its
    two 1x1 layers through encoder.Conv2d (bf16 weight image, cp2_wgrad1x1).  Same weights, same batch: the loss stays within
    2 % of the fp32-neck step's, every neck parameter the flags select receives a finite gradient of the fp32-neck step's size."""
    out = {}
    for neck_amp in (True, False):
        model = small_model(K=512, pretrain_type=PretrainType.DENSECL, cfg_name="config_moco.py", amp_dtype=torch.bfloat16,
                            channels_last=True)
        model.neck_autocast = neck_amp
        model.encoder_q.to(memory_format=torch.channels_last)
        model.encoder_k.to(memory_format=torch.channels_last)
        model.flatten_parameters()
        model.enable_query_shadow()                     # what optim.FlatSGD does: the query convolutions read a bf16 weight image
        batch = synthetic.make_batch(4, 64, 64, DEV, seed=5)
        loss = model(visualize=False, step=0, new_epoch=False, idx_shuffle=torch.arange(4, device=DEV), **batch)
        loss.backward()
        neck = model.encoder_q.neck
        out[neck_amp] = (float(loss), [p.grad.clone() for p in list(neck.local_projector.parameters()) + list(neck.global_projector.parameters())])
        if neck_amp:
            for nk in (neck, model.encoder_k.neck):
                assert nk.local_projector[0].shadow_weight is not None and nk.local_projector[0].shadow_weight.dtype == torch.bfloat16
    assert abs(out[True][0] - out[False][0]) <= 0.02 * abs(out[False][0]), (out[True][0], out[False][0])
    for ga, gf in zip(out[True][1], out[False][1]):
        assert torch.isfinite(ga).all()
        assert (ga - gf).norm() <= 0.1 * gf.norm() + 1e-6, (float((ga - gf).norm()), float(gf.norm()))


@pytest.mark.parametrize("case", [
    dict(ptype=PretrainType.DENSECL, flags={}, step=0),
    dict(ptype=PretrainType.PROPOSED_V2, flags=dict(use_symmetrical_loss=True, lmbd_coordinate=0.3), step=0),
    dict(ptype=PretrainType.PROPOSED_V2, flags=dict(use_symmetrical_loss=True, use_predictor=True), step=1),
    dict(ptype=PretrainType.PROPOSED_V2, flags=dict(use_avgpool_global=True, use_predictor=True), step=0),
], ids=["densecl", "v2-symmetric-coordinate-even", "v2-symmetric-predictor-odd", "v2-avgpool-predictor"])
def test_forward_densecl_matches_oracle_on_same_features(case):
    """Model-level parity of forward_densecl (reference builder.py:667-999): the backbone and neck outputs of both
    encoders are captured by hooks and handed to the oracle's DenseCL losses -- which tensors the flags select
    (:700-716), the un-shuffle of the key features (:739-748), pixel ids at the BACKBONE stride (:913-922), global loss
    vs `queue` and local loss vs `queue2` (:760-772, :808-910), the symmetric second pass and its `.mean()` of a scalar
    (:944-966), which pass feeds the queues at even / odd steps (:968-975), the pooled-local keys going to `queue2`
    (:982-983), and the gradient that reaches the neck outputs.  Tolerances: loss 2e-5, gradients 2e-5 of the largest
    entry, queues 2e-6 (the enqueue is a copy of unit vectors normalised on the GPU)."""
    flags, step = case["flags"], case["step"]
    model = small_model(K=512, pretrain_type=case["ptype"], cfg_name="config_moco.py",
                        **(dict(lmbd_cp2_dense_loss=0.5, dense_logits_temp=0.2) if case["ptype"] != PretrainType.DENSECL else {}), **flags)
    b, hw = 4, 128
    batch = synthetic.make_batch(b, hw, hw, DEV, seed=11)
    bs = model.backbone_output_stride
    cap = {"qb": [], "qn": [], "kb": [], "kn": []}

    def keep_grad(key):
        def hook(mod, inp, out):
            if isinstance(out, dict):
                for v in out.values():
                    if v.requires_grad:
                        v.retain_grad()
            cap[key].append(out)
        return hook
    hooks = [model.encoder_q.backbone.register_forward_hook(keep_grad("qb")), model.encoder_q.neck.register_forward_hook(keep_grad("qn")),
             model.encoder_k.backbone.register_forward_hook(keep_grad("kb")), model.encoder_k.neck.register_forward_hook(keep_grad("kn"))]
    queue0, queue20 = model.queue.clone().cpu(), model.queue2.clone().cpu()
    perm = torch.randperm(b)
    loss = model(visualize=False, step=step, new_epoch=False, idx_shuffle=perm.to(DEV), **batch)
    loss.backward()
    for h in hooks:
        h.remove()
    sym, pred, avg = (bool(flags.get(k)) for k in ("use_symmetrical_loss", "use_predictor", "use_avgpool_global"))
    lam = float(flags.get("lmbd_coordinate", 0.0))
    n_pass = 2 if sym else 1
    assert all(len(v) == n_pass for v in cap.values())
    F = torch.nn.functional
    ids = [O.strided_gather(batch["pixel_ids_a"].cpu(), bs), O.strided_gather(batch["pixel_ids_b"].cpu(), bs)]
    want_g = want_l = 0.0
    leaves, keys = [], []
    for p in range(n_pass):
        qn = {k: v.detach().cpu().requires_grad_(True) for k, v in cap["qn"][p].items()}
        sfx = "pred" if pred else "proj"
        q_local, q_global = qn["x_local_" + sfx], (qn["x_avgpool_local_" + sfx] if avg else qn["x_global_" + sfx])
        eq = F.normalize(cap["qb"][p][3].detach().float().cpu().flatten(2), dim=1)
        kn = {k: O.unshuffle_take(v.detach().float().cpu(), perm, 0, 1) for k, v in cap["kn"][p].items()}
        ek = F.normalize(O.unshuffle_take(cap["kb"][p][3].detach().float().cpu(), perm, 0, 1).flatten(2), dim=1)
        k_pool = F.normalize(kn["x_avgpool_local_proj"], dim=1)
        k_global = k_pool if avg else F.normalize(kn["x_global_proj"], dim=1)
        k_local = F.normalize(kn["x_local_proj"].flatten(2), dim=1)
        want_g = want_g + O.densecl_global_loss(F.normalize(q_global, dim=1), k_global, queue0, 0.2)
        want_l = want_l + O.densecl_local_loss(eq, ek, F.normalize(q_local.flatten(2), dim=1), k_local, ids[p], ids[1 - p],
                                               queue20, 0.2, lam)[0]
        leaves.append((cap["qn"][p], qn, "x_local_" + sfx, ("x_avgpool_local_" if avg else "x_global_") + sfx))
        keys.append((k_global, k_pool))
    want = 0.5 * want_g + 0.5 * want_l
    want.backward()
    assert abs(float(loss) - float(want)) <= 2e-5, (float(loss), float(want))
    logs = model.flush_logs()[0][1]
    assert abs(logs["train/loss_ins_step"] - float(want_g)) <= 2e-5 and abs(logs["train/loss_dense_step"] - float(want_l)) <= 2e-5
    for got, ref, name_l, name_g in leaves:
        for name in {name_l, name_g}:
            leaf_got, leaf_ref = got[name], ref[name]
            g, w = leaf_got.grad.detach().cpu(), leaf_ref.grad
            if avg and name.startswith("x_local"):
                # on the GPU x_avgpool_local_* is computed FROM x_local_* inside the neck, so its gradient flows on into
                # x_local_*; the oracle's leaves are independent: d/d(mean over pixels) spread back by hand
                pooled = ref["x_avgpool_local_" + name.split("_")[-1]].grad
                w = w + pooled[:, :, None, None] / (w.shape[2] * w.shape[3])
            assert (g - w).abs().max() <= 2e-5 * w.abs().max() + 1e-9, name
    upd = keys[1] if (sym and step % 2 == 0) else keys[0]            # reference builder.py:968-975
    q_want, ptr = O.dequeue_and_enqueue(queue0, 0, upd[0])
    q2_want, ptr2 = O.dequeue_and_enqueue(queue20, 0, upd[1])
    assert int(model.queue_ptr) == ptr == b and int(model.queue2_ptr) == ptr2 == b
    assert (model.queue.cpu() - q_want).abs().max() <= 2e-6 and (model.queue2.cpu() - q2_want).abs().max() <= 2e-6
    if sym:                                                          # the other pass's keys are NOT what was written
        other = keys[0] if upd is keys[1] else keys[1]
        assert (model.queue.cpu()[:, :b] - other[0].t()).abs().max() > 1e-3


@pytest.mark.parametrize("flags", [dict(use_symmetrical_loss=True), dict(use_predictor=True),
                                   dict(use_avgpool_global=True, use_symmetrical_loss=True, lmbd_coordinate=0.3)])
def test_proposed_v2_variants_run_and_follow_the_reference_update_rule(flags):
    """PROPOSED_V2 flags of reference builder.py:687-716, 944-972 end to end: finite loss, gradients for exactly the
    parameters the chosen heads use, and at an even step the symmetric pass enqueues the SECOND pass's keys."""
    model = small_model(K=512, pretrain_type=PretrainType.PROPOSED_V2, cfg_name="config_moco.py",
                        lmbd_cp2_dense_loss=0.5, dense_logits_temp=0.2, **flags)
    batch = synthetic.make_batch(4, 64, 64, DEV, seed=9)
    keys = {}
    enq = model._dequeue_and_enqueue
    model._dequeue_and_enqueue = lambda k: (keys.__setitem__("global", k.clone()), enq(k))[1]
    loss = model(visualize=False, step=0, new_epoch=False, **batch)
    loss.backward()
    assert torch.isfinite(loss) and int(model.queue_ptr) == 4 and int(model.queue2_ptr) == 4
    assert torch.equal(model.queue[:, :4].t(), keys["global"])
    pred = [p.grad is not None for p in model.encoder_q.neck.global_predictor.parameters()]
    assert all(pred) == bool(flags.get("use_predictor", False)) and any(pred) == all(pred)
    assert all(p.grad is None for p in model.encoder_q.decode_head.parameters())
    logs = model.flush_logs()[0][1]
    assert abs(0.5 * logs["train/loss_ins_step"] + 0.5 * logs["train/loss_dense_step"] - logs["train/loss_step"]) < 1e-5


def test_flatten_preserves_channels_last_and_values():
    model = small_model()
    model.encoder_q.to(memory_format=torch.channels_last)
    model.encoder_k.to(memory_format=torch.channels_last)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    strides = {n: p.stride() for n, p in model.named_parameters()}
    model.flatten_parameters()
    for n, p in model.named_parameters():
        assert p.stride() == strides[n], n
        assert torch.equal(p.detach(), before[n]), n
    w = model.encoder_q.backbone.layer2[0].conv2.weight
    assert w.is_contiguous(memory_format=torch.channels_last) and not w.is_contiguous()
    # EMA over the flat buffers == per-parameter EMA
    for p in model.encoder_q.parameters():
        p.data.add_(0.01)
    want = O.momentum_update([p.detach().cpu() for p in model.encoder_k.parameters()],
                             [p.detach().cpu() for p in model.encoder_q.parameters()], 0.999)
    model._momentum_update_key_encoder()
    for p, w_ in zip(model.encoder_k.parameters(), want):
        assert torch.equal(p.detach().cpu(), w_)


def _bf16_ulps(a: torch.Tensor, b: torch.Tensor) -> int:
    """Largest distance between two bf16 tensors in units of the last place (sign-magnitude -> ordered integers)."""
    def order(t):
        i = t.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(i >= 0x8000, 0x8000 - i, i)
    return int((order(a) - order(b)).abs().max())


def _leaf_walk(enc, x, use_image: bool):
    """(final output, [(leaf name, module, output)]) of an eval-mode forward under bf16 autocast, with or without the bf16
    weight image of the convolutions."""
    from cp2_amd.encoder import Conv2d
    outs, hooks, saved = [], [], {}
    for name, m in enc.named_modules():
        if len(list(m.children())) == 0:
            hooks.append(m.register_forward_hook(lambda mod, inp, out, name=name: outs.append((name, mod, out.detach().clone()))
                                                 if isinstance(out, torch.Tensor) else None))
    if not use_image:
        for m in enc.modules():
            if isinstance(m, Conv2d):
                saved[m], m.shadow_weight = m.shadow_weight, None
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            y = enc(x)
    finally:
        for m, w in saved.items():
            m.shadow_weight = w
        for h in hooks:
            h.remove()
    return y, outs


def test_key_weight_shadow_is_exact_and_used():
    """The EMA's bf16 image of the key weights equals casting the fp32 weights (bit for bit), and the key encoder computes
    the same thing with and without it.  What "the same" means was measured layer by layer (tools/shadow_layer_walk.py,
    DESIGN.md section 4): the two forwards hand MIOpen identical bf16 operands and are served by the same solvers; every leaf
    module's output is identical, bit for bit, UNLESS the solver is one of MIOpen's global split-K kernels
    (igemm_fwd_gtcx35_nhwc_bf16_..._gkgs: zero-fill, fp32 atomic accumulation, cast back), which is not reproducible from
    one call to the next on the SAME operands -- 1 bf16 ulp in ~0.01 % of a layer's outputs, with the image or without it
    alike.  Which layers get such a solver depends on what MIOpen's find has measured earlier in the process (a fresh
    process in immediate mode: none; after cudnn.benchmark runs: layer3 / layer4's 3x3 convolutions at 4x4 pixels).
    So: with deterministic solvers requested (cudnn.deterministic) and two identical calls agreeing, every leaf must agree
    exactly with and without the image; in any mode the first leaf to differ must be a convolution output at most 2 bf16
    ulps apart, and the encoder output stays within one bf16 rounding step of its largest value."""
    from cp2_amd.encoder import Conv2d
    model = small_model(amp_dtype=torch.bfloat16, channels_last=True)
    model.encoder_q.to(memory_format=torch.channels_last)
    model.encoder_k.to(memory_format=torch.channels_last)
    for p in model.encoder_q.parameters():
        p.data.add_(0.003 * torch.randn_like(p))
    model._momentum_update_key_encoder()
    assert model._flat_k_bf16 is not None
    assert torch.equal(model._flat_k_bf16, model._flat_k.to(torch.bfloat16))
    conv = model.encoder_k.backbone.layer2[0].conv2
    assert conv.shadow_weight is not None and torch.equal(conv.shadow_weight, conv.weight.to(torch.bfloat16))
    assert conv.shadow_weight.stride() == conv.weight.stride()
    x = torch.rand(4, 3, 64, 64, device=DEV).contiguous(memory_format=torch.channels_last)
    model.encoder_k.eval()

    def first_diff(u, v):
        for i, ((name, mod, a), (_, _, b)) in enumerate(zip(u, v)):
            if not torch.equal(a, b):
                return i, name, mod, a, b
        return None

    keep = (torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic, Conv2d.gemm_1x1)
    try:
        for deterministic in (True, False):
            torch.backends.cudnn.deterministic = deterministic
            torch.backends.cudnn.benchmark = not deterministic
            # with the image, the wide 1x1 stride-1 layers of a gradient-free forward run as hipBLASLt GEMMs (another kernel,
            # another summation order than MIOpen's); the exact comparison is between the SAME MIOpen problems
            Conv2d.gemm_1x1 = not deterministic
            _leaf_walk(model.encoder_k, x, True)                         # the first call of a configuration may run MIOpen's find
            y_img, leaves_img = _leaf_walk(model.encoder_k, x, True)
            y_again, leaves_again = _leaf_walk(model.encoder_k, x, True)
            y_cast, leaves_cast = _leaf_walk(model.encoder_k, x, False)
            assert len(leaves_img) == len(leaves_cast) > 40
            noise, diff = first_diff(leaves_img, leaves_again), first_diff(leaves_img, leaves_cast)
            if deterministic and noise is None:
                assert diff is None, f"deterministic solvers: {diff[1]} differs with / without the weight image"
                assert torch.equal(y_img, y_cast)
            if diff is not None:
                i, name, mod, a, b = diff
                assert isinstance(mod, Conv2d) and a.dtype == torch.bfloat16, f"first difference at {name}: not a convolution output"
                assert _bf16_ulps(a, b) <= 2, f"{name}: {_bf16_ulps(a, b)} bf16 ulps apart"
            assert (y_img.float() - y_cast.float()).abs().max().item() <= 2.0 ** -7 * y_cast.float().abs().max().item()
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic, Conv2d.gemm_1x1 = keep


def test_key_forward_graph_equals_eager_key_forward():
    """The key encoder forward replayed from a hipGraph (engine.ForwardGraph: 3 eager calls, capture, replays) gives
    the same features and leaves the same BN running statistics / batch counters as the eager forward.
    Tolerance: MIOpen's bf16 forward solvers may use split-K atomics (not run-to-run reproducible), so features are
    compared to bf16 precision (2e-2 of the largest value) and running statistics to 1e-2; a replay that read a stale
    input or stale weights would be off by O(1): every call uses a different image."""
    models = []
    for use_graph in (True, False):
        m = small_model(amp_dtype=torch.bfloat16, channels_last=True)
        m.encoder_q.to(memory_format=torch.channels_last)
        m.encoder_k.to(memory_format=torch.channels_last)
        m.key_forward_graph = use_graph
        m.flatten_parameters()
        m._momentum_update_key_encoder()                     # fills the bf16 key-weight shadow
        models.append(m)
    for i in range(7):
        img = torch.randn(8, 3, 64, 64, device=DEV, generator=torch.Generator(DEV).manual_seed(i))
        with torch.no_grad():
            ka = models[0]._encode_key(img).clone()
            kb = models[1]._encode_key(img).clone()
        assert (ka.float() - kb.float()).abs().max().item() <= 2e-2 * kb.float().abs().max().item(), i
    g = models[0]._key_graph
    assert g is not None and any(e["graph"] is not None for e in g.entries.values())
    assert models[1]._key_graph is None
    sa, sb = models[0].state_dict(), models[1].state_dict()
    for name in sa:
        if name.startswith("encoder_k."):
            a, b = sa[name].float(), sb[name].float()
            assert (a - b).abs().max().item() <= 1e-2 * b.abs().max().item() + 1e-6, name
    assert int(sa["encoder_k.backbone.bn1.num_batches_tracked"]) == 7
