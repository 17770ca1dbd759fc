"""BASELINE.json's full configuration (64 x 3 x 512 x 512, 80 classes, K = 100) through size-independent properties:
the oracle cannot run the network at this size in test time, so the checks are (i) the CPU oracle's decode on the full
HIP heat map (exact indices / scores / classes for all 64 images), (ii) sortedness and range invariants, (iii) bit-exact
repeatability of the captured graph, (iv) batch-composition independence: images 0..3 give the same bits in a batch of
64 as in a batch of 4 (every kernel is per-pixel / per-image, no cross-image reduction, no atomics in inference)."""
import math

import pytest
import torch

from oracle import ctdet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["f16", "f16x3"])
def setup(request):
    """the whole network at BASELINE's size in the throughput mode and in the bench's headline mode (f16x3)"""
    import bench
    dev = torch.device("cuda:0")
    model, cfg = bench.build_model(request.param, dev, seed=3)
    model.eval()
    model.score_threshold = 0.0
    model.wh[2].bias.data.fill_(4.0)   # positive box sizes so that detections survive the empty-box filter
    images = bench.synthetic_images(64, 512, 0, dev)
    return model, images


def _engine(model, B):
    return next(e for e in model._engines.values() if e.B == B)


def test_fullsize_decode_matches_oracle_and_invariants(setup):
    model, images = setup
    with torch.no_grad():
        out = model.infer_batch_tensor(images)
    eng = _engine(model, 64)
    # kernels only also at this size (a torch reduction over a 4 M element map would bring a memset node: engine/graph_nodes.py)
    assert eng.graph_nodes.get("kernel", 0) > 0 and set(eng.graph_nodes) <= {"kernel", "empty"}, eng.graph_nodes
    hm, wh, reg = [t.float().cpu().permute(0, 3, 1, 2).contiguous() for t in eng.out]
    assert hm.shape == (64, 80, 128, 128)
    assert hm.min() >= 1e-4 and hm.max() <= 1 - 1e-4 and torch.isfinite(wh).all() and torch.isfinite(reg).all()
    boxes, scores, classes, inds = [t.cpu() for t in eng.dec]
    rb, rs, rc, ri = O.ctdet_decode(hm, wh, reg, down_ratio=4, K=100)
    assert torch.equal(scores, rs) and torch.equal(classes, rc) and torch.equal(inds.long(), ri)
    assert torch.allclose(boxes, rb, atol=1e-4, rtol=1e-6)
    assert (scores[:, :-1] >= scores[:, 1:]).all()                 # sorted, descending
    assert inds.min() >= 0 and inds.max() < 128 * 128 and classes.min() >= 0 and classes.max() < 80
    assert len(out) == 64
    for b in (0, 17, 63):
        inst = out[b]["instances"]
        bb, ss, cc = O.inference_single_image(rb[b], rs[b], rc[b], 100, 0.0)
        bb, keep = O.detector_postprocess(bb, (512, 512), 512, 512)
        assert torch.equal(inst.scores.cpu(), ss[keep]) and torch.equal(inst.pred_classes.cpu(), cc[keep])
        assert torch.allclose(inst.pred_boxes.tensor.cpu(), bb[keep], atol=1e-4, rtol=1e-6)
        assert len(inst) > 0


def test_fullsize_replay_is_bit_exact_and_batch_independent(setup):
    model, images = setup
    with torch.no_grad():
        model.infer_batch_tensor(images)
        e64 = _engine(model, 64)
        first = [t.clone() for t in e64.out] + [t.clone() for t in e64.dec]
        model.infer_batch_tensor(images)
        second = list(e64.out) + list(e64.dec)
        for i, (a, b) in enumerate(zip(first, second)):
            assert torch.equal(a, b), f"replay differs in output {i}: max abs diff {(a.float() - b.float()).abs().max().item()}"
        model.infer_batch_tensor(images[:4].contiguous())
        e4 = _engine(model, 4)
        for i, (a, b) in enumerate(zip(first[:3], e4.out)):      # hm, wh, reg of images 0..3
            assert torch.equal(a[:4], b), f"batch 64 vs 4 differs in map {i}: {(a[:4] - b).abs().max().item()}"
        for a, b in zip(first[3:], e4.dec):
            assert torch.equal(a[:4], b)


# Many-round grids: a workgroup that starts while the CU's memory pipeline is busy exposes any LDS-DMA wait that is too
# lenient (one such race in the halo kernel's last chunk only showed from the third round of workgroups per CU on and
# was invisible at the small test sizes).  Full BASELINE-sized layers against torch's own convolution on the GPU.
@pytest.mark.parametrize("case", [(64, 128, 128, 64, 64, False), (64, 128, 128, 64, 27, True), (64, 64, 64, 128, 128, False),
                                  (64, 32, 32, 256, 256, False), (32, 128, 128, 64, 768, False),
                                  (64, 128, 128, 32, 64, False), (64, 16, 16, 512, 512, False), (64, 16, 16, 512, 27, True)])
def test_fullsize_conv3x3_matches_torch(case):
    import torch.nn.functional as F
    from detectron2_centernet_amd import ops
    B, H, W, cin, cout, f32out = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, H, W, cin, generator=g).half().to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
    p = ops.PackedConv(w, None, None, stride=1, pad=1, compute=ops.F16)
    y = ops.conv2d(x, p, out_dtype=torch.float32 if f32out else torch.float16)
    y2 = ops.conv2d(x, p, out_dtype=torch.float32 if f32out else torch.float16)
    assert torch.equal(y, y2)                                   # deterministic
    bad = 0
    for b0 in range(0, B, 8):
        ref = F.conv2d(x[b0:b0 + 8].permute(0, 3, 1, 2).float(), w.half().float(), None, 1, 1).permute(0, 2, 3, 1)
        bad += ((y[b0:b0 + 8, ..., :cout].float() - ref).abs() > 1e-2).sum().item()
    assert bad == 0, f"{bad} elements off by more than 1e-2"


def test_fullsize_dcn_window_matches_oracle_on_sampled_images():
    """the LDS-window DCNv2 kernel on a full 64 x 128 x 128 x 64 layer: deterministic, and images 0 / 40 / 63 agree with
    the CPU oracle (the many-round regime is where an earlier gather-from-global kernel showed a race)"""
    from detectron2_centernet_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 128, 128, 64, generator=g).half()
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24)
    om = torch.randn(64, 128, 128, 28, generator=g)
    om[..., :18] *= 1.5
    pw = ops.PackedConv(w.to(dev), None, None, stride=1, pad=1, compute=ops.F16)
    xd, omd = x.to(dev), om.to(dev)
    y = ops.dcnv2(xd, omd, pw)
    assert torch.equal(y, ops.dcnv2(xd, omd, pw))
    y = y.float().cpu()
    for b in (0, 40, 63):
        xb = x[b:b + 1].float().permute(0, 3, 1, 2)
        omb = om[b:b + 1].permute(0, 3, 1, 2)
        ref = O.dcnv2_forward(xb, omb[:, :18], torch.sigmoid(omb[:, 18:27]), w.half().float(), None, 1, 1, 1)
        err = (y[b] - ref.permute(0, 2, 3, 1)[0]).abs().max().item()
        assert err <= 6e-3 * max(1.0, ref.abs().max().item()), (b, err)


# ---- training-side kernels at the size of the batch-16 training step (many rounds of workgroups per CU) ----
def _close(got, ref, tol, what):
    err = (got - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= tol * scale, f"{what}: max err {err} (scale {scale})"


@pytest.mark.parametrize("case", [(16, 128, 128, 64, 64, 3, 1, 1), (16, 256, 256, 32, 64, 3, 2, 1), (16, 64, 64, 128, 128, 3, 1, 1),
                                  (16, 128, 128, 64, 256, 1, 1, 0),
                                  (4, 512, 512, 8, 16, 7, 1, 3), (4, 512, 512, 16, 16, 3, 1, 1)])
def test_fullsize_wgrad_dgrad_match_torch(case):
    import torch.nn.functional as F
    from detectron2_centernet_amd import ops_train as ot
    B, H, W, Cin, Cout, k, s, p = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g).half().float().to(dev)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).half().float().to(dev)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = (torch.randn(B, Cout, Ho, Wo, generator=g) * 0.05).half().float().to(dev)
    ref_dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride=s, padding=p)
    ref_dx = torch.nn.grad.conv2d_input(x.shape, w, dy, stride=s, padding=p)
    xh, dyh = x.permute(0, 2, 3, 1).contiguous().half(), dy.permute(0, 2, 3, 1).contiguous().half()
    dw = ot.conv_wgrad(xh, dyh, Cout, k, k, s, p, scale=1.0).view(Cout, k, k, Cin).permute(0, 3, 1, 2)
    _close(dw, ref_dw, 3e-3, "dW")
    dx = ot.conv_dgrad(dyh, w, s, p, (H, W))
    _close(dx[..., :Cin].float().permute(0, 3, 1, 2), ref_dx, 3e-3, "dX")


def test_fullsize_bn_train_fwd_bwd_match_torch():
    import torch.nn.functional as F
    from detectron2_centernet_amd import ops_train as ot
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    B, C, H, W = 16, 64, 128, 128
    y = (torch.randn(B, C, H, W, generator=g) * 2 + 0.5).half().float().to(dev).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev).requires_grad_(True)
    beta = torch.randn(C, generator=g).to(dev).requires_grad_(True)
    ref = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5).relu()
    dz = torch.randn(ref.shape, generator=g).half().float().to(dev)
    ref.backward(dz)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    yh = y.detach().permute(0, 2, 3, 1).contiguous().half()
    z, mean, invstd, scale = ot.bn_train_fwd(yh, gamma.detach(), beta.detach(), rm, rv, 1e-5, 0.1, relu=True)
    _close(z.float().permute(0, 3, 1, 2), ref.detach(), 2e-3, "bn fwd")
    dy, _, dgamma, dbeta = ot.bn_train_bwd(dz.permute(0, 2, 3, 1).contiguous().half(), z, yh, mean, invstd, scale, relu=True,
                                           grad_mult=1.0)
    _close(dy.float().permute(0, 3, 1, 2), y.grad, 6e-3, "bn dy")
    _close(dgamma, gamma.grad, 3e-3, "dgamma")
    _close(dbeta, beta.grad, 3e-3, "dbeta")


def test_fullsize_dcn_backward_scatter_window_vs_atomics():
    """the LDS-window scatter kernel against the f32-atomics kernel on a full 16 x 128 x 128 x 64 layer"""
    import os
    from detectron2_centernet_amd import ops_train as ot
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    B, H, W, Cin = 16, 128, 128, 64
    x = torch.randn(B, H, W, Cin, generator=g).half().to(dev)
    dcol = (torch.randn(B, H, W, 9 * Cin, generator=g) * 0.1).half().to(dev)
    om = torch.randn(B, H, W, 28, generator=g)
    om[..., :18] *= 2.0
    om = om.to(dev)
    dx_w, dom_w = ot.dcn_col2im_coord(dcol, x, om)
    from detectron2_centernet_amd import _lib
    with _lib.tuning(_lib.TUNE_NO_COL2IM_WINDOW):
        dx_a, dom_a = ot.dcn_col2im_coord(dcol, x, om)
    assert (dx_w - dx_a).abs().max().item() <= dx_a.abs().max().item() * 2.0 ** -14
    assert (dom_w - dom_a).abs().max().item() <= dom_a.abs().max().item() * 2e-4


def test_resnet50_800x800_bs8_properties(dev):
    """BASELINE.json configs[4] at full size: ResNet-50 CenterNet, 8 x 3 x 800 x 800 per GPU.  Eval: finite outputs, a
    bit-exact graph replay, the decode of the HIP heat map equal to the oracle's decode, batch-composition independence
    (image 0 alone == image 0 in the batch of 8).  Train: one whole step at this size is finite, moves the trainable
    parameters and leaves the frozen stem / res2 untouched."""
    import bench
    from detectron2_centernet_amd.engine.bench_train import synthetic_batch
    from detectron2_centernet_amd.engine.train_loop import SimpleTrainer

    model, cfg = bench.build_model("f16", dev, seed=2, config="r50")
    model.eval()
    model.score_threshold = 0.0
    images = bench.synthetic_images(8, 800, 0, dev)
    out = model.infer_batch_tensor(images)
    eng = _engine(model, 8)
    hm, wh, reg = [t.float().cpu().permute(0, 3, 1, 2) for t in eng.out]
    assert hm.shape == (8, 80, 200, 200) and torch.isfinite(hm).all() and torch.isfinite(wh).all()
    first = [t.clone() for t in eng.out]
    model.infer_batch_tensor(images)
    for a, b in zip(first, eng.out):
        assert torch.equal(a, b)
    boxes, scores, classes, inds = [t.cpu() for t in eng.dec]
    rb, rs, rc, ri = O.ctdet_decode(hm[:2], wh[:2], reg[:2], down_ratio=4, K=100)
    assert torch.equal(scores[:2], rs) and torch.equal(classes[:2], rc) and torch.equal(inds[:2].long(), ri)
    assert (scores[:, :-1] >= scores[:, 1:]).all() and len(out) == 8
    model.infer_batch_tensor(images[:1].contiguous())
    solo = _engine(model, 1)
    assert torch.equal(solo.out[0][0], first[0][0])
    # one full training step
    model._engines = {}
    cfg.SOLVER.IMS_PER_BATCH = 8
    tr = SimpleTrainer(model, None, cfg)
    p0 = tr.optimizer.flat_param.clone()
    stem0 = model.backbone.stem.conv1.weight.detach().clone()
    losses = tr.run_step_tensors(*synthetic_batch(8, 800, 0, dev))
    vals = {k: float(v) for k, v in losses.items()}
    assert all(v == v and abs(v) < 1e9 for v in vals.values()), vals
    assert torch.isfinite(tr.optimizer.flat_param).all() and (tr.optimizer.flat_param - p0).abs().max() > 0
    assert torch.equal(model.backbone.stem.conv1.weight.detach(), stem0)


# ---- f16x3 (f32 tensors, split f16 products) at the BASELINE sizes: many rounds of workgroups per CU ----
@pytest.mark.parametrize("case", [(64, 128, 128, 64, 64), (64, 64, 64, 128, 128), (64, 32, 32, 256, 256), (64, 128, 128, 64, 768),
                                  (64, 128, 128, 64, 27), (64, 16, 16, 512, 512)])
def test_fullsize_conv3x3_f16x3_matches_torch_f32(case):
    """the tap-pair halo kernel (and, for the 16x16 map, the uniform-K kernel) in f16x3 mode against torch's fp32 conv:
    deterministic, every element within 2e-5 of the output scale"""
    import torch.nn.functional as F
    from detectron2_centernet_amd import ops
    B, H, W, cin, cout = case
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(B, H, W, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    p = ops.PackedConv(w, None, bias, stride=1, pad=1, compute=ops.F16X3)
    y = ops.conv2d(x, p, act=ops.ACT_RELU)
    assert torch.equal(y, ops.conv2d(x, p, act=ops.ACT_RELU))   # deterministic
    worst = 0.0
    for b0 in range(0, B, 8):
        ref = F.conv2d(x[b0:b0 + 8].permute(0, 3, 1, 2), w, bias, 1, 1).relu().permute(0, 2, 3, 1)
        worst = max(worst, (y[b0:b0 + 8, ..., :cout] - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    assert worst <= 2e-5, worst


@pytest.mark.parametrize("shape", [(128, 64, 64), (64, 128, 128)])
def test_fullsize_dcn_f16x3_matches_oracle_on_sampled_images(shape):
    """f16x3 DCNv2 (LDS-window kernel, split operands) on full layers of the batch-64 step -- 64 x 128 x 128 x 64 -> 64 (the
    <2,64> tile: four 32-pixel waves) and 64 x 64 x 64 x 128 -> 128 (the <1,128> tile: eight 16-pixel waves) --:
    deterministic, images 0 / 40 / 63 against the CPU oracle within 2e-5"""
    from detectron2_centernet_amd import ops
    dev = torch.device("cuda:0")
    C, H, W = shape
    g = torch.Generator().manual_seed(2)
    x = torch.randn(64, H, W, C, generator=g)
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5)
    om = torch.randn(64, H, W, 28, generator=g)
    om[..., :18] *= 1.5
    pw = ops.PackedConv(w.to(dev), None, None, stride=1, pad=1, compute=ops.F16X3)
    xd, omd = x.to(dev), om.to(dev)
    y = ops.dcnv2(xd, omd, pw)
    assert torch.equal(y, ops.dcnv2(xd, omd, pw))
    y = y.cpu()
    for b in (0, 40, 63):
        xb, omb = x[b:b + 1].permute(0, 3, 1, 2), om[b:b + 1].permute(0, 3, 1, 2)
        ref = O.dcnv2_forward(xb, omb[:, :18], torch.sigmoid(omb[:, 18:27]), w, None, 1, 1, 1)
        err = (y[b] - ref.permute(0, 2, 3, 1)[0]).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (b, err)


def test_fullsize_fused_dcn_offset_matches_oracle_on_sampled_images():
    """the dominant f16 inference kernel -- DCNv2 with its offset / mask conv computed in the same kernel
    (`ops.dcnv2_offset`, dcn_window_rows_kernel<fused>) -- on a full 64 x 128 x 128 x 64 layer: deterministic, and images
    0 / 21 / 40 / 63 against the CPU oracle (offset conv + sampling + contraction on the f16-rounded operands).  The
    round-1 halo race only showed from the third round of workgroups per CU on: this is that regime."""
    import torch.nn.functional as F
    from detectron2_centernet_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 128, 128, 64, generator=g).half()
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    w_off = torch.randn(27, 64, 3, 3, generator=g) * (0.5 / 24)
    b_off = torch.randn(27, generator=g) * 0.5
    p = ops.PackedConv(w.to(dev), None, None, stride=1, pad=1, compute=ops.F16, cout_align=64)
    p_off = ops.PackedConv(w_off.to(dev), None, b_off.to(dev), stride=1, pad=1, compute=ops.F16)
    xd = x.to(dev)
    assert ops.dcnv2_offset_supported(xd, p_off, p)
    y = ops.dcnv2_offset(xd, p_off, p, act=ops.ACT_RELU)
    assert torch.equal(y, ops.dcnv2_offset(xd, p_off, p, act=ops.ACT_RELU))
    y = y.float().cpu()
    for b in (0, 21, 40, 63):
        xb = x[b:b + 1].float().permute(0, 3, 1, 2)
        om = F.conv2d(xb, w_off.half().float(), b_off, 1, 1)
        ref = O.dcnv2_forward(xb, om[:, :18], torch.sigmoid(om[:, 18:27]), w.half().float(), None, 1, 1, 1).relu()
        err = (y[b] - ref.permute(0, 2, 3, 1)[0]).abs().max().item()
        assert err <= 8e-3 * max(1.0, ref.abs().max().item()), (b, err)


@pytest.mark.parametrize("precision", ["f16", "f16x3"])
def test_fullsize_dla34_training_step_16x512(dev, precision):
    """BASELINE configs[2] as a whole under -m gpu: DLA-34 CenterNet, 16 x 3 x 512 x 512, targets + forward + losses +
    backward + SGD, in the throughput mode and in the parity-grade f16x3 mode.  Finite losses, the step replays as a captured
    HIP graph, and the replayed trajectory is the eager one (at this size the step is reproducible to ~1e-7: the f32 atomics'
    order only moves the last bits)."""
    import bench
    from detectron2_centernet_amd.engine.bench_train import synthetic_batch
    from detectron2_centernet_amd.engine.train_loop import SimpleTrainer

    res = {}
    for mode in ("eager", "graph"):
        model, cfg = bench.build_model(precision, dev, seed=3)
        cfg.SOLVER.IMS_PER_BATCH = 16
        tr = SimpleTrainer(model, None, cfg)
        tr.use_hip_graph = mode == "graph"
        p0 = tr.optimizer.flat_param.clone()
        batch = synthetic_batch(16, 512, 0, dev)
        hist = [sum(float(v) for v in tr.run_step_tensors(*batch).values()) for _ in range(5)]
        assert all(math.isfinite(h) for h in hist), hist
        assert tr.graph_state == ("captured" if mode == "graph" else "eager"), tr.graph_state
        for g in (g for g in tr._graphs.values() if g["graph"] is not None):     # kernels only at full size too
            assert g["nodes"].get("kernel", 0) > 0 and set(g["nodes"]) <= {"kernel", "empty"}, g["nodes"]
        res[mode] = (hist, (tr.optimizer.flat_param - p0), tr.optimizer.flat_mom.clone())
        del tr, model
        torch.cuda.empty_cache()
    (he, de, me), (hg, dg, mg) = res["eager"], res["graph"]
    assert he[-1] != he[0]                                           # the trajectory moves
    for a, b in zip(he, hg):
        assert abs(a - b) <= 1e-5 * abs(a), (he, hg)
    # f16 activations snap the f32 atomics' order noise back onto the f16 grid (runs are all but bit-identical); with f32
    # activations it survives and the random-init network amplifies it from step to step: 4e-3 of the largest update after
    # five steps, at losses still equal to 1e-5
    tol = 1e-4 if precision == "f16" else 2e-2
    assert de.abs().max() > 0 and (de - dg).abs().max().item() <= tol * de.abs().max().item()
    assert (me - mg).abs().max().item() <= tol * me.abs().max().item()
