"""Training-side op wrappers (C ABI) and the torch.autograd.Function glue of the CenterNet training path.

Autograd is used as the tape only: every forward and backward body below is one or a few HIP kernel launches.
Three modes.  f16 NHWC activations / activation gradients with f32 parameters and parameter gradients (the throughput mode:
mixed precision with f32 accumulation everywhere) is chosen by f16 activations entering.  f32 activations enter the other
two, told apart by `F32_COMPUTE` (set per forward pass from the model's MODEL.CENTERNET.HIP_PRECISION, remembered by every
autograd node for its backward): F32 -- the reference's own arithmetic, contractions as f32 FMA chains on the f32 matrix
pipe / plain f32 kernels: the mode in which the whole step is compared with the fp32 oracle -- or F16X3 (round 4): the same
f32 tensors, statistics and gradients, every contraction of the step (forward, input gradient, weight gradient, DCNv2's
column GEMMs) as hi*hi + lo*hi + hi*lo on the f16 matrix pipe with f32 accumulation: f32-grade results at several times
the f32 rate.  A static loss scale keeps small heatmap gradients inside the f16 range (f16 tensors in the first mode, the hi /
lo halves of the split in the third): it is applied in the loss backward and removed again on every parameter gradient, so
`param.grad` and the returned loss values are unscaled.
"""
import ctypes as C

import os

import torch

from . import _lib, ops
from ._lib import ConvDesc
from .ops import ACT_NONE, ACT_RELU, F16, F16X3, F32, _nhwc_stride, _ptr, _require_cuda, _stream, dt_of

GRAD_SCALE = 1024.0
# f16x3: DCNv2's d(columns) GEMM inside the scatter kernel (ctdet_dcn_col2im_fused).  Built, parity-green and OFF by default:
# measured 7.4 ms per bs-16 step against 4.9 (scatter) + 2.1 (the sixteen 1x1 GEMM launches) for the two-step path -- the eight
# waves of a workgroup each pull the same W fragments through L1 (no LDS left to share them: 157 KB are the scatter's), and
# nothing overlaps the GEMM phase at one workgroup per CU.  CTDET_FUSED_DCOL=1 turns it on.
FUSE_DCOL = os.environ.get("CTDET_FUSED_DCOL", "0") == "1"
KEEP_COLS = os.environ.get("CTDET_NO_KEEP_COLS", "0") != "1"   # f16x3: DCNv2's forward kernel also writes the sampled columns
F32_COMPUTE = F32     # how contractions over f32 tensors run: F32 or F16X3 (engine/train_step.py sets it per forward pass)


def comp_of(x):
    """compute mode of the node that takes activation x"""
    return F16 if x.dtype == torch.float16 else F32_COMPUTE
# parameter gradients are multiplied by PARAM_GRAD_MULT = 1 / (GRAD_SCALE * world_size): removes the loss scale and
# pre-divides by the data-parallel world size so a SUM all-reduce yields the mean (what DDP does)
PARAM_GRAD_MULT = 1.0 / GRAD_SCALE


def set_world_size(world):
    global PARAM_GRAD_MULT
    PARAM_GRAD_MULT = 1.0 / (GRAD_SCALE * max(1, int(world)))


class ZeroArena:
    """One f32 buffer that is cleared with a single fill at the start of a training step and handed out in slices to the
    kernels that accumulate with atomics (the weight gradients): ~80 fills per step become one.  A slice lives until the
    next `begin_step()`; autograd has added it into the parameter's .grad long before."""

    def __init__(self, numel, device):
        self.buf = torch.zeros(int(numel), dtype=torch.float32, device=device)
        self.used = self.high = 0

    def begin_step(self):
        if self.high:
            self.buf[:self.high].zero_()
        self.used = 0

    def take(self, numel):
        n = (int(numel) + 63) // 64 * 64          # 256-byte granules
        if self.used + n > self.buf.numel():
            return None
        t = self.buf[self.used:self.used + numel]
        self.used += n
        self.high = max(self.high, self.used)
        return t


ARENA = None   # set by the trainer (engine/train_loop.py); None: every accumulator is its own torch.zeros


def _zeros_f32(shape, device):
    n = 1
    for d in shape:
        n *= int(d)
    if ARENA is not None and ARENA.buf.device == device:
        t = ARENA.take(n)
        if t is not None:
            return t.view(*shape)
    return torch.zeros(*shape, dtype=torch.float32, device=device)


def _ws(Cc, device):
    return torch.empty(_lib.lib().ctdet_chan_workspace_bytes(Cc) // 4, dtype=torch.float32, device=device)


# ------------------------------------------------------------------------------------------ raw wrappers
def bn_train_fwd(y, gamma, beta, running_mean, running_var, eps, momentum, res=None, relu=True):
    B, H, W, Cc = y.shape
    dev = y.device
    z = torch.empty(B, H, W, Cc, dtype=y.dtype, device=dev)
    mean, invstd, scale, shift = (torch.empty(Cc, dtype=torch.float32, device=dev) for _ in range(4))
    # algorithmic bytes: y read twice (statistics, apply), z written, the residual read once
    with ops.prof_region("bn_train_fwd", flops=0.0, nbytes=float(B * H * W * Cc * y.element_size() * (3 if res is None else 4))):
        rc = _lib.lib().ctdet_bn_train_fwd(_ptr(y), _nhwc_stride(y), _ptr(res), _nhwc_stride(res) if res is not None else 0,
                                           _ptr(z), _nhwc_stride(z), B * H * W, Cc, _ptr(gamma), _ptr(beta), float(eps),
                                           float(momentum), _ptr(running_mean), _ptr(running_var), _ptr(mean), _ptr(invstd),
                                           _ptr(scale), _ptr(shift), _ptr(_ws(Cc, dev)), int(relu), dt_of(y), _stream())
    _lib.check(rc, "ctdet_bn_train_fwd")
    return z, mean, invstd, scale


def bn_train_bwd(dz, z, y, mean, invstd, scale, relu=True, want_dres=False, grad_mult=None, into=None):
    """dgamma / dbeta come back multiplied by grad_mult (default: PARAM_GRAD_MULT, what every parameter gradient carries).
    into=(dgamma_slot, dbeta_slot): write them (Cc floats each; either may be None) straight into the parameters' gradient
    slices -- each BatchNorm / bias parameter has one producer per step, so a plain store is the accumulation"""
    B, H, W, Cc = dz.shape
    dev = dz.device
    dy = torch.empty(B, H, W, Cc, dtype=dz.dtype, device=dev)
    dres = torch.empty(B, H, W, Cc, dtype=dz.dtype, device=dev) if want_dres else None
    slots = [t if t is not None and t.numel() == Cc and t.is_contiguous() else None for t in (into or (None, None))]
    dgamma, dbeta = slots
    if dgamma is None or dbeta is None:
        dgb = torch.empty(2, Cc, dtype=torch.float32, device=dev)   # written, not accumulated
        dgamma, dbeta = (dgb[0] if dgamma is None else dgamma), (dgb[1] if dbeta is None else dbeta)
    gm = PARAM_GRAD_MULT if grad_mult is None else grad_mult
    # algorithmic bytes: dz, z, y read twice (sums, apply) where present, dy (and dres) written
    nt = 2 * (1 + (z is not None) + (y is not None)) + 1 + int(want_dres)
    with ops.prof_region("bn_train_bwd", flops=0.0, nbytes=float(B * H * W * Cc * dz.element_size() * nt)):
        rc = _lib.lib().ctdet_bn_train_bwd(_ptr(dz), _nhwc_stride(dz), _ptr(z), _nhwc_stride(z) if z is not None else 0,
                                           _ptr(y), _nhwc_stride(y) if y is not None else 0, _ptr(mean), _ptr(invstd),
                                           _ptr(scale), B * H * W, Cc, int(relu), _ptr(dy), _nhwc_stride(dy), _ptr(dres),
                                           _nhwc_stride(dres) if dres is not None else 0, _ptr(dgamma), _ptr(dbeta),
                                           float(gm), _ptr(_ws(Cc, dev)), dt_of(dz), _stream())
    _lib.check(rc, "ctdet_bn_train_bwd")
    return dy, dres, dgamma, dbeta


def conv_wgrad(x, dy, Cout, R, S, stride, pad, dil=1, scale=None, into=None, comp=None):
    """scale * dW, f32 [Cout, R*S*Cin] (tap-major) for y = conv(x, W); x, dy NHWC, both f16 or both f32 (comp F32: plain f32
    FMAs; F16X3: split products on the f16 matrix pipe).  scale defaults to PARAM_GRAD_MULT.
    into=(grad, taps, cin_k): accumulate into `grad`, the parameter's own OIHW gradient [Cout_real, Cin_real, kh, kw] (its
    slice of the optimizer's flat buffer), with k = tap*cin_k + c -- nothing is returned"""
    B, H, W, Cin = x.shape
    _, Ho, Wo, Cd = dy.shape
    assert Cd >= Cout
    d = ConvDesc()
    d.B, d.H, d.W, d.Cin, d.in_stride = B, H, W, Cin, _nhwc_stride(x)
    d.Cout, d.Ho, d.Wo, d.out_stride = Cout, Ho, Wo, _nhwc_stride(dy)
    d.R, d.S, d.stride, d.pad, d.dil = R, S, stride, pad, dil
    d.compute_dtype = dt_of(x) if comp is None or x.dtype == torch.float16 else comp
    assert dy.dtype == x.dtype
    sc = float(PARAM_GRAD_MULT if scale is None else scale)
    with ops.prof_region(f"conv_wgrad<{R}x{S}>", flops=2.0 * B * Ho * Wo * Cout * R * S * Cin, nbytes=0.0,
                         info=f"M={B * Ho * Wo} {Cin}->{Cout} k{R} s{stride}"):
        if into is not None:
            grad, taps, cin_k = into
            assert grad.dtype == torch.float32 and grad.is_contiguous() and grad.dim() == 4
            assert grad.shape[2] * grad.shape[3] == taps and grad.shape[0] <= Cout and grad.shape[1] <= cin_k
            rc = _lib.lib().ctdet_conv_wgrad_oihw(C.byref(d), _ptr(x), _ptr(dy), _ptr(grad), sc, taps, cin_k, grad.shape[1],
                                                  grad.shape[0], _stream())
            dw = None
        else:
            dw = _zeros_f32((Cout, R * S * Cin), x.device)
            rc = _lib.lib().ctdet_conv_wgrad(C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), sc, _stream())
    _lib.check(rc, "ctdet_conv_wgrad")
    return dw


def grad_slot(p):
    """the parameter's slice of the optimizer's flat gradient buffer when backward may accumulate into it directly (set up by
    solver.FlatSGD), else None.  Writing there replaces `return dw` -> autograd's AccumulateGrad (`p.grad += dw`, one
    elementwise kernel per parameter and step)."""
    g = getattr(p, "_ctdet_flat_grad", None)
    if g is None or p.grad is None or p.grad.data_ptr() != g.data_ptr():
        return None
    return g


def grad_done(p):
    """what AccumulateGrad's post hook would have announced (engine/reducer.py counts parameters per bucket)"""
    h = getattr(p, "_ctdet_grad_hook", None)
    if h is not None:
        h(p)


def grad_into_slot(p, g):
    """a finished (already scaled) gradient g of parameter p added straight to p's slice of the flat gradient buffer instead of
    being returned to autograd; True if that was possible.  Besides saving AccumulateGrad's kernel this keeps AccumulateGrad
    nodes out of the backward pass altogether: such a node runs on the stream that was current when IT was created, and a
    node that an older autograd graph still keeps alive (a loss tensor the caller holds on to) drags the default stream into
    a HIP-graph capture of the step -- hipStreamEndCapture then crashes (round 3's VoVNet capture: engine/train_loop.py)."""
    slot = grad_slot(p)
    if slot is None or slot.numel() != g.numel():
        return False
    slot.view(-1).add_(g.reshape(-1))
    grad_done(p)
    return True


PENDING = []   # (slot, tap-major dw, taps, cin_k): finished weight gradients not yet added to their OIHW slots


class SideLane:
    """A second HIP stream for the weight-gradient branch of backward.  dX of a layer feeds the next layer's backward, dW feeds
    nothing until the optimizer: the weight-gradient kernels (one workgroup per CU, atomics-bound) run here next to the
    dX / BatchNorm chain instead of in front of it.  Captured into the training step's HIP graph as a parallel branch.
    Operands are kept referenced until `join()`: a tensor freed on the main stream could be handed out again there while
    this stream still reads it."""

    def __init__(self):
        self.enabled = os.environ.get("CTDET_TRAIN_SIDE_STREAM", "0") == "1"
        self.stream, self.keep, self.active = None, [], False

    def run(self, fn, *operands):
        if not self.enabled:
            return fn()
        main = torch.cuda.current_stream()
        if self.stream is None or self.stream.device != main.device:
            self.stream = torch.cuda.Stream(device=main.device)
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream):
            out = fn()
        self.keep.append(operands)
        self.active = True
        return out

    def join(self):
        if self.active:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.active = False
        self.keep.clear()


SIDE = SideLane()
_END_QUEUED = [False]


def _end_of_backward():
    _END_QUEUED[0] = False
    SIDE.join()
    flush_param_grads()


def _queue_end_of_backward():
    """once per backward pass: join the side stream and flush the deferred weight gradients when the pass ends"""
    if not _END_QUEUED[0]:
        _END_QUEUED[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)


def flush_param_grads(ptr_lo=None, ptr_hi=None):
    """adds the pending tap-major weight gradients (all, or those whose slot starts inside [ptr_lo, ptr_hi)) to the parameters'
    gradients, 24 tensors per launch.  Runs by itself when a backward pass ends (queued on the autograd engine by the first
    deferred gradient), and from the reducer before a bucket is exchanged."""
    todo = [e for e in PENDING if ptr_lo is None or ptr_lo <= e[0].data_ptr() < ptr_hi]
    if not todo:
        return
    SIDE.join()
    PENDING[:] = [e for e in PENDING if not (ptr_lo is None or ptr_lo <= e[0].data_ptr() < ptr_hi)]
    n = len(todo)
    vp, i32 = C.c_void_p * n, C.c_int32 * n
    rc = _lib.lib().ctdet_grad_scatter_oihw(vp(*[e[1].data_ptr() for e in todo]), vp(*[e[0].data_ptr() for e in todo]),
                                            i32(*[e[0].shape[0] for e in todo]), i32(*[e[0].shape[1] for e in todo]),
                                            i32(*[e[3] for e in todo]), i32(*[e[2] for e in todo]), n, _stream())
    _lib.check(rc, "ctdet_grad_scatter_oihw")


def wgrad_to_param(p, x, dy, Cout_k, R, S, stride, pad, taps, cin_k, scale=None, make_x=None, keep=(), comp=None):
    """The weight gradient of parameter p (OIHW [Cout, Cin, kh, kw], kh*kw = taps; k = tap*cin_k + c in the kernel's order)
    accumulated into p's slice of the optimizer's flat gradient buffer instead of being handed to autograd (whose
    AccumulateGrad would launch one strided add per parameter): tap-major partial sums in the step's zeroed arena as always
    (coalesced atomics), added to the slot by flush_param_grads -- one launch per bucket.  (Letting the 1x1 kernels accumulate
    in the slot itself, `conv_wgrad(into=...)`, measured 0.37 ms per step SLOWER than arena + scatter.)  False: p has no slot
    (stand-alone use) -- the caller returns the gradient to autograd.  make_x: the kernel's input is produced on the side stream
    too (DCNv2: the sampled columns); keep: its operands."""
    slot = grad_slot(p)
    if slot is None or scale is not None:
        return False
    _queue_end_of_backward()

    def work():
        xin = make_x() if make_x is not None else x
        return conv_wgrad(xin, dy, Cout_k, R, S, stride, pad, comp=comp), xin
    dw, xin = SIDE.run(work, x, dy, *keep)
    if make_x is not None and SIDE.enabled:    # (with the side stream off nothing outlives the call: round 4 found the eager
        SIDE.keep.append((xin,))               # step holding every DCNv2 layer's columns here, 5 GB per step)
    PENDING.append((slot, dw, taps, cin_k))
    grad_done(p)
    return True


def maxpool2x2_bwd(x, dz):
    B, H, W, Cc = x.shape
    dx = torch.empty(B, H, W, Cc, dtype=x.dtype, device=x.device)
    rc = _lib.lib().ctdet_maxpool2x2_bwd(_ptr(x), _nhwc_stride(x), _ptr(dz), _nhwc_stride(dz), _ptr(dx), _nhwc_stride(dx),
                                         B, H, W, Cc, dt_of(x), _stream())
    _lib.check(rc, "ctdet_maxpool2x2_bwd")
    return dx


def dwconvT_bwd(x, dz, weight, f, wk=None, raw=False):
    """wk: the [2f, 2f, C] f32 form of `weight` if the caller already has it (the forward of the same step made it);
    raw: return dw as the kernel accumulates it, [2f, 2f, C], instead of the parameter's [C, 1, 2f, 2f] view"""
    B, H, W, Cc = x.shape
    w = wk if wk is not None else weight.detach().reshape(Cc, 2 * f, 2 * f).to(torch.float32).permute(1, 2, 0).contiguous()
    dx = torch.empty(B, H, W, Cc, dtype=x.dtype, device=x.device)
    dw = _zeros_f32((2 * f, 2 * f, Cc), x.device)
    rc = _lib.lib().ctdet_dwconvT_bwd(_ptr(x), _nhwc_stride(x), _ptr(dz), _nhwc_stride(dz), _ptr(w), _ptr(dx),
                                      _nhwc_stride(dx), _ptr(dw), B, H, W, Cc, f, dt_of(x), _stream())
    _lib.check(rc, "ctdet_dwconvT_bwd")
    return dx, (dw if raw else dw.permute(2, 0, 1).reshape(Cc, 1, 2 * f, 2 * f))


def dcn_cols(x, om, mask_is_prob=False):
    B, H, W, Cin = x.shape
    col = torch.empty(B, H, W, 9 * Cin, dtype=x.dtype, device=x.device)
    with ops.prof_region("dcn_cols", flops=0.0, nbytes=float(B * H * W * Cin * x.element_size() * 10 + B * H * W * 27 * 4)):
        rc = _lib.lib().ctdet_dcn_cols(_ptr(x), _nhwc_stride(x), _ptr(om), _nhwc_stride(om), _ptr(col), B, H, W, Cin, int(mask_is_prob), dt_of(x), _stream())
    _lib.check(rc, "ctdet_dcn_cols")
    return col


def dcn_weight_matrix(weight, Cw=None, chunked=False):
    """[Cw, 9*Cin, 1, 1] matrix W with d(columns) = dY . W: column index tap*Cin + c, or chunked (c/32)*288 + tap*32 + c%32 --
    the layout dcn_col2im_coord(dcol_chunked=True) reads in contiguous runs"""
    Cout, Cin = weight.shape[:2]
    w = weight.detach()
    if chunked:
        wmat = w.reshape(Cout, Cin // 32, 32, 9).permute(0, 1, 3, 2).reshape(Cout, 9 * Cin, 1, 1)
    else:
        wmat = w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin, 1, 1)
    if Cw is not None and Cw != Cout:
        wmat = ops.kpad(wmat, (0, 0, 0, 0, 0, 0, 0, Cw - Cout))
    return wmat


def dcn_dcol(dyp, weight, chunked, comp=None):
    """d(columns) [B, H, W, 9*Cin] = dY . W of a DCNv2 layer (deform_conv_cuda.cu:1003-1009), rows tap-major or chunk-major.
    f16 / f16x3: the operand is packed straight from the [Cout, Cin, 3, 3] parameter (a planned pack: no permuted copy per step);
    f32: the permuted weight matrix through the generic input-gradient path."""
    Cout, Cin = weight.shape[:2]
    comp = comp_of(dyp) if comp is None else comp
    if comp == F32:
        return conv_dgrad(dyp, dcn_weight_matrix(weight, dyp.shape[3], chunked), 1, 0, dyp.shape[1:3], comp=F32)
    p = ops.PackedConv(weight.detach(), None, None, stride=1, pad=0, compute=comp, cin_pad=dyp.shape[3],
                       transposed="dcn_cols_chunked" if chunked else "dcn_cols")
    dcol = torch.empty(dyp.shape[0], dyp.shape[1], dyp.shape[2], p.Cout_eff, dtype=dyp.dtype, device=dyp.device)
    ops.conv2d(dyp, p, out=dcol)
    return dcol


def dcn_col2im_coord(dcol, x, om, mask_is_prob=False, dom_channels=None, dcol_chunked=False, comp=None):
    """dx (f32, atomically accumulated) and dom = d(offsets, mask logits).  dom_channels=None: f32, the shape of om;
    dom_channels=C (f16 data): an f16 [B, H, W, C] tensor whose channels 27.. are zero -- directly the dY of the offset conv's
    backward, without a cast or a channel pad in between"""
    B, H, W, Cin = x.shape
    dx = torch.zeros(B, H, W, Cin, dtype=torch.float32, device=x.device)
    if dom_channels is None or x.dtype == torch.float32:
        dom = torch.empty(B, H, W, om.shape[3] if dom_channels is None else dom_channels, dtype=torch.float32, device=x.device)
    else:
        dom = torch.empty(B, H, W, dom_channels, dtype=torch.float16, device=x.device)
    es = x.element_size()
    with ops.prof_region("dcn_col2im", flops=0.0, nbytes=float(B * H * W * Cin * (9 * es + es + 4) + B * H * W * 27 * 8)):
        rc = _lib.lib().ctdet_dcn_col2im_coord(_ptr(dcol), _ptr(x), _nhwc_stride(x), _ptr(om), _nhwc_stride(om), _ptr(dx),
                                               _ptr(dom), dom.shape[3], dt_of(dom), B, H, W, Cin, int(mask_is_prob),
                                               int(dcol_chunked), dt_of(x) if comp is None or x.dtype == torch.float16 else comp,
                                               _stream())
    _lib.check(rc, "ctdet_dcn_col2im_coord")
    return dx, dom


def dcn_col2im_fused(dyp, weight, x, om, mask_is_prob=False, dom_channels=None):
    """the f16x3 mode's DCNv2 backward through the sampler with the d(columns) GEMM inside the scatter kernel
    (ctdet_dcn_col2im_fused): (dx f32, dom f32) or None when the shape does not qualify (the caller then materialises
    d(columns)).  dyp f32 [B, H, W, K]: K % 32 == 0, channels beyond the layer's couts zero."""
    B, H, W, Cin = x.shape
    Cout, K = weight.shape[0], dyp.shape[3]
    if (x.dtype != torch.float32 or dyp.dtype != torch.float32 or K % 32 or K < Cout or Cin % 32 or H % 8 or W % 16
            or weight.dtype != torch.float32 or not weight.is_contiguous()):
        return None
    wpk, wscale = ops.pack_x3(weight.detach(), (Cout, Cin, 1, 1, K, 9 * Cin, K, 5, 3, 9 * Cin))
    dx = torch.zeros(B, H, W, Cin, dtype=torch.float32, device=x.device)
    dom = torch.empty(B, H, W, om.shape[3] if dom_channels is None else dom_channels, dtype=torch.float32, device=x.device)
    with ops.prof_region("dcn_col2im_fused", flops=2.0 * B * H * W * K * 9 * Cin, nbytes=float(B * H * W * (Cin * 8 + K * 4 + 27 * 8))):
        rc = _lib.lib().ctdet_dcn_col2im_fused(_ptr(dyp), _nhwc_stride(dyp), K, _ptr(wpk), _ptr(wscale), _ptr(x), _nhwc_stride(x),
                                               _ptr(om), _nhwc_stride(om), _ptr(dx), _ptr(dom), dom.shape[3], B, H, W, Cin,
                                               int(mask_is_prob), _stream())
    if rc == 1:
        return None
    _lib.check(rc, "ctdet_dcn_col2im_fused")
    return dx, dom


# ------------------------------------------------------------------------------------------ helpers
def _fwd_pack(weight, stride, pad, cin_pad=None, bias=None, compute=F16):
    return ops.PackedConv(weight, None, bias, stride=stride, pad=pad, compute=compute, cin_pad=cin_pad)


_PHASE_TAPS = {}


def _phase_tap_index(device):
    """index into the 9 taps (+ a zero tap at 9) of the [phase (row parity, col parity)][a][b] 2x2 kernels: a row of
    parity 0 is fed by kernel row 1 of dY row i only, a row of parity 1 by kernel row 2 of dY row i (a = 0) and kernel row 0
    of dY row i + 1 (a = 1); columns alike"""
    t = _PHASE_TAPS.get(device)
    if t is None:
        rmap = {(0, 0): -1, (0, 1): 1, (1, 0): 2, (1, 1): 0}
        idx = []
        for ph in range(2):
            for pw in range(2):
                for a in range(2):
                    for b in range(2):
                        r, s_ = rmap[(ph, a)], rmap[(pw, b)]
                        idx.append(9 if r < 0 or s_ < 0 else r * 3 + s_)
        t = _PHASE_TAPS[device] = torch.tensor(idx, dtype=torch.int64, device=device)
    return t


def _conv_dgrad_s2_phases(dy, weight, in_hw, comp=F16):
    """input gradient of a 3x3 / stride 2 / pad 1 conv without zero-stuffing: one 2x2 conv over dy to the four output phases
    (4*Cin channels), then ctdet_depth_to_space2"""
    Cout, Cin, _, _ = weight.shape
    B, Ho, Wo, Cd = dy.shape
    H, W = in_hw
    Cp = (Cin + 7) // 8 * 8
    w9 = weight.detach().reshape(Cout, Cin, 9)
    if Cd != Cout:
        w9 = ops.kpad(w9, (0, 0, 0, 0, 0, Cd - Cout))
    w10 = ops.kpad(w9, (0, 1, 0, Cp - Cin))                         # zero tap, channel padding -> [Cd, Cp, 10]
    wd = w10.index_select(2, _phase_tap_index(dy.device)).view(Cd, Cp, 4, 2, 2)     # [co, ci, phase, a, b]
    wd = wd.permute(2, 1, 0, 3, 4).reshape(4 * Cp, Cd, 2, 2).contiguous()
    p = ops.PackedConv(wd, None, None, stride=1, pad=1, compute=comp)
    ph = ops.conv2d(dy, p)                                                           # [B, Ho+1, Wo+1, 4*Cp]
    dx = torch.empty(B, H, W, Cp, dtype=dy.dtype, device=dy.device)
    rc = _lib.lib().ctdet_depth_to_space2(_ptr(ph), _nhwc_stride(ph), _ptr(dx), _nhwc_stride(dx), B, H, W, Cp, ph.shape[1],
                                          ph.shape[2], dt_of(dx), _stream())
    _lib.check(rc, "ctdet_depth_to_space2")
    return dx if Cp == Cin else dx[..., :Cin]


def conv_dgrad(dy, weight, stride, pad, in_hw, cin_pad=None, comp=None):
    """dx of y = conv(x, weight): a conv over dy with the taps flipped and in/out channels swapped;
    cin_pad (f16 / f16x3): dy carries that many channels (>= Cout, the extra ones zero) -- the operand gets zero columns for them;
    f16 / f16x3: 3x3 / stride 2 / pad 1 goes through the four-phase form, other strides read dy as zero-stuffed (in_dil);
    f32: always the zero-stuffed form on the f32 MFMA kernel."""
    Cout, Cin, R, S = weight.shape
    comp = comp_of(dy) if comp is None else comp
    if comp == F32:
        wt = ops.flip_taps(weight.detach()).permute(1, 0, 2, 3).contiguous()      # [Cin, Cout, R, S]: dX = conv(dY, wt)
        p = ops.PackedConv(wt, None, None, stride=1, pad=R - 1 - pad, compute=F32)
        p.in_dil = stride
        dx = torch.empty(dy.shape[0], in_hw[0], in_hw[1], p.Cout_eff, dtype=torch.float32, device=dy.device)
        ops.conv2d(dy, p, out=dx)
        return dx if p.Cout_eff == Cin else dx[..., :Cin]
    if stride == 2 and R == 3 and S == 3 and pad == 1 and dy.shape[3] % (8 if comp == F16 else 16) == 0:
        return _conv_dgrad_s2_phases(dy, weight, in_hw, comp)     # pads the operand to dy's channel count itself
    wsrc = weight.detach()
    if comp == F16X3 and not (wsrc.dtype == torch.float32 and wsrc.is_contiguous()):
        wsrc = wsrc.float().contiguous()
    p = ops.PackedConv(wsrc, None, None, stride=1, pad=R - 1 - pad, compute=comp, tap_major=stride > 1,
                       transposed=True, cin_pad=cin_pad)
    p.in_dil = stride
    B = dy.shape[0]
    dx = torch.empty(B, in_hw[0], in_hw[1], p.Cout_eff, dtype=dy.dtype, device=dy.device)
    ops.conv2d(dy, p, out=dx)
    return dx if p.Cout_eff == Cin else dx[..., :Cin]


def _wgrad_to_oihw(dw, Cout, Cin_real, Cin_used, R, S):
    return dw.view(Cout, R, S, Cin_used)[..., :Cin_real].permute(0, 3, 1, 2)   # already scaled by the kernel; a view


# ------------------------------------------------------------------------------------------ autograd Functions
class ConvFn(torch.autograd.Function):
    """y = conv(x, weight) (+bias)(+relu) without BatchNorm; x NHWC, f16 or f32 (the mode)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, relu, out_f32, param_grad_mult=None):
        """param_grad_mult: multiplier on the parameter gradients; None = PARAM_GRAD_MULT (the training step's loss-scale /
        world-size protocol), 1.0 for stand-alone use behind the reference's module interface"""
        ctx.pgm = param_grad_mult
        f32 = x.dtype == torch.float32
        comp = ctx.comp = comp_of(x)
        p = _fwd_pack(weight, stride, pad, cin_pad=x.shape[3] if x.shape[3] != weight.shape[1] else None, bias=bias,
                      compute=comp)
        y = ops.conv2d(x, p, act=ACT_RELU if relu else ACT_NONE,
                       out_dtype=torch.float32 if (out_f32 or f32) else torch.float16)
        ctx.cfg = (stride, pad, relu, weight.shape[0], bias is not None)
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, weight, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        stride, pad, relu, Cout, has_bias = ctx.cfg
        Cout, Cin, R, S = weight.shape
        comp = ctx.comp
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dy = _pad_c(dy.contiguous(), comp)    # channel count -> multiple of 8 / 4 (padded channels carry zeros)
        Cw = dy.shape[3]
        direct = ctx.pgm is None                  # the training step's protocol: gradients go straight into the flat buffer
        wparam, bparam = ctx.params
        dbias = None
        if relu or has_bias:
            sb = grad_slot(bparam) if (direct and has_bias) else None
            dy, _, _, db = bn_train_bwd(dy, _pad_c(y, comp) if relu else None, None, None, None, None, relu=relu, grad_mult=ctx.pgm,
                                        into=(None, sb) if sb is not None and sb.numel() == Cw else None)
            dbias = db[:Cout] if has_bias else None
            if sb is not None and sb.numel() == Cw:
                grad_done(bparam)
                dbias = None
            elif direct and has_bias and grad_into_slot(bparam, dbias):
                dbias = None
        if direct and wgrad_to_param(wparam, x, dy, Cw, R, S, stride, pad, R * S, x.shape[3], comp=comp):
            dwt = None
        else:
            dw = conv_wgrad(x, dy, Cw, R, S, stride, pad, scale=ctx.pgm, comp=comp)[:Cout]
            dwt = _wgrad_to_oihw(dw, Cout, Cin, x.shape[3], R, S)
        dx = None
        if ctx.needs_input_grad[0]:
            if x.shape[3] != Cin or comp == F32:
                wpad = weight if Cw == Cout else ops.kpad(weight.detach(), (0, 0, 0, 0, 0, 0, 0, Cw - Cout))
                if x.shape[3] != Cin:         # input channels were padded (the 3 -> 8 channel image): so is dX
                    wpad = ops.kpad(wpad.detach(), (0, 0, 0, 0, 0, x.shape[3] - Cin))
                dx = conv_dgrad(dy, wpad, stride, pad, x.shape[1:3], comp=comp)
            else:                             # padded dY channels meet zero operand columns: no padded copy of the weight
                dx = conv_dgrad(dy, weight, stride, pad, x.shape[1:3], cin_pad=Cw, comp=comp)
        return dx, dwt, dbias, None, None, None, None, None


class FrozenConvFn(torch.autograd.Function):
    """z = act(conv(x, w) * scale + bias + res): a conv followed by FrozenBatchNorm2d -- an affine whose scale / bias are
    buffers (detectron2/layers/batch_norm.py:13-99) -- with the optional residual add and ReLU of the ResNet blocks
    (modeling/backbone/resnet.py:100-112, 196-214), one kernel launch forward.  Backward: g = dz * (z > 0); d res = g;
    d conv = g * scale; the weight gradient only if the conv is not frozen (FREEZE_AT)."""

    @staticmethod
    def forward(ctx, x, weight, scale, bias, res, stride, pad, relu):
        comp = ctx.comp = comp_of(x)
        p = ops.PackedConv(weight.detach(), scale, bias, stride=stride, pad=pad, compute=comp,
                           cin_pad=x.shape[3] if x.shape[3] != weight.shape[1] else None)
        z = ops.conv2d(x, p, act=ACT_RELU if relu else ACT_NONE, residual=res)
        ctx.cfg = (stride, pad, relu, res is not None)
        ctx.wparam = weight
        ctx.save_for_backward(x, weight, scale, z if relu else None)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, weight, scale, z = ctx.saved_tensors
        stride, pad, relu, has_res = ctx.cfg
        Cout, Cin, R, S = weight.shape
        comp = ctx.comp
        dz = _pad_c(dz.contiguous().to(x.dtype), comp)
        g = dz
        if relu:
            g, _, _, _ = bn_train_bwd(dz, _pad_c(z, comp), None, None, None, None, relu=True)
        dres = (g if g.shape[3] == Cout else g[..., :Cout]) if has_res else None
        sc = scale if g.shape[3] == Cout else ops.kpad(scale, (0, g.shape[3] - Cout))
        dconv = g * sc.to(g.dtype)
        Cw = dconv.shape[3]
        dwt = None
        if ctx.needs_input_grad[1] and not wgrad_to_param(ctx.wparam, x, dconv, Cw, R, S, stride, pad, R * S, x.shape[3], comp=comp):
            dw = conv_wgrad(x, dconv, Cw, R, S, stride, pad, comp=comp)[:Cout]
            dwt = _wgrad_to_oihw(dw, Cout, Cin, x.shape[3], R, S)
        dx = None
        if ctx.needs_input_grad[0]:
            wpad = weight if Cw == Cout else ops.kpad(weight.detach(), (0, 0, 0, 0, 0, 0, 0, Cw - Cout))
            dx = conv_dgrad(dconv, wpad, stride, pad, x.shape[1:3], comp=comp)
        return dx, dwt, None, None, dres, None, None, None


class ConvTransposeFn(torch.autograd.Function):
    """dense nn.ConvTranspose2d (weight [Cin, Cout, k, k], no bias, output_padding 0) of CenterNet._make_deconv_layer
    (centernet.py:268-293).  Forward: a conv over the input read as zero-stuffed in place.  Backward: dX = conv2d(dY, W,
    stride, padding) -- the transposed conv's adjoint is the plain strided conv with the same weight read as
    [out = Cin][in = Cout] -- and dW[ci][co] = the weight gradient of that conv with the roles of input (dY) and output
    gradient (X) as they stand."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        comp = ctx.comp = comp_of(x)
        y = ops.conv_transpose2d(x, weight.detach(), None, None, stride, pad, comp)
        ctx.cfg = (stride, pad)
        ctx.wparam = weight
        ctx.save_for_backward(x, weight)
        return y if y.shape[3] == weight.shape[1] else y[..., :weight.shape[1]]

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad = ctx.cfg
        Cin, Cout, k, _ = weight.shape
        comp = ctx.comp
        dy = _pad_c(dy.contiguous().to(x.dtype), comp)
        Cw = dy.shape[3]
        wpad = weight.detach() if Cw == Cout else ops.kpad(weight.detach(), (0, 0, 0, 0, 0, Cw - Cout))
        p = ops.PackedConv(wpad, None, None, stride=stride, pad=pad, compute=comp)   # rows = Cin, channels = Cw
        dx = ops.conv2d(dy, p)
        dx = dx if dx.shape[3] == Cin else dx[..., :Cin]
        xp = _pad_c(x, comp)
        # the parameter is [Cin, Cout, k, k]: "output channels" of this weight gradient = Cin, "input channels" = Cout
        if wgrad_to_param(ctx.wparam, dy, xp, xp.shape[3], k, k, stride, pad, k * k, Cw, comp=comp):
            return dx, None, None, None
        dw = conv_wgrad(dy, xp, xp.shape[3], k, k, stride, pad, comp=comp)[:Cin]         # [Cin, k*k*Cw]
        dwt = dw.view(Cin, k, k, Cw)[..., :Cout].permute(0, 3, 1, 2)
        return dx, dwt, None, None


def _pad_c(t, comp=None):
    """channel count up to a multiple of 8 (the f16 and f16x3 kernels work on 8-channel pieces) / 4 (f32)"""
    q = 8 if t is not None and (t.dtype == torch.float16 or comp == F16X3) else 4
    if t is None or t.shape[3] % q == 0:
        return t
    return ops.kpad(t, (0, q - t.shape[3] % q))


class BNActFn(torch.autograd.Function):
    """z = relu?(BatchNorm_train(y) + res) with batch statistics (running stats updated in place)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, res, running_mean, running_var, eps, momentum, relu):
        z, mean, invstd, scale = bn_train_fwd(y, gamma.detach(), beta.detach(), running_mean, running_var, eps, momentum,
                                              res=res, relu=relu)
        ctx.relu, ctx.has_res = relu, res is not None
        ctx.params = (gamma, beta)
        ctx.save_for_backward(y, z, mean, invstd, scale)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, z, mean, invstd, scale = ctx.saved_tensors
        gamma, beta = ctx.params
        sg, sb = grad_slot(gamma), grad_slot(beta)
        if sg is not None and sg.numel() != y.shape[3]:
            sg = None
        if sb is not None and sb.numel() != y.shape[3]:
            sb = None
        dy, dres, dgamma, dbeta = bn_train_bwd(dz.contiguous(), z, y, mean, invstd, scale, relu=ctx.relu,
                                               want_dres=ctx.has_res, into=(sg, sb))
        if sg is not None:
            grad_done(gamma)
        if sb is not None:
            grad_done(beta)
        return dy, None if sg is not None else dgamma, None if sb is not None else dbeta, dres, None, None, None, None, None


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.maxpool2x2(x)

    @staticmethod
    def backward(ctx, dz):
        (x,) = ctx.saved_tensors
        return maxpool2x2_bwd(x, dz.contiguous())


def maxpool3x3s2_bwd(x, dz, ceil_nopad):
    B, H, W, Cc = x.shape
    dx = torch.empty(B, H, W, Cc, dtype=x.dtype, device=x.device)
    rc = _lib.lib().ctdet_maxpool3x3s2_bwd(_ptr(x), _nhwc_stride(x), _ptr(dz), _nhwc_stride(dz), _ptr(dx), _nhwc_stride(dx),
                                           dt_of(x), B, H, W, Cc, int(ceil_nopad), _stream())
    _lib.check(rc, "ctdet_maxpool3x3s2_bwd")
    return dx


class MaxPool3x3s2Fn(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, ceil_mode=True) (VoVNet stage pooling, vovnet.py:291-292; ceil=True) or F.max_pool2d(x, 3, 2, 1)
    (BasicStem; ceil=False) with its autograd, NHWC, both directions on HIP kernels"""

    @staticmethod
    def forward(ctx, x, ceil):
        ctx.ceil = bool(ceil)
        ctx.save_for_backward(x)
        return ops.maxpool3x3s2_ceil(x) if ceil else ops.maxpool3x3s2(x)

    @staticmethod
    def backward(ctx, dz):
        (x,) = ctx.saved_tensors
        return maxpool3x3s2_bwd(x, dz.contiguous(), ctx.ceil), None


class EseFn(torch.autograd.Function):
    """eSEModule (vovnet.py:200-213): y = x * hsigmoid(fc(mean over pixels of x)) (+ identity).  The two passes over the map in
    each direction are HIP kernels (ctdet_global_avgpool / ctdet_ese_scale forward, ctdet_ese_dot / ctdet_ese_bwd backward); the
    B x C numbers in between -- the C x C fc layer, hsigmoid and its slope -- are device-side torch ops (multiply + reduce, no
    BLAS call: the step is captured as a HIP graph).  The fc layer's gradients go straight into the optimizer's flat buffer."""

    @staticmethod
    def forward(ctx, x, fc_w, fc_b, identity):
        Cc = fc_w.shape[0]
        pooled = ops.global_avgpool(x)                                               # f32 [B, C]
        w2 = fc_w.detach().view(Cc, Cc).float()
        s = (pooled[:, None, :] * w2[None]).sum(dim=2) + fc_b.detach().float()
        ctx.params = (fc_w, fc_b)
        ctx.has_identity = identity is not None
        ctx.save_for_backward(x, pooled, s, w2)
        return ops.ese_scale(x, s, identity)

    @staticmethod
    def backward(ctx, dy):
        x, pooled, s, w2 = ctx.saved_tensors
        fc_w, fc_b = ctx.params
        dy = dy.contiguous()
        B, H, W, Cc = x.shape
        r = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
        rc = _lib.lib().ctdet_ese_dot(_ptr(dy), _nhwc_stride(dy), _ptr(x), _nhwc_stride(x), dt_of(x), B, H * W, Cc, _ptr(r), _stream())
        _lib.check(rc, "ctdet_ese_dot")
        gate = torch.clamp(s + 3.0, 0.0, 6.0) / 6.0
        gs = r * ((s > -3.0) & (s < 3.0)).float() / 6.0                              # through relu6(s + 3) / 6
        gw = (gs[:, :, None] * pooled[:, None, :]).sum(dim=0)                        # [C_out, C_in]
        gb = gs.sum(dim=0)
        gp = ((gs[:, :, None] * w2[None]).sum(dim=1) / float(H * W)).contiguous()    # d(mean) spread over the pixels
        dx = torch.empty_like(x)
        rc = _lib.lib().ctdet_ese_bwd(_ptr(dy), _nhwc_stride(dy), _ptr(gate.contiguous()), _ptr(gp), _ptr(dx), _nhwc_stride(dx),
                                      dt_of(x), B, H * W, Cc, _stream())
        _lib.check(rc, "ctdet_ese_bwd")
        gw = (gw * PARAM_GRAD_MULT).view_as(fc_w)
        gb = gb * PARAM_GRAD_MULT
        gw = None if grad_into_slot(fc_w, gw) else gw
        gb = None if grad_into_slot(fc_b, gb) else gb
        return dx, gw, gb, (dy if ctx.has_identity else None)


class DwConvTAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, skip, f):
        ctx.f = f
        ctx.wparam = weight
        ctx.wk = ops._dw_weight(weight, x.shape[3], f, True)      # [2f, 2f, C] f32: made once per step, used by both passes
        ctx.save_for_backward(x, weight)
        return ops.dwconvT_add(x, ctx.wk.permute(2, 0, 1), f, skip=skip, prepared=ctx.wk)

    @staticmethod
    def backward(ctx, dz):
        x, weight = ctx.saved_tensors
        dz = dz.contiguous()
        slot = grad_slot(ctx.wparam)
        dx, dw = dwconvT_bwd(x, dz, weight, ctx.f, wk=ctx.wk, raw=slot is not None)
        if slot is not None:
            # tap-major [2f*2f][C] sums -> the [C, 1, 2f, 2f] gradient through the per-bucket scatter (as one "output channel"
            # of C inputs), scaled like every parameter gradient
            k2 = 4 * ctx.f * ctx.f
            src = dw.view(k2, -1) * PARAM_GRAD_MULT
            _queue_end_of_backward()
            PENDING.append((slot.view(1, slot.shape[0], slot.shape[2], slot.shape[3]), src, k2, slot.shape[0]))
            grad_done(ctx.wparam)
            return dx, None, dz, None
        return dx, dw * PARAM_GRAD_MULT, dz, None


class DCNFn(torch.autograd.Function):
    """modulated deformable conv (3x3/s1/p1) for training.  Forward: the fused sampling + MFMA kernel of the inference path
    (no column tensor).  Backward: the columns are materialised once there (as the reference does for both directions,
    deform_conv_cuda.cu:874-917) so dW and d(columns) are plain 1x1 contractions on the MFMA kernels; nothing of size
    9*Cin per pixel lives between forward and backward.  om: raw f32 [.., >=27] output of conv_offset_mask (mask logits);
    mask_is_prob: the reference's functional form passes sigmoid-ed masks instead (deform_conv.py:182-194)."""

    @staticmethod
    def forward(ctx, x, om, weight, bias, mask_is_prob=False, param_grad_mult=None):
        ctx.pgm = param_grad_mult
        f32 = x.dtype == torch.float32
        comp = ctx.comp = comp_of(x)
        p = ops.PackedConv(weight.detach(), None, bias, stride=1, pad=1, compute=comp, cout_align=None if f32 else 64)
        y = ops.dcnv2(x, om, p, mask_is_prob=mask_is_prob)
        ctx.mask_is_prob = mask_is_prob
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, om, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, om, weight = ctx.saved_tensors
        Cout, Cin = weight.shape[:2]
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if dy.shape[3] != Cout:
            dy = dy[..., :Cout].contiguous()
        comp = ctx.comp
        dyp = _pad_c(dy, comp)
        col = dcn_cols(x, om, ctx.mask_is_prob)
        chunked = comp != F32 and Cin % 32 == 0
        _, _, _, dbias = bn_train_bwd(dyp, None, None, None, None, None, relu=False, grad_mult=ctx.pgm)
        dw = conv_wgrad(col, dyp, dyp.shape[3], 1, 1, 1, 0, scale=ctx.pgm, comp=comp)[:Cout]         # [Cout, 9*Cin]
        dwt = dw.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        fused = dcn_col2im_fused(dyp, weight, x, om, ctx.mask_is_prob) if (comp == F16X3 and FUSE_DCOL) else None
        if fused is not None:
            dx32, dom = fused
        else:
            dcol = dcn_dcol(dyp, weight, chunked, comp)                    # [M, 9*Cin]
            dx32, dom = dcn_col2im_coord(dcol.contiguous(), x, om, ctx.mask_is_prob, dcol_chunked=chunked, comp=comp)
        return dx32.to(x.dtype), dom, dwt, dbias[:Cout] if ctx.has_bias else None, None, None


class DeformConvFn(torch.autograd.Function):
    """The DCN wrapper as ONE autograd node: om = conv_offset_mask(x) (3x3, 27 channels, f32 out), y = dcn(x, om).  x feeds both
    convs; as two nodes autograd adds their two input gradients with a generic elementwise kernel after a separate
    f32 -> f16 cast of the scatter result -- here the offset conv's input gradient is produced in f32 ON TOP of the scatter
    buffer (the conv kernel's residual input) and cast once."""

    @staticmethod
    def forward(ctx, x, w_off, b_off, weight, bias):
        f32 = x.dtype == torch.float32
        comp = ctx.comp = comp_of(x)
        p_off = _fwd_pack(w_off, 1, 1, bias=b_off, compute=comp)
        p = ops.PackedConv(weight.detach(), None, bias, stride=1, pad=1, compute=comp, cout_align=None if f32 else 64)
        if not f32 and x.shape[3] % 32 == 0 and ops.dcnv2_offset_supported(x, p_off, p):
            # one kernel for both convs; the offsets / mask logits are kept for the backward pass
            om = torch.empty(x.shape[0], x.shape[1], x.shape[2], 28, dtype=torch.float32, device=x.device)
            y, cols = ops.dcnv2_offset(x, p_off, p, om_out=om), None
        else:
            om = ops.conv2d(x, p_off, out_dtype=torch.float32)
            # f16x3: the forward kernel writes the sampled columns on the side (kept for the weight gradient: no second sampling
            # pass over the layer in the backward; 9*Cin floats per pixel held between the two passes)
            y, cols = ops.dcnv2(x, om, p, want_cols=True) if (comp == F16X3 and KEEP_COLS) else (ops.dcnv2(x, om, p), None)
        ctx.params = (w_off, b_off, weight, bias)
        ctx.cols = cols
        ctx.save_for_backward(x, om, w_off, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, om, w_off, weight = ctx.saved_tensors
        Cout, Cin = weight.shape[:2]
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        comp = ctx.comp
        dyp = _pad_c(dy if dy.shape[3] == Cout else dy[..., :Cout].contiguous(), comp)
        # ---- main conv: dW, d(columns) -> scatter (d input, f32) + d(offset / mask logits)
        chunked = comp != F32 and Cin % 32 == 0
        p_woff, p_boff, p_w, p_b = ctx.params
        sb = grad_slot(p_b) if p_b is not None else None
        if sb is not None and sb.numel() != dyp.shape[3]:
            sb = None
        _, _, _, dbias = bn_train_bwd(dyp, None, None, None, None, None, relu=False, into=(None, sb))
        # k = tap*Cin + c of the columns, which are sampled on the weight-gradient stream as well
        kept, ctx.cols = ctx.cols, None
        if wgrad_to_param(p_w, None, dyp, dyp.shape[3], 1, 1, 1, 0, 9, Cin, make_x=lambda: kept if kept is not None else dcn_cols(x, om),
                          keep=(x, om), comp=comp):
            dwt = None
        else:
            col = kept if kept is not None else dcn_cols(x, om)
            dw = conv_wgrad(col, dyp, dyp.shape[3], 1, 1, 1, 0, comp=comp)[:Cout]
            dwt = dw.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        if sb is not None:
            grad_done(p_b)
        # dom comes back in the data type and channel padding the offset conv's backward kernels take (f16 / f16x3: 32 channels)
        n_om = w_off.shape[0]
        f32 = x.dtype == torch.float32
        domc = (n_om + 3) // 4 * 4 if comp == F32 else (n_om + 7) // 8 * 8
        fused = dcn_col2im_fused(dyp, weight, x, om, dom_channels=domc) if (comp == F16X3 and FUSE_DCOL) else None
        if fused is not None:
            dx32, dom_p = fused
        else:
            dcol = dcn_dcol(dyp, weight, chunked, comp)
            dx32, dom_p = dcn_col2im_coord(dcol.contiguous(), x, om, dom_channels=domc, dcol_chunked=chunked, comp=comp)
        # ---- offset / mask conv: bias and weight gradients from dom; its input gradient lands on top of dx32
        Cw = dom_p.shape[3]
        _, _, _, db_off = bn_train_bwd(dom_p, None, None, None, None, None, relu=False)
        if wgrad_to_param(p_woff, x, dom_p, Cw, 3, 3, 1, 1, 9, x.shape[3], comp=comp):
            dw_off_t = None
        else:
            dw_off = conv_wgrad(x, dom_p, Cw, 3, 3, 1, 1, comp=comp)[:n_om]
            dw_off_t = _wgrad_to_oihw(dw_off, n_om, Cin, x.shape[3], 3, 3)
        if comp == F32:
            wpad = ops.kpad(w_off.detach(), (0, 0, 0, 0, 0, 0, 0, Cw - n_om)) if Cw != n_om else w_off.detach()
            wt = ops.flip_taps(wpad).permute(1, 0, 2, 3).contiguous()
            pt = ops.PackedConv(wt, None, None, stride=1, pad=1, compute=F32)
        else:   # the padded dY channels meet zero operand columns: packed from the parameter itself (a planned pack)
            pt = ops.PackedConv(w_off.detach(), None, None, stride=1, pad=1, compute=comp, transposed=True, cin_pad=Cw)
        dx = ops.conv2d(dom_p, pt, out=dx32 if pt.Cout_eff == Cin else None, residual=dx32 if pt.Cout_eff == Cin else None,
                        out_dtype=torch.float32)
        if pt.Cout_eff != Cin:
            dx = dx[..., :Cin] + dx32
        # no gradient for a bias that is not there (autograd raises on a tensor returned for a None input) or that was written
        # straight into its slot
        db_off_ret = None if grad_into_slot(p_boff, db_off[:n_om]) else db_off[:n_om]
        return dx.to(x.dtype), dw_off_t, db_off_ret, dwt, None if (sb is not None or p_b is None) else dbias[:Cout]


class FocalLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, gt, alpha):
        loss, stats, grad = ops.focal_loss(logits, gt, alpha, want_grad=True, grad_scale=GRAD_SCALE)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


class RegL1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, mask, ind, target):
        grad = torch.zeros_like(pred)  # pred may carry padded channels (only the first two are used)
        loss, grad = ops.reg_l1_loss(pred, mask, ind, target, want_grad=True, grad_scale=GRAD_SCALE, grad=grad)
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None
