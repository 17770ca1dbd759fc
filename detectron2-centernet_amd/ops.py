"""Host-side wrappers over the C ABI: torch tensors are used only as device-memory handles
(data_ptr + shape); every computation below happens in libctdet_hip.so on the current HIP stream.

Activation convention: NHWC tensors ``[B, H, W, C]`` that may be channel-slices of a wider NHWC
buffer (that is how Root's ``torch.cat`` (reference dla.py:88) is made free: producers write straight
into their slice of the concat buffer).
"""
import ctypes as C
import os
import math

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_SIGMOID_CLAMP, F16, F16X3, F32, U8, ConvDesc, HeadDesc

_TORCH_DT = {F16: torch.float16, F32: torch.float32}

# bench.py's roofline leg: when PROFILE_ON, every conv-shaped launch is bracketed by HIP events on the
# launch stream and (kernel instantiation, algorithmic FLOPs, start, end, algorithmic bytes, info, repetitions) is
# appended to PROFILE.
PROFILE_ON = False
PROFILE = []
PROFILE_REP = 5   # each profiled launch is issued this many times back to back between one event pair


def _kernel_name(p, M, deform, out_dt, x_shape=None, nsrc=1):
    """mirrors the kernel selection of launch_conv_f16_t() in csrc/conv_igemm.hip"""
    bc = _lib.lib().ctdet_conv_cout_tile(p.Cout_eff)
    if p.compute == F16X3:       # the f32 kernels' split instantiations: same selection, tagged
        H, W = (x_shape[1], x_shape[2]) if x_shape is not None else (0, 0)
        if (not deform and p.R == 3 and p.S == 3 and p.stride == 1 and p.pad == 1 and p.dil == 1 and p.in_dil == 1 and nsrc <= 1
                and p.Cin % 16 == 0 and p.Kpad == p.K and H and ((H % 8 == 0 and W % 32 == 0) or (H % 16 == 0 and W % 16 == 0 and (p.Cin // 16) % 2 == 0))
                and bc in (32, 64, 128)
                and not (_lib.lib().ctdet_get_tuning_flags() & _lib.TUNE_NO_HALO)):
            bc = min(bc, 64)
            return f"conv3x3_halo_pair{'2' if (p.Cin // 16) % 2 == 0 else ''}_kernel<256x{max(bc, 32)},f16x3>"
        p32 = _F32View(p)
        return _kernel_name(p32, M, deform, out_dt, x_shape, nsrc).replace("_f32_", "_f16x3_", 1)
    if p.compute != F16:
        big = ((M + 255) // 256) * (p.Cout_pad // bc) >= 512
        if deform:
            H, W = (x_shape[1], x_shape[2]) if x_shape is not None else (0, 0)
            win = (p.R == 3 and p.stride == 1 and p.pad == 1 and p.dil == 1 and H and H % 8 == 0 and W % 16 == 0 and
                   p.Kpad == p.K and p.Cout_pad % 64 == 0 and p.Cin % 16 == 0 and
                   not (_lib.lib().ctdet_get_tuning_flags() & _lib.TUNE_NO_F32_DCN_WINDOW))
            return f"dcn_f32_window_kernel<8x16,{p.Cout_pad}>" if win else f"dcn_f32_mfma_kernel<128x{bc}>"
        bp = 256 if (big or bc == 16) else 128
        H, W = (x_shape[1], x_shape[2]) if x_shape is not None else (0, 0)
        if p.Cin in (4, 8, 16) and p.R == p.S and p.dil == 1 and nsrc <= 1 and H:
            Ho, Wo = p.out_hw(H, W)
            if ((p.R, p.Cin, bc, p.stride) in ((7, 4, 16, 1), (7, 8, 16, 1), (3, 16, 16, 1)) and Ho % 8 == 0 and Wo % 64 == 0) or \
                    ((p.R, p.Cin, bc, p.stride) == (3, 16, 32, 2) and Ho % 4 == 0 and Wo % 32 == 0):
                return f"conv_f32_win_kernel<{p.R}x{p.R},Cin{p.Cin},Cout{bc},s{p.stride}>"
        uk = p.R * p.S <= 32 and p.in_dil == 1 and p.Kpad == p.K and p.Cin % 16 == 0
        return f"conv_f32_{'uk' if uk else 'mfma'}_kernel<{bp}x{bc}>"
    o = "f16" if out_dt == F16 else "f32"
    if deform:
        return f"dcn_window_kernel<128x{bc},{o}>"
    H, W = (x_shape[1], x_shape[2]) if x_shape is not None else (0, 0)
    big = ((M + 255) // 256) * (p.Cout_pad // bc) >= 512
    bp = 256 if (big or bc == 16) else 128
    if (p.R == 3 and p.S == 3 and p.stride == 1 and p.pad == 1 and p.dil == 1 and p.in_dil == 1 and nsrc <= 1 and p.korder == 1
            and H and H % 16 == 0 and W % 16 == 0 and W % 32 != 0 and bc in (32, 64, 128) and p.Cin % 64 == 0 and p.Kpad == 9 * p.Cin
            and not (_lib.lib().ctdet_get_tuning_flags() & (_lib.TUNE_NO_HALO_TAP2 | _lib.TUNE_NO_HALO))):
        return f"conv3x3_halo_tap2_kernel<16x16x{min(bc, 64)},{o}>"
    if (p.R == 3 and p.S == 3 and p.stride == 1 and p.pad == 1 and p.dil == 1 and p.in_dil == 1 and nsrc <= 1
            and p.korder == 1 and H % 8 == 0 and W % 32 == 0 and bc in (32, 64, 128)):
        if p.Cin % 64 == 0 and p.Kpad == 9 * p.Cin and not (_lib.lib().ctdet_get_tuning_flags() & _lib.TUNE_NO_HALO_TAP2):
            return f"conv3x3_halo_tap2_kernel<256x{min(bc, 64)},{o}>"
        small = not (_lib.lib().ctdet_get_tuning_flags() & _lib.TUNE_NO_SMALL_GRID_TILES)
        if bc == 128 and small and (M // 256) * (p.Cout_pad // 128) < 512:
            bc = 64                      # fewer workgroups than the chip holds: 64-cout tiles
        return f"conv3x3_halo_kernel<256x{bc},{o}>"
    Wo = (W + 2 * p.pad - p.dil * (p.S - 1) - 1) // p.stride + 1 if W else 0
    if p.Cin in (8, 16) and p.korder == 0 and nsrc <= 1 and Wo and Wo % 64 == 0 and p.Cout_pad <= 32:
        Ho = (H + 2 * p.pad - p.dil * (p.R - 1) - 1) // p.stride + 1
        win = ((p.R, p.Cin, bc, p.stride) in ((7, 8, 16, 1), (3, 16, 16, 1)) and Ho % 16 == 0) or \
              ((p.R, p.Cin, bc, p.stride) == (3, 16, 32, 2) and Ho % 8 == 0 and Wo % 32 == 0)
        if win and p.R == p.S and p.dil == 1 and not (_lib.lib().ctdet_get_tuning_flags() & _lib.TUNE_NO_WIN):
            return f"conv_win_kernel<{p.R}x{p.R},Cin{p.Cin},Cout{bc},s{p.stride},{o}>"
        return f"conv_smallc_kernel<Cout{bc},K{p.Kpad},{o}>"
    if p.Kpad == p.K and p.R * p.S <= 32 and p.in_dil == 1 and (p.korder == 1 or p.R * p.S == 1) and p.Cin % 32 == 0:
        if (bc == 128 and not big and ((M + 127) // 128) * (p.Cout_pad // 128) < 512
                and not (_lib.lib().ctdet_get_tuning_flags() & _lib.TUNE_NO_SMALL_GRID_TILES)):
            bc = 64
        return f"conv_igemm_uk_kernel<{bp}x{bc},{'cat' if nsrc > 1 else 'conv'},{o}>"
    return f"conv_igemm_dma_kernel<{bp}x{bc},{o}>"


class _F32View:
    """a PackedConv seen as its F32 twin (kernel selection of the F16X3 mode mirrors the F32 one)"""

    def __init__(self, p):
        self.__dict__.update(p.__dict__)
        self.compute = F32
        self.out_hw = p.out_hw


class _Prof:
    def __init__(self, p, M, deform, out_dt, x_shape=None, nsrc=1, name=None, flops=None):
        self.on = PROFILE_ON
        if self.on:
            self.name = name if name is not None else _kernel_name(p, M, deform, out_dt, x_shape, nsrc)
            self.flops = flops if flops is not None else 2.0 * M * p.Cout * p.R * p.S * p.Cin_real
            osz = 2 if out_dt == F16 else 4
            if p is not None:   # algorithmic bytes: input once + output once (+ offsets/masks for the deformable conv)
                isz = 2 if p.compute == F16 else 4
                in_px = M * (p.stride * p.stride if not deform else 1)
                self.bytes = in_px * p.Cin_real * isz + M * p.Cout * osz + (M * 27 * 4 if deform else 0) + p.Cout_pad * p.Kpad * isz
                self.info = f"M={M} {p.Cin_real}->{p.Cout} k{p.R} s{p.stride}" + (" dcn" if deform else "") + (f" cat{nsrc}" if nsrc > 1 else "")
            else:
                self.bytes, self.info = 0.0, f"M={M}"
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def reps(self):
        return PROFILE_REP if self.on else 1

    def done(self):
        if self.on:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.append((self.name, self.flops, self.e0, e1, self.bytes, self.info, PROFILE_REP))


class prof_region:
    """`with prof_region(name, flops=, nbytes=):` brackets the launches inside with HIP events on the launch stream when
    PROFILE_ON (bench.py's live per-kernel timing of the decode and of the training step); free otherwise."""

    def __init__(self, name, flops=0.0, nbytes=0.0, info=""):
        self.rec = (name, flops, nbytes, info) if PROFILE_ON else None

    def __enter__(self):
        if self.rec is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            name, flops, nbytes, info = self.rec
            PROFILE.append((name, flops, self.e0, e1, nbytes, info, 1))
        return False


# CTDET_RANGE_CHECK=1 (debug): after every f16x3 / f16 contraction the output's largest magnitude is read back and a value the
# NEXT layer cannot carry raises.  f16x3 holds an activation as hi + lo with hi = f16(x): beyond 65504 hi is inf, the layer's
# sums become inf / NaN -- and the ReLU of the epilogue (fmaxf) turns NaN into 0, so the network's outputs stay finite and
# wrong.  The check costs a host synchronisation per layer, so the eval engines run eagerly (no graph) while it is on.
RANGE_CHECK = os.environ.get("CTDET_RANGE_CHECK", "0") == "1"
F16_MAX = 65504.0


def _range_check(out, p, what):
    if RANGE_CHECK and p.compute in (F16, F16X3):
        amax = out.detach().float().abs().amax().item()
        if not (amax <= F16_MAX):
            raise FloatingPointError(f"{what}: output magnitude {amax:.4g} leaves the range the {'f16x3' if p.compute == F16X3 else 'f16'} "
                                     f"mode can carry into the next layer (|x| <= {F16_MAX:.0f}; Cin={p.Cin_real}, Cout={p.Cout}, k={p.R})")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            # same contract as the reference's native op (deform_conv.py:203-204): no CPU implementation
            raise NotImplementedError("the CenterNet HIP path has no CPU implementation; tensors must be on a ROCm device")


def _nhwc_stride(t):
    """pixel stride (elements) of an NHWC tensor or channel-slice view; validates the layout."""
    assert t.dim() == 4, f"expected NHWC 4-D tensor, got {tuple(t.shape)}"
    B, H, W, Cc = t.shape
    s = t.stride()
    ps = s[2] if W > 1 else (s[1] if H > 1 else (s[0] if B > 1 else Cc))
    if Cc > 1:
        assert s[3] == 1, "channel dim must be contiguous"
    assert ps >= Cc
    if W > 1 and H > 1:
        assert s[1] == W * ps, "rows must be dense"
    if B > 1 and H > 1:
        assert s[0] == H * W * ps, "images must be dense"
    return ps


def dt_of(t):
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.uint8:
        return U8
    raise TypeError(f"unsupported dtype {t.dtype}")


def round_up(a, b):
    return (a + b - 1) // b * b


def flip_taps(w):
    """w.flip(2, 3) of an OIHW weight; a 1x1 weight is returned as it is (torch's flip of size-1 dimensions is a plain copy, a
    hipMemcpyAsync: see kpad)"""
    return w if w.shape[2] * w.shape[3] == 1 else w.flip(2, 3)


def kpad(t, pad, value=0.0):
    """torch.nn.functional.pad(t, pad, value=value) for padding after the END of dimensions (pad = (0, n_last, 0, n_before_last,
    ...)), made of kernels only.  F.pad is fill + narrow().copy_(), and a contiguous narrow (a 1-D vector, dimension 0 of a
    weight) is copied by hipMemcpyAsync: a memcpy node in a captured step, which must hold kernels only (engine/graph_nodes.py)"""
    for i in range(0, len(pad), 2):
        assert pad[i] == 0, "kpad pads after the end of a dimension only"
        n, dim = int(pad[i + 1]), t.dim() - 1 - i // 2
        if n:
            shape = list(t.shape)
            shape[dim] = n
            t = torch.cat([t, t.new_full(shape, value)], dim)
    return t


class PackPlan:
    """The packed operands of every conv weight a training step packs (forward and input-gradient forms; f16 images and the
    split f16x3 images with their row scales), kept in persistent buffers and refreshed by ONE kernel per kind after the
    optimizer update (`run()`, called by solver.FlatSGD.step) instead of one pack launch per layer, direction and step.  An entry
    is valid while the weight's version counter still has the value `run()` (or the recording pack) saw; a stale or unknown
    weight is packed on the spot as before and (re)recorded."""

    def __init__(self, flat_param):
        """flat_param: the optimizer's parameter buffer -- only weights that live inside it are planned (their address is
        stable and their version counter is the parameter's; a temporary could reuse an address with a fresh counter).

        A captured training step holds raw pointers to every entry's packed buffer (its conv kernels) and to the descriptor
        table (its pack launch), so neither may ever be freed or moved while the plan lives: a stale entry is re-packed INTO
        its buffer (`repack`), a table that the entry set outgrew is retired, not dropped."""
        self.lo = flat_param.data_ptr()
        self.hi = self.lo + flat_param.numel() * flat_param.element_size()
        self.entries = {}        # key -> [weight, packed, args, version, scale buffer (f16x3 entries) or None]
        self.table = {0: None, 1: None}     # device descriptor tables (f16 / f16x3), rebuilt when the set of entries changes
        self.blocks = {0: 0, 1: 0}
        self._retired = []       # earlier tables: a captured graph may still launch the pack kernel on them

    def covers(self, weight):
        return self.lo <= weight.data_ptr() < self.hi

    def lookup(self, key, weight):
        e = self.entries.get(key)
        if e is not None and e[3] == weight._version:
            return e[1] if e[4] is None else (e[1], e[4])
        return None

    def stale_buffer(self, key):
        """the packed buffer (f16x3: the (buffer, scale) pair) of a known entry whose weight has changed since it was packed
        (None: unknown key)"""
        e = self.entries.get(key)
        if e is None:
            return None
        return e[1] if e[4] is None else (e[1], e[4])

    def record(self, key, weight, packed, args, scale=None):
        e = self.entries.get(key)
        if e is not None:                      # re-packed in place: same buffer, same table
            assert e[1].data_ptr() == packed.data_ptr()
            e[0], e[3] = weight, weight._version
            return
        self.entries[key] = [weight, packed, args, weight._version, scale]
        kind = 0 if scale is None else 1
        if self.table[kind] is not None:
            self._retired.append(self.table[kind])
        self.table[kind] = None

    def _build(self, kind, es):
        if kind == 0:
            arr = (_lib.PackDesc * len(es))()
            blk = 0
            for d, (w, wp, a, _, _) in zip(arr, es):
                d.w, d.packed = w.data_ptr(), wp.data_ptr()
                d.O, d.I, d.R, d.S, d.chans_pad, d.rows_pad, d.Kpad, d.korder, d.transposed = a
                d.blk0 = blk
                blk += (d.rows_pad * d.Kpad + 255) // 256
        else:
            arr = (_lib.Pack3Desc * len(es))()
            blk = 0
            for d, (w, wp, a, _, sc) in zip(arr, es):
                d.w, d.packed, d.scale_out = w.data_ptr(), wp.data_ptr(), sc.data_ptr()
                d.O, d.I, d.R, d.S, d.chans_pad, d.rows_pad, d.Kpad, d.layout, d.transposed, d.scale_n = a
                d.blk0 = blk
                blk += d.rows_pad
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table[kind], self.blocks[kind] = raw.to(es[0][1].device), blk

    def run(self):
        for kind in (0, 1):
            es = [e for e in self.entries.values() if (e[4] is None) == (kind == 0)]
            if not es:
                continue
            if self.table[kind] is None:
                self._build(kind, es)
            fn = _lib.lib().ctdet_pack_weights_batch if kind == 0 else _lib.lib().ctdet_pack_weights_x3_batch
            _lib.check(fn(_ptr(self.table[kind]), len(es), self.blocks[kind], _stream()), "ctdet_pack_weights_batch")
            for e in es:
                e[3] = e[0]._version


PACK_PLAN = None    # set by solver.FlatSGD on a GPU; None: every PackedConv packs for itself


def pack_x3(weight, args):
    """(packed f32-viewed image [rows_pad, Kpad], row scale [scale_n]) of ctdet_pack_weights_x3 for the f32 contiguous parameter
    `weight`; args = (O, I, R, S, chans_pad, rows_pad, Kpad, layout, transposed, scale_n).  Planned (persistent buffers, refreshed
    by the batched pack after every optimizer step) when the weight lives in the optimizer's flat buffer."""
    rows_pad, Kpad, scale_n = args[5], args[6], args[9]
    key = ("x3",) + tuple(args) + (weight.data_ptr(),)
    plan = PACK_PLAN if (PACK_PLAN is not None and PACK_PLAN.covers(weight)) else None
    bufs = plan.lookup(key, weight) if plan is not None else None
    if bufs is None:
        bufs = plan.stale_buffer(key) if plan is not None else None
        if bufs is None:
            bufs = (torch.empty(rows_pad, Kpad, dtype=torch.float32, device=weight.device),
                    torch.empty(scale_n, dtype=torch.float32, device=weight.device))
        rc = _lib.lib().ctdet_pack_weights_x3(_ptr(weight), _ptr(bufs[0]), _ptr(bufs[1]), *args, _stream())
        _lib.check(rc, "ctdet_pack_weights_x3")
        if plan is not None:
            plan.record(key, weight, bufs[0], tuple(args), scale=bufs[1])
    return bufs


class PackedConv:
    """Weights of one conv-shaped contraction, packed for the kernels, plus its folded epilogue.

    weight: [Cout, Cin, R, S] (PyTorch OIHW, the reference's parameter layout).
    scale/bias: per-output-channel f32 epilogue (folded BatchNorm and/or conv bias) or None.
    compute: F16 (f16 tensors, f16 MFMA), F32 (f32 tensors, f32 MFMA: the reference's arithmetic) or F16X3 (f32 tensors,
    every product as three f16 products on the f16 matrix pipe: ctdet_conv_desc in include/ctdet_hip.h).
    """

    def __init__(self, weight, scale=None, bias=None, stride=1, pad=0, dil=1, compute=F16, cin_pad=None,
                 tap_major=False, transposed=False, cout_align=None):
        """transposed: pack the input-gradient operand of `weight` (rows = its input channels, taps flipped) -- the
        conv that maps dY to dX; f16 / f16x3.  cout_align: pad the packed rows to a multiple of this (the DCNv2 window kernel
        works on 64-cout tiles whatever Cout is)."""
        _require_cuda(weight)
        # transposed = "dcn_cols" / "dcn_cols_chunked": the operand of DCNv2's d(columns) contraction for a [O, I, 3, 3] weight,
        # a 1x1 conv from dY's O channels to 9*I column channels (ctdet_pack_weights transposed = 2 / 3)
        dcn_mode = {"dcn_cols": 2, "dcn_cols_chunked": 3}[transposed] if isinstance(transposed, str) else 0
        if dcn_mode:
            assert compute in (F16, F16X3) and scale is None and bias is None and weight.dtype == torch.float32 and weight.is_contiguous()
            assert tuple(weight.shape[2:]) == (3, 3) and not tap_major
            Cin, Cout, R, S = weight.shape[0], 9 * weight.shape[1], 1, 1
        elif transposed:
            assert compute in (F16, F16X3) and scale is None and bias is None
            assert compute == F16 or (weight.dtype == torch.float32 and weight.is_contiguous())
            Cin, Cout, R, S = weight.shape
        else:
            Cout, Cin, R, S = weight.shape
        self.Cout, self.R, self.S = Cout, R, S
        self.Cin = cin_pad if cin_pad is not None else Cin
        self.Cin_real = Cin
        assert self.Cin >= Cin
        self.stride, self.pad, self.dil, self.compute = stride, pad, dil, compute
        self.in_dil = 1  # >1: input read as zero-stuffed (input gradient of a strided conv); set by the caller
        self.Cout_eff = round_up(Cout, 4)
        K = R * S * self.Cin
        self.K = K
        dev = weight.device
        self.korder = 1 if (compute == F16 and not tap_major and self.Cin % 32 == 0 and R * S > 1) else 0
        if compute == F16 and weight.dtype == torch.float32 and weight.is_contiguous():
            # one packing kernel instead of the torch chain below
            if self.Cin % 8:
                raise ValueError(f"f16 path needs Cin % 8 == 0 (got {self.Cin}); pass cin_pad")
            tile = max(_lib.lib().ctdet_conv_cout_tile(self.Cout_eff), cout_align or 1)
            self.Kpad = round_up(K, 32)
            self.Cout_pad = round_up(self.Cout_eff, tile)
            O, I = weight.shape[0], weight.shape[1]
            args = (O, I, weight.shape[2], weight.shape[3], self.Cin, self.Cout_pad, self.Kpad, self.korder,
                    dcn_mode if dcn_mode else int(bool(transposed)))
            plan = PACK_PLAN if (PACK_PLAN is not None and PACK_PLAN.covers(weight)) else None
            wp = plan.lookup(args + (weight.data_ptr(),), weight) if plan is not None else None
            if wp is None:
                # a stale planned entry is re-packed into its own buffer: a captured step reads that address
                wp = plan.stale_buffer(args + (weight.data_ptr(),)) if plan is not None else None
                if wp is None:
                    wp = torch.empty(self.Cout_pad, self.Kpad, dtype=torch.float16, device=dev)
                rc = _lib.lib().ctdet_pack_weights(_ptr(weight.detach()), _ptr(wp), *args, _stream())
                _lib.check(rc, "ctdet_pack_weights")
                if plan is not None:
                    plan.record(args + (weight.data_ptr(),), weight, wp, args)
            self.w = wp
            self.scale = self._pad_vec(scale, 1.0, dev)
            self.bias = self._pad_vec(bias, 0.0, dev)
            return
        if compute == F16X3 and weight.dtype == torch.float32 and weight.is_contiguous():
            # the split operands come from ctdet_pack_weights_x3, one launch per layout, built when a kernel first asks for
            # one (`w` / `scale`: the tap-major split image; `w_pair` / `scale_pair`: the tap-pair image of the 3x3 halo kernels)
            tile = max(_lib.lib().ctdet_conv_cout_tile(self.Cout_eff), cout_align or 1)
            self.Kpad = round_up(K, 16)
            self.Cout_pad = round_up(self.Cout_eff, tile)
            self._x3_src = (weight.detach(), dcn_mode if dcn_mode else int(bool(transposed)))
            self._x3 = {}
            self._user_scale = self._pad_vec(scale, 1.0, dev)
            self._pair_capable = (R, S, stride, pad, dil) == (3, 3, 1, 1, 1) and self.Cin % 16 == 0 and not dcn_mode
            self.bias = self._pad_vec(bias, 0.0, dev)
            return
        if transposed:
            weight = flip_taps(weight.detach()).permute(1, 0, 2, 3).contiguous()
        w = weight.detach().to(torch.float32).permute(0, 2, 3, 1)  # [Cout,R,S,Cin]
        if self.Cin != Cin:
            w = kpad(w, (0, self.Cin - Cin))
        # k ordering (ctdet_conv_desc.korder): chunk-major keeps the R*S taps of one 32-channel chunk adjacent
        if self.korder == 1:
            w = w.reshape(Cout, R * S, self.Cin // 32, 32).permute(0, 2, 1, 3)
        w = w.reshape(Cout, K)
        if compute == F16:
            if self.Cin % 8:
                raise ValueError(f"f16 path needs Cin % 8 == 0 (got {self.Cin}); pass cin_pad")
            tile = max(_lib.lib().ctdet_conv_cout_tile(self.Cout_eff), cout_align or 1)
            self.Kpad = round_up(K, 32)
            self.Cout_pad = round_up(self.Cout_eff, tile)
            wp = kpad(w.to(torch.float16), (0, self.Kpad - K, 0, self.Cout_pad - Cout))    # kernels only: see kpad
        else:
            # f32 MFMA path: the same [row = cout][k] image as the f16 operand, 16 k per LDS row
            tile = max(_lib.lib().ctdet_conv_cout_tile(self.Cout_eff), cout_align or 1)
            self.Kpad = round_up(K, 16)
            self.Cout_pad = round_up(self.Cout_eff, tile)
            wp = kpad(w.to(torch.float32), (0, self.Kpad - K, 0, self.Cout_pad - Cout))
            if wp.data_ptr() == weight.data_ptr():    # nothing to pad or reorder (1x1): the image is still a snapshot of the weight,
                wp = torch.mul(wp, 1.0)               # made by a kernel (clone() of a contiguous tensor is a hipMemcpyAsync)
            if compute == F16X3:   # same image, every group of 4 k as {w_hi[4], w_lo[4]} f16
                # rows are first scaled by a power of two (exact) so that their largest weight lies in [1024, 2048): the lo
                # halves of a row's significant weights then stay f16 normals (2^-22 of the row maximum instead of the 3e-8
                # absolute floor of the f16 subnormals, which is 1e-6 of a typical 0.02 weight); the epilogue scale undoes it
                amax = wp.abs().amax(dim=1)
                e = torch.floor(torch.log2(amax.clamp_min(1e-30)))
                pw = torch.where(amax > 0, torch.exp2(10.0 - e), torch.ones_like(amax))
                wp = wp * pw[:, None]
                inv = (1.0 / pw)[:self.Cout_eff]
                scale = inv if scale is None else self._pad_vec(scale, 1.0, dev) * inv
                if (R, S, stride, pad, dil) == (3, 3, 1, 1, 1) and self.Cin % 16 == 0 and not transposed:
                    self._wp_scaled = wp          # kept until the first plain-conv use builds the pair image from it
                ws = torch.empty_like(wp)
                _lib.check(_lib.lib().ctdet_split_weights(_ptr(wp), _ptr(ws), wp.numel(), _stream()), "ctdet_split_weights")
                wp = ws
        self.w = wp.contiguous()

        self.scale = self._pad_vec(scale, 1.0, dev)
        self.bias = self._pad_vec(bias, 0.0, dev)

    pair_korder = 2
    w_pair = None      # F16X3, 3x3 / s1 / p1: the tap-pair image of the halo-resident kernels (ctdet_conv_desc.korder 2 / 3),
    scale_pair = None  # its epilogue scale (the kernel-packed path; the torch path shares `scale`)
    _wp_scaled = None  # built on the first conv2d() that can use it (a DCNv2 weight never does)
    _x3_src = None     # (f32 OIHW parameter, transposed mode) when the f16x3 operands come from ctdet_pack_weights_x3
    _pair_capable = False
    _w = _scale = None

    @property
    def w(self):
        return self._w if self._x3_src is None else self._x3_operand(0)[0]

    @w.setter
    def w(self, v):
        self._w = v

    @property
    def scale(self):
        return self._scale if self._x3_src is None else self._x3_operand(0)[1]

    @scale.setter
    def scale(self, v):
        self._scale = v

    def _x3_operand(self, layout):
        """(packed image viewed as f32 [rows_pad, Kpad], epilogue scale) of one layout of ctdet_pack_weights_x3; planned
        (persistent buffers refreshed in one launch after the optimizer step) when the weight lives in the optimizer's buffer"""
        e = self._x3.get(layout)
        if e is not None:
            return e
        weight, tmode = self._x3_src
        nch = self.Cin // 16
        if layout == 0:
            rows_pad, Kpad = self.Cout_pad, self.Kpad
        else:
            rows_pad, Kpad = round_up(self.Cout_pad, 32), (nch // 2 * 288 if layout == 3 else nch * 160)
        O, I = weight.shape[0], weight.shape[1]
        R, S = (1, 1) if tmode >= 2 else (weight.shape[2], weight.shape[3])
        bufs = pack_x3(weight, (O, I, R, S, self.Cin, rows_pad, Kpad, layout, tmode, self.Cout_eff))
        sc = bufs[1] if self._user_scale is None else bufs[1] * self._user_scale
        e = self._x3[layout] = (bufs[0], sc)
        return e

    def _pack_pairs(self, wp):
        """wp: the scaled tap-major f32 image [Cout_pad, 9*Cin] -> pair image viewed as f32, rows padded to a multiple of 32 (the
        narrowest tile of the kernel).  A 128-byte step {X, Y} multiplies two (tap, 16-channel chunk) operands a, b:
        X = for q in 0..3 {w_hi[a][4q..4q+3], w_hi[b][4q..4q+3]}, Y likewise from w_lo.
        Cin % 32 == 0 (korder 3, [rows, Cin/32*288]): per chunk PAIR (A, B) nine steps -- taps (2s, 2s+1) of A, s = 0..3; tap 8
        of A with tap 8 of B; taps (2s, 2s+1) of B.  Otherwise (korder 2, [rows, Cin/16*160]): per chunk five steps, taps
        (2s, 2s+1), the tenth tap zero.  Returns (image, korder)."""
        rows = round_up(self.Cout_pad, 32)
        nch = self.Cin // 16
        w = torch.zeros(rows, 10, self.Cin, dtype=torch.float32, device=wp.device)
        w[:wp.shape[0], :9] = wp.reshape(wp.shape[0], 9, self.Cin)
        hi = w.to(torch.float16)
        lo = (w - hi.float()).to(torch.float16)
        if nch % 2 == 0:
            hl = torch.stack([hi, lo], 0).reshape(2, rows, 10, nch // 2, 2, 4, 4)   # [X/Y, row, tap, chunk pair, A/B, q, j]
            steps = ([((2 * s, 0), (2 * s + 1, 0)) for s in range(4)] + [((8, 0), (8, 1))]
                     + [((2 * s, 1), (2 * s + 1, 1)) for s in range(4)])
            tap = torch.tensor([[a[0], b[0]] for a, b in steps], device=wp.device)    # [9, 2]
            ab = torch.tensor([[a[1], b[1]] for a, b in steps], device=wp.device)
            g = hl[:, :, tap, :, ab]                # [9, 2, X/Y, row, chunk pair, q, j] (advanced indices first)
            img = g.permute(3, 4, 0, 2, 5, 1, 6).contiguous()                       # [row, chunk pair, step, X/Y, q, operand, j]
            return img.reshape(rows, nch // 2 * 576).view(torch.float32), 3
        hl = torch.stack([hi, lo], 0).reshape(2, rows, 5, 2, nch, 4, 4)     # [X/Y, row, pair, tap of pair, chunk, q, j]
        img = hl.permute(1, 4, 2, 0, 5, 3, 6).contiguous()                  # [row, chunk, pair, X/Y, q, tap of pair, j]
        return img.reshape(rows, nch * 320).view(torch.float32), 2

    def pair_ok(self, x):
        """may the halo pair kernel take this input? (mirrors launch_halo_pair in csrc/conv_igemm.hip)"""
        # (a 16-channel input on a 64-divisible map is the LDS-window kernel's: level0 of DLA-34, 492 vs 825 us per 64 images)
        tile32 = x.shape[1] % 8 == 0 and x.shape[2] % 32 == 0
        tile16 = x.shape[1] % 16 == 0 and x.shape[2] % 16 == 0 and (self.Cin // 16) % 2 == 0     # (16x16 tiles: korder 3 only)
        ok = ((self.w_pair is not None or self._wp_scaled is not None or self._pair_capable) and self.in_dil == 1 and (tile32 or tile16)
              and _nhwc_stride(x) % 4 == 0 and x.data_ptr() % 16 == 0
              and (self.Cin > 16 or x.shape[2] % 64 != 0 or _nhwc_stride(x) != self.Cin)
              and not (_lib.lib().ctdet_get_tuning_flags() & _lib.TUNE_NO_HALO))
        if ok and self.w_pair is None:
            if self._x3_src is not None:
                self.pair_korder = 3 if (self.Cin // 16) % 2 == 0 else 2
                self.w_pair, self.scale_pair = self._x3_operand(self.pair_korder)
            else:
                (self.w_pair, self.pair_korder), self._wp_scaled = self._pack_pairs(self._wp_scaled), None
        return ok

    def _pad_vec(self, v, fill, dev):
        if v is None:
            return None
        v = v.detach().to(torch.float32)
        if v.shape[0] == self.Cout_eff:
            return v.contiguous()
        return kpad(v.contiguous(), (0, self.Cout_eff - v.shape[0]), value=float(fill))   # not a slice assignment: see kpad

    @property
    def act_dt(self):
        """dtype enum of the activations this contraction takes"""
        return F16 if self.compute == F16 else F32

    def out_hw(self, H, W):
        Ho = (H + 2 * self.pad - (self.dil * (self.R - 1) + 1)) // self.stride + 1
        Wo = (W + 2 * self.pad - (self.dil * (self.S - 1) + 1)) // self.stride + 1
        return Ho, Wo

    def desc(self, x, out, act, residual, clamp=(0.0, 1.0), allow_pair=False):
        B, H, W, Cx = x.shape
        assert Cx == self.Cin, f"conv expects {self.Cin} input channels, got {Cx}"
        if self.in_dil > 1:
            Ho, Wo = out.shape[1], out.shape[2]
        else:
            Ho, Wo = self.out_hw(H, W)
        assert tuple(out.shape[:3]) == (B, Ho, Wo) and out.shape[3] >= self.Cout_eff, (out.shape, (B, Ho, Wo, self.Cout_eff))
        d = ConvDesc()
        d.B, d.H, d.W, d.Cin, d.in_stride = B, H, W, self.Cin, _nhwc_stride(x)
        d.Cout, d.Ho, d.Wo, d.out_stride = self.Cout_eff, Ho, Wo, _nhwc_stride(out)
        d.R, d.S, d.stride, d.pad, d.dil = self.R, self.S, self.stride, self.pad, self.dil
        d.Kpad, d.Cout_pad = self.Kpad, self.Cout_pad
        d.compute_dtype, d.out_dtype, d.act = self.compute, dt_of(out), act
        d.res_stride = _nhwc_stride(residual) if residual is not None else 0
        d.clamp_lo, d.clamp_hi = clamp
        d.korder = self.korder
        d.in_dil = self.in_dil
        if allow_pair and self.pair_ok(x):
            d.korder, d.Kpad, d.Cout_pad = self.pair_korder, self.w_pair.shape[1], self.w_pair.shape[0]
        return d


def _alloc_out(x, p, out, out_dtype):
    B, H, W, _ = x.shape
    Ho, Wo = p.out_hw(H, W)
    assert out is not None or p.in_dil == 1, "input-dilated convs need an explicit output buffer"
    if out is None:
        dt = out_dtype if out_dtype is not None else (torch.float16 if p.compute == F16 else torch.float32)
        out = torch.empty(B, Ho, Wo, p.Cout_eff, dtype=dt, device=x.device)
    return out


def conv2d(x, p, out=None, act=ACT_NONE, residual=None, out_dtype=None, clamp=(0.0, 1.0)):
    """y = act(conv(x) * scale + bias + residual); x NHWC. Returns the NHWC output buffer."""
    _require_cuda(x, residual, out)
    assert dt_of(x) == p.act_dt, "input dtype must match the packed compute dtype"
    out = _alloc_out(x, p, out, out_dtype)
    if residual is not None:
        assert residual.dtype == out.dtype and residual.shape[:3] == out.shape[:3]
    d = p.desc(x, out, act, residual, clamp, allow_pair=True)
    prof = _Prof(p, d.B * d.Ho * d.Wo, False, d.out_dtype, x.shape)
    if d.korder >= 2:
        w, sc = p.w_pair, (p.scale_pair if p.scale_pair is not None else p.scale)
    else:
        w, sc = p.w, p.scale
    for _ in range(prof.reps()):
        rc = _lib.lib().ctdet_conv2d_fwd(C.byref(d), _ptr(x), _ptr(w), _ptr(sc), _ptr(p.bias), _ptr(residual),
                                         _ptr(out), _stream())
    _lib.check(rc, "ctdet_conv2d_fwd")
    prof.done()
    _range_check(out, p, "conv2d")
    return out


def conv1x1_cat(xs, p, out=None, act=ACT_NONE, residual=None, out_dtype=None):
    """1x1 conv over the channel-concatenation of the NHWC tensors `xs` (<= 4) without building the concat."""
    _require_cuda(*xs)
    assert 1 <= len(xs) <= 4 and p.R == 1 and p.S == 1 and p.stride == 1 and p.pad == 0
    B, H, W, _ = xs[0].shape
    for t in xs:
        assert tuple(t.shape[:3]) == (B, H, W) and dt_of(t) == p.act_dt
    cins = [t.shape[3] for t in xs]
    assert sum(cins) == p.Cin, (cins, p.Cin)
    if out is None:
        dt = out_dtype if out_dtype is not None else (torch.float16 if p.compute == F16 else torch.float32)
        out = torch.empty(B, H, W, p.Cout_eff, dtype=dt, device=xs[0].device)
    d = ConvDesc()
    d.B, d.H, d.W, d.Cin, d.in_stride = B, H, W, p.Cin, p.Cin
    d.Cout, d.Ho, d.Wo, d.out_stride = p.Cout_eff, H, W, _nhwc_stride(out)
    d.R = d.S = d.stride = d.dil = 1
    d.pad = 0
    d.Kpad, d.Cout_pad = p.Kpad, p.Cout_pad
    d.compute_dtype, d.out_dtype, d.act = p.compute, dt_of(out), act
    d.res_stride = _nhwc_stride(residual) if residual is not None else 0
    d.clamp_lo, d.clamp_hi = 0.0, 1.0
    d.korder = 0
    d.in_dil = 1
    n = len(xs)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in xs])
    cin_a = (C.c_int32 * n)(*cins)
    str_a = (C.c_int32 * n)(*[_nhwc_stride(t) for t in xs])
    prof = _Prof(p, B * H * W, False, d.out_dtype, xs[0].shape, len(xs))
    for _ in range(prof.reps()):
        rc = _lib.lib().ctdet_conv1x1_cat_fwd(C.byref(d), ptrs, cin_a, str_a, n, _ptr(p.w), _ptr(p.scale),
                                              _ptr(p.bias), _ptr(residual), _ptr(out), _stream())
    _lib.check(rc, "ctdet_conv1x1_cat_fwd")
    prof.done()
    _range_check(out, p, "conv1x1_cat")
    return out


HEADS_FUSED = os.environ.get("CTDET_NO_FUSED_HEADS", "0") != "1"


def dcnv2(x, offset_mask, p, out=None, act=ACT_NONE, out_dtype=None, mask_is_prob=False, want_cols=False):
    """Modulated deformable conv: offset_mask is the raw f32 NHWC output of conv_offset_mask (>= 27 ch);
    with mask_is_prob the 9 mask channels already went through sigmoid.
    want_cols (f16x3, training): returns (y, cols) where cols f32 [B,H,W,9*Cin] are the sampled columns written by the same
    kernel, or (y, None) when the layer is not served by the LDS-window kernel."""
    _require_cuda(x, offset_mask, out)
    assert dt_of(x) == p.act_dt and offset_mask.dtype == torch.float32
    if p.compute == F16 and p.Cout_pad % 64:
        raise ValueError(f"dcnv2 (f16) works on 64-cout tiles: pack the weights with PackedConv(..., cout_align=64) "
                         f"(Cout={p.Cout}, packed rows {p.Cout_pad})")
    if out is None and p.compute == F16 and p.Cout_eff % 8:
        # the window kernel stores 16 bytes at a time: give the buffer a pixel stride that is a multiple of 8 channels and
        # hand back the view of the real ones
        B, H, W, _ = x.shape
        dt = out_dtype if out_dtype is not None else torch.float16
        out = torch.empty(B, H, W, round_up(p.Cout_eff, 8), dtype=dt, device=x.device)[..., :p.Cout_eff]
    out = _alloc_out(x, p, out, out_dtype)
    assert tuple(offset_mask.shape[:3]) == tuple(out.shape[:3])
    p._wp_scaled = None          # a deformable conv's weights: no pair image will be needed
    d = p.desc(x, out, act, None)
    cols = None
    if want_cols and p.compute == F16X3 and _lib.lib().ctdet_dcnv2_cols_supported(C.byref(d), _ptr(x), _ptr(out)):
        cols = torch.empty(x.shape[0], x.shape[1], x.shape[2], 9 * p.Cin, dtype=torch.float32, device=x.device)
    prof = _Prof(p, d.B * d.Ho * d.Wo, True, d.out_dtype, x.shape)     # (after the allocation: a 600 MB hipMalloc is not kernel time)
    if prof.on and cols is not None:
        prof.bytes += cols.numel() * 4
    for _ in range(prof.reps()):
        if cols is not None:
            rc = _lib.lib().ctdet_dcnv2_fwd_cols(C.byref(d), _ptr(x), _ptr(offset_mask), _nhwc_stride(offset_mask), int(mask_is_prob),
                                                 _ptr(p.w), _ptr(p.scale), _ptr(p.bias), _ptr(out), _ptr(cols), _stream())
        else:
            rc = _lib.lib().ctdet_dcnv2_fwd(C.byref(d), _ptr(x), _ptr(offset_mask), _nhwc_stride(offset_mask),
                                            int(mask_is_prob), _ptr(p.w), _ptr(p.scale), _ptr(p.bias), _ptr(out), _stream())
    _lib.check(rc, "ctdet_dcnv2_fwd")
    prof.done()
    _range_check(out, p, "dcnv2")
    return (out, cols) if want_cols else out


def dcnv2_offset_supported(x, p_off, p):
    """may `dcnv2_offset` serve this layer? (f16, 3x3/s1/p1 both convs, at most 64 couts packed to 64 rows, map divisible by
    the 8x16 tile, Cin % 32 == 0, 27 offset / mask channels packed chunk-major into 32 rows)"""
    if p.compute != F16 or p_off.compute != F16 or p.Cout_pad != 64 or p_off.Cout != 27 or p_off.Cout_pad != 32:
        return False
    if (p_off.R, p_off.S, p_off.stride, p_off.pad, p_off.dil, p_off.korder, p_off.Kpad) != (3, 3, 1, 1, 1, 1, p.Kpad):
        return False
    B, H, W, _ = x.shape
    out = torch.empty(0, H, W, round_up(p.Cout_eff, 8), dtype=torch.float16, device=x.device)
    d = p.desc(x[:0], out, ACT_NONE, None)
    d.B = B
    return bool(_lib.lib().ctdet_dcnv2_offset_supported(C.byref(d)))


def dcnv2_offset(x, p_off, p, out=None, act=ACT_NONE, out_dtype=None, om_out=None):
    """act(dcn(x, conv_offset_mask(x))) in one kernel (ctdet_dcnv2_offset_fwd): p_off = the packed 3x3 offset / mask conv with
    its bias, p = the packed deformable conv.  om_out (f32 [B,H,W,>=28]) receives the offsets / mask logits if given."""
    _require_cuda(x, out, om_out)
    assert dt_of(x) == F16 and p.compute == F16 and p_off.bias is not None and p_off.scale is None
    if out is None and p.Cout_eff % 8:
        B, H, W, _ = x.shape
        dt = out_dtype if out_dtype is not None else torch.float16
        out = torch.empty(B, H, W, round_up(p.Cout_eff, 8), dtype=dt, device=x.device)[..., :p.Cout_eff]
    out = _alloc_out(x, p, out, out_dtype)
    assert p_off.bias.shape[0] >= 28                     # Cout_eff floats: the kernel reads channels 0..27
    d = p.desc(x, out, act, None)
    prof = _Prof(p, d.B * d.Ho * d.Wo, True, d.out_dtype, x.shape, name=f"dcn_window_rows_kernel<128x64,offset conv fused>",
                 flops=2.0 * d.B * d.Ho * d.Wo * p.K * (p.Cout + 27))
    if prof.on:
        prof.bytes -= d.B * d.Ho * d.Wo * 27 * 4      # no offset tensor is read
    for _ in range(prof.reps()):
        rc = _lib.lib().ctdet_dcnv2_offset_fwd(C.byref(d), _ptr(x), _ptr(p_off.w), _ptr(p_off.bias), _ptr(om_out),
                                               _nhwc_stride(om_out) if om_out is not None else 0, _ptr(p.w), _ptr(p.scale),
                                               _ptr(p.bias), _ptr(out), _stream())
    _lib.check(rc, "ctdet_dcnv2_offset_fwd")
    prof.done()
    return out


def preprocess(images, mean, std, Hp, Wp, out_dtype=torch.float16, out=None, partial=False, border=0):
    """images: [B,3,H,W] uint8/float32 CHW on device -> normalised NHWC [B,Hp,Wp,8] (3 channels used).
    `out` may be a batch-slice of a larger padded buffer (ragged batches: one call per image).
    border > 0: `out` is a caller-owned [B,Hp+2b,Wp+2b,8] buffer whose zero frame was cleared once."""
    _require_cuda(images)
    B, Cc, H, W = images.shape
    assert Cc == 3 and images.stride(3) == 1 and images.stride(2) == W and images.stride(1) == H * W
    if out is None:
        if border:
            out = torch.zeros(B, Hp + 2 * border, Wp + 2 * border, 8, dtype=out_dtype, device=images.device)
        else:
            out = torch.empty(B, Hp, Wp, 8, dtype=out_dtype, device=images.device)
    assert out.shape[3] % 8 == 0 or (out.shape[3] == 4 and out.dtype == torch.float32)
    assert tuple(out.shape[1:3]) == (Hp + 2 * border, Wp + 2 * border)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    rc = _lib.lib().ctdet_preprocess(_ptr(images), dt_of(images), _ptr(out), dt_of(out), B, H, W, Hp, Wp,
                                     images.stride(0), m, s, _nhwc_stride(out), int(border), _stream())
    _lib.check(rc, "ctdet_preprocess")
    return out


class PackedDlaBase:
    """operands of ctdet_dla_base_fwd: the 7x7 stem [16,3,7,7], level0 [16,16,3,3] and level1 [32,16,3,3] weights with
    their folded BatchNorm (scale, bias) pairs."""

    def __init__(self, w_stem, sb_stem, w_l0, sb_l0, w_l1, sb_l1):
        assert tuple(w_stem.shape) == (16, 3, 7, 7) and tuple(w_l0.shape) == (16, 16, 3, 3) and tuple(w_l1.shape) == (32, 16, 3, 3)
        dev = w_stem.device
        # stem operand: k = (r*8 + s)*4 + c -- 4-channel pixels, kernel rows padded to 8 taps (one MFMA K step per row)
        w4 = torch.zeros(16, 7, 8, 4, dtype=torch.float32, device=dev)
        w4[:, :, :7, :3] = w_stem.detach().float().permute(0, 2, 3, 1)
        self.w0 = w4.reshape(16, 224).to(torch.float16).contiguous()
        self.p1 = PackedConv(w_l0, sb_l0[0], sb_l0[1], stride=1, pad=1, compute=F16)
        self.p2 = PackedConv(w_l1, sb_l1[0], sb_l1[1], stride=2, pad=1, compute=F16)
        assert self.p1.korder == 0 and self.p1.Kpad == 160 and self.p2.korder == 0 and self.p2.Kpad == 160
        self.s0 = sb_stem[0].detach().float().contiguous()
        self.b0 = sb_stem[1].detach().float().contiguous()


class PackedDlaBaseX3:
    """operands of ctdet_dla_base_x3_fwd: the three layers as f16x3 PackedConvs (the split tap-major images of
    ctdet_pack_weights_x3, the stem on 4-channel pixels) -- exactly what the layer-by-layer f16x3 path contracts with."""
    x3 = True

    def __init__(self, w_stem, sb_stem, w_l0, sb_l0, w_l1, sb_l1):
        assert tuple(w_stem.shape) == (16, 3, 7, 7) and tuple(w_l0.shape) == (16, 16, 3, 3) and tuple(w_l1.shape) == (32, 16, 3, 3)
        # stem operand: k = (r*8 + s)*4 + c -- 4-channel pixels, kernel rows padded to 8 taps (a K step = four consecutive
        # taps of a row); packed as the [16, 224, 1, 1] weight of a 1x1 contraction
        w4 = torch.zeros(16, 7, 8, 4, dtype=torch.float32, device=w_stem.device)
        w4[:, :, :7, :3] = w_stem.detach().float().permute(0, 2, 3, 1)
        self.p0 = PackedConv(w4.reshape(16, 224, 1, 1), sb_stem[0], sb_stem[1], compute=F16X3)
        self.p1 = PackedConv(w_l0.detach().float().contiguous(), sb_l0[0], sb_l0[1], stride=1, pad=1, compute=F16X3)
        self.p2 = PackedConv(w_l1.detach().float().contiguous(), sb_l1[0], sb_l1[1], stride=2, pad=1, compute=F16X3)
        for p, kpad, rows in ((self.p0, 224, 16), (self.p1, 144, 16), (self.p2, 144, 32)):
            assert p.korder == 0 and p.Kpad == kpad and p.w.shape[0] >= rows and p.w.shape[1] == kpad, (p.korder, p.Kpad, p.w.shape)
            assert p.scale.numel() >= rows and p.bias.numel() >= rows


BASE_FUSED = os.environ.get("CTDET_NO_FUSED_BASE", "0") != "1"


def dla_base_fused_ok(Hp, Wp):
    return BASE_FUSED and Hp % 16 == 0 and Wp % 32 == 0


def dla_base_fused(images, mean, std, Hp, Wp, p, out=None, pooled=None):
    """images [B,3,H,W] uint8/f32 on device -> level1 output of DLA (NHWC [B,Hp/2,Wp/2,32]; f16, or f32 when p is a
    PackedDlaBaseX3: f16x3 arithmetic) in one launch: normalisation, 7x7 stem, level0, level1 (stride 2), BatchNorm folded,
    ReLU after each.  pooled: optional [B,Hp/4,Wp/4,>=32] buffer of the output's dtype that receives MaxPool2d(2) of the
    output (what level2's Tree starts with)."""
    _require_cuda(images, out, pooled)
    B, Cc, H, W = images.shape
    assert Cc == 3 and images.stride(3) == 1 and images.stride(2) == W and images.stride(1) == H * W
    x3 = getattr(p, "x3", False)
    odt = torch.float32 if x3 else torch.float16
    if out is None:
        out = torch.empty(B, Hp // 2, Wp // 2, 32, dtype=odt, device=images.device)
    assert out.dtype == odt and tuple(out.shape[:3]) == (B, Hp // 2, Wp // 2) and out.shape[3] >= 32
    d = _lib.DlaBaseDesc()
    d.B, d.H, d.W, d.Hp, d.Wp, d.img_dtype = B, H, W, Hp, Wp, dt_of(images)
    d.img_batch_stride = images.stride(0)
    for i in range(3):
        d.mean[i], d.std[i] = float(mean[i]), float(std[i])
    d.out_stride = _nhwc_stride(out)
    if pooled is not None:
        assert pooled.dtype == odt and tuple(pooled.shape[:3]) == (B, Hp // 4, Wp // 4) and pooled.shape[3] >= 32
        d.pool_stride = _nhwc_stride(pooled)
    px = B * Hp * Wp
    prof = _Prof(None, px, False, F16X3 if x3 else F16,
                 name="dla_base_x3_kernel<u8|f32 -> 32ch,f16x3>" if x3 else "dla_base_fused_kernel<u8|f32 -> 32ch,f16>",
                 flops=2.0 * (px * 16 * 147 + px * 16 * 144 + (px // 4) * 32 * 144))
    for _ in range(prof.reps()):
        if x3:
            rc = _lib.lib().ctdet_dla_base_x3_fwd(C.byref(d), _ptr(images), _ptr(p.p0.w), _ptr(p.p0.scale), _ptr(p.p0.bias),
                                                  _ptr(p.p1.w), _ptr(p.p1.scale), _ptr(p.p1.bias), _ptr(p.p2.w),
                                                  _ptr(p.p2.scale), _ptr(p.p2.bias), _ptr(out), _ptr(pooled), _stream())
        else:
            rc = _lib.lib().ctdet_dla_base_fwd(C.byref(d), _ptr(images), _ptr(p.w0), _ptr(p.s0), _ptr(p.b0), _ptr(p.p1.w),
                                               _ptr(p.p1.scale), _ptr(p.p1.bias), _ptr(p.p2.w), _ptr(p.p2.scale),
                                               _ptr(p.p2.bias), _ptr(out), _ptr(pooled), _stream())
    _lib.check(rc, "ctdet_dla_base_fwd")
    if prof.on:
        prof.bytes = images.numel() * images.element_size() + (px // 4) * 32 * out.element_size() * (1.25 if pooled is not None else 1.0)
        prof.info = f"M={px} 3->16->16->32 fused"
    prof.done()
    return out


class PackedHeads:
    """weights of the fused CenterNet heads (ctdet_head_fused_fwd): first convs [256, Cin, 3, 3] + bias per head, final
    1x1 convs [cout, 256, 1, 1] + bias per head, acts per head."""

    HID = 256

    def __init__(self, first_weights, first_biases, final_weights, final_biases, acts):
        n = len(first_weights)
        assert 1 <= n <= 4 and all(w.shape[0] == self.HID and tuple(w.shape[2:]) == (3, 3) for w in first_weights)
        assert all(tuple(w.shape[1:]) == (self.HID, 1, 1) for w in final_weights)
        dev = first_weights[0].device
        self.n, self.Cin, self.acts = n, first_weights[0].shape[1], list(acts)
        self.p1 = PackedConv(torch.cat([w.detach().float() for w in first_weights], 0).contiguous(), None, None, stride=1,
                             pad=1, compute=F16)
        assert self.p1.korder == 1 and self.p1.Kpad == 9 * self.Cin
        self.b1 = torch.cat([b.detach().float() for b in first_biases]).contiguous()
        self.couts = [w.shape[0] for w in final_weights]
        self.w2, self.b2 = [], []
        for w, b in zip(final_weights, final_biases):
            c = w.shape[0]
            wp = torch.zeros(round_up(c, 16), self.HID, dtype=torch.float16, device=dev)
            wp[:c] = w.detach().reshape(c, self.HID).half()
            bp = torch.zeros(round_up(c, 16), dtype=torch.float32, device=dev)
            bp[:c] = b.detach().float()
            self.w2.append(wp)
            self.b2.append(bp)


def heads_fused(x, ph, clamp=(0.0, 1.0), outs=None):
    """x f16 NHWC [B,H,W,Cin] -> list of f32 NHWC maps [B,H,W,round_up(cout,4)], one per head."""
    _require_cuda(x)
    assert x.dtype == torch.float16 and x.shape[3] == ph.Cin
    B, H, W, _ = x.shape
    if outs is None:
        outs = [torch.empty(B, H, W, round_up(c, 4), dtype=torch.float32, device=x.device) for c in ph.couts]
    d = HeadDesc()
    d.nheads, d.B, d.H, d.W, d.Cin, d.in_stride = ph.n, B, H, W, ph.Cin, _nhwc_stride(x)
    for h in range(ph.n):
        d.w2[h], d.b2[h], d.y[h] = ph.w2[h].data_ptr(), ph.b2[h].data_ptr(), outs[h].data_ptr()
        d.y_stride[h], d.cout[h], d.act[h] = _nhwc_stride(outs[h]), ph.couts[h], ph.acts[h]
    d.clamp_lo, d.clamp_hi = clamp
    M = B * H * W
    prof = _Prof(None, M, False, F32, name="head_fused_kernel<128x256,f16>",
                 flops=sum(2.0 * M * PackedHeads.HID * (9 * ph.Cin + c) for c in ph.couts))
    for _ in range(prof.reps()):
        rc = _lib.lib().ctdet_head_fused_fwd(C.byref(d), _ptr(x), _ptr(ph.p1.w), _ptr(ph.b1), _stream())
    _lib.check(rc, "ctdet_head_fused_fwd")
    prof.done()
    return outs


def maxpool3x3s2(x, out=None):
    """F.max_pool2d(x, 3, stride=2, padding=1) on NHWC (BasicStem, resnet.py:341-345)."""
    _require_cuda(x, out)
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, Cc, dtype=x.dtype, device=x.device)
    rc = _lib.lib().ctdet_maxpool3x3s2(_ptr(x), _ptr(out), dt_of(x), B, H, W, Cc, _nhwc_stride(x), _nhwc_stride(out),
                                       _stream())
    _lib.check(rc, "ctdet_maxpool3x3s2")
    return out


def maxpool3x3s2_ceil(x, out=None):
    """nn.MaxPool2d(3, stride=2, ceil_mode=True) on NHWC (VoVNet stages, vovnet.py:291-292)."""
    _require_cuda(x, out)
    B, H, W, Cc = x.shape
    Ho, Wo = -(-(H - 3) // 2) + 1, -(-(W - 3) // 2) + 1
    if (Ho - 1) * 2 >= H:
        Ho -= 1
    if (Wo - 1) * 2 >= W:
        Wo -= 1
    if out is None:
        out = torch.empty(B, Ho, Wo, Cc, dtype=x.dtype, device=x.device)
    assert tuple(out.shape[:3]) == (B, Ho, Wo)
    rc = _lib.lib().ctdet_maxpool3x3s2_ceil(_ptr(x), _ptr(out), dt_of(x), B, H, W, Cc, _nhwc_stride(x), _nhwc_stride(out),
                                            _stream())
    _lib.check(rc, "ctdet_maxpool3x3s2_ceil")
    return out


def finite_flag(*tensors):
    """int32 [1] on the device: 1 if every value of the f32 NHWC maps (channel slices of wider buffers allowed) is finite, else
    0 -- kernels only (no torch reduction: those clear their semaphores with a memset, which a captured step must not hold)"""
    _require_cuda(*tensors)
    flag = torch.ones(1, dtype=torch.int32, device=tensors[0].device)
    for t in tensors:
        assert t.dtype == torch.float32 and t.dim() == 4 and t.stride(3) == 1
        B, H, W, Cc = t.shape
        _lib.check(_lib.lib().ctdet_finite_flag(_ptr(t), B * H * W, Cc, _nhwc_stride(t), _ptr(flag), _stream()), "ctdet_finite_flag")
    return flag


def global_avgpool(x):
    """NHWC x -> f32 [B, C]: the mean over the pixels (nn.AdaptiveAvgPool2d(1) of the eSE module, vovnet.py:203)."""
    _require_cuda(x)
    B, H, W, Cc = x.shape
    out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    rc = _lib.lib().ctdet_global_avgpool(_ptr(x), dt_of(x), B, H * W, Cc, _nhwc_stride(x), _ptr(out), _stream())
    _lib.check(rc, "ctdet_global_avgpool")
    return out


def ese_scale(x, s, identity=None, out=None):
    """y = x * hsigmoid(s[b, c]) (+ identity); s f32 [B, C] (vovnet.py:186-213, 268-271)."""
    _require_cuda(x, s, identity, out)
    B, H, W, Cc = x.shape
    s = s.contiguous()
    assert s.dtype == torch.float32 and tuple(s.shape) == (B, Cc)
    if out is None:
        out = torch.empty(B, H, W, Cc, dtype=x.dtype, device=x.device)
    if identity is not None:
        assert identity.dtype == x.dtype and tuple(identity.shape) == tuple(x.shape)
    rc = _lib.lib().ctdet_ese_scale(_ptr(x), _nhwc_stride(x), _ptr(s), _ptr(identity),
                                    _nhwc_stride(identity) if identity is not None else 0, _ptr(out), _nhwc_stride(out),
                                    dt_of(x), B, H * W, Cc, _stream())
    _lib.check(rc, "ctdet_ese_scale")
    return out


def conv_transpose2d(x, weight, scale, bias, stride, padding, compute, act=ACT_NONE, cache=None):
    """Dense nn.ConvTranspose2d (weight [Cin, Cout, k, k], output_padding 0) + per-channel scale/bias + act on NHWC:
    a stride-1 convolution with the flipped kernel over the zero-stuffed input (pad k-1-p); the kernels read the
    input as zero-stuffed in place (`in_dil`)."""
    _require_cuda(x, weight)
    Cin, Cout, k, k2 = weight.shape
    assert k == k2 and x.shape[3] == Cin
    p = cache.get("p") if cache is not None else None
    if p is None:
        wc = weight.detach().flip(2, 3).permute(1, 0, 2, 3).contiguous()  # [Cout, Cin, k, k]
        p = PackedConv(wc, scale, bias, stride=1, pad=k - 1 - padding, compute=compute, tap_major=True)
        if cache is not None:
            cache["p"] = p
    B, H, W, _ = x.shape
    Ho, Wo = (H - 1) * stride - 2 * padding + k, (W - 1) * stride - 2 * padding + k
    out = torch.empty(B, Ho, Wo, p.Cout_eff, dtype=x.dtype, device=x.device)
    p.in_dil = stride
    return conv2d(x, p, out=out, act=act)


def maxpool2x2(x, out=None):
    _require_cuda(x, out)
    B, H, W, Cc = x.shape
    if out is None:
        out = torch.empty(B, H // 2, W // 2, Cc, dtype=x.dtype, device=x.device)
    rc = _lib.lib().ctdet_maxpool2x2(_ptr(x), _ptr(out), dt_of(x), B, H, W, Cc, _nhwc_stride(x), _nhwc_stride(out),
                                     _stream())
    _lib.check(rc, "ctdet_maxpool2x2")
    return out


def _dw_weight(weight, Cc, f, fresh=False):
    """[C,1,2f,2f] -> f32 [2f,2f,C]; cached on the tensor object itself (an address-keyed cache would go stale
    when the allocator reuses a freed block)."""
    if fresh:
        # training: always re-derive from the live parameter -- a captured training step would otherwise replay with
        # the copy made at capture time
        return weight.detach().reshape(Cc, 2 * f, 2 * f).to(torch.float32).permute(1, 2, 0).contiguous()
    hit = getattr(weight, "_ctdet_dw", None)
    if hit is None or hit[0] != (weight.data_ptr(), weight._version):
        packed = weight.detach().reshape(Cc, 2 * f, 2 * f).to(torch.float32).permute(1, 2, 0).contiguous()
        hit = ((weight.data_ptr(), weight._version), packed)
        try:
            weight._ctdet_dw = hit
        except AttributeError:
            pass
    return hit[1]


def dwconvT_add(x, weight, f, skip=None, out=None, fresh_weight=False, prepared=None):
    """ConvTranspose2d(C,C,2f,stride=f,padding=f//2,groups=C)(x) + skip; weight f32 [C,1,2f,2f] or [C,2f,2f];
    prepared: its [2f,2f,C] f32 form if the caller already made it"""
    _require_cuda(x, weight, skip, out)
    B, H, W, Cc = x.shape
    w = prepared if prepared is not None else _dw_weight(weight, Cc, f, fresh_weight)
    if out is None:
        out = torch.empty(B, H * f, W * f, Cc, dtype=x.dtype, device=x.device)
    rc = _lib.lib().ctdet_dwconvT_add(_ptr(x), _ptr(w), _ptr(skip), _ptr(out), dt_of(x), B, H, W, Cc, f,
                                      _nhwc_stride(x), _nhwc_stride(skip) if skip is not None else 0,
                                      _nhwc_stride(out), _stream())
    _lib.check(rc, "ctdet_dwconvT_add")
    return out


class DecodeWorkspace:
    """candidate lists of ctdet_decode for a (B, H, W, C, K) problem"""

    def __init__(self, B, H, W, Cc, K, device):
        n = _lib.lib().ctdet_decode_workspace_bytes(B, H, W, Cc, K)
        self.key = (B, H, W, Cc, K)
        self.buf = torch.empty(n // 4, dtype=torch.int32, device=device)


SIGMOID_CLAMP_FLOOR = 1e-4     # `_sigmoid`'s lower clamp (centernet.py:13-15); the head kernels' epilogue writes exactly this f32


def decode(heat, wh, reg, K, down_ratio, workspace=None, check_status=False, heat_floor=0.0):
    """Batched ctdet_decode. heat f32 NHWC [B,H,W,C] (may be the channel slice [..., :C] of a wider buffer: padded head
    outputs are decoded in place); wh/reg f32 NHWC (2 channels, may be slices).  heat_floor: a lower bound of the positive
    heat values the caller vouches for (SIGMOID_CLAMP_FLOOR for a clamped map; same results, the background plateau is
    skipped).
    Returns boxes [B,K,4], scores [B,K], classes [B,K] (int32), inds [B,K] (int32)."""
    _require_cuda(heat, wh, reg)
    assert heat.dtype == torch.float32
    B, H, W, Cc = heat.shape
    if workspace is None or workspace.key != (B, H, W, Cc, K):
        workspace = DecodeWorkspace(B, H, W, Cc, K, heat.device)
    dev = heat.device
    boxes = torch.empty(B, K, 4, dtype=torch.float32, device=dev)
    scores = torch.empty(B, K, dtype=torch.float32, device=dev)
    classes = torch.empty(B, K, dtype=torch.int32, device=dev)
    inds = torch.empty(B, K, dtype=torch.int32, device=dev)
    with prof_region("decode", nbytes=float(B * H * W * Cc * 4), info=f"{B}x{H}x{W}x{Cc} K={K}"):
        rc = _lib.lib().ctdet_decode(_ptr(heat), _nhwc_stride(heat), _ptr(wh), _nhwc_stride(wh), _ptr(reg),
                                     _nhwc_stride(reg) if reg is not None else 0, B, H, W, Cc, K, float(down_ratio),
                                     float(heat_floor), _ptr(workspace.buf), _ptr(boxes), _ptr(scores), _ptr(classes), _ptr(inds), _stream())
    _lib.check(rc, "ctdet_decode")
    if check_status:
        _lib.check(_lib.lib().ctdet_decode_status(_ptr(workspace.buf), B, H, W, Cc, K, _stream()), "ctdet_decode_status")
    return boxes, scores, classes, inds


def postprocess(boxes, scores, classes, max_det, score_thresh, img_params):
    """Batched threshold + rescale + clip + drop-empty, order preserving. img_params f32 [B,4] on device
    = (scale_x, scale_y, out_w, out_h).  Returns compacted (boxes, scores, classes, counts[B] i32)."""
    _require_cuda(boxes, scores, classes, img_params)
    B, K = scores.shape
    ob, os_, oc = torch.empty_like(boxes), torch.empty_like(scores), torch.empty_like(classes)
    counts = torch.empty(B, dtype=torch.int32, device=boxes.device)
    rc = _lib.lib().ctdet_postprocess(_ptr(boxes), _ptr(scores), _ptr(classes), B, K, int(max_det), float(score_thresh),
                                      _ptr(img_params), _ptr(ob), _ptr(os_), _ptr(oc), _ptr(counts), _stream())
    _lib.check(rc, "ctdet_postprocess")
    return ob, os_, oc, counts


def gaussian_targets(boxes, classes, counts, H, W, num_classes, hm=None):
    """Batched gen_heatmap. boxes f32 [B,Nmax,4], classes i64 [B,Nmax], counts i32 [B].
    Returns dict(hm [B,H,W,C] NHWC f32, wh [B,128,2], reg [B,128,2], ind [B,128] i64, reg_mask [B,128] u8)."""
    _require_cuda(boxes, classes, counts)
    B, Nmax, _ = boxes.shape
    dev = boxes.device
    assert boxes.dtype == torch.float32 and classes.dtype == torch.int64 and counts.dtype == torch.int32
    boxes, classes = boxes.contiguous(), classes.contiguous()
    if hm is None:
        hm = torch.empty(B, H, W, num_classes, dtype=torch.float32, device=dev)
    wh = torch.empty(B, 128, 2, dtype=torch.float32, device=dev)
    reg = torch.empty(B, 128, 2, dtype=torch.float32, device=dev)
    ind = torch.empty(B, 128, dtype=torch.int64, device=dev)
    reg_mask = torch.empty(B, 128, dtype=torch.uint8, device=dev)
    rc = _lib.lib().ctdet_gaussian_targets(_ptr(boxes), _ptr(classes), _ptr(counts), B, Nmax, H, W, num_classes,
                                           _ptr(hm), _ptr(wh), _ptr(reg), _ptr(ind), _ptr(reg_mask), _stream())
    _lib.check(rc, "ctdet_gaussian_targets")
    return {"hm": hm, "wh": wh, "reg": reg, "ind": ind, "reg_mask": reg_mask}


def gaussian_radius(hw_pairs):
    """hw_pairs i32 [n,2] (h, w) on device -> (radius f64 [n], int radius i32 [n])."""
    _require_cuda(hw_pairs)
    n = hw_pairs.shape[0]
    r = torch.empty(n, dtype=torch.float64, device=hw_pairs.device)
    ri = torch.empty(n, dtype=torch.int32, device=hw_pairs.device)
    rc = _lib.lib().ctdet_gaussian_radius(_ptr(hw_pairs.contiguous()), n, _ptr(r), _ptr(ri), _stream())
    _lib.check(rc, "ctdet_gaussian_radius")
    return r, ri


def focal_loss(logits, gt, alpha, want_grad=True, grad_scale=1.0):
    """_neg_loss on sigmoid+clamp of logits. logits/gt f32 NHWC [B,H,W,C]; alpha f32 [C].
    Returns (loss [1], stats [4] = pos, neg, num_pos, 1/num_pos, grad or None)."""
    _require_cuda(logits, gt, alpha)
    assert logits.is_contiguous() and gt.is_contiguous() and logits.shape == gt.shape
    B, H, W, Cc = logits.shape
    dev = logits.device
    ws = torch.empty(_lib.lib().ctdet_focal_loss_workspace_bytes(logits.numel()) // 4, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    stats = torch.empty(4, dtype=torch.float32, device=dev)
    grad = torch.empty_like(logits) if want_grad else None
    rc = _lib.lib().ctdet_focal_loss(_ptr(logits), _ptr(gt), _ptr(alpha), B, H, W, Cc, float(grad_scale), _ptr(ws),
                                     _ptr(loss), _ptr(stats), _ptr(grad), _stream())
    _lib.check(rc, "ctdet_focal_loss")
    return loss, stats, grad


def reg_l1_loss(pred, mask, ind, target, want_grad=True, grad_scale=1.0, grad=None):
    """RegL1Loss. pred f32 NHWC [B,H,W,2] (may be a channel slice); mask u8 [B,N]; ind i64 [B,N]; target [B,N,2]."""
    _require_cuda(pred, mask, ind, target)
    B, H, W, _ = pred.shape
    N = mask.shape[1]
    dev = pred.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    if want_grad and grad is None:
        grad = torch.zeros(B, H, W, 2, dtype=torch.float32, device=dev)
    rc = _lib.lib().ctdet_reg_l1_loss(_ptr(pred), _nhwc_stride(pred), _ptr(mask.contiguous()), _ptr(ind.contiguous()),
                                      _ptr(target.contiguous()), B, N, H * W, float(grad_scale), _ptr(loss),
                                      _ptr(grad) if want_grad else C.c_void_p(0),
                                      _nhwc_stride(grad) if want_grad else 0, _stream())
    _lib.check(rc, "ctdet_reg_l1_loss")
    return loss, grad


def sgd_momentum_(param, grad, buf, lr_dev, momentum, weight_decay, first_step):
    """In-place fused SGD step on flat f32 tensors; lr_dev is a 1-element device tensor."""
    _require_cuda(param, grad, buf, lr_dev)
    assert param.is_contiguous() and grad.is_contiguous() and buf.is_contiguous()
    rc = _lib.lib().ctdet_sgd_momentum(_ptr(param), _ptr(grad), _ptr(buf), param.numel(), _ptr(lr_dev),
                                       float(momentum), float(weight_decay), int(bool(first_step)), _stream())
    _lib.check(rc, "ctdet_sgd_momentum")


def sgd_momentum_runs_(param, grad, buf, run_end, run_lr_index, run_wd, lr_table, momentum, first_step):
    """one launch over a flat buffer of consecutive hyper-parameter runs (see ctdet_sgd_momentum_runs)"""
    _require_cuda(param, grad, buf, run_end, run_lr_index, run_wd, lr_table)
    assert param.is_contiguous() and grad.is_contiguous() and buf.is_contiguous()
    assert run_end.dtype == torch.int64 and run_lr_index.dtype == torch.int32
    rc = _lib.lib().ctdet_sgd_momentum_runs(_ptr(param), _ptr(grad), _ptr(buf), param.numel(), _ptr(run_end),
                                            _ptr(run_lr_index), _ptr(run_wd), _ptr(lr_table), run_end.numel(),
                                            float(momentum), int(bool(first_step)), _stream())
    _lib.check(rc, "ctdet_sgd_momentum_runs")
