#!/usr/bin/env python
"""Known-answer vectors for modulated deformable convolution (DCNv2), derived BY HAND from the cited lines of the
reference's CUDA kernels -- the reference has no CPU path and no test for this operator, so nothing it
holds can pin the oracle; these vectors are the only independent anchor: each expected output is computed with plain
`torch.nn.functional.conv2d` on explicitly shifted / averaged inputs, never with a deformable-conv implementation.
They do NOT formally pin the oracle (parity for DCNv2 stays "unpinned").

Derivations (detectron2/layers/csrc/deformable/deform_conv_cuda_kernel.cu):
  * sampling position of tap (i, j) at output (h, w), 3x3 / stride 1 / pad 1 / dilation 1:
      h_im = h - 1 + i + offset[2k],  w_im = w - 1 + j + offset[2k+1],  k = 3i + j                      (:838-851)
    value = mask[k] * bilinear(x, h_im, w_im) if -1 < h_im < H and -1 < w_im < W else 0                  (:852-862)
    bilinear (:666-699): corners floor(h_im), floor(h_im)+1 (same in w), each corner 0 when outside the image.
  * zero offsets, unit mask: every tap samples an integer position inside or exactly ON the zero frame
    (h_im = -1 -> excluded by the strict test, which is also what zero padding gives): out = conv2d(x, w, padding=1).
  * the same integer offset (dy, dx) for every tap and pixel: the tap reads x[h-1+i+dy, w-1+j+dx], 0 outside:
    out[h, w] = sum_ij w[i, j] * X[h-1+i+dy, w-1+j+dx], X = x extended by zeros in every direction (a position with
    h_im = -1 or h_im >= H is excluded by :852, which is the same zero) -- `conv_shifted` below.  (NOT conv2d(shift(x),
    padding=1): the padding frame of a shifted image sits in the wrong place.)
  * offset (dy + 1/2, dx): lh = 1/2, lw = 0 -> value = (x[floor] + x[floor + 1]) / 2 with out-of-image corners 0:
    out = (conv_shifted(dy, dx) + conv_shifted(dy + 1, dx)) / 2.   Likewise in w.  (-1 < h_im = -1/2: only the
    lower corner exists: x[0] / 2 -- the zero-filled shift gives the same.)
  * masks: out = sum_k mask_k * (tap k's contribution): per-tap convs with a one-hot 3x3 kernel.
  * `-0.0` offsets equal `+0.0` offsets (floor(-0.0) = -0 -> corner index 0).
  * h_im in (H-1, H): value = (H - h_im) * x[H-1] (upper corner guarded out, :689-693); h_im = -1 exactly: 0 (:852).
Backward (deform_conv_cuda.cu:929-1129) for the integer-shift cases, unit mask: the operator is the linear map
  out = conv_shifted(x, w, dy, dx): grad_input / grad_weight / grad_bias are torch autograd's of that plain convolution.  For the half-pixel case additionally d out / d offset_h = mask * sum_c col-grad *
  (x[floor+1] - x[floor]) (:1041-1049 with lw = 0: dmcn_get_coordinate_weight reduces to v3 - v1).

Run:  python tests/golden/make_dcn_known_answers.py   (writes tests/golden/g11_dcn_known_answers.npz)
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))


PADX = 4


def conv_shifted(x, w, dy, dx, bias=None):
    """out[h, w'] = sum_ij w[i, j] * X[h - 1 + i + dy, w' - 1 + j + dx] with X = x extended by zeros in every direction
    (dy, dx integers, |.| < PADX): a valid convolution over the zero-extended input, read at the shifted window"""
    B, C, H, W = x.shape
    big = F.conv2d(F.pad(x, (PADX, PADX, PADX, PADX)), w, bias)          # [B, Co, H + 2*PADX - 2, W + 2*PADX - 2]
    y0, x0 = PADX - 1 + dy, PADX - 1 + dx
    return big[:, :, y0:y0 + H, x0:x0 + W]


def shift(x, dy, dx):
    """shift(x)[h, w] = x[h + dy, w + dx], zero outside"""
    B, C, H, W = x.shape
    out = torch.zeros_like(x)
    ys, xs = slice(max(0, -dy), min(H, H - dy)), slice(max(0, -dx), min(W, W - dx))
    yd, xd = slice(max(0, dy), min(H, H + dy)), slice(max(0, dx), min(W, W + dx))
    out[:, :, ys, xs] = x[:, :, yd, xd]
    return out


def main():
    g = torch.Generator().manual_seed(20261004)
    B, C, Co, H, W = 2, 16, 8, 6, 7
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(Co, C, 3, 3, generator=g, dtype=torch.float64) / 12
    bias = torch.randn(Co, generator=g, dtype=torch.float64)
    go = torch.randn(B, Co, H, W, generator=g, dtype=torch.float64)
    out = {"x": x.numpy(), "w": w.numpy(), "bias": bias.numpy(), "grad_out": go.numpy()}
    ones = torch.ones(B, 9, H, W, dtype=torch.float64)

    def case(name, off, mask, expect, grads=None):
        out[f"{name}_offset"], out[f"{name}_mask"], out[f"{name}_out"] = off.numpy(), mask.numpy(), expect.numpy()
        for k, v in (grads or {}).items():
            out[f"{name}_{k}"] = v.numpy()

    def conv_grads(xs, dy, dx):
        """gradients of out = conv_shifted(x, w, dy, dx) + bias"""
        xs = xs.clone().requires_grad_(True)
        ww = w.clone().requires_grad_(True)
        bb = bias.clone().requires_grad_(True)
        (conv_shifted(xs, ww, dy, dx, bb) * go).sum().backward()
        return {"grad_input": xs.grad, "grad_weight": ww.grad, "grad_bias": bb.grad}

    zero = torch.zeros(B, 18, H, W, dtype=torch.float64)
    case("zero", zero, ones, F.conv2d(x, w, bias, padding=1), conv_grads(x, 0, 0))
    case("negzero", -zero, ones, F.conv2d(x, w, bias, padding=1))
    for dy, dx in ((1, 0), (0, -2), (-1, 3), (2, 2)):
        off = zero.clone()
        off[:, 0::2] = dy
        off[:, 1::2] = dx
        case(f"int_{dy}_{dx}".replace("-", "m"), off, ones, conv_shifted(x, w, dy, dx, bias),
             conv_grads(x, dy, dx))
    # half-pixel in h
    off = zero.clone()
    off[:, 0::2] = 0.5
    half = 0.5 * (conv_shifted(x, w, 0, 0) + conv_shifted(x, w, 1, 0)) + bias.view(1, -1, 1, 1)
    xs = x.clone().requires_grad_(True)
    (((0.5 * (conv_shifted(xs, w, 0, 0) + conv_shifted(xs, w, 1, 0))) * go).sum()).backward()
    # d out / d offset_h of tap k at (h, w) = sum_co go * sum_c w[co, c, k] * (X[c, floor+1] - X[c, floor])
    goff = torch.zeros(B, 18, H, W, dtype=torch.float64)
    for k in range(9):
        onehot = torch.zeros_like(w)
        onehot[:, :, k // 3, k % 3] = w[:, :, k // 3, k % 3]
        goff[:, 2 * k] = ((conv_shifted(x, onehot, 1, 0) - conv_shifted(x, onehot, 0, 0)) * go).sum(1)
    case("half_h", off, ones, half, {"grad_input": xs.grad, "grad_offset_h": goff[:, 0::2]})
    # half-pixel in w with per-tap masks
    off = zero.clone()
    off[:, 1::2] = -0.5
    mk = torch.linspace(0.1, 0.9, 9, dtype=torch.float64).view(1, 9, 1, 1).expand(B, 9, H, W).contiguous()
    exp = bias.view(1, -1, 1, 1).expand(B, Co, H, W).clone()
    for k in range(9):
        onehot = torch.zeros_like(w)
        onehot[:, :, k // 3, k % 3] = w[:, :, k // 3, k % 3]
        exp = exp + mk[0, k, 0, 0] * 0.5 * (conv_shifted(x, onehot, 0, -1) + conv_shifted(x, onehot, 0, 0))
    case("half_w_masked", off, mk, exp)
    # borders: the centre tap (k = 4) of the first / last row pushed to h_im = -1 exactly and to H - 1/4
    off = zero.clone()
    off[:, 8, 0, :] = -1.0            # h_im = 0 - 1 + 1 - 1 = -1  -> contributes 0
    off[:, 8, H - 1, :] = 0.75        # h_im = H - 1 + 0.75        -> 0.25 * x[H-1]
    onehot = torch.zeros_like(w)
    onehot[:, :, 1, 1] = w[:, :, 1, 1]
    centre = F.conv2d(x, onehot, None, padding=1)
    exp = F.conv2d(x, w, bias, padding=1)
    exp[:, :, 0, :] -= centre[:, :, 0, :]
    exp[:, :, H - 1, :] -= 0.75 * centre[:, :, H - 1, :]
    case("border", off, ones, exp)
    np.savez_compressed(os.path.join(HERE, "g11_dcn_known_answers.npz"), **out)
    print("wrote g11_dcn_known_answers.npz:", sorted(out))


if __name__ == "__main__":
    main()
