"""aten_backend.py -- TEST INFRASTRUCTURE ONLY.  The oracle's operator set on PyTorch's CPU (ATen / oneDNN)
kernels: with ``hdrtvnet_oracle.use_backend("aten")`` the same numpy graphs of AGCM / LE / HG run as a plain
eager-PyTorch fp32 network, which is what the reference executes on a CPU (SURVEY.md 8d "CPU baseline":
reference CPU-eager semantics rebuilt from the weight pack).  Two uses:
  * bench.py's ``cpu_baseline`` leg times this backend (threads = ``torch.get_num_threads()``),
  * tests cross-check the plain-C operators against ATen's on the golden inputs.
The reference's own Python never travels to the GPU box; this file only needs ``torch``."""
from __future__ import annotations

import warnings

import numpy as np
import torch
import torch.nn.functional as F

warnings.filterwarnings("ignore", message="The given NumPy array is not writable")   # weight views of the pack


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def _n(t):
    return t.contiguous().numpy()


@torch.inference_mode()
def conv2d(x, w, b=None, stride=1, pad=0):
    return _n(F.conv2d(_t(x)[None], _t(w), None if b is None else _t(b), stride=stride, padding=pad)[0])


@torch.inference_mode()
def avgpool3s2p1(x):
    return _n(F.avg_pool2d(_t(x)[None], 3, stride=2, padding=1, count_include_pad=True)[0])


@torch.inference_mode()
def instnorm(x, gamma, beta, eps=1e-5):
    return _n(F.instance_norm(_t(x)[None], weight=_t(gamma), bias=_t(beta), eps=eps)[0])


@torch.inference_mode()
def batchnorm(x, gamma, beta, mean, var, eps=1e-5):
    return _n(F.batch_norm(_t(x)[None], _t(mean), _t(var), _t(gamma), _t(beta), training=False, eps=eps)[0])


@torch.inference_mode()
def maxpool2(x):
    return _n(F.max_pool2d(_t(x)[None], 2)[0])


@torch.inference_mode()
def pixelshuffle2(x):
    return _n(F.pixel_shuffle(_t(x)[None], 2)[0])


@torch.inference_mode()
def bicubic_aa_quarter(x):
    return _n(F.interpolate(_t(x)[None], scale_factor=0.25, mode="bicubic", align_corners=False,
                            recompute_scale_factor=False, antialias=True)[0])


@torch.inference_mode()
def relu(x):
    return _n(F.relu(_t(x)))


@torch.inference_mode()
def leaky(x, slope):
    return _n(F.leaky_relu(_t(x), slope))


OPS = {"relu": relu, "leaky": leaky, "conv2d": conv2d, "avgpool3s2p1": avgpool3s2p1, "instnorm": instnorm, "batchnorm": batchnorm,
       "maxpool2": maxpool2, "pixelshuffle2": pixelshuffle2, "bicubic_aa_quarter": bicubic_aa_quarter}
