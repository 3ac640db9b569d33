"""AssociationHead -- counterpart of /root/reference/dcnn/networks/association_head.py:5-35.

Same constructor and ``forward`` contract (``(N, C, roi, roi)`` -> ``(N, embedding_dim)`` unit
vectors: flatten, ``Linear``, ``F.normalize(p=2, dim=1)``), computed by the HIP library: the
``Linear`` runs as an ``roi x roi`` valid convolution on the MFMA implicit-GEMM kernel and
the normalisation as a wave reduction.  On the tracker's per-frame path the head is fused
into the detector context instead (``TrackRCNN.attach_association_head``), so ``forward``
here serves stand-alone use and the parity tests.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib


class _FC:
    def __init__(self, out_f, in_f):
        self.weight = torch.zeros(out_f, in_f)
        self.bias = torch.zeros(out_f)


class AssociationHead:
    def __init__(self, roi_size, input_depth, embedding_dim=128):
        self.embedding_dim = embedding_dim
        self.roi_size = roi_size
        self.input_depth = input_depth
        self.fc = _FC(embedding_dim, input_depth * roi_size * roi_size)
        self._packed = None
        self._device = torch.device("cpu")

    # nn.Module-like surface used by the reference (rcnn_tracker.py:55-57)
    def state_dict(self):
        return {"fc.weight": self.fc.weight, "fc.bias": self.fc.bias}

    def load_state_dict(self, sd):
        w, b = sd["fc.weight"], sd["fc.bias"]
        if tuple(w.shape) != tuple(self.fc.weight.shape) or tuple(b.shape) != tuple(self.fc.bias.shape):
            raise RuntimeError("size mismatch for fc: %s vs %s" % (tuple(w.shape), tuple(self.fc.weight.shape)))
        self.fc.weight = w.detach().to(torch.float32).cpu().contiguous()
        self.fc.bias = b.detach().to(torch.float32).cpu().contiguous()
        self._packed = None

    def to(self, device):
        self._device = torch.device(device)
        return self

    def eval(self):
        return self

    def num_flat_features(self, x):
        n = 1
        for s in x.size()[1:]:
            n *= s
        return n

    def _desc(self, n):
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.Cin = n, self.roi_size, self.roi_size, self.input_depth
        d.Cout, d.KH, d.KW, d.stride, d.pad = self.embedding_dim, self.roi_size, self.roi_size, 1, 0
        d.relu, d.res_mode, d.cfg, d.splitk = 0, 0, -1, 0
        return d

    def forward(self, x):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.ApseError("AssociationHead.forward needs a GPU tensor (no CPU fallback)")
        n = x.shape[0]
        if n == 0:
            return torch.zeros((0, self.embedding_dim), device=x.device)
        d = self._desc(n)
        if self._packed is None or self._packed[0].device != x.device:
            packed = np.zeros(lib.apse_conv_packed_elems(C.byref(d)), np.float32)
            w = np.ascontiguousarray(self.fc.weight.numpy())       # [out][c*h*w] == OIHW
            _lib.check(lib.apse_conv_pack_weight(C.byref(d), _lib.ptr(w), self.input_depth, None, _lib.ptr(packed)), None,
                       "apse_conv_pack_weight")
            bias = np.zeros(((self.embedding_dim + 127) // 128) * 128, np.float32)
            bias[: self.embedding_dim] = self.fc.bias.numpy()
            self._packed = (torch.from_numpy(packed).to(x.device), torch.from_numpy(bias).to(x.device))
        xn = x.to(torch.float32).permute(0, 2, 3, 1).contiguous()            # NCHW -> NHWC (plumbing)
        y = torch.empty((n, self.embedding_dim), device=x.device, dtype=torch.float32)
        steps = self.roi_size * ((self.roi_size * self.input_depth + 31) // 32)
        ws = torch.empty((64 * n * self.embedding_dim,), device=x.device, dtype=torch.float32)
        _lib.check(lib.apse_conv2d(C.byref(d), _lib.ptr(xn), _lib.ptr(self._packed[0]), _lib.ptr(self._packed[1]), None,
                                   _lib.ptr(y), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), None, "apse_conv2d")
        out = torch.empty_like(y)
        _lib.check(lib.apse_l2_normalize(_lib.ptr(y), _lib.ptr(out), n, self.embedding_dim, _lib.stream_ptr()), None,
                   "apse_l2_normalize")
        return out

    __call__ = forward
