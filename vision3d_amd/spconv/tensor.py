"""SparseConvTensor: active-site features + (b,z,y,x) indices + cached rulebooks."""
import torch

from .. import _lib as L


def _dense_scatter(out, idx, features):
    out = out.permute(0, 2, 3, 4, 1).contiguous()          # (B, D, H, W, C): one row per site
    out = out.index_put((idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]), features)
    return out.permute(0, 4, 1, 2, 3).contiguous()


class Rulebook(object):
    """Output-stationary neighbour table of one sparse convolution geometry.

    nbr (K, cap) int32, k-major: nbr[k, o] = input row feeding output row o through kernel offset k, or
    -1.  `n_dev` is the live number of output rows (device int32), `n` the same on the host."""

    def __init__(self, nbr, cap, n, n_dev, out_indices, out_shape):
        self.nbr, self.cap, self.n, self.n_dev = nbr, cap, n, n_dev
        self.out_indices, self.out_shape = out_indices, out_shape


class SparseConvTensor(object):

    def __init__(self, features, indices, spatial_shape, batch_size):
        """features (N, C) float32, indices (N, 4) int32 rows (batch, z, y, x), spatial_shape [D, H, W]."""
        L.require_gpu("SparseConvTensor", features, indices)
        if indices.dtype != torch.int32:
            raise RuntimeError("SparseConvTensor: indices must be int32 (reference passes coordinates.int())")
        self.features = features
        self.indices = indices.contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self._n_dev = None

    @property
    def n_dev(self):
        if self._n_dev is None:
            self._n_dev = torch.tensor([self.features.shape[0]], dtype=torch.int32, device=self.features.device)
        return self._n_dev

    @property
    def spatial_size(self):
        d, h, w = self.spatial_shape
        return d * h * w

    def find_indice_pair(self, key):
        return None if key is None else self.indice_dict.get(key)

    def replace_feature(self, features):
        out = SparseConvTensor.__new__(SparseConvTensor)
        out.__dict__.update(self.__dict__)
        out.features = features
        return out

    def dense(self, channels_first=True):
        """Scatter into zeros: (B, C, D, H, W) (or (B, D, H, W, C)); detector/sparse_cnn.py:130."""
        if torch.is_grad_enabled() and self.features.requires_grad:
            # training: differentiable scatter (gradient = gather of the dense gradient at the active sites)
            n, c = self.features.shape
            d, h, w = self.spatial_shape
            out = self.features.new_zeros((self.batch_size, c, d, h, w))
            out = _dense_scatter(out, self.indices.long(), self.features)
            return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()
        feat = L.as_f32("dense", self.features)
        n, c = feat.shape
        d, h, w = self.spatial_shape
        out = torch.empty((self.batch_size, c, d, h, w), dtype=torch.float32, device=feat.device)
        if n == 0:
            return out.zero_() if channels_first else out.zero_().permute(0, 2, 3, 4, 1).contiguous()
        with L.device_guard(feat.device):
            L.check(L.lib().v3d_densify(L.ptr(feat), L.ptr(self.indices), L.ptr(self.n_dev), n, self.batch_size, c,
                                        L.host_i32(self.spatial_shape), L.ptr(out), L.stream_ptr()), "densify")
        return out if channels_first else out.permute(0, 2, 3, 4, 1).contiguous()
