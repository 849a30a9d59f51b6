"""SECOND detector (interface of vision3d/detector/second.py:10-94): vfe -> sparse cnn -> dense RPN -> head.

`Second(cfg).forward(item)` adds P_cls (B,1,2,200,176) / P_reg (B,1,2,200,176,7);
`.inference(item)` returns (boxes, batch_idx, class_idx, scores).  `item` is the dict produced by
vision3d_amd.core.Preprocessor (same keys as the reference's).  Parameter tree = the reference's
(vfe | cnn.blocks.* | rpn.down_block.* / rpn.up_block.* | head.conv_cls / head.conv_reg).
"""
from torch import nn
from torch.nn.modules.batchnorm import _BatchNorm

from .. import spconv
from .layers import VoxelFeatureExtractor
from .proposal import ProposalLayer
from .sparse_cnn import SpMiddleFHD


class Middle(SpMiddleFHD):
    """Sparse backbone straight to the BEV map (skips the metric-coordinate outputs)."""

    def forward(self, features, coordinates, batch_size):
        x = spconv.SparseConvTensor(features, coordinates.int(), self.grid_shape, batch_size)
        return self.to_bev(self.blocks(x))


class RPN(nn.Module):
    """One-stage dense RPN: ZeroPad+Conv3x3, `blocks` x Conv3x3 (pad 1), then a `stride`x`stride` conv
    (1x1 at stride 1), every conv bias-free and followed by BatchNorm2d(eps 1e-3, mom 0.01) + ReLU."""

    def __init__(self, C_in=128, C_up=128, C_down=128, blocks=5):
        super().__init__()
        self.down_block, C_in = self._make_down_block(C_in, C_down, blocks)
        self.up_block = self._make_up_block(C_in, C_up)
        self._init_weights()

    @staticmethod
    def _bn(planes):
        return nn.BatchNorm2d(planes, eps=1e-3, momentum=0.01)

    def _make_down_block(self, inplanes, planes, num_blocks, stride=1):
        layers = [nn.ZeroPad2d(1), nn.Conv2d(inplanes, planes, 3, stride=stride, bias=False), self._bn(planes), nn.ReLU()]
        for _ in range(num_blocks):
            layers += [nn.Conv2d(planes, planes, 3, padding=1, bias=False), self._bn(planes), nn.ReLU()]
        return nn.Sequential(*layers), planes

    def _make_up_block(self, inplanes, planes, stride=1):
        return nn.Sequential(nn.Conv2d(inplanes, planes, stride, stride=stride, bias=False), self._bn(planes), nn.ReLU())

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_normal_(m.weight)
            elif isinstance(m, _BatchNorm):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return self.up_block(self.down_block(x))


class Second(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.vfe = VoxelFeatureExtractor()
        self.cnn = Middle(cfg)
        self.rpn = RPN()
        self.head = ProposalLayer(cfg)
        self.cfg = cfg

    def feature_extract(self, item):
        if "voxel_mean" in item:  # fused into the device voxelizer
            features = item["voxel_mean"]
        else:
            features = self.vfe(item["features"], item["occupancy"])
        features = self.cnn(features, item["coordinates"], item["batch_size"])
        return self.rpn(features)

    def forward(self, item):
        scores, boxes = self.head(self.feature_extract(item))
        item.update(dict(P_cls=scores, P_reg=boxes))
        return item

    def inference(self, item):
        return self.head.inference(self.feature_extract(item), item["anchors"])
