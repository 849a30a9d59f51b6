"""GPU parity: the bf16x3 MFMA convolution (csrc/dense_conv.hip) vs fp32 torch convolutions -- features
within 1e-4 relative (BASELINE north_star) -- standalone, as the whole RPN + heads stack, and end to end."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import assert_features_close, randomize_bn
from vision3d_amd import synth
from vision3d_amd.core.config import second_car_cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,h,w,cin,cout,k", [(2, 37, 29, 64, 128, 3), (1, 20, 16, 128, 128, 3), (1, 33, 50, 128, 128, 1),
                                             (3, 9, 7, 32, 256, 3), (1, 40, 31, 128, 16, 1), (1, 61, 53, 256, 128, 1),
                                             (2, 24, 12, 128, 256, 3), (1, 30, 44, 192, 128, 3), (1, 19, 23, 64, 64, 3), (1, 25, 17, 96, 128, 1)])
def test_conv_matches_torch_fp32(b, h, w, cin, cout, k):
    """The shape list reaches all three kernels through the dispatch of v3d_conv2d_nhwc_bf16x3: the 144-pixel kernel (Cin >= 64,
    Cout > 32, W >= 16), the 64-pixel kernel (Cin = 32, Cout <= 32, or W < 16: the (2, 24, 12, 128, 256) and (3, 9, 7, 32, 256)
    rows) and the streaming 1x1 head kernel (128 -> 16)."""
    from vision3d_amd.runtime import conv2d_split, pack_conv_weight, to_split_nhwc
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    x = torch.randn(b, cin, h, w, generator=g).cuda()
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).cuda()
    scale = (torch.rand(cout, generator=g) + 0.5).cuda()
    bias = torch.randn(cout, generator=g).cuda() * 0.2
    ref = F.relu(F.conv2d(x.double(), (wt * scale.view(-1, 1, 1, 1)).double(), bias.double(), padding=k // 2)).float()
    hi, lo = to_split_nhwc(x)
    img = pack_conv_weight(wt, scale)
    (y_hi, y_lo), y = conv2d_split(hi, lo, img, bias, True, cin, cout, k, out_split=(cout % 8 == 0), out_nchw=True)
    assert_features_close(y.cpu().numpy(), ref.cpu().numpy(), f"conv {cin}->{cout} k{k} fp32 NCHW output")
    if y_hi is not None:  # split planes: hi + lo reproduces the fp32 value to ~2^-17
        back = (y_hi.view(torch.bfloat16).float() + y_lo.view(torch.bfloat16).float()).permute(0, 3, 1, 2)
        assert_features_close(back.cpu().numpy(), ref.cpu().numpy(), "split planes output")
    # without bias / relu
    _, y2 = conv2d_split(hi, lo, pack_conv_weight(wt), None, False, cin, cout, k, out_split=False, out_nchw=True)
    assert_features_close(y2.cpu().numpy(), F.conv2d(x, wt, None, padding=k // 2).cpu().numpy(), "plain conv")


def build_model(seed=0):
    from vision3d_amd.detector import Second
    torch.manual_seed(seed)
    model = Second(second_car_cfg())
    randomize_bn(model, seed)
    with torch.no_grad():
        model.head.conv_cls.weight.normal_(0, 0.05)
        model.head.conv_reg.weight.normal_(0, 0.02)
    return model.cuda().eval()


def test_dense_head_stack_matches_torch():
    """Real BEV map -> 7 RPN convs + heads: MFMA path vs torch fp32 modules."""
    from vision3d_amd.runtime import to_split_nhwc
    model = build_model(1)
    clouds = [torch.from_numpy(synth.make_cloud(s)).cuda() for s in (0, 1)]
    with torch.no_grad():
        bev = model.bev_from_points(clouds)
        ref_feat = model.rpn.up_block(model.rpn.down_block(bev))
        ref_cls, ref_reg = model.head(ref_feat)
        maps, feats = model.dense_plan().forward(*to_split_nhwc(bev), want_features=True)
        cls_map, reg_map = model.head.maps_from_fused(maps)
        cls2, reg2 = model.head_maps_from_points(clouds)   # + the split densify path
    assert_features_close(feats.cpu().numpy(), ref_feat.cpu().numpy(), "RPN features")
    assert_features_close(cls_map.cpu().numpy(), ref_cls.cpu().numpy(), "cls map")
    assert_features_close(reg_map.cpu().numpy(), ref_reg.cpu().numpy(), "reg map")
    np.testing.assert_array_equal(cls2.cpu().numpy(), cls_map.cpu().numpy())
    np.testing.assert_array_equal(reg2.cpu().numpy(), reg_map.cpu().numpy())


def test_native_path_matches_cpu_oracle_end_to_end():
    from gpu_util import numpy_state_dict
    from oracle import second_cpu
    cfg = second_car_cfg()
    model = build_model(2)
    cloud = synth.make_cloud(3)
    with torch.no_grad():
        cls_map, reg_map = model.head_maps_from_points([torch.from_numpy(cloud).cuda()])
    ref = second_cpu.second_forward(numpy_state_dict(model), [cloud], cfg.VOXEL_SIZE, cfg.GRID_BOUNDS, cfg.MAX_OCCUPANCY,
                                    cfg.MAX_VOXELS)
    assert_features_close(cls_map.cpu().numpy().reshape(ref["cls"].shape), ref["cls"], "P_cls native vs oracle")
    reg = reg_map.permute(0, 1, 5, 2, 3, 4).reshape(ref["reg"].shape)
    assert_features_close(reg.cpu().numpy(), ref["reg"], "P_reg native vs oracle")


def test_inference_points_paths_agree():
    from vision3d_amd.core import AnchorGenerator
    model = build_model(3)
    anchors = AnchorGenerator(second_car_cfg()).anchors.cuda()
    clouds = [torch.from_numpy(synth.make_cloud(5)).cuda()]
    with torch.no_grad():
        a = model.inference_points(clouds, anchors, dense="mfma")
        b = model.inference_points(clouds, anchors, dense="torch")
    assert a[0].shape[1] == 7 and abs(len(a[0]) - len(b[0])) <= max(2, len(b[0]) // 10)
