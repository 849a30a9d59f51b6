"""CPU: the host-side logic around the native training paths -- which module stacks the dense training plan accepts
(vision3d_amd/dense_train.py) and that ProposalLoss falls back to the torch expressions when the fused maps are not usable
(CPU tensors, unexpected target layouts)."""
import torch
from torch import nn

from vision3d_amd import dense_train
from vision3d_amd.core.config import second_car_cfg
from vision3d_amd.detector import ProposalLoss, Second


def test_rpn_of_second_parses_into_seven_conv_bn_pairs():
    model = Second(second_car_cfg())
    convs, bns, clean = dense_train.rpn_pairs(model.rpn)
    assert clean and len(convs) == len(bns) == 7
    assert [c.kernel_size[0] for c in convs] == [3, 3, 3, 3, 3, 3, 1]
    assert all(c.in_channels == c.out_channels == 128 and c.bias is None for c in convs)


def test_supported_refuses_what_the_kernels_are_not_built_for():
    model = Second(second_car_cfg())
    bev = torch.zeros(1, 128, 8, 16, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert not dense_train.supported(model.rpn, model.head, bev)  # not on a GPU
    fake = bev.to("meta")
    assert not dense_train.supported(model.rpn, model.head, fake)
    # structural checks do not depend on the device: exercise them through a stand-in with is_cuda = True
    class OnGpu(object):
        is_cuda, dtype, shape = True, torch.bfloat16, (1, 128, 8, 16)
        def dim(self): return 4
        def is_contiguous(self, memory_format=None): return True
    assert dense_train.supported(model.rpn, model.head, OnGpu())
    convs, _, _ = dense_train.rpn_pairs(model.rpn)
    convs[2].stride = (2, 2)
    assert not dense_train.supported(model.rpn, model.head, OnGpu())
    convs[2].stride = (1, 1)
    model.rpn.up_block = nn.Sequential(*list(model.rpn.up_block), nn.Sigmoid())  # an op the kernels do not know
    assert not dense_train.supported(model.rpn, model.head, OnGpu())


def test_plan_cache_evicts_the_oldest_geometry(monkeypatch):
    """train_head_maps keeps at most MAX_CACHED_PLANS arenas per model: the eviction runs here with the plan and the autograd node
    stubbed out (the real ones need a GPU)."""
    built = []

    class StubPlan(object):
        def __init__(self, rpn, head, B, H, W, device, precision="bf16"):
            built.append((B, H, W))

        def parameters(self):
            return []

    class StubFunction(object):
        @staticmethod
        def apply(plan, bev, *params):
            return plan

    monkeypatch.setattr(dense_train, "DenseTrainPlan", StubPlan)
    monkeypatch.setattr(dense_train, "DenseTrainFunction", StubFunction)
    cache = {}
    n = dense_train.MAX_CACHED_PLANS
    for b in range(1, n + 3):
        dense_train.train_head_maps(None, None, torch.zeros(b, 128, 4, 4), cache)
        assert len(cache) <= n
    assert len(built) == n + 2
    assert [k[1][0] for k in cache] == list(range(3, n + 3))  # the two oldest geometries were dropped
    first = dense_train.train_head_maps(None, None, torch.zeros(n + 2, 128, 4, 4), cache)
    assert len(built) == n + 2 and first is cache[("cpu", (n + 2, 128, 4, 4), "bf16")]  # a cached geometry is reused


def test_supported_refuses_frozen_batchnorm_and_cumulative_momentum():
    model = Second(second_car_cfg())

    class OnGpu(object):
        is_cuda, dtype, shape = True, torch.bfloat16, (1, 128, 8, 16)
        def dim(self): return 4
        def is_contiguous(self, memory_format=None): return True
    _, bns, _ = dense_train.rpn_pairs(model.rpn)
    assert dense_train.supported(model.rpn, model.head, OnGpu())
    bns[3].eval()
    assert "frozen" in dense_train.why_unsupported(model.rpn, model.head, OnGpu())
    bns[3].train()
    bns[1].momentum = None
    assert "momentum" in dense_train.why_unsupported(model.rpn, model.head, OnGpu())
    bns[1].momentum = 0.01
    bns[0].running_var = bns[0].running_var.double()
    assert "running statistics" in dense_train.why_unsupported(model.rpn, model.head, OnGpu())


def test_forward_drops_fused_maps_of_an_earlier_forward():
    """A re-used item dict must not carry the fused training maps of an earlier forward into ProposalLoss."""
    cfg = second_car_cfg()
    model = Second(cfg)
    stale = torch.zeros(1, 16, 4, 4)

    class Stop(Exception):
        pass

    def boom(item):
        raise Stop()
    model.feature_extract = boom
    model.eval()
    item = {"_head_maps": stale, "features": None}
    try:
        with torch.no_grad():
            model(item)
    except Stop:
        pass
    assert "_head_maps" not in item


def test_proposal_loss_checks_that_fused_maps_belong_to_the_items_outputs():
    cfg = second_car_cfg()
    loss = ProposalLoss(cfg)
    maps, p_cls, p_reg = torch.zeros(1, 16, 4, 4), torch.zeros(1), torch.zeros(1)
    # the outputs were replaced after the forward: the fused maps no longer speak for them
    assert loss._fused({"_head_maps": (maps, p_cls, p_reg), "P_cls": p_cls.clone(), "P_reg": p_reg}) is None


def test_proposal_loss_ignores_fused_maps_it_cannot_use():
    cfg = second_car_cfg()
    b, h, w = 1, 6, 8
    na = cfg.NUM_CLASSES * cfg.NUM_YAW
    g = torch.Generator().manual_seed(0)
    maps = torch.randn(b, na * 8, h, w, generator=g)
    P_cls = maps[:, :na].reshape(b, cfg.NUM_CLASSES, cfg.NUM_YAW, h, w)
    P_reg = maps[:, na:].reshape(b, cfg.NUM_CLASSES, 7, cfg.NUM_YAW, h, w).permute(0, 1, 3, 4, 5, 2)
    shape = (b, cfg.NUM_CLASSES, cfg.NUM_YAW, h, w)
    G_cls = (torch.rand(shape, generator=g) > 0.8).long()
    item = dict(P_cls=P_cls, P_reg=P_reg, G_cls=G_cls, M_cls=torch.ones(shape, dtype=torch.bool),
                G_reg=torch.randn(shape + (7,), generator=g), M_reg=(G_cls == 1).unsqueeze(-1))
    ref = ProposalLoss(cfg)(item)
    got = ProposalLoss(cfg)(dict(item, _head_maps=maps))  # CPU maps: the native pass does not apply
    assert torch.equal(ref["loss"], got["loss"]) and torch.equal(ref["cls_loss"], got["cls_loss"])
