"""Dense half of a SECOND train step on hand-written kernels (csrc/dense_train.hip).

The RPN (7 x [Conv2d 128 -> 128 + BatchNorm2d(batch statistics) + ReLU], vision3d/detector/second.py:58-94) and the fused
[cls | reg] 1x1 head (detector/proposal.py:19-22) run forward and backward as ONE native call each way
(`v3d_dense_train_forward / _backward`): bf16 storage / fp32 accumulation, i.e. the arithmetic of the same stack under
`torch.autocast(bfloat16)` -- which is how rounds 1-2 ran it, through MIOpen.  `DenseTrainFunction` is the autograd node;
`Second.forward` takes this path in training mode under bf16 autocast when the shapes are the ones the kernels are built for
(`supported()`), otherwise the torch modules run as before.

Two arithmetics (`precision`):
  "bf16"    bf16 storage, one MFMA term per product: the contract of `torch.autocast(bfloat16)`; input = bf16 channels_last map.
  "bf16x3"  the fp32 step of the reference's train.py:58-66 (no autocast anywhere): every tensor of the step is a split pair of bf16
            planes (hi + lo, 16 significant bits), every product three MFMA terms with fp32 accumulation (2^-17 per product, no
            scales to calibrate), statistics / normalisation / head gradients in fp32 on hi + lo
            (`v3d_dense_train_forward_split / _backward_split`); input = the fp32 BEV map, gradient returned in fp32.
            What `Second.forward` runs for a training step OUTSIDE autocast: no MIOpen convolution in the step.
"""
import ctypes as C

import torch
from torch import nn

from . import _lib as L


MAX_CACHED_PLANS = 4  # (device, batch geometry) -> plan, per model


def rpn_pairs(rpn):
    mods = [m for m in list(rpn.down_block) + list(rpn.up_block) if not isinstance(m, (nn.ZeroPad2d, nn.ReLU))]
    convs = [m for m in mods if isinstance(m, nn.Conv2d)]
    bns = [m for m in mods if isinstance(m, nn.modules.batchnorm._BatchNorm)]
    return convs, bns, len(convs) + len(bns) == len(mods)


def supported(rpn, head, bev, precision="bf16"):
    return why_unsupported(rpn, head, bev, precision) is None


def why_unsupported(rpn, head, bev, precision="bf16"):
    """None when the kernels cover the stack, else the reason.  Covered: bf16 channels_last CUDA input with 128 channels; every
    RPN conv 128 -> 128, 3x3 (pad 1 / ZeroPad2d + pad 0) or 1x1, stride 1, bias-free, fp32 weights, each followed by a
    BatchNorm2d (affine, in TRAINING mode with a numeric momentum: the kernels always normalise with batch statistics and update
    the running statistics by `momentum`; fp32 running statistics on the input's device) + ReLU; head = two biased fp32 1x1
    convs with 8 * n <= 64 fused outputs."""
    if precision == "bf16x3":
        if not (bev.is_cuda and bev.dtype == torch.float32 and bev.dim() == 4 and bev.shape[1] == 128 and bev.shape[3] >= 4):
            return "input is not an fp32 CUDA map with 128 channels"
    elif not (bev.is_cuda and bev.dtype == torch.bfloat16 and bev.dim() == 4 and bev.shape[1] == 128 and bev.shape[3] >= 4
              and bev.is_contiguous(memory_format=torch.channels_last)):
        return "input is not a bf16 channels_last CUDA map with 128 channels"
    convs, bns, clean = rpn_pairs(rpn)
    if not clean or len(convs) != len(bns) or not convs or len(convs) > 16:
        return "the RPN is not a stack of (Conv2d, BatchNorm2d, ReLU) triples"
    dev = getattr(bev, "device", None)

    def off_device(t):  # (stand-in inputs of the CPU tests carry no device: nothing to compare then)
        return dev is not None and t.device != dev
    mods = list(rpn.down_block) + list(rpn.up_block)
    for i, m in enumerate(mods):  # a ZeroPad2d may only sit in front of a pad-0 3x3 conv (together = pad 1)
        if isinstance(m, nn.ZeroPad2d):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if m.padding != (1, 1, 1, 1) or not isinstance(nxt, nn.Conv2d) or nxt.kernel_size != (3, 3) or nxt.padding != (0, 0):
                return "a ZeroPad2d that is not ZeroPad2d(1) in front of a pad-0 3x3 convolution"
    for i, (c, b) in enumerate(zip(convs, bns)):
        k = c.kernel_size[0]
        if (c.in_channels != 128 or c.out_channels != 128 or c.kernel_size not in ((1, 1), (3, 3)) or c.stride != (1, 1)
                or c.bias is not None or c.groups != 1 or c.dilation != (1, 1) or c.weight.dtype != torch.float32):
            return f"RPN convolution {i} is not a bias-free fp32 128 -> 128 3x3 / 1x1 stride-1 convolution"
        padded = c.padding == (k // 2, k // 2) or (k == 3 and c.padding == (0, 0))  # the latter only behind a ZeroPad2d (checked above)
        if not padded or not isinstance(b, nn.BatchNorm2d) or not b.affine or b.num_features != 128:
            return f"RPN layer {i}: padding / BatchNorm2d(128, affine) pattern not covered"
        if not b.training:
            return f"RPN BatchNorm {i} is frozen (eval mode inside a training model): the kernels use batch statistics"
        if b.track_running_stats and b.running_mean is not None:
            if b.momentum is None:
                return f"RPN BatchNorm {i} has momentum=None (cumulative average): the kernels take a numeric momentum"
            if (b.running_mean.dtype != torch.float32 or b.running_var.dtype != torch.float32 or off_device(b.running_mean)
                    or off_device(b.running_var)):
                return f"RPN BatchNorm {i}: running statistics are not fp32 tensors on the input's device"
        if b.weight.dtype != torch.float32 or off_device(b.weight) or off_device(c.weight):
            return f"RPN layer {i}: parameters are not fp32 tensors on the input's device"
    for conv in (head.conv_cls, head.conv_reg):
        if conv.kernel_size != (1, 1) or conv.in_channels != 128 or conv.bias is None or conv.stride != (1, 1):
            return "head convolutions are not biased 1x1 convolutions on 128 channels"
        if (conv.weight.dtype != torch.float32 or conv.bias.dtype != torch.float32 or off_device(conv.weight)
                or off_device(conv.bias)):
            return "head parameters are not fp32 tensors on the input's device"
    if (head.conv_cls.out_channels + head.conv_reg.out_channels) not in (8, 16, 24, 32, 48, 64):
        return "fused head width is not one of 8, 16, 24, 32, 48, 64"
    if precision == "bf16x3" and head.conv_cls.out_channels + head.conv_reg.out_channels > 16:
        return "fused head wider than 16 channels (the split path's head kernel)"
    return None


class DenseTrainPlan(object):
    """Arena + parameter plumbing for one (B, H, W) geometry.  The arena carries the activations of ONE forward to its backward."""

    def __init__(self, rpn, head, B, H, W, device, precision="bf16"):
        self.rpn, self.head = rpn, head
        self.convs, self.bns, _ = rpn_pairs(rpn)
        self.B, self.H, self.W, self.device = int(B), int(H), int(W), device
        self.O = head.conv_cls.out_channels + head.conv_reg.out_channels
        self.split = precision == "bf16x3"
        lib = L.lib()
        if self.split:
            n = int(lib.v3d_dense_train_arena_bytes_split(self.B, self.H, self.W, len(self.convs), self.O))
            self.arena = torch.empty(n, dtype=torch.uint8, device=device)
        else:
            n = int(lib.v3d_dense_train_arena_bytes(self.B, self.H, self.W, len(self.convs), self.O))
            self.arena = torch.empty(n, dtype=torch.uint8, device=device)
            with torch.cuda.device(device):
                L.check(lib.v3d_dense_train_arena_init(L.ptr(self.arena), self.B, self.H, self.W, len(self.convs), self.O, L.stream_ptr()),
                        "dense_train_arena_init")
        self.generation = 0

    def parameters(self):
        ps = []
        for c, b in zip(self.convs, self.bns):
            ps += [c.weight, b.weight, b.bias]
        return ps + [self.head.conv_cls.weight, self.head.conv_cls.bias, self.head.conv_reg.weight, self.head.conv_reg.bias]

    def _io(self, grads=None):
        io = (L.DenseTrainLayer * len(self.convs))()
        self._keep = []  # contiguous copies of parameters stored in another memory format (e.g. a channels_last module)
        for i, (d, c, b) in enumerate(zip(io, self.convs, self.bns)):
            ptrs = []
            for t in (c.weight, b.weight, b.bias):
                if t.dtype != torch.float32 or t.device != self.device:
                    raise RuntimeError("dense train plan: parameters must be float32 tensors on the plan's device")
                t = t.detach()
                if not t.is_contiguous():
                    t = t.contiguous()
                    self._keep.append(t)
                ptrs.append(t.data_ptr())
            d.weight, d.gamma, d.beta = ptrs
            if b.track_running_stats and b.running_mean is not None:
                d.running_mean, d.running_var = b.running_mean.data_ptr(), b.running_var.data_ptr()
                d.num_batches_tracked = b.num_batches_tracked.data_ptr()
            d.eps = float(b.eps)
            d.momentum = float(b.momentum) if b.momentum is not None else 0.1
            d.ksize = int(c.kernel_size[0])
            if grads is not None:
                d.grad_weight, d.grad_gamma, d.grad_beta = (g.data_ptr() for g in grads[3 * i:3 * i + 3])
        return io

    def _head(self):
        w = torch.cat((self.head.conv_cls.weight.detach().reshape(-1, 128), self.head.conv_reg.weight.detach().reshape(-1, 128)), 0)
        b = torch.cat((self.head.conv_cls.bias.detach(), self.head.conv_reg.bias.detach()), 0)
        return w.float().contiguous(), b.float().contiguous()

    def forward(self, bev):
        """bev bf16 (B, 128, H, W) channels_last [split plan: fp32 (B, 128, H, W)] -> fused head maps fp32 (B, O, H, W)."""
        maps = torch.empty((self.B, self.O, self.H, self.W), dtype=torch.float32, device=self.device)
        self._hw, self._hb = self._head()
        io = self._io()
        with torch.cuda.device(self.device):
            if self.split:
                from .runtime import to_split_nhwc
                bev = to_split_nhwc(bev.float().contiguous())  # (hi, lo) bf16 NHWC planes, kept for the backward (weight gradient)
                L.check(L.lib().v3d_dense_train_forward_split(L.ptr(bev[0]), L.ptr(bev[1]), self.B, self.H, self.W, io, len(self.convs),
                                                              L.ptr(self._hw), L.ptr(self._hb), self.O, L.ptr(self.arena), L.ptr(maps),
                                                              L.stream_ptr()), "dense_train_forward_split")
            else:
                L.check(L.lib().v3d_dense_train_forward(L.ptr(bev), self.B, self.H, self.W, io, len(self.convs), L.ptr(self._hw),
                                                        L.ptr(self._hb), self.O, L.ptr(self.arena), L.ptr(maps), L.stream_ptr()),
                        "dense_train_forward")
        stats = [t for b in self.bns if b.track_running_stats and b.running_mean is not None
                 for t in (b.running_mean, b.running_var, b.num_batches_tracked)]
        if stats:  # updated through raw pointers: bump the version counters (host side only)
            torch._C._autograd._unsafe_set_version_counter(stats, [t._version + 1 for t in stats])
        self.generation += 1
        self._bev = bev  # the first layer's input is read again by the backward (weight gradient)
        return maps

    def backward(self, dmaps):
        """dmaps fp32 (B, O, H, W) -> (dbev bf16 channels_last, [grads in the order of parameters()])."""
        dmaps = dmaps.float().contiguous()
        params = self.parameters()
        n_rpn = 3 * len(self.convs)
        flat = torch.empty(sum(p.numel() for p in params[:n_rpn]) + self.O * 129, dtype=torch.float32, device=self.device)
        grads, off = [], 0
        for p in params[:n_rpn]:
            grads.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        dhw, dhb = flat[off:off + self.O * 128].view(self.O, 128), flat[off + self.O * 128:off + self.O * 129]
        io = self._io(grads)
        with torch.cuda.device(self.device):
            if self.split:
                planes = torch.empty((2, self.B, self.H, self.W, 128), dtype=torch.bfloat16, device=self.device)
                L.check(L.lib().v3d_dense_train_backward_split(L.ptr(self._bev[0]), L.ptr(self._bev[1]), L.ptr(dmaps), self.B, self.H, self.W,
                                                               io, len(self.convs), L.ptr(self._hw), self.O, L.ptr(self.arena), L.ptr(dhw),
                                                               L.ptr(dhb), L.ptr(planes[0]), L.ptr(planes[1]), L.stream_ptr()),
                        "dense_train_backward_split")
                dbev = torch.empty((self.B, 128, self.H, self.W), dtype=torch.float32, device=self.device)  # what the sparse plan takes
                L.check(L.lib().v3d_split_nhwc_to_nchw(L.ptr(planes[0]), L.ptr(planes[1]), self.B, 128, self.H, self.W, L.ptr(dbev),
                                                       L.stream_ptr()), "split_nhwc_to_nchw")
            else:
                dbev = torch.empty((self.B, 128, self.H, self.W), dtype=torch.bfloat16, device=self.device, memory_format=torch.channels_last)
                L.check(L.lib().v3d_dense_train_backward(L.ptr(self._bev), L.ptr(dmaps), self.B, self.H, self.W, io, len(self.convs),
                                                         L.ptr(self._hw), self.O, L.ptr(self.arena), L.ptr(dhw), L.ptr(dhb), L.ptr(dbev),
                                                         L.stream_ptr()), "dense_train_backward")
        nc = self.head.conv_cls.out_channels
        grads += [dhw[:nc].reshape(self.head.conv_cls.weight.shape), dhb[:nc], dhw[nc:].reshape(self.head.conv_reg.weight.shape), dhb[nc:]]
        return dbev, grads


class DenseTrainFunction(torch.autograd.Function):
    """Autograd node of the dense half: forward / backward are one native call each (DenseTrainPlan)."""

    @staticmethod
    def forward(ctx, plan, bev, *params):
        ctx.plan = plan
        out = plan.forward(bev.detach())
        ctx.generation = plan.generation
        ctx.needs_bev_grad = bev.requires_grad
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dmaps):
        plan = ctx.plan
        if ctx.generation != plan.generation:
            raise RuntimeError("dense train plan: this backward belongs to forward #%d but the plan's arena now holds forward #%d "
                               "(one forward/backward pair at a time per plan; V3D_DENSE_TRAIN=torch, or Second.native_dense_train "
                               "= False, keeps the torch modules, which accept several forwards before a backward)"
                               % (ctx.generation, plan.generation))
        dbev, grads = plan.backward(dmaps)
        return (None, dbev if ctx.needs_bev_grad else None) + tuple(grads)


def train_head_maps(rpn, head, bev, cache, precision="bf16"):
    """bev (bf16 channels_last | fp32 for "bf16x3") -> fused fp32 head maps through the native plan; `cache`: dict owned by the caller
    (plans by geometry and arithmetic)."""
    key = (str(bev.device), tuple(bev.shape), precision)
    plan = cache.get(key)
    if plan is None:
        while len(cache) >= MAX_CACHED_PLANS:  # an arena is ~150 MB (split: ~300 MB) per image of the batch: keep the most recent only
            cache.pop(next(iter(cache)))
        plan = cache[key] = DenseTrainPlan(rpn, head, bev.shape[0], bev.shape[2], bev.shape[3], bev.device, precision)
    return DenseTrainFunction.apply(plan, bev, *plan.parameters())
