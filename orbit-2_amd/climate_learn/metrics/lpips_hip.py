"""LPIPS-VGG16 and the `perceptual` loss (L1 + 0.5 * mean LPIPS) on the HIP backend.

Reference: metrics/functional.py:17-33 (`perceptual`), metrics/metrics.py:119-187 (`PERCEPTUAL`, which builds
`lpips.LPIPS(net='vgg')`, freezes it and wraps it in bf16 FSDP).  `lpips` is third-party and its pretrained weights
are not in the reference tree: the graph is restated here (ScalingLayer, torchvision VGG16 features, taps after
relu1_2/2_2/3_3/4_3/5_3, channel unit-normalisation, squared difference, `lin` 1x1 convs, spatial mean, sum);
weights come from a user-supplied state dict (lpips / torchvision key names) or are random (synthetic throughput).

Feature maps are NHWC bf16; a 3x3 convolution is im2col + the MFMA GEMM (bias + ReLU epilogue).  Only the
prediction receives a gradient (the LPIPS parameters are frozen, the target needs none)."""
import math
from typing import Dict, Optional

import torch

from .. import _hip

BF, F32 = torch.bfloat16, torch.float32
VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
VGG16_TAPS = (1, 3, 6, 9, 12)              # conv indices whose ReLU output feeds an LPIPS `lin`
# torchvision `features` indices of the 13 convolutions, and lpips' slice naming (net.sliceK.<idx>)
_TV_CONV_IDX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)


def random_lpips_state(seed: int = 0) -> Dict[str, torch.Tensor]:
    """He-normal convolutions, small positive lins -- same recipe as the stand-in weights the parity tests use."""
    g = torch.Generator(device="cpu").manual_seed(seed)      # CPU draws even inside a `with torch.device("cuda")` block
    sd, cin, i = {}, 3, 0
    for c in VGG16_CFG:
        if c == "M":
            continue
        sd["conv%d.weight" % i] = torch.randn(c, cin, 3, 3, generator=g, device="cpu") * math.sqrt(2.0 / (9 * cin))
        sd["conv%d.bias" % i] = torch.randn(c, generator=g, device="cpu") * 0.05
        cin = c
        i += 1
    for k, ci in enumerate(VGG16_TAPS):
        ch = sd["conv%d.weight" % ci].shape[0]
        sd["lin%d.weight" % k] = torch.rand(ch, generator=g, device="cpu") * (2.0 / ch)
    return sd


def _canonical(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """accepts conv{i}/lin{k} names, torchvision `features.{idx}` names or lpips `net.slice{s}.{idx}` / `lin{k}.model.1`"""
    out = {}
    for key, v in sd.items():
        k = key
        for pre in ("module.", "loss_fn.", "_fsdp_wrapped_module."):
            if k.startswith(pre):
                k = k[len(pre):]
        parts = k.split(".")
        if parts[0].startswith("conv") and parts[0][4:].isdigit():
            out[k] = v
        elif parts[0].startswith("lin") and parts[-1] == "weight":
            out["lin%s.weight" % parts[0][3:]] = v.reshape(-1)
        elif parts[0] in ("features",) or (parts[0] == "net" and parts[1].startswith("slice")):
            idx = int(parts[-2])
            if idx in _TV_CONV_IDX:
                out["conv%d.%s" % (_TV_CONV_IDX.index(idx), parts[-1])] = v
    return out


class LPIPSVGG16:
    """frozen network; call `perceptual(pred, target)` (autograd-aware) or `distance(pred, target)` (no grad)"""

    def __init__(self, device, state: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0):
        sd = _canonical(state) if state is not None else random_lpips_state(seed)
        self.device = torch.device(device)
        w0 = sd["conv0.weight"].float()                                   # [64,3,3,3] = [co][ci][ky][kx]
        self.w1 = w0.permute(2, 3, 1, 0).reshape(27, 64).contiguous().to(self.device)    # [(t*3+ci)][co]
        self.b1 = sd["conv0.bias"].float().contiguous().to(self.device)
        self.layers = []                                                  # convs 1..12
        for i in range(1, 13):
            w = sd["conv%d.weight" % i].float()                           # [co][ci][3][3]
            co, ci = w.shape[:2]
            wg = w.permute(0, 2, 3, 1).reshape(co, 9 * ci).contiguous()   # [co][(t, ci)]  = GEMM B operand (K contiguous)
            self.layers.append(dict(ci=ci, co=co, w=wg.to(self.device, BF), wt=wg.t().contiguous().to(self.device, BF),
                                    b=sd["conv%d.bias" % i].float().to(self.device, BF)))
        self.lins = [sd["lin%d.weight" % k].float().reshape(-1).contiguous().to(self.device) for k in range(5)]

    # ---- forward over a batch of images; returns the per-layer activations --------------------------------------
    def _features(self, img):
        N, _, H, W = img.shape
        acts = []                                # (kind, tensor, N, H, W, C): outputs of conv / pool stages in order
        h = _hip.lpips_conv1_fwd(img, self.w1, self.b1)
        acts.append(("conv", h, H, W, 64))
        li = 0
        for c in VGG16_CFG[1:]:
            ph, pw, pc = acts[-1][2], acts[-1][3], acts[-1][4]
            if c == "M":
                acts.append(("pool", _hip.maxpool2_fwd(acts[-1][1], N, ph, pw, pc), ph // 2, pw // 2, pc))
                continue
            L = self.layers[li]
            li += 1
            col = _hip.im2col3x3(acts[-1][1], N, ph, pw, pc)
            M = N * ph * pw
            out = torch.empty(M, L["co"], dtype=BF, device=img.device)
            _hip.gemm(col, L["w"], out, M, L["co"], 9 * pc, 9 * pc, 9 * pc, L["co"], bias=L["b"], act=2)
            del col
            acts.append(("conv", out, ph, pw, L["co"]))
        return acts

    @staticmethod
    def _check(pred, target):
        if pred.shape != target.shape or pred.dim() != 4 or pred.shape[1] != 3:
            raise ValueError("LPIPS needs two [B,3,H,W] tensors of the same shape, got %s and %s"
                             % (tuple(pred.shape), tuple(target.shape)))
        if pred.shape[2] % 16 or pred.shape[3] % 16:
            raise ValueError("this LPIPS build needs H and W to be multiples of 16 (four 2x2 max-pools)")

    def perceptual(self, pred, target):
        if target.shape[2] > pred.shape[2] or target.shape[3] > pred.shape[3]:
            # the caller hands over the uncropped target (trainer.training_step); the reference crops it to the
            # prediction before every loss (examples/intermediate_downscaling.py:295-296)
            target = target[:, :, : pred.shape[2], : pred.shape[3]]
        self._check(pred, target)
        return _PerceptualFn.apply(pred, target, self)


def _tap_positions(acts):
    conv_seen, pos = -1, {}
    for j, a in enumerate(acts):
        if a[0] == "conv":
            conv_seen += 1
            if conv_seen in VGG16_TAPS:
                pos[j] = VGG16_TAPS.index(conv_seen)
    return pos


class _PerceptualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, net):
        B, _, H, W = pred.shape
        p32 = pred.detach().float().contiguous()
        t32 = target.detach().float().contiguous()
        acts = net._features(torch.cat([p32, t32], 0))
        taps = _tap_positions(acts)
        val = torch.zeros(B, dtype=F32, device=pred.device)
        for j, k in taps.items():
            _, f, h, w, c = acts[j]
            _hip.lpips_tap_fwd(f, net.lins[k], val, B, h * w, c)
        l1 = torch.zeros(1, dtype=F32, device=pred.device)
        _hip.l1_mean(p32, t32, l1)
        ctx.net, ctx.acts, ctx.taps, ctx.dims = net, acts, taps, (B, H, W)
        ctx.save_for_backward(p32, t32)
        return (l1 + 0.5 * val.mean()).reshape(())

    @staticmethod
    def backward(ctx, gout):
        net, acts, taps = ctx.net, ctx.acts, ctx.taps
        B, H, W = ctx.dims
        p32, t32 = ctx.saved_tensors
        # the scalar upstream gradient (loss scale included) stays ON THE DEVICE: the tap / first-conv kernels read it through
        # a pointer, so the backward has no host read (no pipeline stall, capturable in a hipGraph)
        go = gout.detach().reshape(1).to(F32).contiguous()

        def tap(j):
            """gradient of 0.5*mean_b(val) w.r.t. the pre-ReLU output of stage j (prediction half), or None"""
            if j not in taps:
                return None
            _, f, h, w, c = acts[j]
            return _hip.lpips_tap_bwd(f, net.lins[taps[j]], 0.5 / (B * h * w), B, h * w, c, gscale=go)

        def pred_half(j):
            _, f, h, w, c = acts[j]
            return f[: B * h * w]

        # gz = gradient w.r.t. the pre-ReLU output of conv stage j (prediction images only); start at relu5_3
        j = len(acts) - 1
        gz = tap(j)
        conv_id = 12
        while j > 0:
            L = net.layers[conv_id - 1]
            conv_id -= 1
            _, _, h, w, _ = acts[j]
            M = B * h * w
            dcol = torch.empty(M, 9 * L["ci"], dtype=BF, device=gz.device)
            _hip.gemm(gz, L["wt"], dcol, M, 9 * L["ci"], L["co"], L["co"], L["co"], 9 * L["ci"])
            if acts[j - 1][0] == "conv":
                gz = _hip.col2im3x3(dcol, B, h, w, L["ci"], act=pred_half(j - 1), tapg=tap(j - 1))
                j -= 1
            else:                                          # pool below: its input is the conv stage j-2
                gp = _hip.col2im3x3(dcol, B, h, w, L["ci"])
                _, _, sh, sw, sc = acts[j - 2]
                gz = _hip.maxpool2_bwd(gp, pred_half(j - 2), B, sh, sw, sc, tapg=tap(j - 2))
                j -= 2
            del dcol
        dimg = _hip.lpips_conv1_bwd(gz, net.w1, p32, t32, 1.0 / p32.numel(), gscale=go)
        return dimg, None, None
