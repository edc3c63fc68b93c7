"""TEST INFRASTRUCTURE (lives beside the oracle, outside the product package): builds a HIP model and the CPU oracle with
identical weights, and smoke_step() = one tiny training step of the HIP path on cuda:0 checked against the oracle.
Imported only by tests/ and __graft_entry__.smoke()."""
import os
import sys

import numpy as np
import torch


def build_pair(D=128, depth=1, heads=2, dd=1, grid=(16, 32), B=2, seed=0, out_vars=("total_precipitation_24hr",),
               in_vars=None):
    """(HIP model on cuda, oracle state dict + config on CPU) holding identical weights + a seeded batch.
    in_vars: the input variable list (default: the four constants + the output variables)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "orbit-2_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import orbit2_oracle as O
    from climate_learn.models.hub import Res_Slim_ViT
    consts = ["land_sea_mask", "orography", "lattitude", "landcover"]
    in_vars = list(in_vars) if in_vars is not None else consts + [v for v in out_vars]
    cfg = O.Config(in_vars, grid, len(out_vars), D, depth, dd, heads, spatial_resolution=156.0)
    sd = O.init_state_dict(cfg, len(in_vars), seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in ("var_embed", "var_query"):
        sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    for k in sd:
        if k.endswith(".bias") and "norm" not in k:
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=g)
    model = Res_Slim_ViT(in_vars, grid, len(in_vars), len(out_vars), 1, patch_size=2, embed_dim=D, depth=depth,
                         decoder_depth=dd, num_heads=heads, drop_path=0.0, drop_rate=0.0, learn_pos_emb=True)
    model.load_state_dict(sd, strict=True)
    model.data_config(156.0, grid, len(in_vars), len(out_vars))
    x = torch.randn(B, len(in_vars), *grid, generator=g)
    y = torch.randn(B, len(out_vars), grid[0] * 4 + 1, grid[1] * 4 + 3, generator=g)
    pi = list(out_vars).index("total_precipitation_24hr")
    y[:, pi] = torch.log1p(torch.relu(y[:, pi]))
    return model, sd, cfg, O, x, y, in_vars, list(out_vars)


def nerr(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


TOL_FLOOR, TOL_FACTOR = 2e-2, 1.5


def grad_tolerance(spread):
    """the tolerance contract: a HIP result may differ from the fp32 oracle by the contract's 2e-2 (normalised max
    error) or by 1.5 x what running the SAME math in plain bf16 moves that tensor, whichever is larger.  `spread` is that
    measured bf16-vs-fp32 figure: the reference's own (tests/golden/bf16_spread.npz, written from the reference's modules by
    tests/golden/make_golden_bf16_spread.py) where a fixture exists, the oracle's (oracle_bf16_spread) elsewhere."""
    return max(TOL_FLOOR, TOL_FACTOR * float(spread))


def oracle_bf16_spread(O, sd, cfg, x, y, in_vars, out_vars, loss="bayesian_tv", vw=None, fp32_grads=None, **kw):
    """the yardstick where no reference fixture exists for the configuration: the oracle (pinned to the reference by
    tests/test_oracle_golden.py) run once more with weights, inputs and arithmetic in plain bf16; returns
    {name: normalised max error of its bf16 gradient against its fp32 gradient}, and the same for rel. L2 under 'l2.'+name"""
    def cast(o, dt):
        if torch.is_tensor(o):
            return o.to(dt) if o.is_floating_point() else o
        if isinstance(o, dict):
            return {k: cast(v, dt) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return type(o)(cast(v, dt) for v in o)
        return o

    def run(dt):
        s = {k: v.clone().to(dt).requires_grad_() for k, v in sd.items()}
        yy = y.to(dt) if loss.startswith("perceptual") else y       # (the LPIPS network takes the target as an input image)
        l = O.training_loss(s, cfg, x.to(dt), yy, in_vars, out_vars, loss, vw, **cast(kw, dt))
        l.float().backward()
        return {k: v.grad.detach().float() for k, v in s.items() if v.grad is not None}
    g32 = fp32_grads if fp32_grads is not None else run(torch.float32)
    g16 = run(torch.bfloat16)
    out = {}
    for k, g in g32.items():
        if k in g16:
            out[k] = nerr(g16[k], g)
            out["l2." + k] = float((g16[k].double() - g.double()).norm() / g.double().norm().clamp_min(1e-30))
    return out


def smoke_step():
    from climate_learn import _hip
    from climate_learn.metrics import Bayesian_TV
    from climate_learn.trainer import training_step
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    assert _hip.selftest() == 0, "MFMA / LDS layout self-test failed"
    model, sd, cfg, O, x, y, in_vars, out_vars = build_pair()
    dev = torch.device("cuda:0")
    model = model.to(dev).eval()
    vw = {"total_precipitation_24hr": 1.0}
    loss = training_step((x, y, in_vars, out_vars), 0, model, dev, vw, Bayesian_TV(aggregate_only=True))
    loss.backward()
    torch.cuda.synchronize()
    sdo = {k: v.clone().requires_grad_() for k, v in sd.items()}
    ref = O.training_loss(sdo, cfg, x, y, in_vars, out_vars, "bayesian_tv", vw)
    ref.backward()
    e_loss = abs(float(loss) - float(ref)) / abs(float(ref))
    e_g = nerr(model.head[0].weight.grad, sdo["head.0.weight"].grad)
    e_q = nerr(model.blocks[0].attn.qkv.weight.grad, sdo["blocks.0.attn.qkv.weight"].grad)
    # tolerance per tensor: 2e-2, or 1.5 x what plain bf16 arithmetic moves this tensor of this very model (measured here)
    sp = oracle_bf16_spread(O, sd, cfg, x, y, in_vars, out_vars, "bayesian_tv", vw,
                            fp32_grads={k: v.grad.detach() for k, v in sdo.items() if v.grad is not None})
    t_g, t_q = grad_tolerance(sp["head.0.weight"]), grad_tolerance(sp["blocks.0.attn.qkv.weight"])
    print("[smoke] loss hip=%.6f oracle=%.6f rel=%.2e | grad err head.0=%.2e (bf16 spread %.2e, tol %.2e) qkv=%.2e "
          "(bf16 spread %.2e, tol %.2e)" % (float(loss), float(ref), e_loss, e_g, sp["head.0.weight"], t_g, e_q,
                                            sp["blocks.0.attn.qkv.weight"], t_q), flush=True)
    assert e_loss < 2e-2 and e_g <= t_g and e_q <= t_q, "HIP step disagrees with the CPU oracle"
