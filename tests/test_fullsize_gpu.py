"""Parity at the BASELINE configuration's FULL sizes (interm_1b: B x 8192 tokens x 3072 channels, 24 heads of 128),
where the CPU oracle cannot run the whole problem: each kernel is checked (a) on a random SAMPLE of its outputs
against an fp64 restatement that needs only the inputs of those outputs, and (b) through size-independent properties
(rows of softmax sum to one, gradient column-sum identities, linearity, batch independence, determinism)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import orbit2_oracle as O
from tests.hashmask import keep_mask

BF = torch.bfloat16
B_, L_, D_, H_, HID_ = 2, 8192, 3072, 24, 12288
M_ = B_ * L_


@pytest.fixture(scope="module")
def hip():
    from climate_learn import _hip
    _hip.lib()
    return _hip


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, generator=g, device="cuda") * scale).to(BF)


def test_gemm_forward_and_dw_forms_sampled_entries(hip):
    """fc1 forward (ring kernel, bias + GELU + dropout epilogue) and the four grouped dW GEMMs at the 1b shapes:
    4096 random output entries each against fp64 dot products of the very same bf16 inputs"""
    x, w1, b1 = rnd(M_, D_, seed=1), rnd(HID_, D_, scale=0.03, seed=2), rnd(HID_, seed=3)
    out = torch.empty(M_, HID_, dtype=BF, device="cuda")
    pre = torch.empty(M_, HID_, dtype=BF, device="cuda")
    p, seed = 0.1, 424242
    hip.gemm(x, w1, out, M_, HID_, D_, D_, D_, HID_, bias=b1, act=1, save_pre=pre, drop_p=p, seed=seed)
    g = torch.Generator().manual_seed(5)
    mi = torch.randint(0, M_, (4096,), generator=g).cuda()
    ni = torch.randint(0, HID_, (4096,), generator=g).cuda()
    ref_pre = (x[mi].double() * w1[ni].double()).sum(1) + b1[ni].double()
    assert float((pre[mi, ni].double() - ref_pre).abs().max() / ref_pre.abs().max()) < 6e-3
    # dropout mask of element (m, n): byte ((m*N+n) & 3) of hash((m*N+n) >> 2); evaluate the replica only on the sample
    from tests.hashmask import o2_hash64
    idx = (mi.cpu().numpy().astype(np.uint64) * np.uint64(HID_) + ni.cpu().numpy().astype(np.uint64))
    hh = o2_hash64(seed, idx >> np.uint64(2))
    keep = ((hh >> ((idx & np.uint64(3)) * np.uint64(8))) & np.uint64(0xFF)) >= np.uint64(26)
    ref_out = F.gelu(pre[mi, ni].double().cpu()) * torch.from_numpy(keep.astype(np.float64)) * (256.0 / 230.0)
    assert float((out[mi, ni].double().cpu() - ref_out).abs().max() / ref_out.abs().max()) < 8e-3
    # grouped weight-gradient GEMMs: dW[n_out, n_in] = sum_t dY[t, n_out] X[t, n_in]
    probs, chk = [], []
    for i, (no, ni_) in enumerate(((3 * D_, D_), (D_, D_), (HID_, D_), (D_, HID_))):
        dy, xx = rnd(M_, no, scale=0.05, seed=10 + i), rnd(M_, ni_, seed=20 + i)
        dw = torch.empty(no, ni_, dtype=BF, device="cuda")
        probs.append((dy, xx, dw, no, ni_, M_, no, ni_, ni_, dict(a_kc=False, b_kc=False)))
        chk.append((dy, xx, dw, no, ni_))
    hip.gemm_grouped(probs)
    for dy, xx, dw, no, ni_ in chk:
        a = torch.randint(0, no, (512,), generator=g).cuda()
        b = torch.randint(0, ni_, (512,), generator=g).cuda()
        ref = (dy[:, a].double() * xx[:, b].double()).sum(0)
        assert float((dw[a, b].double() - ref).abs().max() / ref.abs().max()) < 6e-3


def test_attention_full_length_rows_and_identities(hip):
    """L = 8192, 24 heads of 128: sampled query rows of out / lse / dQ against an fp64 row-wise restatement, plus the
    identities sum_k dV[k] = sum_q dO[q] (rows of P sum to one) and sum_k dK[k] = 0 when all queries are equal"""
    B, L, H, d = 1, L_, H_, 128
    qkv = rnd(B, L, 3 * H * d, scale=0.7, seed=31)
    do = rnd(B, L, H * d, seed=32)
    out, lse = hip.attn_fwd(qkv, B, L, H, d, 0.0, 0)
    dqkv = hip.attn_bwd(qkv, out, do, lse, B, L, H, d, 0.0, 0)
    q5 = qkv.view(B, L, 3, H, d)
    g = torch.Generator().manual_seed(7)
    for _ in range(24):
        h, q = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, L, (1,), generator=g))
        qv, K, V = q5[0, q, 0, h].double(), q5[0, :, 1, h].double(), q5[0, :, 2, h].double()
        s = (K @ qv) * d ** -0.5
        pr = torch.softmax(s, 0)
        o_ref = pr @ V
        assert float((out.view(B, L, H, d)[0, q, h].double() - o_ref).abs().max() / o_ref.abs().max()) < 1.5e-2
        assert abs(float(lse[0, h, q]) - float(torch.logsumexp(s, 0))) < 2e-3
        dov = do.view(B, L, H, d)[0, q, h].double()
        dp = V @ dov
        ds = pr * (dp - (pr * dp).sum())
        dq_ref = (ds @ K) * d ** -0.5
        got = dqkv.view(B, L, 3, H, d)[0, q, 0, h].double()
        assert float((got - dq_ref).abs().max() / dq_ref.abs().max()) < 3e-2
    dv_sum = dqkv.view(B, L, 3, H, d)[0, :, 2].double().sum(0)              # [H, d]
    do_sum = do.view(B, L, H, d)[0].double().sum(0)
    assert float((dv_sum - do_sum).abs().max() / do_sum.abs().max()) < 2e-2
    # all queries equal -> every row of dS sums to zero and dK[k] = scale * q * sum_q dS[q, k] sums to zero over k
    qkv2 = qkv.clone()
    qkv2.view(B, L, 3, H, d)[:, :, 0] = qkv2.view(B, L, 3, H, d)[:, :1, 0]
    out2, lse2 = hip.attn_fwd(qkv2, B, L, H, d, 0.0, 0)
    dq2 = hip.attn_bwd(qkv2, out2, do, lse2, B, L, H, d, 0.0, 0).view(B, L, 3, H, d)
    dk = dq2[0, :, 1].double()
    assert float(dk.sum(0).abs().max() / dk.abs().sum(0).max()) < 2e-3
    # dropout on: same seed -> bit-identical, another seed -> another mask, E[out] preserved to the dropout noise level
    o1, _ = hip.attn_fwd(qkv, B, L, H, d, 0.1, 77)
    o2, _ = hip.attn_fwd(qkv, B, L, H, d, 0.1, 77)
    o3, _ = hip.attn_fwd(qkv, B, L, H, d, 0.1, 78)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    assert float((o1.double() - out.double()).mean().abs()) < 1e-3


def test_attention_at_the_bench_batch_last_sample(hip):
    """per-GPU batch 16 (the bench's): qkv is 2.4 GB, so the last samples' rows lie past 2^31 bytes from its base -- sampled query
    rows of sample 15 (out / lse / dQ, and dK / dV of sampled keys) against the fp64 row-wise restatement, and sample 15 computed
    alone gives the same bits (batch independence across the 2 GiB line), with and without dropout"""
    B, L, H, d = 16, L_, H_, 128
    qkv = rnd(B, L, 3 * H * d, scale=0.7, seed=51)
    do = rnd(B, L, H * d, seed=52)
    out, lse = hip.attn_fwd(qkv, B, L, H, d, 0.0, 0)
    dqkv = hip.attn_bwd(qkv, out, do, lse, B, L, H, d, 0.0, 0)
    b = B - 1
    q5 = qkv.view(B, L, 3, H, d)
    g = torch.Generator().manual_seed(9)
    for _ in range(8):
        h, q = int(torch.randint(0, H, (1,), generator=g)), int(torch.randint(0, L, (1,), generator=g))
        qv, K, V = q5[b, q, 0, h].double(), q5[b, :, 1, h].double(), q5[b, :, 2, h].double()
        s = (K @ qv) * d ** -0.5
        pr = torch.softmax(s, 0)
        o_ref = pr @ V
        assert float((out.view(B, L, H, d)[b, q, h].double() - o_ref).abs().max() / o_ref.abs().max()) < 1.5e-2
        assert abs(float(lse[b, h, q]) - float(torch.logsumexp(s, 0))) < 2e-3
        dov = do.view(B, L, H, d)[b, q, h].double()
        dp = V @ dov
        ds = pr * (dp - (pr * dp).sum())
        dq_ref = (ds @ K) * d ** -0.5
        got = dqkv.view(B, L, 3, H, d)[b, q, 0, h].double()
        assert float((got - dq_ref).abs().max() / dq_ref.abs().max()) < 3e-2
    one = qkv[b:b + 1].contiguous()
    for p, seed in ((0.0, 0), (0.1, 77)):
        if p:
            out, lse = hip.attn_fwd(qkv, B, L, H, d, p, seed)
            dqkv = hip.attn_bwd(qkv, out, do, lse, B, L, H, d, p, seed)
        if p == 0.0:                                   # (with dropout the masks are a function of the global row index b*H*L ...)
            o1, l1 = hip.attn_fwd(one, 1, L, H, d, p, seed)
            d1 = hip.attn_bwd(one, o1, do[b:b + 1].contiguous(), l1, 1, L, H, d, p, seed)
            assert torch.equal(o1[0], out[b]) and torch.equal(l1[0], lse[b]) and torch.equal(d1[0], dqkv[b])
        else:
            o2, l2 = hip.attn_fwd(qkv, B, L, H, d, p, seed)
            assert torch.equal(o2, out) and torch.equal(l2, lse)
            dv_sum = dqkv.view(B, L, 3, H, d)[b, :, 2].double().sum(0)
            assert torch.isfinite(dv_sum).all()


def test_generated_attention_kernels_at_the_bench_batch(hip):
    """the GENERATED forward / dQ / dK+dV kernels (q pre-scaled, d = 128, L % 256 == 0: csrc/attn_*_asm.h) at the bench's shape,
    per-GPU batch 16: rows of the LAST sample (past 2^31 bytes of qkv) of out / lse / dQ and sampled keys of dK / dV against fp64
    restatements over the whole sequence, without and with dropout (mask replica for the sampled rows / keys); the identity
    sum_k dV[k] = sum_q dO[q]; bit-repeatability; and agreement with the compiler-scheduled kernels (ORBIT2_ATTN_NO_W4)"""
    from tests.hashmask import ATTN_KEY_SALT, o2_hash64
    import numpy as np
    B, L, H, d = 16, L_, H_, 128
    PRE = hip.ATTN_Q_PRESCALED
    x = rnd(B, L, 3 * H * d, scale=0.7, seed=61).view(B, L, 3, H * d)
    x[:, :, 0] = (x[:, :, 0].float() * (1.4426950408889634 / d ** 0.5)).to(BF)          # what the qkv GEMM's column scale stores
    qkv = x.view(B, L, 3 * H * d)
    do = rnd(B, L, H * d, seed=62)
    b = B - 1
    q5 = qkv.view(B, L, 3, H, d)
    g = torch.Generator().manual_seed(19)
    for p, seed in ((0.0, 0), (0.1, 0xABCDEF12345)):
        out, lse = hip.attn_fwd(qkv, B, L, H, d, p, seed, flags=PRE)
        dqkv = hip.attn_bwd(qkv, out, do, lse, B, L, H, d, p, seed, flags=PRE)
        thr = int(p * 256 + 0.5)
        sc = 256.0 / (256.0 - thr)

        def keep(h, rows, keys):           # [len(rows), len(keys)] keep mask of (sample b, head h)
            if thr == 0:
                return torch.ones(len(rows), len(keys), dtype=torch.float64, device="cuda")
            R = o2_hash64(seed, ((b * H + h) * L + rows.cpu().numpy()).astype(np.uint64)).astype(np.uint64)[:, None]
            K = o2_hash64((seed ^ ATTN_KEY_SALT) & 0xFFFFFFFFFFFFFFFF, (keys.cpu().numpy() >> 2).astype(np.uint64)).astype(np.uint64)[None, :]
            xx = ((R ^ K) * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)
            xx ^= xx >> np.uint64(16)
            byte = (xx >> ((keys.cpu().numpy().astype(np.uint64) & np.uint64(3)) * np.uint64(8))[None, :]) & np.uint64(0xFF)
            return torch.from_numpy((byte >= np.uint64(thr)).astype(np.float64)).cuda()

        allk = torch.arange(L, device="cuda")
        for _ in range(3):
            h = int(torch.randint(0, H, (1,), generator=g))
            Qt, K, V = q5[b, :, 0, h].double(), q5[b, :, 1, h].double(), q5[b, :, 2, h].double()
            dO = do.view(B, L, H, d)[b, :, h].double()
            rows = torch.randint(0, L, (6,), generator=g).cuda()
            s2 = Qt[rows] @ K.t()                                     # exp2 domain
            lse2 = torch.logsumexp(s2 * 0.6931471805599453, 1)
            P = torch.exp(s2 * 0.6931471805599453 - lse2[:, None])
            M = keep(h, rows, allk)
            A = P * M * sc
            o_ref = A @ V
            assert float((out.view(B, L, H, d)[b, rows, h].double() - o_ref).abs().max() / o_ref.abs().max()) < 1.5e-2
            assert float((lse[b, h, rows].double() - lse2).abs().max()) < 2e-3
            dA = dO[rows] @ V.t()
            delta = (dO[rows] * o_ref).sum(1)
            dS = P * (dA * M * sc - delta[:, None])
            dq_ref = (dS @ K) * d ** -0.5
            got = dqkv.view(B, L, 3, H, d)[b, rows, 0, h].double()
            assert float((got - dq_ref).abs().max() / dq_ref.abs().max()) < 3e-2
            # sampled keys: columns of P over ALL queries (row statistics from the kernel's own lse / out)
            keys = torch.randint(0, L, (6,), generator=g).cuda()
            lse_all = lse[b, h].double()
            Pk = torch.exp((Qt @ K[keys].t()) * 0.6931471805599453 - lse_all[:, None])          # [L, 6]
            Mk = keep(h, allk, keys)
            O_all = out.view(B, L, H, d)[b, :, h].double()
            delta_all = (dO * O_all).sum(1)
            dv_ref = (Pk * Mk * sc).t() @ dO
            dAk = dO @ V[keys].t()
            dSk = Pk * (dAk * Mk * sc - delta_all[:, None])
            dk_ref = (dSk.t() @ Qt) * 0.6931471805599453               # w.r.t. the unscaled k: ln 2 * dS^T q~
            gk = dqkv.view(B, L, 3, H, d)[b, keys, 1, h].double()
            gv = dqkv.view(B, L, 3, H, d)[b, keys, 2, h].double()
            assert float((gv - dv_ref).abs().max() / dv_ref.abs().max()) < 3e-2
            assert float((gk - dk_ref).abs().max() / dk_ref.abs().max()) < 3e-2
        out2, lse2_ = hip.attn_fwd(qkv, B, L, H, d, p, seed, flags=PRE)
        dq2 = hip.attn_bwd(qkv, out2, do, lse2_, B, L, H, d, p, seed, flags=PRE)
        assert torch.equal(out, out2) and torch.equal(lse, lse2_) and torch.equal(dqkv, dq2)          # bit-repeatable
        old, lse_o = hip.attn_fwd(qkv, B, L, H, d, p, seed, flags=PRE | hip.ATTN_NO_W4)
        dq_o = hip.attn_bwd(qkv, old, do, lse_o, B, L, H, d, p, seed, flags=PRE | hip.ATTN_NO_W4)
        assert float((out[b].float() - old[b].float()).abs().max() / old[b].float().abs().max()) < 1e-2
        assert float((dqkv[b].float() - dq_o[b].float()).abs().max() / dq_o[b].float().abs().max()) < 2e-2
        if p == 0.0:
            dv_sum = dqkv.view(B, L, 3, H, d)[b, :, 2].double().sum(0)
            do_sum = do.view(B, L, H, d)[b].double().sum(0)
            assert float((dv_sum - do_sum).abs().max() / do_sum.abs().max()) < 2e-2
        del out2, lse2_, dq2, old, lse_o, dq_o


def test_layernorm_adamw_sampled(hip):
    x = rnd(M_, D_, scale=2.0, seed=41)
    gam, bet = rnd(D_, seed=42), rnd(D_, seed=43)
    y, mean, rstd = hip.layernorm_fwd(x, gam, bet)
    g = torch.Generator().manual_seed(9)
    rows = torch.randint(0, M_, (256,), generator=g).cuda()
    ref = F.layer_norm(x[rows].double(), (D_,), gam.double(), bet.double(), 1e-5)
    assert float((y[rows].double() - ref).abs().max() / ref.abs().max()) < 6e-3
    # fused AdamW over 2^28 elements, 3 steps, sampled against the oracle's formula
    n = 1 << 28
    gg = torch.Generator(device="cuda").manual_seed(44)
    p = torch.randn(n, generator=gg, device="cuda")
    grad = (torch.randn(n, generator=gg, device="cuda") * 3e-3).to(BF)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    p16 = torch.empty(n, dtype=BF, device="cuda")
    idx = torch.randint(0, n, (4096,), generator=g).cuda()
    pr, mr, vr = p[idx].double().cpu(), torch.zeros(4096, dtype=torch.float64), torch.zeros(4096, dtype=torch.float64)
    gr = grad[idx].double().cpu() * 0.5
    for step in (1, 2, 3):
        hip.adamw(p, m, v, grad, p16, n, 5e-4, 0.9, 0.99, 1e-8, 1e-5, step, 0.5, None)
        O.adamw_step(pr, gr, mr, vr, step, 5e-4, 0.9, 0.99, 1e-8, 1e-5)       # in place on (p, m, v)
    assert float((p[idx].double().cpu() - pr).abs().max()) < 2e-6
    assert torch.equal(p16[idx].float(), p[idx].to(BF).float())


def test_varagg_full_grid_sampled_tokens_and_linearity(hip):
    """V = 23 variables on the 128 x 256 grid (8192 tokens / sample), D = 3072, 24 heads: sampled tokens of the folded
    forward against the table algebra in fp64, and linearity of the MFMA backward in dz"""
    V, h, w = 23, 128, 256
    g = torch.Generator(device="cuda").manual_seed(51)
    x = torch.randn(B_, V, h, w, generator=g, device="cuda")
    stab = torch.randn(H_, V, 5, generator=g, device="cuda") * 0.4
    gtab = torch.randn(V, 5, D_, generator=g, device="cuda") * 0.1
    z, attw = hip.varagg_fwd(x, stab, gtab, H_, D_)
    gc = torch.Generator().manual_seed(3)
    dh = D_ // H_
    for _ in range(16):
        tok = int(torch.randint(0, M_, (1,), generator=gc))
        b, l = divmod(tok, L_)
        pr_, pc = divmod(l, w // 2)
        pt = torch.ones(V, 5, dtype=torch.float64)
        pt[:, :4] = x[b, :, 2 * pr_:2 * pr_ + 2, 2 * pc:2 * pc + 2].reshape(V, 4).double().cpu()
        a = torch.softmax((stab.double().cpu() * pt).sum(-1), -1)                       # [H, V]
        val = (gtab.double().cpu() * pt[:, :, None]).sum(1)                             # [V, D]
        ref = torch.cat([(a[hh, :, None] * val[:, hh * dh:(hh + 1) * dh]).sum(0) for hh in range(H_)])
        assert float((z[tok].double().cpu() - ref).abs().max() / ref.abs().max()) < 6e-3
        assert float((attw[tok].double().cpu() - a).abs().max()) < 1e-5
    dz1, dz2 = rnd(M_, D_, seed=52), rnd(M_, D_, seed=53)
    s1, g1 = hip.varagg_bwd(x, gtab, attw, dz1, H_, D_)
    s2, g2 = hip.varagg_bwd(x, gtab, attw, dz2, H_, D_)
    s12, g12 = hip.varagg_bwd(x, gtab, attw, (dz1.float() * 0.5 + dz2.float() * 0.25).to(BF), H_, D_)
    # dz1/2 + dz2/4 is rounded to bf16 once more: linear up to that rounding (2^-9 relative per element, averaged)
    assert float((g12 - (0.5 * g1 + 0.25 * g2)).abs().max() / g12.abs().max()) < 2e-3
    assert float((s12 - (0.5 * s1 + 0.25 * s2)).abs().max() / s12.abs().max()) < 2e-3


def test_interm_1b_model_batch_independence_and_determinism():
    """the whole interm_1b network at its real size (1.0 B parameters, 128 x 256 grid): eval-mode predictions of a
    sample do not depend on its batch neighbours, and a repeated forward is bit-identical"""
    from climate_learn.models.hub import Res_Slim_ViT
    consts = ["land_sea_mask", "orography", "lattitude", "landcover"]
    dv = consts + ["v%d" % i for i in range(18)] + ["total_precipitation_24hr"]
    outs = ["total_precipitation_24hr", "v0", "v1"]
    with torch.device("cuda"):
        m = Res_Slim_ViT(dv, (128, 256), len(dv), 3, 4, patch_size=2, embed_dim=3072, depth=8, decoder_depth=4,
                         num_heads=24, drop_path=0.1, drop_rate=0.1)
    m = m.cuda().eval()
    assert 0.99e9 < sum(p.numel() for p in m.parameters()) < 1.02e9
    g = torch.Generator(device="cuda").manual_seed(61)
    x = torch.randn(2, len(dv), 128, 256, generator=g, device="cuda")
    with torch.no_grad():
        y2 = m(x, dv, outs)
        y2b = m(x, dv, outs)
        y1 = m(x[1:2].contiguous(), dv, outs)
    assert y2.shape == (2, 3, 512, 1024) and torch.isfinite(y2).all()
    assert torch.equal(y2, y2b)
    assert torch.equal(y2[1:2], y1)


def test_balanced_weight_gradient_group_at_the_bench_batch(hip, monkeypatch):
    """round 6, _ops._dw_balance at the headline size (131072 tokens; the Block's four weight gradients = 1728 tiles = 6.75 rounds):
    the tail tiles (all of proj + the last 4 tile rows of qkv) are split 4 ways over the tokens, bf16 partials summed in fp32.
    Against the unbalanced group: every full-length tile bit for bit, the split rows to the partials' bf16 rounding; with beta = 1 the
    sum lands on top of an existing gradient; bit-repeatable; sampled entries against fp64 dot products."""
    from climate_learn import _ops
    T, D = 131072, D_
    g = torch.Generator(device="cuda").manual_seed(9)
    mk = lambda c: (torch.randn(T, _ops._ld_pad(c), device="cuda", generator=g) * 0.25).to(BF)[:, :c]
    shapes = ((3 * D, D), (D, D), (4 * D, D), (D, 4 * D))
    ops = [(mk(no), mk(ni)) for no, ni in shapes]

    def run(balance, beta, init):
        monkeypatch.setattr(_ops, "_DW_BALANCE", balance)
        outs = [init[i].clone() if init is not None else torch.empty(no, ni, dtype=BF, device="cuda") for i, (no, ni) in enumerate(shapes)]
        probs = [(dy, x, o, no, ni, T, dy.stride(0), x.stride(0), ni, dict(a_kc=False, b_kc=False, beta=beta))
                 for (dy, x), o, (no, ni) in zip(ops, outs, shapes)]
        probs2, sums = _ops._dw_balance(probs)
        assert (len(probs2) > len(probs)) == bool(balance) and (len(sums) == 2) == bool(balance)
        hip.gemm_grouped(probs2)
        for parts, S, rows, K, dst, b in sums:
            hip.batch_sum(parts, S, rows, K, dst, beta=b)
        torch.cuda.synchronize()
        return outs

    plain = run(0, 0.0, None)
    bal = run(4, 0.0, None)
    bal2 = run(4, 0.0, None)
    for a, b in zip(bal, bal2):
        assert torch.equal(a, b)                                          # fixed summation order
    assert torch.equal(plain[2], bal[2]) and torch.equal(plain[3], bal[3])     # fc1, fc2: untouched problems
    cut = 3 * D - 4 * 256
    assert torch.equal(plain[0][:cut], bal[0][:cut])                      # qkv's full-length tile rows
    for a, b in ((plain[0][cut:], bal[0][cut:]), (plain[1], bal[1])):   # the split rows: four bf16-rounded partials
        err = (a.float() - b.float()).abs().max() / a.float().abs().max()
        assert float(err) < 1.5e-2, float(err)
    dy, x = ops[1]                                                        # proj: sampled entries against fp64
    rows_, cols_ = torch.randint(0, D, (64,), device="cuda"), torch.randint(0, D, (64,), device="cuda")
    ref = (dy[:, rows_].double() * x[:, cols_].double()).sum(0)
    got = bal[1][rows_, cols_].double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-2
    init = [torch.full((no, ni), 0.5, dtype=BF, device="cuda") for no, ni in shapes]
    acc = run(4, 1.0, init)                                               # accumulation into an existing gradient
    d = (acc[1].float() - 0.5 - bal[1].float()).abs().max() / bal[1].float().abs().max()
    assert float(d) < 2e-2, float(d)
