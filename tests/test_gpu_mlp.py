"""GPU: the one-kernel feed-forward block (GEMM, GELU, GEMM, residual, LayerNorm) against fp64 PyTorch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer(m, inter, seed):
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(seed)
    bf = torch.bfloat16
    x = torch.randn((m, 384), generator=g, device=dev).to(bf)
    w1 = (torch.randn((inter, 384), generator=g, device=dev) * 0.05).to(bf)
    b1 = (torch.randn((inter,), generator=g, device=dev) * 0.1).to(bf)
    w2 = (torch.randn((384, inter), generator=g, device=dev) * 0.03).to(bf)
    b2 = (torch.randn((384,), generator=g, device=dev) * 0.1).to(bf)
    gamma = (1 + 0.1 * torch.randn((384,), generator=g, device=dev)).to(bf)
    beta = (0.1 * torch.randn((384,), generator=g, device=dev)).to(bf)
    return x, w1, b1, w2, b2, gamma, beta


def _exact(x, w1, b1, w2, b2, gamma, beta):
    d = torch.float64
    hmid = torch.nn.functional.gelu(x.to(d) @ w1.to(d).t() + b1.to(d))
    v = x.to(d) + hmid @ w2.to(d).t() + b2.to(d)
    return torch.nn.functional.layer_norm(v, (384,), gamma.to(d), beta.to(d), 1e-12)


@pytest.mark.parametrize('m,inter', [(1000, 1536), (128, 1536), (1, 1536), (129, 64), (4097, 1536), (300, 32)])
def test_mlp_block_matches_exact(m, inter):
    import rag_hip
    args = _layer(m, inter, 7 * m + inter)
    y = rag_hip.enc_mlp_dev(*args, 1e-12)
    ref = _exact(*args)
    err = (y.double() - ref).abs()
    # outputs are O(1) after LayerNorm and stored as bf16 (half an ulp = 4e-3 at 1..2); the
    # intermediate activation is rounded to bf16 once, as in the unfused layer
    assert float(err.max()) < 4e-2
    assert float(err.mean()) < 4e-3
    # the unfused route (library GEMMs, GELU and residual+LayerNorm kernels) rounds three more times;
    # the one-kernel block must be at least as close to the exact result
    x, w1, b1, w2, b2, gamma, beta = args
    mid = rag_hip.enc_gelu_((x @ w1.t() + b1).contiguous())
    unf = rag_hip.enc_add_layernorm_dev((mid @ w2.t() + b2).contiguous(), x, gamma, beta, 1e-12)
    assert float(err.mean()) <= float((unf.double() - ref).abs().mean()) * 1.05


@pytest.mark.parametrize('m,inter', [(1000, 1536), (128, 1536), (1, 1536), (129, 64), (4097, 1536)])
def test_attention_output_and_mlp_block_matches_exact(m, inter):
    """Attention-output projection + LayerNorm + feed-forward block in one kernel."""
    import rag_hip
    x, w1, b1, w2, b2, gamma2, beta2 = _layer(m, inter, 11 * m + inter)
    dev, bf = x.device, torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(5 * m + inter)
    ctx = torch.randn((m, 384), generator=g, device=dev).to(bf)
    wo = (torch.randn((384, 384), generator=g, device=dev) * 0.05).to(bf)
    bo = (torch.randn((384,), generator=g, device=dev) * 0.1).to(bf)
    gamma1 = (1 + 0.1 * torch.randn((384,), generator=g, device=dev)).to(bf)
    beta1 = (0.1 * torch.randn((384,), generator=g, device=dev)).to(bf)
    y = rag_hip.enc_attn_out_mlp_dev(ctx, x, wo, bo, gamma1, beta1, w1, b1, w2, b2, gamma2, beta2, 1e-12)
    d = torch.float64
    v = x.to(d) + ctx.to(d) @ wo.to(d).t() + bo.to(d)
    x1 = torch.nn.functional.layer_norm(v, (384,), gamma1.to(d), beta1.to(d), 1e-12)
    # the kernel rounds x1 to bf16 where the layer-by-layer forward stores it; follow that rounding
    x1b = x1.to(bf)
    ref = _exact(x1b, w1, b1, w2, b2, gamma2, beta2)
    err = (y.double() - ref).abs()
    # x1 is O(1) with a bf16 ulp of 8e-3: a one-ulp difference in a few of its 384 values moves the output
    assert float(err.mean()) < 6e-3
    assert float(err.max()) < 8e-2
    # against the layer-by-layer route on the same inputs
    lin = (ctx @ wo.t() + bo).contiguous()
    x1u = rag_hip.enc_add_layernorm_dev(lin, x, gamma1, beta1, 1e-12)
    yu = rag_hip.enc_mlp_dev(x1u, w1, b1, w2, b2, gamma2, beta2, 1e-12)
    assert float(err.mean()) <= float((yu.double() - ref).abs().mean()) * 1.1 + 1e-4


@pytest.mark.parametrize('m,inter', [(1000, 1536), (128, 1536), (1, 1536), (129, 64), (4097, 1536)])
def test_residual_layernorm_and_mlp_block_in_one_kernel(m, inter):
    """LN(x + proj) in front of the feed-forward block, in its kernel: against the residual+LayerNorm kernel
    followed by the feed-forward kernel (nearly always the same bf16 values: the two LayerNorms sum their
    rows in different orders) and against fp64."""
    import rag_hip
    x, w1, b1, w2, b2, gamma2, beta2 = _layer(m, inter, 13 * m + inter)
    dev, bf = x.device, torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(3 * m + inter)
    proj = torch.randn((m, 384), generator=g, device=dev).to(bf)
    gamma1 = (1 + 0.1 * torch.randn((384,), generator=g, device=dev)).to(bf)
    beta1 = (0.1 * torch.randn((384,), generator=g, device=dev)).to(bf)
    y = rag_hip.enc_norm_mlp_dev(proj, x, gamma1, beta1, w1, b1, w2, b2, gamma2, beta2, 1e-12)
    x1u = rag_hip.enc_add_layernorm_dev(proj, x, gamma1, beta1, 1e-12)
    yu = rag_hip.enc_mlp_dev(x1u, w1, b1, w2, b2, gamma2, beta2, 1e-12)
    assert float((y.float() - yu.float()).abs().mean()) < 2e-3
    assert float((y == yu).float().mean()) > 0.9
    d = torch.float64
    x1 = torch.nn.functional.layer_norm(x.to(d) + proj.to(d), (384,), gamma1.to(d), beta1.to(d), 1e-12)
    ref = _exact(x1.to(bf), w1, b1, w2, b2, gamma2, beta2)
    err = (y.double() - ref).abs()
    assert float(err.mean()) < 6e-3
    assert float(err.max()) < 8e-2
    assert float(err.mean()) <= float((yu.double() - ref).abs().mean()) * 1.1 + 1e-4


@pytest.mark.parametrize('m,n', [(1000, 1152), (128, 1152), (1, 32), (129, 64), (4097, 1152), (77, 384)])
def test_linear384_matches_exact(m, n):
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(3 * m + n)
    x = torch.randn((m, 384), generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn((n, 384), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    b = (torch.randn((n,), generator=g, device=dev) * 0.1).to(torch.bfloat16)
    y = rag_hip.enc_linear384_dev(x, w, b)
    ref = x.double() @ w.double().t() + b.double()
    # bf16 output of an fp32 accumulation: half an ulp of bf16 relative, plus accumulation noise
    assert float(((y.double() - ref).abs() / (ref.abs() + 1.0)).max()) < 6e-3
    lib_y = (x @ w.t() + b)
    assert float((y.double() - ref).abs().mean()) <= float((lib_y.double() - ref).abs().mean()) * 1.05


@pytest.mark.parametrize('m', [1000, 128, 1, 129, 4097])
def test_linear384_residual_layernorm_tail(m):
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(13 * m)
    bf = torch.bfloat16
    x = torch.randn((m, 384), generator=g, device=dev).to(bf)
    res = torch.randn((m, 384), generator=g, device=dev).to(bf)
    w = (torch.randn((384, 384), generator=g, device=dev) * 0.05).to(bf)
    b = (torch.randn((384,), generator=g, device=dev) * 0.1).to(bf)
    gamma = (1 + 0.1 * torch.randn((384,), generator=g, device=dev)).to(bf)
    beta = (0.1 * torch.randn((384,), generator=g, device=dev)).to(bf)
    y = rag_hip.enc_linear384_dev(x, w, b, res, gamma, beta, 1e-12)
    v = x.double() @ w.double().t() + b.double() + res.double()
    ref = torch.nn.functional.layer_norm(v, (384,), gamma.double(), beta.double(), 1e-12)
    err = (y.double() - ref).abs()
    assert float(err.max()) < 5e-2 and float(err.mean()) < 5e-3
    # the separate route: bf16 GEMM output, then the residual + LayerNorm kernel
    unf = rag_hip.enc_add_layernorm_dev(rag_hip.enc_linear384_dev(x, w, b), res, gamma, beta, 1e-12)
    assert float(err.mean()) <= float((unf.double() - ref).abs().mean()) * 1.1 + 1e-4


def test_mlp_rows_are_independent():
    """A row's result does not depend on which workgroup or lane computes it."""
    import rag_hip
    x, w1, b1, w2, b2, gamma, beta = _layer(700, 1536, 3)
    y = rag_hip.enc_mlp_dev(x, w1, b1, w2, b2, gamma, beta, 1e-12)
    perm = torch.randperm(700, device=x.device)
    y2 = rag_hip.enc_mlp_dev(x[perm].contiguous(), w1, b1, w2, b2, gamma, beta, 1e-12)
    assert torch.equal(y[perm], y2)


def test_misaligned_pointers_are_rejected():
    import rag_hip
    x, w1, b1, w2, b2, gamma, beta = _layer(8, 64, 1)
    flat = torch.zeros(8 * 384 + 8, dtype=torch.bfloat16, device=x.device)
    x_off = flat[1:1 + 8 * 384].view(8, 384)                # 2 bytes past a 16-byte boundary
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.enc_mlp_dev(x_off, w1, b1, w2, b2, gamma, beta)
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.enc_linear384_dev(x_off, w1, b1)


def test_mlp_argument_errors():
    import rag_hip
    x, w1, b1, w2, b2, gamma, beta = _layer(8, 48, 1)
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.enc_mlp_dev(x, w1, b1, w2, b2, gamma, beta)           # intermediate not a multiple of 32
    x, w1, b1, w2, b2, gamma, beta = _layer(8, 64, 1)
    with pytest.raises(ValueError):
        rag_hip.enc_mlp_dev(x, w1, b1[:32], w2, b2, gamma, beta)
    with pytest.raises(ValueError):
        rag_hip.enc_mlp_dev(x.float(), w1, b1, w2, b2, gamma, beta)


@pytest.mark.parametrize('m,k', [(1000, 1536), (128, 384), (37, 64), (1, 32), (4099, 1536)])
def test_linear_ln_split_matches_float64(m, k):
    """rag_enc_linear_ln_split_dev, both operand sources, against float64:
    LayerNorm(residual + GELU(h + b1) W^T + b2) from the fp32 first product, and
    LayerNorm(residual + a W^T + b2) from a pre-split fp16 operand; plus the split written for the next GEMM."""
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(1000 + m + k)
    h = torch.randn((m, k), generator=g, device=dev) * 1.2
    b1 = torch.randn((k,), generator=g, device=dev) * 0.3
    w = torch.randn((384, k), generator=g, device=dev) * 0.05
    b2, gamma, beta = (torch.randn((384,), generator=g, device=dev) * s + o for s, o in ((0.2, 0), (0.3, 1.0), (0.2, 0)))
    res = torch.randn((m, 384), generator=g, device=dev)
    wh, wl = rag_hip.split_weight_halves(w)
    # the packed images hold exactly fp16(w) and fp16(w - fp16(w))
    hi = w.to(torch.float16)
    assert wh.dtype == torch.float16 and torch.equal(rag_hip.unpack_weight_halves(wh), hi)
    assert torch.equal(rag_hip.unpack_weight_halves(wl), (w - hi.float()).to(torch.float16))

    def ref(a):
        pre = res.double() + a @ w.double().t() + b2.double()
        return torch.nn.functional.layer_norm(pre, (384,), gamma.double(), beta.double(), 1e-12)
    y, ys = rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, res, gamma, beta, 1e-12)
    want = ref(torch.nn.functional.gelu(h.double() + b1.double()))
    err = float((y.double() - want).abs().max())
    # the same block in plain fp32 PyTorch ops, for scale: the kernel must be in the same round-off class
    plain = torch.nn.functional.layer_norm(res + torch.nn.functional.gelu(h + b1) @ w.t() + b2, (384,), gamma, beta, 1e-12)
    err_plain = float((plain.double() - want).abs().max())
    print(f'm={m} k={k}: max error vs float64: kernel {err:.2e}, fp32 PyTorch ops {err_plain:.2e}')
    assert err < max(3 * err_plain, 2e-6), (err, err_plain)
    assert ys.dtype == torch.float16 and tuple(ys.shape) == (m, 1152) and torch.equal(ys[:, :384], ys[:, 384:768])
    assert float((ys[:, :384].double() + ys[:, 768:].double() - y.double()).abs().max()) < 2e-6
    assert torch.equal(ys[:, :384], y.to(torch.float16))
    y2, none = rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, res, gamma, beta, 1e-12, want_split=False)
    assert none is None and torch.equal(y2, y)
    # pre-split operand (what the attention kernel hands to the output projection)
    a3 = rag_hip.enc_split_bf16x3_dev(h, torch.float16)
    y3, _ = rag_hip.enc_linear_ln_split_dev(a3, None, wh, wl, b2, res, gamma, beta, 1e-12)
    err3 = float((y3.double() - ref(h.double())).abs().max())
    assert err3 < max(3 * err_plain, 2e-6), (err3, err_plain)
    # the same operand as [hi | lo] rows (in_is_split == 2): the same result bit for bit
    a2 = torch.cat([a3[:, :k], a3[:, 2 * k:]], dim=1).contiguous()
    y4, s4 = rag_hip.enc_linear_ln_split_dev(a2, None, wh, wl, b2, res, gamma, beta, 1e-12)
    assert torch.equal(y4, y3) and torch.equal(s4, _)
    # chained form (rag_enc_linear_ln_split_chain_dev): the residual given as the [hi | hi | lo] rows of a previous
    # call; equals the fp32-residual call on the residual hi + lo bit for bit, with and without the fp32 row
    r3 = rag_hip.enc_split_bf16x3_dev(res, torch.float16)
    r_back = r3[:, :384].float() + r3[:, 768:].float()
    assert float((r_back - res).abs().max()) <= 2 ** -22 * float(res.abs().max())
    y5, s5 = rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, r_back, gamma, beta, 1e-12)
    y6, s6 = rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, r3, gamma, beta, 1e-12)
    none7, s7 = rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, r3, gamma, beta, 1e-12, want_y=False)
    assert torch.equal(y6, y5) and torch.equal(s6, s5) and none7 is None and torch.equal(s7, s5)
    assert float((y6.double() - want).abs().max()) < max(3 * err_plain, 2e-6)
    none8, s8 = rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, res, gamma, beta, 1e-12, want_y=False)
    assert none8 is None and torch.equal(s8, ys)
    with pytest.raises(ValueError):
        rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, res, gamma, beta, 1e-12, want_y=False, want_split=False)
    # the separate route it replaces: library GEMM on [hi|hi|lo] operands + bias/residual/LayerNorm kernel
    w3 = rag_hip.split_weight_bf16x3(w, torch.float16)
    sep, _ = rag_hip.enc_bias_add_layernorm_split_dev(torch.mm(a3, w3.t(), out_dtype=torch.float32), b2, res, gamma, beta,
                                                      1e-12, half=torch.float16)
    assert float((sep - y3).abs().max()) < max(3 * err_plain, 2e-6)
    with pytest.raises(ValueError):
        rag_hip.enc_linear_ln_split_dev(h, None, wh, wl, b2, res, gamma, beta, 1e-12)


def test_cross_encoder_fp32_forward_same_with_and_without_the_fused_output_kernels():
    from indexing import bert
    dev = torch.device('cuda:0')
    m, _, cfg = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32, native=True, split_gemm=True,
                                split_dtype=torch.float16)
    g = torch.Generator(device=dev).manual_seed(5)
    lens = torch.randint(5, 150, (300,), generator=g, device=dev, dtype=torch.int32)
    cu = torch.zeros((301,), dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(lens, 0)
    t = int(cu[-1])
    ids = torch.randint(1000, cfg.vocab_size, (t,), generator=g, device=dev, dtype=torch.int32)
    types = torch.zeros_like(ids)
    pos = torch.cat([torch.arange(int(n), dtype=torch.int32, device=dev) for n in lens])
    with torch.no_grad():
        assert m.bert.fused_split_out
        a = m.forward_packed(ids, types, pos, cu, int(lens.max()))
        m.bert.fused_split_out = False
        b = m.forward_packed(ids, types, pos, cu, int(lens.max()))
    assert float((a - b).abs().max()) < 1e-6


def test_cross_encoder_fp32_forward_with_the_residual_kept_split(monkeypatch):
    """The fused route keeps the residual stream as the [hi | hi | lo] rows its kernels write (22 of an fp32 row's 24
    bits) instead of an fp32 row beside them: logits against the fp32-residual route and against a float64 forward of
    the same weights."""
    from indexing import bert
    dev = torch.device('cuda:0')
    m, _, cfg = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32, native=True, split_gemm=True,
                                split_dtype=torch.float16)
    plain, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32)
    g = torch.Generator(device=dev).manual_seed(6)
    lens = torch.randint(5, 150, (300,), generator=g, device=dev, dtype=torch.int32)
    cu = torch.zeros((301,), dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(lens, 0)
    t = int(cu[-1])
    assert t >= 12288
    ids = torch.randint(1000, cfg.vocab_size, (t,), generator=g, device=dev, dtype=torch.int32)
    types = torch.zeros_like(ids)
    pos = torch.cat([torch.arange(int(n), dtype=torch.int32, device=dev) for n in lens])
    with torch.no_grad():
        a = m.forward_packed(ids, types, pos, cu, int(lens.max())).clone()
        monkeypatch.setenv('RAG_CHAIN_RESIDUAL', '0')
        b = m.forward_packed(ids, types, pos, cu, int(lens.max())).clone()
        ids_l = [ids[int(cu[i]):int(cu[i + 1])].tolist() for i in range(300)]
        it, mt = bert.pad_batch(ids_l, 0, dev)
        plain64 = plain.double()
        want = plain64(it, torch.zeros_like(it), mt).float()
        plain32 = plain64.float()
        ref32 = plain32(it, torch.zeros_like(it), mt)
    err_chain, err_fp32res = float((a - want).abs().max()), float((b - want).abs().max())
    err_torch = float((ref32 - want).abs().max())
    print(f'max logit error vs float64: split residual {err_chain:.2e}, fp32 residual {err_fp32res:.2e}, PyTorch fp32 {err_torch:.2e}')
    assert float((a - b).abs().max()) < 2e-6
    assert err_chain < max(3 * err_torch, 2e-6)


def test_linear_ln_split_argument_errors_and_row_independence():
    """Limits of rag_enc_linear_ln_split_dev / rag_enc_linear_ln_split_pack_dev are refused with an error (no launch);
    rows do not influence each other (a row count that ends inside a 128-row tile, rows past the end never written)."""
    import ctypes as C
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(77)
    k = 96
    w = torch.randn((384, k), generator=g, device=dev) * 0.05
    wh, wl = rag_hip.split_weight_halves(w)
    b1 = torch.zeros((k,), device=dev)
    b2, gamma, beta = torch.zeros((384,), device=dev), torch.ones((384,), device=dev), torch.zeros((384,), device=dev)
    h = torch.randn((200, k), generator=g, device=dev)
    res = torch.randn((200, 384), generator=g, device=dev)
    y_all, s_all = rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, res, gamma, beta, 1e-12)
    for m in (1, 127, 129, 199):
        y, s = rag_hip.enc_linear_ln_split_dev(h[:m].contiguous(), b1, wh, wl, b2, res[:m].contiguous(), gamma, beta, 1e-12)
        assert torch.equal(y, y_all[:m]) and torch.equal(s, s_all[:m])
    # outputs larger than the row count keep their tail: nothing is written past row m
    lib = rag_hip.load_library()
    y_buf = torch.full((130, 384), 7.0, device=dev)
    s_buf = torch.full((130, 1152), 3.0, dtype=torch.float16, device=dev)
    f = lambda t: C.c_void_p(t.data_ptr())
    args = lambda m_, k_, n_, wh_=wh: (f(h), 0, f(b1), f(wh_), f(wl), f(b2), f(res), f(gamma), f(beta), C.c_float(1e-12), m_, k_,
                                         n_, f(y_buf), f(s_buf), None)
    torch.cuda.synchronize()
    assert lib.rag_enc_linear_ln_split_dev(*args(100, k, 384)) == 0
    torch.cuda.synchronize()
    assert torch.equal(y_buf[:100], y_all[:100]) and bool((y_buf[100:] == 7.0).all()) and bool((s_buf[100:] == 3.0).all())
    # limits
    assert lib.rag_enc_linear_ln_split_dev(*args(100, k, 383)) == rag_hip.lib().rag_enc_linear_ln_split_dev(*args(100, k, 383)) != 0
    assert 'output width' in rag_hip.last_error()
    assert lib.rag_enc_linear_ln_split_dev(*args(100, 40, 384)) != 0 and 'multiple of 32' in rag_hip.last_error()
    assert lib.rag_enc_linear_ln_split_dev(*args(100, 4096, 384)) != 0
    assert lib.rag_enc_linear_ln_split_dev(*args(-1, k, 384)) != 0
    assert lib.rag_enc_linear_ln_split_dev(*args(0, k, 384)) == 0                       # nothing to do
    odd = torch.empty((384 * k + 8,), dtype=torch.float16, device=dev)[1:]            # 2-byte offset: not 16-byte aligned
    assert lib.rag_enc_linear_ln_split_dev(*args(100, k, 384, odd)) != 0 and 'aligned' in rag_hip.last_error()
    assert lib.rag_enc_linear_ln_split_pack_dev(f(w), 40, 384, f(wh), f(wl), None) != 0
    assert lib.rag_enc_linear_ln_split_pack_dev(f(w), k, 100, f(wh), f(wl), None) != 0
    with pytest.raises(ValueError):
        rag_hip.enc_linear_ln_split_dev(h, b1, wh, wl, b2, res[:10], gamma, beta, 1e-12)
    rag_hip.check_faults()


@pytest.mark.parametrize('m,inter', [(1000, 1536), (1, 1536), (129, 64)])
def test_mlp_block_float16_instance_matches_exact(m, inter):
    """The same kernel with float16 elements (the cross-encoder's screening pass): 11 significant bits instead of 8."""
    import rag_hip
    args = [a.float().to(torch.float16) for a in _layer(m, inter, 11 * m + inter)]
    y = rag_hip.enc_mlp_dev(*args, 1e-12)
    assert y.dtype == torch.float16
    ref = _exact(*args)
    err = (y.double() - ref).abs()
    assert float(err.max()) < 6e-3 and float(err.mean()) < 6e-4          # an eighth of the bfloat16 bounds
    bf = rag_hip.enc_mlp_dev(*[a.to(torch.bfloat16) for a in args], 1e-12)
    assert float(err.mean()) < 0.3 * float((bf.double() - ref).abs().mean())


@pytest.mark.parametrize('m,n', [(1000, 1152), (1, 32), (77, 384)])
def test_linear384_float16_instance_matches_exact(m, n):
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(5 * m + n)
    h = torch.float16
    x = torch.randn((m, 384), generator=g, device=dev).to(h)
    w = (torch.randn((n, 384), generator=g, device=dev) * 0.05).to(h)
    b = (torch.randn((n,), generator=g, device=dev) * 0.1).to(h)
    y = rag_hip.enc_linear384_dev(x, w, b)
    assert y.dtype == h
    ref = x.double() @ w.double().t() + b.double()
    assert float(((y.double() - ref).abs() / (ref.abs() + 1.0)).max()) < 8e-4
    if n == 384:
        gamma = (1 + 0.1 * torch.randn((384,), generator=g, device=dev)).to(h)
        beta = (0.1 * torch.randn((384,), generator=g, device=dev)).to(h)
        res = torch.randn((m, 384), generator=g, device=dev).to(h)
        y2 = rag_hip.enc_linear384_dev(x, w, b, res, gamma, beta, 1e-12)
        ref2 = torch.nn.functional.layer_norm(ref + res.double(), (384,), gamma.double(), beta.double(), 1e-12)
        assert float((y2.double() - ref2).abs().max()) < 8e-3
