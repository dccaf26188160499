"""GPU: the encoder kernels between the GEMMs (fused attention, residual + LayerNorm) against
plain PyTorch fp32 references of the same operations."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_attention(qkv, lens, heads):
    b, s, h3 = qkv.shape
    h = h3 // 3
    q, k, v = qkv.float().view(b, s, 3, heads, h // heads).permute(2, 0, 3, 1, 4)
    bias = torch.zeros((b, 1, 1, s), device=qkv.device)
    bias.masked_fill_(torch.arange(s, device=qkv.device)[None, None, None, :] >= lens[:, None, None, None], -1e30)
    p = torch.softmax(q @ k.transpose(-1, -2) / (h // heads) ** 0.5 + bias, dim=-1)
    return (p @ v).transpose(1, 2).reshape(b, s, h)


@pytest.mark.parametrize('b,s,heads,d', [(7, 48, 12, 32), (5, 160, 12, 32), (3, 33, 4, 64), (2, 512, 2, 32),
                                         (2, 300, 3, 64), (9, 8, 12, 32)])
def test_attention_matches_fp32_reference(b, s, heads, d):
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(b * 1000 + s)
    qkv = (torch.randn((b, s, 3 * heads * d), generator=g, device=dev) * 1.5).to(torch.bfloat16)
    lens = torch.randint(1, s + 1, (b,), generator=g, device=dev, dtype=torch.int32)
    lens[0] = s
    if b > 2:
        lens[1] = 1
        lens[2] = 0                                   # an empty slot: defined (zero) output
    out = rag_hip.enc_attention_dev(qkv, lens, heads).float()
    ref = _ref_attention(qkv, lens, heads)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    for i in range(b):
        n = int(lens[i])
        if n == 0:
            assert float(out[i].abs().max()) == 0.0
            continue
        # real tokens: bf16 output of an fp32 softmax over bf16 products
        err = (out[i, :n] - ref[i, :n]).abs().max()
        assert float(err) < 2e-2, (i, n, float(err))
        # padding rows of a block that holds real tokens attend over the real keys too
        blk_end = min(s, (n + 31) // 32 * 32)
        assert float((out[i, n:blk_end] - ref[i, n:blk_end]).abs().max() if blk_end > n else 0.0) < 2e-2
        assert float(out[i, blk_end:].abs().max() if blk_end < s else 0.0) == 0.0


def test_attention_argument_errors():
    import rag_hip
    dev = torch.device('cuda:0')
    qkv = torch.zeros((1, 8, 3 * 48), dtype=torch.bfloat16, device=dev)
    lens = torch.ones(1, dtype=torch.int32, device=dev)
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.enc_attention_dev(qkv, lens, 3)          # head size 16
    with pytest.raises(ValueError):
        rag_hip.enc_attention_dev(qkv.double(), lens, 1)    # float64 is not covered
    big = torch.zeros((1, 513, 3 * 32), dtype=torch.bfloat16, device=dev)
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.enc_attention_dev(big, lens, 1)


@pytest.mark.parametrize('dtype,tol', [(torch.bfloat16, 2e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize('rows,hidden', [(1000, 384), (37, 768), (5, 2048), (3, 8)])
def test_add_layernorm_matches_reference(dtype, tol, rows, hidden):
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(rows + hidden)
    x = (torch.randn((rows, hidden), generator=g, device=dev) * 2 + 0.5).to(dtype)
    res = torch.randn((rows, hidden), generator=g, device=dev).to(dtype)
    gamma = (1 + 0.1 * torch.randn((hidden,), generator=g, device=dev)).to(dtype)
    beta = (0.1 * torch.randn((hidden,), generator=g, device=dev)).to(dtype)
    for r in (res, None):
        out = rag_hip.enc_add_layernorm_dev(x, r, gamma, beta, 1e-12).float()
        v = x.double() + (r.double() if r is not None else 0)
        ref = torch.nn.functional.layer_norm(v, (hidden,), gamma.double(), beta.double(), 1e-12)
        assert float((out - ref).abs().max()) < tol


def test_cross_encoder_with_native_kernels_tracks_fp32():
    from indexing import bert
    dev = torch.device('cuda:0')
    ref, _, cfg = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32)
    plain, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.bfloat16)
    native, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.bfloat16, native=True)
    g = torch.Generator(device=dev).manual_seed(3)
    b, s = 64, 96
    ids = torch.randint(1000, cfg.vocab_size, (b, s), generator=g, device=dev)
    lens = torch.randint(5, s + 1, (b,), generator=g, device=dev)
    mask = (torch.arange(s, device=dev)[None, :] < lens[:, None]).long()
    types = (torch.arange(s, device=dev)[None, :] >= (lens[:, None] // 3)).long() * mask
    with torch.no_grad():
        a = ref(ids, types, mask)
        p = plain(ids, types, mask)
        n = native(ids, types, mask)
    torch.cuda.synchronize()
    err_plain = float((p - a).abs().max())
    err_native = float((n - a).abs().max())
    spread = float(a.std())
    # the kernels keep more of the arithmetic in fp32 than the PyTorch bf16 ops do
    assert err_native <= max(1.5 * err_plain, 0.02 * spread + 1e-3), (err_native, err_plain, spread)


def test_embed_layernorm_matches_reference():
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(11)
    for dtype, tol in ((torch.bfloat16, 3e-2), (torch.float32, 2e-5)):
        word = torch.randn((500, 384), generator=g, device=dev).to(dtype)
        position = torch.randn((64, 384), generator=g, device=dev).to(dtype)
        type_emb = torch.randn((2, 384), generator=g, device=dev).to(dtype)
        gamma = (1 + 0.1 * torch.randn((384,), generator=g, device=dev)).to(dtype)
        beta = (0.1 * torch.randn((384,), generator=g, device=dev)).to(dtype)
        ids = torch.randint(0, 500, (777,), generator=g, device=dev, dtype=torch.int32)
        pos = torch.randint(0, 64, (777,), generator=g, device=dev, dtype=torch.int32)
        types = torch.randint(0, 2, (777,), generator=g, device=dev, dtype=torch.int32)
        out = rag_hip.enc_embed_layernorm_dev(ids, types, pos, word, position, type_emb, gamma, beta, 1e-12).float()
        v = word.double()[ids.long()] + position.double()[pos.long()] + type_emb.double()[types.long()]
        ref = torch.nn.functional.layer_norm(v, (384,), gamma.double(), beta.double(), 1e-12)
        assert float((out - ref).abs().max()) < tol
    # an id beyond the table is clamped and reported, not read
    ids[5] = 10 ** 6
    rag_hip.enc_embed_layernorm_dev(ids, types, pos, word, position, type_emb, gamma, beta, 1e-12)
    assert rag_hip.fault_flags() == 256


def test_packed_attention_equals_padded():
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(21)
    b, s, heads, d = 40, 130, 12, 32
    qkv = torch.randn((b, s, 3 * heads * d), generator=g, device=dev).to(torch.bfloat16)
    lens = torch.randint(1, s + 1, (b,), generator=g, device=dev, dtype=torch.int32)
    padded = rag_hip.enc_attention_dev(qkv, lens, heads)
    cu = torch.zeros((b + 1,), dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(lens, 0)
    keep = torch.arange(s, device=dev)[None, :] < lens[:, None]
    packed = rag_hip.enc_attention_packed_dev(qkv[keep].contiguous(), cu, int(lens.max()), heads)
    torch.cuda.synchronize()
    assert torch.equal(packed, padded[keep])
    # offsets that do not describe the batch are reported and the sequence is skipped
    cu[-1] += 5
    rag_hip.enc_attention_packed_dev(qkv[keep].contiguous(), cu, int(lens.max()) + 5, heads)
    assert rag_hip.fault_flags() == 128


def test_packed_pairs_match_padded_pairs_and_logits():
    import rag_hip
    from indexing import bert
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(5)
    nq, nc = 9, 50
    qlen = rng.integers(3, 20, nq)
    clen = rng.integers(1, 90, nc)
    q_off = torch.from_numpy(np.concatenate([[0], np.cumsum(qlen)]).astype(np.int64)).to(dev)
    c_off = torch.from_numpy(np.concatenate([[0], np.cumsum(clen)]).astype(np.int64)).to(dev)
    q_tok = torch.from_numpy(rng.integers(1000, 30000, int(qlen.sum())).astype(np.int32)).to(dev)
    c_tok = torch.from_numpy(rng.integers(1000, 30000, int(clen.sum())).astype(np.int32)).to(dev)
    n = 300
    pair_q = torch.from_numpy(rng.integers(0, nq, n).astype(np.int32)).to(dev)
    pair_r = torch.from_numpy(rng.integers(0, nc, n).astype(np.int64)).to(dev) + 7     # row base 7
    pair_r[::17] = -1                                                                    # empty slots
    max_len = 64                                                                         # truncates some chunks
    plen = rag_hip.ce_pair_len_dev(q_off, c_off, 7, pair_q, pair_r, max_len)
    cu = torch.zeros((n + 1,), dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(plen, 0)
    total = int(cu[-1])
    ids, types, pos = rag_hip.ce_pack_packed_dev(q_tok, q_off, c_tok, c_off, 7, pair_q, pair_r, cu, total, max_len)
    pid, ptyp, pmask = rag_hip.ce_pack_dev(q_tok, q_off, c_tok, c_off, 7, pair_q, pair_r, None, n, max_len)
    keep = pmask.bool()
    assert torch.equal(pmask.sum(dim=1).int(), plen)
    assert torch.equal(ids.long(), pid[keep]) and torch.equal(types.long(), ptyp[keep])
    assert torch.equal(pos.long(), (torch.cumsum(pmask, 1) - 1)[keep])
    # the same logits from the unpadded stream as from the padded batch (same kernels, same math)
    model, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.bfloat16, native=True)
    live = torch.nonzero(plen > 0)[:, 0]
    cu_live = torch.cat([cu[:-1][live], cu[-1:]])
    with torch.no_grad():
        packed = model.forward_packed(ids, types, pos, cu_live, int(plen.max()))
        padded = model(pid[live], ptyp[live], pmask[live])
    torch.cuda.synchronize()
    assert float((packed - padded).abs().max()) < 2e-2 * max(1.0, float(padded.abs().max()))


@pytest.mark.parametrize('b,s,heads,d', [(6, 26, 12, 64), (3, 150, 4, 32), (2, 70, 2, 64)])
def test_fp32_attention_matches_reference(b, s, heads, d):
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(b + s)
    qkv = torch.randn((b, s, 3 * heads * d), generator=g, device=dev)
    lens = torch.randint(1, s + 1, (b,), generator=g, device=dev, dtype=torch.int32)
    lens[0] = s
    out = rag_hip.enc_attention_dev(qkv, lens, heads)
    ref = _ref_attention(qkv.double(), lens, heads)
    for i in range(b):
        n = int(lens[i])
        assert float((out[i, :n].double() - ref[i, :n]).abs().max()) < 2e-6
    cu = torch.zeros((b + 1,), dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(lens, 0)
    keep = torch.arange(s, device=dev)[None, :] < lens[:, None]
    packed = rag_hip.enc_attention_packed_dev(qkv[keep].contiguous(), cu, int(lens.max()), heads)
    assert torch.equal(packed, out[keep])


def test_packed_fp32_embedder_matches_pytorch_ops():
    from indexing import bert
    dev = torch.device('cuda:0')
    plain, _, cfg = bert.load_model('random:bert-base', 'bi', dev)
    native, _, _ = bert.load_model('random:bert-base', 'bi', dev, native=True)
    g = torch.Generator(device=dev).manual_seed(8)
    b, s = 50, 26
    ids = torch.randint(1000, cfg.vocab_size, (b, s), generator=g, device=dev)
    lens = torch.randint(3, s + 1, (b,), generator=g, device=dev)
    mask = (torch.arange(s, device=dev)[None, :] < lens[:, None]).long()
    with torch.no_grad():
        a = plain(ids, None, mask, normalize=True)
        p = native.forward_packed(*native.bert.pack(ids, mask), normalize=True)
    torch.cuda.synchronize()
    # embeddings are unit vectors: a per-component bound is a bound on any cosine score
    assert float((a - p).abs().max()) < 2e-5
    assert float((a * p).sum(dim=1).min()) > 1 - 1e-6


def test_split_bf16_gemm_embedder_holds_the_cosine_tolerance():
    import rag_hip
    from indexing import bert
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn((1000, 768), generator=g, device=dev) * 3
    w = torch.randn((512, 768), generator=g, device=dev) * 0.05
    x3 = rag_hip.enc_split_bf16x3_dev(x)
    hi = x.bfloat16()
    assert torch.equal(x3[:, :768], hi) and torch.equal(x3[:, 768:1536], hi)
    assert torch.equal(x3[:, 1536:], (x - hi.float()).bfloat16())
    y = torch.mm(x3, rag_hip.split_weight_bf16x3(w).t(), out_dtype=torch.float32)
    ref = x.double() @ w.double().t()
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    plain, _, cfg = bert.load_model('random:bert-base', 'bi', dev)
    split, _, _ = bert.load_model('random:bert-base', 'bi', dev, native=True, split_gemm=True)
    b, s = 64, 26
    ids = torch.randint(1000, cfg.vocab_size, (b, s), generator=g, device=dev)
    lens = torch.randint(3, s + 1, (b,), generator=g, device=dev)
    mask = (torch.arange(s, device=dev)[None, :] < lens[:, None]).long()
    with torch.no_grad():
        a = plain(ids, None, mask, normalize=True)
        p = split.forward_packed(*split.bert.pack(ids, mask), normalize=True)
    corpus = torch.nn.functional.normalize(torch.randn((50000, 768), generator=g, device=dev), dim=1)
    # the bar of the path: cosine scores within 1e-4 of the fp32 forward (here: 10x inside it)
    assert float(((a - p) @ corpus.t()).abs().max()) < 1e-5
    assert float((a - p).abs().max()) < 2e-5


def test_split_tails_equal_the_separate_kernels_bit_for_bit():
    """bias + GELU + split and bias + residual + LayerNorm + split against the kernels they stand for;
    the forward built from them gives the same embeddings as the one built from the separate kernels."""
    import rag_hip
    from indexing import bert
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(12)
    for rows, k in ((1000, 3072), (37, 768), (1, 8)):
        y = torch.randn((rows, k), generator=g, device=dev) * 2
        bias = torch.randn((k,), generator=g, device=dev)
        want = rag_hip.enc_split_bf16x3_dev(rag_hip.enc_gelu_((y + bias).contiguous()))
        assert torch.equal(rag_hip.enc_bias_act_split_dev(y, bias, True), want)
        assert torch.equal(rag_hip.enc_bias_act_split_dev(y, bias, False), rag_hip.enc_split_bf16x3_dev(y + bias))
    for rows, h in ((1000, 768), (5, 384), (3, 2048)):
        y = torch.randn((rows, h), generator=g, device=dev)
        res = torch.randn((rows, h), generator=g, device=dev)
        bias, gamma, beta = (torch.randn((h,), generator=g, device=dev) for _ in range(3))
        want = rag_hip.enc_add_layernorm_dev(y + bias, res, gamma, beta, 1e-12)
        out, split = rag_hip.enc_bias_add_layernorm_split_dev(y, bias, res, gamma, beta, 1e-12)
        assert torch.equal(out, want) and torch.equal(split, rag_hip.enc_split_bf16x3_dev(want))
        out, split = rag_hip.enc_bias_add_layernorm_split_dev(y, bias, res, gamma, beta, 1e-12, want_split=False)
        assert torch.equal(out, want) and split is None
    with pytest.raises(ValueError):
        rag_hip.enc_bias_act_split_dev(y, bias[:-1].contiguous(), True)
    model, _, cfg = bert.load_model('random:bert-base', 'bi', dev, native=True, split_gemm=True)
    b, s = 48, 21
    ids = torch.randint(1000, cfg.vocab_size, (b, s), generator=g, device=dev)
    lens = torch.randint(3, s + 1, (b,), generator=g, device=dev)
    mask = (torch.arange(s, device=dev)[None, :] < lens[:, None]).long()
    packed = model.bert.pack(ids, mask)
    with torch.no_grad():
        assert model.bert.fused_split_tails and model.bert.split_attention
        with_mfma_attention = model.forward_packed(*packed, normalize=True)
        model.bert.split_attention = False          # plain fp32 attention in both forms
        fused = model.forward_packed(*packed, normalize=True)
        model.bert.fused_split_tails = False
        separate = model.forward_packed(*packed, normalize=True)
    assert torch.equal(fused, separate)
    # attention as split-bf16 matrix-core products: the same embeddings to fp32 round-off
    assert float((with_mfma_attention - fused).abs().max()) < 2e-6


@pytest.mark.parametrize('half', ['fp16', 'bf16'])
@pytest.mark.parametrize('heads,head_dim,max_len,n_seq,mag', [
    (12, 32, 157, 61, 0.3), (12, 32, 157, 61, 1.5), (12, 64, 40, 33, 0.3), (4, 32, 300, 9, 0.5), (2, 64, 288, 5, 0.3),
    (12, 32, 1, 3, 1.0)])
def test_split_attention_matches_fp64_and_first_only(heads, head_dim, max_len, n_seq, mag, half):
    """rag_enc_attention_split_dev: fp32 q|k|v (+ bias) -> [hi | hi | lo] of the attention output, against a
    float64 softmax(QK^T/sqrt(d))V.  With fp16 halves (22 significant bits) the result is as good as plain fp32
    arithmetic; with bf16 halves (16 bits) a product is good to 2^-16 of sum |q_i||k_i|, so the error grows with
    the operands' magnitude (`mag` 0.3 is what LayerNorm-ed inputs times 0.02-0.05 weights give; 1.5 means
    attention logits of +-10)."""
    import rag_hip
    dev = torch.device('cuda:0')
    hdt = torch.float16 if half == 'fp16' else torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(100 + heads + max_len)
    hid = heads * head_dim
    lens = torch.randint(1, max_len + 1, (n_seq,), generator=g, device=dev, dtype=torch.int32)
    lens[0] = max_len
    cu = torch.zeros((n_seq + 1,), dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(lens, 0)
    t = int(cu[-1])
    qkv = torch.randn((t, 3 * hid), generator=g, device=dev) * mag
    bias = torch.randn((3 * hid,), generator=g, device=dev) * (mag / 3)
    assert rag_hip.enc_attention_split_supported(head_dim, max_len)
    out3 = rag_hip.enc_attention_split_dev(qkv, bias, cu, max_len, heads, half=hdt)
    first3 = rag_hip.enc_attention_split_dev(qkv, bias, cu, max_len, heads, first_only=True, half=hdt)
    nobias3 = rag_hip.enc_attention_split_dev(qkv + bias, None, cu, max_len, heads, half=hdt)
    torch.cuda.synchronize()
    assert out3.dtype == hdt and tuple(out3.shape) == (t, 3 * hid) and tuple(first3.shape) == (n_seq, 3 * hid)
    assert torch.equal(out3[:, :hid], out3[:, hid:2 * hid])
    assert torch.equal(first3, out3[cu[:-1].long()])
    assert torch.equal(nobias3, out3)
    # rag_enc_attention_split_hl_dev: the same halves as [hi | lo] rows
    out2 = rag_hip.enc_attention_split_dev(qkv, bias, cu, max_len, heads, half=hdt, hi_lo=True)
    first2 = rag_hip.enc_attention_split_dev(qkv, bias, cu, max_len, heads, first_only=True, half=hdt, hi_lo=True)
    torch.cuda.synchronize()
    assert tuple(out2.shape) == (t, 2 * hid) and tuple(first2.shape) == (n_seq, 2 * hid)
    assert torch.equal(out2[:, :hid], out3[:, :hid]) and torch.equal(out2[:, hid:], out3[:, 2 * hid:])
    assert torch.equal(first2, out2[cu[:-1].long()])
    got = out3[:, :hid].double() + out3[:, 2 * hid:].double()
    full = (qkv + bias).double()
    cu_h = cu.cpu().tolist()
    worst = scale = 0.0
    for i in range(n_seq):
        x = full[cu_h[i]:cu_h[i + 1]].view(-1, 3, heads, head_dim)
        q, k, v = x[:, 0].transpose(0, 1), x[:, 1].transpose(0, 1), x[:, 2].transpose(0, 1)
        p = torch.softmax(q @ k.transpose(1, 2) / head_dim ** 0.5, dim=-1)
        ref = (p @ v).transpose(0, 1).reshape(-1, hid)
        worst = max(worst, float((got[cu_h[i]:cu_h[i + 1]] - ref).abs().max()))
        scale = max(scale, float(ref.abs().max()))
    # fp16 halves: a few fp32 ulps of the largest value (exp2 and fp32 accumulation included)
    tol = 4e-6 * max(scale, 1.0) * (1 + mag * mag * 4) if half == 'fp16' else 2e-4 * max(mag, 0.3) ** 2 / 0.09 * 0.1
    assert worst < tol, (worst, tol)
    # hi is the 16-bit rounding of the result, lo the rounding of what is left
    assert float((out3[:, :hid].float() - got.float()).abs().max()) <= 2 ** -8 * float(got.abs().max())


def test_cross_encoder_fp32_split_path_holds_the_fp32_order():
    """The float32 cross-encoder with split-bf16 GEMMs and attention (the default of the batched path) against
    the plain PyTorch float32 forward: logits within 5e-6, and the same top-5 of every 50-pair pool wherever the
    float32 logits that decide it are 2e-5 apart (reference: CrossEncoder.predict, src/reranker/rerank.py:79-84)."""
    from indexing import bert
    dev = torch.device('cuda:0')
    fast, _, cfg = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32, native=True, split_gemm=True,
                                   split_dtype=torch.float16)
    plain, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32)
    g = torch.Generator(device=dev).manual_seed(77)
    n_q, pool = 24, 50
    ids_l, typ_l = [], []
    for _ in range(n_q):
        q = torch.randint(1000, cfg.vocab_size, (int(torch.randint(8, 25, (1,), generator=g, device=dev)),), generator=g, device=dev)
        for _ in range(pool):
            c = torch.randint(1000, cfg.vocab_size, (int(torch.randint(4, 130, (1,), generator=g, device=dev)),), generator=g, device=dev)
            ids_l.append([101] + q.tolist() + [102] + c.tolist() + [102])
            typ_l.append([0] * (len(q) + 2) + [1] * (len(c) + 1))
    cu = torch.tensor([0] + [len(x) for x in ids_l], device=dev).cumsum(0).to(torch.int32)
    flat = lambda l: torch.tensor([v for x in l for v in x], dtype=torch.int32, device=dev)
    pos = torch.cat([torch.arange(len(x), dtype=torch.int32, device=dev) for x in ids_l])
    with torch.no_grad():
        got = fast.forward_packed(flat(ids_l), flat(typ_l), pos, cu, max(len(x) for x in ids_l))
        want = []
        for s0 in range(0, len(ids_l), 200):
            it, mt = bert.pad_batch(ids_l[s0:s0 + 200], 0, dev)
            tt, _ = bert.pad_batch(typ_l[s0:s0 + 200], 0, dev)
            want.append(plain(it, tt, mt))
        want = torch.cat(want)
    assert float((got - want).abs().max()) < 5e-6
    got, want = got.view(n_q, pool).double(), want.view(n_q, pool).double()
    srt = torch.sort(want, dim=1, descending=True).values
    decided = (srt[:, :5] - srt[:, 1:6]).min(dim=1).values >= 2e-5
    same = (torch.argsort(got, dim=1, descending=True, stable=True)[:, :5]
            == torch.argsort(want, dim=1, descending=True, stable=True)[:, :5]).all(dim=1)
    assert int(decided.sum()) >= n_q // 2           # the check is not vacuous
    assert bool(same[decided].all())


def test_first_token_attention_equals_row_zero_of_the_full_kernel():
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(31)
    lens = torch.randint(1, 140, (77,), generator=g, device=dev, dtype=torch.int32)
    cu = torch.zeros((78,), dtype=torch.int32, device=dev)
    cu[1:] = torch.cumsum(lens, 0)
    qkv = torch.randn((int(cu[-1]), 3 * 384), generator=g, device=dev).to(torch.bfloat16)
    full = rag_hip.enc_attention_packed_dev(qkv, cu, int(lens.max()), 12)
    first = rag_hip.enc_attention_packed_first_dev(qkv, cu, int(lens.max()), 12)
    assert torch.equal(first, full[cu[:-1].long()])


def test_gelu_in_place_matches_erf_gelu():
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn((1000, 1536), generator=g, device=dev) * 3
    ref = torch.nn.functional.gelu(x.double())
    y = rag_hip.enc_gelu_(x.clone())
    assert float((y.double() - ref).abs().max()) < 1e-6
    xb = x.to(torch.bfloat16)
    yb = rag_hip.enc_gelu_(xb.clone())
    assert torch.equal(yb, torch.nn.functional.gelu(xb.float()).to(torch.bfloat16))


def test_medical_embedder_packed_path_matches_padded_fp32():
    from indexing.embedder import MedicalEmbedder
    texts = ['aspirin side effects in adults', 'metformin', '', 'type 2 diabetes mellitus first line treatment options '
             'and contraindications in chronic kidney disease ' * 3, 'warfarin inr monitoring'] * 7
    fast = MedicalEmbedder('random:bert-base', device='cuda', tokens_per_batch=256)
    assert fast.model.bert.native and fast.model.bert.split_gemm
    plain = MedicalEmbedder('random:bert-base', device='cuda')
    plain.model.bert.native = False
    a, b = fast.encode(texts), plain.encode(texts)
    assert a.shape == (len(texts), 768) and a.dtype == np.float32 and a.flags['C_CONTIGUOUS']
    assert float(np.abs(a - b).max()) < 2e-5
    assert float(np.abs(np.linalg.norm(a, axis=1) - 1).max()) < 1e-5


def test_cross_encoder_reranker_packed_path_matches_padded_fp32():
    from reranker.rerank import CrossEncoderReranker
    fast = CrossEncoderReranker('random:minilm-l6', device='cuda')
    assert fast.model.bert.native
    plain = CrossEncoderReranker('random:minilm-l6', device='cuda')
    plain.model.bert.native = False
    pairs = [('what treats hypertension', 'ace inhibitors lower blood pressure ' * k) for k in range(1, 30)]
    pairs.append(('q', ''))
    a, b = fast.score_pairs(pairs), plain.score_pairs(pairs)
    assert len(a) == len(pairs)
    assert max(abs(x - y) for x, y in zip(a, b)) < 1e-4


@pytest.mark.parametrize('half', [torch.bfloat16, torch.float16])
def test_embed_layernorm_split_equals_the_two_passes(half):
    """rag_enc_embed_layernorm_split_dev: the fp32 rows and their [hi | hi | lo] split, bit for bit what
    rag_enc_embed_layernorm_dev followed by rag_enc_split_bf16x3_dev give (ragged token count, clamped ids flagged)."""
    import rag_hip
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(3)
    h, vocab, max_pos, n_types, t = 384, 1000, 128, 2, 1001
    word = torch.randn((vocab, h), generator=g, device=dev)
    position = torch.randn((max_pos, h), generator=g, device=dev) * 0.1
    type_emb = torch.randn((n_types, h), generator=g, device=dev) * 0.1
    gamma = torch.rand((h,), generator=g, device=dev) + 0.5
    beta = torch.randn((h,), generator=g, device=dev) * 0.1
    ids = torch.randint(0, vocab, (t,), generator=g, device=dev, dtype=torch.int32)
    types = torch.randint(0, n_types, (t,), generator=g, device=dev, dtype=torch.int32)
    pos = torch.randint(0, max_pos, (t,), generator=g, device=dev, dtype=torch.int32)
    want = rag_hip.enc_embed_layernorm_dev(ids, types, pos, word, position, type_emb, gamma, beta, 1e-12)
    want_split = rag_hip.enc_split_bf16x3_dev(want, half)
    got, got_split = rag_hip.enc_embed_layernorm_split_dev(ids, types, pos, word, position, type_emb, gamma, beta, 1e-12, half)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert got_split.dtype == half and torch.equal(got_split.view(torch.int16), want_split.view(torch.int16))
    rag_hip.check_faults()
    with pytest.raises(ValueError):
        rag_hip.enc_embed_layernorm_split_dev(ids, types, pos, word.bfloat16(), position, type_emb, gamma, beta, 1e-12, half)


def test_float16_cross_encoder_with_native_kernels_is_eight_times_closer_than_bfloat16():
    """The float16 instances of the 16-bit kernels (the screening pass of the rerank): packed forward against the
    float32 PyTorch forward, beside the bfloat16 instances on the same pairs."""
    from indexing import bert
    dev = torch.device('cuda:0')
    ref, _, cfg = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32, init='trained-scale')
    f16, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float16, native=True, init='trained-scale')
    b16, _, _ = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.bfloat16, native=True, init='trained-scale')
    g = torch.Generator(device=dev).manual_seed(9)
    b, s = 96, 120
    ids = torch.randint(1000, cfg.vocab_size, (b, s), generator=g, device=dev)
    lens = torch.randint(5, s + 1, (b,), generator=g, device=dev)
    mask = (torch.arange(s, device=dev)[None, :] < lens[:, None]).long()
    types = (torch.arange(s, device=dev)[None, :] >= (lens[:, None] // 3)).long() * mask
    p_ids, p_types, p_pos, cu, longest = ref.bert.pack(ids, mask, types)
    with torch.no_grad():
        a = ref(ids, types, mask)
        h = f16.forward_packed(p_ids, p_types, p_pos, cu, longest)
        l = b16.forward_packed(p_ids, p_types, p_pos, cu, longest)
    torch.cuda.synchronize()
    e16, eb = float((h - a).abs().max()), float((l - a).abs().max())
    assert h.dtype == torch.float32 and e16 < 0.35 * eb, (e16, eb)
    assert e16 < 0.02 * float(a.std()) + 1e-3, (e16, float(a.std()))


# ---- q|k|v projection + attention in one kernel ------------------------------------------------------------
def _qa_inputs(lens, dt, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    t = int(np.sum(lens))
    x = torch.randn((t, 384), generator=g, device=dev).to(dt)
    w = (torch.randn((1152, 384), generator=g, device=dev) * 0.05).to(dt)
    b = (torch.randn((1152,), generator=g, device=dev) * 0.1).to(dt)
    cu = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
    return x, w, b, cu


def _qa_exact(x, w, b, lens):
    """fp64 attention of the 16-bit operands (projection rounded to the 16-bit type as both GPU routes store it)."""
    qkv = (x.double() @ w.double().t() + b.double()).to(x.dtype).double()
    out = torch.zeros((x.shape[0], 384), dtype=torch.float64, device=x.device)
    p0 = 0
    for n in lens:
        n = int(n)
        q, k, v = (qkv[p0:p0 + n, i * 384:(i + 1) * 384].reshape(n, 12, 32).transpose(0, 1) for i in range(3))
        a = torch.softmax(q @ k.transpose(1, 2) / np.sqrt(32.0), dim=-1) @ v
        out[p0:p0 + n] = a.transpose(0, 1).reshape(n, 384)
        p0 += n
    return out


@pytest.mark.parametrize('dt,tol', [(torch.float16, 3e-3), (torch.bfloat16, 2.5e-2)])
@pytest.mark.parametrize('packed', [False, True])
def test_qkv_attention_in_one_kernel(dt, tol, packed):
    import rag_hip
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    rng = np.random.default_rng(5)
    lens = np.concatenate([[256, 1, 255, 1, 2, 31, 32, 33, 64, 128, 127, 129, 200, 56, 57],
                           rng.integers(1, 257, size=40), rng.integers(8, 90, size=120)]).astype(np.int32)
    if packed:
        order, tiles = rag_hip.enc_qkv_attention_pack(lens)
        assert sorted(order.tolist()) == list(range(len(lens)))
        lens = lens[order]
    else:
        tiles = rag_hip.enc_qkv_attention_tiles(np.concatenate([[0], np.cumsum(lens)]))
    cu_h = np.concatenate([[0], np.cumsum(lens)])
    assert tiles[0] == 0 and tiles[-1] == len(lens) and np.all(np.diff(tiles) > 0)
    assert int((cu_h[tiles[1:]] - cu_h[tiles[:-1]]).max()) <= 256
    if packed:                                               # nearly full tiles
        assert len(tiles) - 1 <= int(np.ceil(cu_h[-1] / 256 * 1.08)) + 1
    x, w, b, cu = _qa_inputs(lens, dt, 11, dev)
    got = rag_hip.enc_qkv_attention_dev(x, w, b, cu, torch.from_numpy(tiles).to(dev), 12)
    two = rag_hip.enc_attention_packed_dev(rag_hip.enc_linear384_dev(x, w, b), cu, int(lens.max()), 12)
    torch.cuda.synchronize()
    rag_hip.check_faults()
    ref = _qa_exact(x, w, b, lens)
    err, err_two = (got.double() - ref).abs(), (two.double() - ref).abs()
    assert float(err.max()) < tol and float(err.mean()) < tol / 20
    assert float(err.mean()) <= 1.1 * float(err_two.mean()) + 1e-6       # no less accurate than the two kernels
    assert got.dtype == dt and tuple(got.shape) == (x.shape[0], 384)
    # only the first token of every sequence as a query
    first = rag_hip.enc_qkv_attention_dev(x, w, b, cu, torch.from_numpy(tiles).to(dev), 12, first_only=True)
    torch.cuda.synchronize()
    rag_hip.check_faults()
    rows = torch.from_numpy(cu_h[:-1].astype(np.int64)).to(dev)
    assert tuple(first.shape) == (len(lens), 384)
    ef = (first.double() - ref[rows]).abs()
    assert float(ef.max()) < tol and float(ef.mean()) < tol / 20


def test_qkv_attention_rows_behind_the_batch_and_empty_tail():
    """x may hold more rows than the sequences cover (the packed route rounds its token count up); sequences without
    tokens behind the last tile are in no tile."""
    import rag_hip
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    lens = np.array([70, 90, 64, 33, 100, 0, 0], np.int32)
    order, tiles = rag_hip.enc_qkv_attention_pack(lens)
    assert order[-2:].tolist() == [5, 6] and tiles[-1] == 5
    lens = lens[order]
    x, w, b, cu = _qa_inputs(lens, torch.float16, 3, dev)
    pad = torch.zeros((59, 384), dtype=torch.float16, device=dev)
    xp = torch.cat([x, pad])
    live = cu[:6].contiguous()                               # the five sequences that have tokens
    got = rag_hip.enc_qkv_attention_dev(xp, w, b, live, torch.from_numpy(tiles).to(dev), 12)
    torch.cuda.synchronize()
    rag_hip.check_faults()
    ref = _qa_exact(x, w, b, lens[:5])
    assert float((got[:x.shape[0]].double() - ref).abs().max()) < 3e-3


def test_qkv_attention_refuses_what_is_not_a_tile():
    import rag_hip
    dev = torch.device('cuda:0')
    rag_hip.init(0)
    lens = np.array([200, 200, 10], np.int32)
    x, w, b, cu = _qa_inputs(lens, torch.float16, 4, dev)
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.enc_qkv_attention_tiles(np.array([0, 300], np.int64))       # a sequence longer than a tile
    assert rag_hip.enc_qkv_attention_pack(np.array([10, 300], np.int32)) is None
    bad = torch.tensor([0, 2, 3], dtype=torch.int32, device=dev)            # first tile: 400 tokens
    out = rag_hip.enc_qkv_attention_dev(x, w, b, cu, bad, 12)
    torch.cuda.synchronize()
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.check_faults()
    good = rag_hip.enc_qkv_attention_dev(x, w, b, cu, torch.tensor([0, 1, 3], dtype=torch.int32, device=dev), 12)
    torch.cuda.synchronize()
    rag_hip.check_faults()
    assert float((good[400:].double() - _qa_exact(x, w, b, lens)[400:]).abs().max()) < 3e-3
    with pytest.raises(ValueError):
        rag_hip.enc_qkv_attention_dev(x.float(), w, b, cu, bad, 12)
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.enc_qkv_attention_dev(x, w, b, cu, torch.tensor([0, 1, 3], dtype=torch.int32, device=dev), 6)   # heads of 64
