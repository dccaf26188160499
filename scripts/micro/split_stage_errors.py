"""Where the split-precision forward loses accuracy: each stage of one MiniLM-L6 layer (random weights, bench-shaped
pairs) computed with bf16 and fp16 halves, against float64 from the same float32 inputs."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'domain-rag-system_amd'))
import numpy as np, torch
import rag_hip
from indexing import bert
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
from ce_precision import make_pairs
dev = torch.device('cuda:0'); rag_hip.init(0)
m, _, cfg = bert.load_model('random:minilm-l6', 'cross', dev, dtype=torch.float32)
ids, types, pos, cu = make_pairs(32, 50, 5)
t = lambda parts: torch.from_numpy(np.concatenate(parts).astype(np.int32)).to(dev)
cu_d = torch.from_numpy(cu.astype(np.int32)).to(dev); longest = max(len(s) for s in ids)
b = m.bert; l = b.layers[0]
x = rag_hip.enc_embed_layernorm_dev(t(ids), t(types), t(pos), b.word.weight, b.position.weight, b.token_type.weight,
                                    b.emb_norm.weight, b.emb_norm.bias, b.emb_norm.eps)
def stat(a, ref):
    e = (a.double() - ref).abs().flatten()
    idx = torch.randint(0, e.numel(), (min(e.numel(), 2_000_000),), device=dev)
    return '%9.2e max %9.2e p999 %9.2e mean  (|ref| mean %.3f)' % (float(e.max()), float(torch.quantile(e[idx], 0.999)), float(e.mean()), float(ref.abs().mean()))
for name, half in (('bf16', torch.bfloat16), ('fp16', torch.float16)):
    print('==== halves:', name)
    w3 = lambda mod: rag_hip.split_weight_bf16x3(mod.weight.detach(), half)
    mm = lambda a3, mod: torch.mm(a3, w3(mod).t(), out_dtype=torch.float32)
    xs = rag_hip.enc_split_bf16x3_dev(x, half)
    H = x.shape[1]
    print('split x  ', stat(xs[:, :H].double() + xs[:, 2*H:].double(), x.double()))
    qkv = mm(xs, l.qkv)
    qkv_ref = x.double() @ l.qkv.weight.double().t()
    print('qkv gemm ', stat(qkv, qkv_ref))
    ctx3 = rag_hip.enc_attention_split_dev(qkv, l.qkv.bias, cu_d, longest, l.heads, half=half)
    ctx = ctx3[:, :H].double() + ctx3[:, 2*H:].double()
    # float64 attention from the same fp32 qkv
    full = (qkv + l.qkv.bias).double(); ref = torch.empty_like(full[:, :H]); cu_h = cu.tolist(); hd = H // l.heads
    for i in range(len(cu_h) - 1):
        z = full[cu_h[i]:cu_h[i+1]].view(-1, 3, l.heads, hd)
        q, k, v = z[:, 0].transpose(0, 1), z[:, 1].transpose(0, 1), z[:, 2].transpose(0, 1)
        p = torch.softmax(q @ k.transpose(1, 2) / hd ** 0.5, dim=-1)
        ref[cu_h[i]:cu_h[i+1]] = (p @ v).transpose(0, 1).reshape(-1, H)
    print('attention', stat(ctx, ref))
    y = mm(ctx3, l.attn_out)
    print('out gemm ', stat(y, ctx @ l.attn_out.weight.double().t()))
    x1, x1s = rag_hip.enc_bias_add_layernorm_split_dev(y, l.attn_out.bias, x, l.attn_norm.weight, l.attn_norm.bias, l.attn_norm.eps, half=half)
    x1_ref = torch.nn.functional.layer_norm(y.double() + l.attn_out.bias.double() + x.double(), (H,), l.attn_norm.weight.double(), l.attn_norm.bias.double(), l.attn_norm.eps)
    print('add+LN   ', stat(x1, x1_ref))
    h = mm(x1s, l.ffn_in)
    print('ffn1 gemm', stat(h, x1.double() @ l.ffn_in.weight.double().t()))
    hs = rag_hip.enc_bias_act_split_dev(h, l.ffn_in.bias, True, half)
    I = h.shape[1]
    g_ref = torch.nn.functional.gelu(h.double() + l.ffn_in.bias.double())
    print('gelu+splt', stat(hs[:, :I].double() + hs[:, 2*I:].double(), g_ref))
    e = (hs[:, :I].double() + hs[:, 2*I:].double() - g_ref).abs()
    for i in torch.topk(e.flatten(), 5).indices.tolist():
        r, c = divmod(i, I)
        print('   worst: row %d col %d  h %.9g  ref %.9g  hi %.9g  mid %.9g  lo %.9g' % (r, c, float(h[r, c]), float(g_ref[r, c]), float(hs[r, c]), float(hs[r, I + c]), float(hs[r, 2 * I + c])))
    y2 = mm(hs, l.ffn_out)
    print('ffn2 gemm', stat(y2, g_ref @ l.ffn_out.weight.double().t()))
