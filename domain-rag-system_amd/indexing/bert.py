"""BERT-family encoder forward passes in plain PyTorch (ROCm), plus checkpoint and
tokenizer loading from a local directory.

PyTorch is used here only for the two transformer forwards of the retrieval path: the
bi-encoder behind ``MedicalEmbedder.encode`` (reference ``src/indexing/embedder.py:65-71``,
a sentence-transformers pipeline: BERT -> mean pooling over the attention mask -> L2
normalise) and the cross-encoder behind ``CrossEncoderReranker.score_pairs`` (reference
``src/reranker/rerank.py:79-84``, a ``BertForSequenceClassification`` with one label).

Model specs understood by ``load_model``:
  * a local directory holding ``config.json`` + ``model.safetensors`` or
    ``pytorch_model.bin`` (+ ``tokenizer.json`` or ``vocab.txt``);
  * ``random:<arch>`` - seeded random weights of a named architecture and a hash
    tokenizer, for benchmarks and tests on machines without checkpoints
    (``bert-base``: 12 layers / 768 hidden; ``minilm-l6``: 6 layers / 384 hidden;
    ``tiny``: 2 layers / 64 hidden);
  * anything else is treated as a hub id and resolved from the local Hugging Face cache
    only; a miss raises (there is no silent fallback model).
"""
import json
import math
import os
import re
import zlib
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class BertConfig:
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    num_labels: int = 1

    @classmethod
    def from_json(cls, path):
        with open(path, 'r', encoding='utf-8') as f:
            raw = json.load(f)
        keys = cls.__dataclass_fields__.keys()
        cfg = cls(**{k: raw[k] for k in keys if k in raw})
        if 'id2label' in raw:
            cfg.num_labels = len(raw['id2label'])
        return cfg


ARCHS = {
    'bert-base': BertConfig(),
    'minilm-l6': BertConfig(hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536),
    'tiny': BertConfig(vocab_size=4096, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                       intermediate_size=128, max_position_embeddings=512),
}


class BertLayer(nn.Module):
    def __init__(self, c: BertConfig):
        super().__init__()
        self.heads = c.num_attention_heads
        self.qkv = nn.Linear(c.hidden_size, 3 * c.hidden_size)
        self.attn_out = nn.Linear(c.hidden_size, c.hidden_size)
        self.attn_norm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.ffn_in = nn.Linear(c.hidden_size, c.intermediate_size)
        self.ffn_out = nn.Linear(c.intermediate_size, c.hidden_size)
        self.ffn_norm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)

    def forward(self, x, attn_bias, lens=None):
        """`lens` (int32 [B], real tokens per sequence) selects the librag_hip kernels for what
        sits between the GEMMs: fused attention straight from the packed projection, and
        residual + LayerNorm in one pass.  Without it (CPU, or a shape the kernels do not
        cover) the same arithmetic runs as PyTorch ops."""
        b, s, h = x.shape
        if lens is not None:
            import rag_hip
            ctx = rag_hip.enc_attention_dev(self.qkv(x), lens, self.heads)
            x = rag_hip.enc_add_layernorm_dev(self.attn_out(ctx), x, self.attn_norm.weight, self.attn_norm.bias,
                                              self.attn_norm.eps)
            return rag_hip.enc_add_layernorm_dev(self.ffn_out(rag_hip.enc_gelu_(self.ffn_in(x))), x, self.ffn_norm.weight,
                                                 self.ffn_norm.bias, self.ffn_norm.eps)
        qkv = self.qkv(x).view(b, s, 3, self.heads, h // self.heads).permute(2, 0, 3, 1, 4)
        ctx = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=attn_bias)
        ctx = ctx.transpose(1, 2).reshape(b, s, h)
        x = self.attn_norm(x + self.attn_out(ctx))
        return self.ffn_norm(x + self.ffn_out(F.gelu(self.ffn_in(x))))


class BertEncoder(nn.Module):
    def __init__(self, c: BertConfig):
        super().__init__()
        self.config = c
        self.word = nn.Embedding(c.vocab_size, c.hidden_size)
        self.position = nn.Embedding(c.max_position_embeddings, c.hidden_size)
        self.token_type = nn.Embedding(c.type_vocab_size, c.hidden_size)
        self.emb_norm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.layers = nn.ModuleList([BertLayer(c) for _ in range(c.num_hidden_layers)])
        self.native = False          # set by load_model(native=True): use the librag_hip encoder kernels
        self.split_gemm = False      # fp32 weights only: run the GEMMs as split-bf16 products (see _linear_split)
        self._split_cache = {}
        self.fused_split_tails = True   # with split_gemm: one kernel for the work between two GEMMs
        self.split_attention = True     # with split_gemm: attention as split-precision MFMA products (rag_enc_attention_split_dev)
        self.split_dtype = torch.bfloat16   # 16-bit type of the split halves: bfloat16 (16 bits between them) or float16 (22)
        self.trim_last_q = os.environ.get('RAG_TRIM_LAST_Q', '1') != '0'   # first_token_only: last layer projects queries for the first rows only
        self.fused_split_out = True     # fp16 halves, hidden 384: output projection / second feed-forward product with
        #                                 everything around them in one kernel (rag_enc_linear_ln_split_dev)
        self.native_linear = True    # bfloat16, hidden 384: q|k|v and output projections through rag_enc_linear384_dev
        self.fused_attn_norm = False # ... with the residual + LayerNorm after the output projection in the same kernel
        self.fused_mlp = True        # bfloat16, hidden 384: feed-forward block in one kernel (rag_enc_mlp_dev)
        self.fused_norm_mlp = False  # ... with the residual + LayerNorm in front of it in the same kernel (rag_enc_norm_mlp_dev): no faster
        self.fused_tail = False      # ... and the attention-output projection + LayerNorm in the same kernel
        self.fused_qkv_attention = True   # 16-bit, hidden 384, head size 32, a tile list given: q|k|v projection and
        #                                   self-attention in one kernel (rag_enc_qkv_attention_dev)

    @staticmethod
    def pack(input_ids, attention_mask, token_type_ids=None):
        """Padded [B, S] ids + left-aligned mask -> the unpadded stream forward_packed takes:
        (ids, types, pos int32 [T], cu_seqlens int32 [B+1], longest).  One host round trip."""
        keep = attention_mask.bool()
        lens = keep.sum(dim=1)
        cu = torch.zeros((input_ids.shape[0] + 1,), dtype=torch.int32, device=input_ids.device)
        cu[1:] = torch.cumsum(lens, 0)
        pos = torch.arange(input_ids.shape[1], device=input_ids.device, dtype=torch.int32)[None, :].expand_as(input_ids)
        types = token_type_ids if token_type_ids is not None else torch.zeros_like(input_ids)
        return (input_ids[keep].to(torch.int32), types[keep].to(torch.int32), pos[keep].contiguous(), cu,
                int(lens.max()) if lens.numel() else 0)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None):
        b, s = input_ids.shape
        pos = torch.arange(s, device=input_ids.device)
        x = self.word(input_ids) + self.position(pos)[None, :, :]
        x = x + self.token_type(token_type_ids if token_type_ids is not None else torch.zeros_like(input_ids))
        x = self.emb_norm(x)
        c = self.config
        if self.native and x.is_cuda:
            import rag_hip
            if not rag_hip.enc_attention_supported(x.dtype, c.hidden_size // c.num_attention_heads, s):
                raise ValueError(f"the librag_hip encoder kernels do not cover dtype {x.dtype}, head size "
                                 f"{c.hidden_size // c.num_attention_heads}, sequence length {s}")
            # BERT masks by position validity; tokenised batches are left-aligned, so a length per
            # sequence carries the same information as the mask
            lens = (attention_mask.sum(dim=1) if attention_mask is not None
                    else torch.full((b,), s, device=x.device)).to(torch.int32)
            for layer in self.layers:
                x = layer(x, None, lens)
            return x
        bias = None
        if attention_mask is not None:
            bias = torch.zeros((b, 1, 1, s), dtype=x.dtype, device=x.device)
            bias.masked_fill_(attention_mask[:, None, None, :] == 0, torch.finfo(x.dtype).min)
        for layer in self.layers:
            x = layer(x, bias)
        return x


    def forward_packed(self, ids, types, pos, cu_seqlens, max_seq_len, first_token_only=False, tiles=None):
        """Unpadded batch on the GPU: ids/types/pos int32 [T], sequence i = tokens
        [cu_seqlens[i], cu_seqlens[i+1]); tiles (int32 on the device, rag_hip.enc_qkv_attention_tiles / _pack of the same
        sequences) lets 16-bit 384-wide layers run projection and attention in one kernel.  Returns hidden states [T, H], or with
        first_token_only (bfloat16) just the first token's row of every sequence [n_seq, H]:
        the last layer then computes its output projection, feed-forward block and LayerNorms
        for those rows only (identical values; a classifier head reads nothing else).
        Everything between the GEMMs runs in librag_hip kernels (bfloat16 or float32 weights)."""
        import rag_hip
        c = self.config
        if not rag_hip.enc_attention_supported(self.word.weight.dtype, c.hidden_size // c.num_attention_heads,
                                               max_seq_len):
            raise ValueError("the packed encoder path needs bfloat16 or float32 weights, head size 32 or 64 and "
                             f"sequences up to {rag_hip.ENC_MAX_SEQ} tokens")
        if self.split_gemm and self.fused_split_tails and self.word.weight.dtype == torch.float32:
            # embedding + LayerNorm + the split for the first GEMM in one pass
            # (where the layers keep the residual stream in the split rows, nothing reads the fp32 row)
            fused_out, chain = self._fused_route(ids.shape[0])
            x, xs = rag_hip.enc_embed_layernorm_split_dev(ids, types, pos, self.word.weight, self.position.weight,
                                                          self.token_type.weight, self.emb_norm.weight, self.emb_norm.bias,
                                                          self.emb_norm.eps, self.split_dtype,
                                                          want_out=not (fused_out and chain))
            return self._forward_packed_split(x, cu_seqlens, max_seq_len, first_token_only, xs)
        x = rag_hip.enc_embed_layernorm_dev(ids, types, pos, self.word.weight, self.position.weight,
                                            self.token_type.weight, self.emb_norm.weight, self.emb_norm.bias,
                                            self.emb_norm.eps)
        lin = self._linear_split if (self.split_gemm and x.dtype == torch.float32) else self._linear
        last = len(self.layers) - 1
        one_kernel = (tiles is not None and self.fused_qkv_attention and self.native_linear
                      and x.dtype in (torch.bfloat16, torch.float16) and c.hidden_size == 384
                      and c.num_attention_heads * 32 == 384)
        for li, l in enumerate(self.layers):
            if first_token_only and li == last and x.dtype in (torch.bfloat16, torch.float16):
                if one_kernel:
                    ctx = rag_hip.enc_qkv_attention_dev(x.contiguous(), l.qkv.weight, l.qkv.bias, cu_seqlens, tiles, l.heads,
                                                        first_only=True)
                else:
                    ctx = rag_hip.enc_attention_packed_first_dev(lin(l.qkv, x), cu_seqlens, max_seq_len, l.heads)
                x = x[cu_seqlens[:-1].long()]
                if self._layer_tail_fused(l, x):
                    return self._layer_tail(l, ctx, x)
                x = self._attn_output(l, ctx, x, lin)
                return self._feed_forward(l, x, lin)
            if one_kernel:
                ctx = rag_hip.enc_qkv_attention_dev(x.contiguous(), l.qkv.weight, l.qkv.bias, cu_seqlens, tiles, l.heads)
            else:
                ctx = rag_hip.enc_attention_packed_dev(lin(l.qkv, x), cu_seqlens, max_seq_len, l.heads)
            if self._layer_tail_fused(l, x):
                x = self._layer_tail(l, ctx, x)
                continue
            if self._norm_mlp_fused(l, x, ctx):
                # output projection, then residual + LayerNorm + feed-forward block in one kernel
                x = rag_hip.enc_norm_mlp_dev(self._linear(l.attn_out, ctx), x.contiguous(), l.attn_norm.weight,
                                             l.attn_norm.bias, l.ffn_in.weight, l.ffn_in.bias, l.ffn_out.weight,
                                             l.ffn_out.bias, l.ffn_norm.weight, l.ffn_norm.bias, l.ffn_norm.eps)
                continue
            x = self._attn_output(l, ctx, x, lin)
            x = self._feed_forward(l, x, lin)
        return x

    def _linear(self, module, x):
        """nn.Linear; 384-wide bfloat16 inputs go through rag_enc_linear384_dev."""
        import rag_hip
        if self.native_linear and x.dim() == 2 and rag_hip.enc_linear384_supported(x, module.weight):
            return rag_hip.enc_linear384_dev(x.contiguous(), module.weight, module.bias)
        return module(x)

    def _attn_output(self, l, ctx, x, lin):
        """LayerNorm(x + attention-output projection of ctx): one kernel for 384-wide bfloat16
        layers (projection with the residual + LayerNorm tail), else GEMM + residual/LayerNorm kernel."""
        import rag_hip
        w = l.attn_out.weight
        if (self.native_linear and self.fused_attn_norm and ctx.dim() == 2 and w.shape[0] == 384
                and rag_hip.enc_linear384_supported(ctx, w)):
            return rag_hip.enc_linear384_dev(ctx.contiguous(), w, l.attn_out.bias, x.contiguous(), l.attn_norm.weight,
                                             l.attn_norm.bias, l.attn_norm.eps)
        return rag_hip.enc_add_layernorm_dev(lin(l.attn_out, ctx), x, l.attn_norm.weight, l.attn_norm.bias,
                                             l.attn_norm.eps)

    def _layer_tail_fused(self, l, x):
        import rag_hip
        return (self.fused_mlp and self.fused_tail and l.attn_norm.eps == l.ffn_norm.eps
                and rag_hip.enc_mlp_supported(x, l.ffn_in.weight, l.ffn_out.weight)
                and l.attn_out.weight.dtype in (torch.bfloat16, torch.float16))

    def _norm_mlp_fused(self, l, x, ctx):
        import rag_hip
        return (self.fused_mlp and self.fused_norm_mlp and self.native_linear and ctx.dim() == 2
                and l.attn_norm.eps == l.ffn_norm.eps
                and rag_hip.enc_mlp_supported(x, l.ffn_in.weight, l.ffn_out.weight)
                and rag_hip.enc_linear384_supported(ctx, l.attn_out.weight))

    def _layer_tail(self, l, ctx, x):
        """Everything after self-attention (output projection, residual + LayerNorm, feed-forward
        block, residual + LayerNorm) in one kernel."""
        import rag_hip
        return rag_hip.enc_attn_out_mlp_dev(ctx, x.contiguous(), l.attn_out.weight, l.attn_out.bias, l.attn_norm.weight,
                                            l.attn_norm.bias, l.ffn_in.weight, l.ffn_in.bias, l.ffn_out.weight,
                                            l.ffn_out.bias, l.ffn_norm.weight, l.ffn_norm.bias, l.ffn_norm.eps)

    def _feed_forward(self, l, x, lin):
        """LayerNorm(x + W2 GELU(W1 x)): one kernel where rag_enc_mlp_dev covers the shape, else
        library GEMMs with the GELU and residual+LayerNorm kernels between them."""
        import rag_hip
        if self.fused_mlp and rag_hip.enc_mlp_supported(x, l.ffn_in.weight, l.ffn_out.weight):
            return rag_hip.enc_mlp_dev(x.contiguous(), l.ffn_in.weight, l.ffn_in.bias, l.ffn_out.weight, l.ffn_out.bias,
                                       l.ffn_norm.weight, l.ffn_norm.bias, l.ffn_norm.eps)
        return rag_hip.enc_add_layernorm_dev(lin(l.ffn_out, rag_hip.enc_gelu_(lin(l.ffn_in, x))), x, l.ffn_norm.weight,
                                             l.ffn_norm.bias, l.ffn_norm.eps)

    def _split_weight(self, module):
        import rag_hip
        w3 = self._split_cache.get(id(module))
        if w3 is None:
            w3 = rag_hip.split_weight_bf16x3(module.weight.detach(), self.split_dtype)
            self._split_cache[id(module)] = w3
        return w3

    def _weight_halves(self, module):
        import rag_hip
        key = ('halves', id(module))
        w = self._split_cache.get(key)
        if w is None:
            w = rag_hip.split_weight_halves(module.weight.detach())
            self._split_cache[key] = w
        return w

    def _fused_route(self, n_tokens):
        """(whether a batch of n_tokens runs its layers through the fused output kernels, whether those keep the residual
        stream in the split rows)"""
        import rag_hip
        c = self.config
        # (a 128-token tile of the fused output kernel takes ~100 us however empty the chip is: below ~96 tiles the
        # library GEMM + tail kernels, which spread a small batch over the chip, are faster: 1.08 against 1.53 ms for
        # one query's 50 pairs, equal at 14 k tokens, slower from there on)
        fused_out = (self.fused_split_out and n_tokens >= int(os.environ.get('RAG_FUSED_OUT_MIN_TOKENS', '12288'))
                     and rag_hip.enc_linear_ln_split_supported(c.hidden_size, c.hidden_size, self.split_dtype)
                     and rag_hip.enc_linear_ln_split_supported(c.intermediate_size, c.hidden_size, self.split_dtype))
        return fused_out, os.environ.get('RAG_CHAIN_RESIDUAL', '1') != '0'

    def _forward_packed_split(self, x, cu_seqlens, max_seq_len, first_token_only=False, xs=None):
        """The fp32 layers with every GEMM as a split-bf16 product (see _linear_split) and the work between
        two GEMMs in one kernel each: bias + residual + LayerNorm + split of the result, bias + GELU + split.
        Same arithmetic in the same order as the separate kernels, so the same hidden states bit for bit.
        Attention reads the fp32 q|k|v projection (its bias added on the way in) and writes the [hi|hi|lo]
        operand of the output projection directly where the MFMA attention kernel covers the head size.
        first_token_only: the last layer computes everything after its attention for the first row of every
        sequence only ([n_seq, H] comes back); a classifier head reads nothing else."""
        import rag_hip
        mm = lambda a3, module: torch.mm(a3, self._split_weight(module).t(), out_dtype=torch.float32)
        half = self.split_dtype
        if xs is None:
            xs = rag_hip.enc_split_bf16x3_dev(x, half)
        last = len(self.layers) - 1
        head_dim = self.config.hidden_size // self.config.num_attention_heads
        mfma_attn = self.split_attention and rag_hip.enc_attention_split_supported(head_dim, max_seq_len)
        c = self.config
        fused_out, chain = self._fused_route(xs.shape[0])       # (x is None where nothing reads the fp32 rows)
        # (the library's GEMM tiles are 256 columns wide: [T, 1152] pads its last tile by half, [T, 768] + [T, 384] ran
        # 3-4 ms of 277 faster on 750 k tokens; 512 | 640 and three products of 384 were no faster, five of 256 slower)
        qkv_cuts = os.environ.get('RAG_QKV_SPLIT_N')
        if qkv_cuts is None:
            qkv_cuts = [2 * c.hidden_size] if (fused_out and c.hidden_size % 128 == 0 and c.hidden_size % 256) else []
        else:
            qkv_cuts = [int(v) for v in qkv_cuts.split(',') if int(v) > 0]
        for li, l in enumerate(self.layers):
            cls_only = first_token_only and li == last
            if mfma_attn:
                # q|k|v bias inside the attention kernel; output already split for the next GEMM
                if cls_only and self.trim_last_q and xs.shape[0] > 4 * (cu_seqlens.shape[0] - 1):
                    # only the first row of every sequence is a query in this layer: keys and values for every token
                    # (the GEMM writes the k|v columns of the usual [T, 3H] layout in place), queries for those rows
                    w3 = self._split_weight(l.qkv)
                    hdim = c.hidden_size
                    qkv = torch.empty((xs.shape[0], 3 * hdim), dtype=torch.float32, device=xs.device)
                    torch.mm(xs, w3[hdim:].t(), out_dtype=torch.float32, out=qkv[:, hdim:])
                    first = cu_seqlens[:-1].long()
                    qkv[first, :hdim] = torch.mm(xs[first], w3[:hdim].t(), out_dtype=torch.float32)
                elif qkv_cuts:
                    # q|k apart from v: each product's width is then a whole number of the library's 256-wide tiles
                    w3 = self._split_weight(l.qkv)
                    qkv = torch.empty((xs.shape[0], w3.shape[0]), dtype=torch.float32, device=xs.device)
                    for a, z in zip([0] + qkv_cuts, qkv_cuts + [w3.shape[0]]):
                        torch.mm(xs, w3[a:z].t(), out_dtype=torch.float32, out=qkv[:, a:z])
                else:
                    qkv = mm(xs, l.qkv)
                # (the fused output kernel reads the halves where they lie: [hi | lo] rows, a third less for attention to
                # write; a library GEMM needs [hi | hi | lo])
                ctx3 = rag_hip.enc_attention_split_dev(qkv, l.qkv.bias, cu_seqlens, max_seq_len, l.heads,
                                                       first_only=cls_only, half=half,
                                                       hi_lo=fused_out and os.environ.get('RAG_ATT_HI_LO', '1') != '0')
            else:
                qkv = mm(xs, l.qkv).add_(l.qkv.bias)
                ctx = rag_hip.enc_attention_packed_dev(qkv, cu_seqlens, max_seq_len, l.heads)
                if cls_only:
                    ctx = ctx[cu_seqlens[:-1].long()]
                ctx3 = rag_hip.enc_split_bf16x3_dev(ctx, half)
            if fused_out and chain:
                # the residual stream stays in its split form between the fused kernels: hi + lo of the [hi | hi | lo] rows they
                # write anyway (the value the GEMMs are given), so no fp32 row is written or read until the last one (the
                # fp32 row's stores are 9 % of the output projection and 4 % of the second product)
                res = xs[cu_seqlens[:-1].long()] if cls_only else xs
                _, xs = rag_hip.enc_linear_ln_split_dev(ctx3, None, *self._weight_halves(l.attn_out), l.attn_out.bias, res,
                                                        l.attn_norm.weight, l.attn_norm.bias, l.attn_norm.eps, want_y=False)
                x, xs = rag_hip.enc_linear_ln_split_dev(mm(xs, l.ffn_in), l.ffn_in.bias, *self._weight_halves(l.ffn_out),
                                                        l.ffn_out.bias, xs, l.ffn_norm.weight, l.ffn_norm.bias,
                                                        l.ffn_norm.eps, want_split=li != last, want_y=li == last)
                continue
            if cls_only:
                x = x[cu_seqlens[:-1].long()].contiguous()
            if fused_out:
                # projection + bias + residual + LayerNorm + split in one kernel; then the first feed-forward product
                # as a library GEMM, and bias + GELU + second product + bias + residual + LayerNorm + split in one
                x, xs = rag_hip.enc_linear_ln_split_dev(ctx3, None, *self._weight_halves(l.attn_out), l.attn_out.bias, x,
                                                        l.attn_norm.weight, l.attn_norm.bias, l.attn_norm.eps)
                x, xs = rag_hip.enc_linear_ln_split_dev(mm(xs, l.ffn_in), l.ffn_in.bias, *self._weight_halves(l.ffn_out),
                                                        l.ffn_out.bias, x, l.ffn_norm.weight, l.ffn_norm.bias,
                                                        l.ffn_norm.eps, want_split=li != last)
                continue
            x, xs = rag_hip.enc_bias_add_layernorm_split_dev(mm(ctx3, l.attn_out), l.attn_out.bias, x,
                                                             l.attn_norm.weight, l.attn_norm.bias, l.attn_norm.eps, half=half)
            hs = rag_hip.enc_bias_act_split_dev(mm(xs, l.ffn_in), l.ffn_in.bias, True, half)
            x, xs = rag_hip.enc_bias_add_layernorm_split_dev(mm(hs, l.ffn_out), l.ffn_out.bias, x, l.ffn_norm.weight,
                                                             l.ffn_norm.bias, l.ffn_norm.eps, want_split=li != last, half=half)
        return x

    def _linear_split(self, module, x):
        """fp32 Linear on the bf16 matrix cores: three bf16 products accumulated in fp32 inside one
        GEMM (operands split into bf16 high and low parts and laid side by side along K).  The
        result differs from the fp32 GEMM by about 1e-5 relative, far inside the 1e-4 the
        embeddings have to hold on cosine scores, at a fraction of the fp32 matrix-core time."""
        import rag_hip
        y = torch.mm(rag_hip.enc_split_bf16x3_dev(x, self.split_dtype), self._split_weight(module).t(), out_dtype=torch.float32)
        return y.add_(module.bias)


class Graphed:
    """fn(*inputs) -> tensor, captured once in a HIP graph and replayed: for the launch-bound encoder forwards of very
    small batches (a single query is ~100 launches of the bi-encoder and ~45 of the cross-encoder, each of them a few
    microseconds of GPU work behind 10-20 of host work).  Shapes are static: inputs are copied into the captured
    buffers, the result is the captured output buffer (valid until the next call).  Everything fn launches must be
    capturable: no host synchronisation, no allocation outside PyTorch's allocator (the librag_hip entry points it
    uses only launch kernels; work buffers and per-device LDS grants are settled by the warm-up calls)."""

    def __init__(self, fn, inputs):
        self.fn = fn                                      # keeps alive whatever the captured kernels read through it
        self.inputs = [t.clone() for t in inputs]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                fn(*self.inputs)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = fn(*self.inputs)

    def __call__(self, *inputs):
        for dst, src in zip(self.inputs, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.out


class GraphCache:
    """Captured graphs by shape key, at most `limit` of them (the oldest is dropped); a key whose capture failed
    once is remembered and never tried again."""

    def __init__(self, limit=32):
        self.limit = limit
        self.graphs = {}

    def get(self, key, fn, inputs):
        if key in self.graphs:
            return self.graphs[key]
        if os.environ.get('RAG_GRAPHS') == '0' or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return None
        try:
            g = Graphed(fn, inputs)
        except Exception as e:                            # noqa: BLE001 - any capture problem: run eagerly from now on
            import warnings
            warnings.warn(f"HIP graph capture of {key[0]} failed ({type(e).__name__}: {e}); running it eagerly")
            torch.cuda.synchronize()
            g = None
        while len(self.graphs) >= self.limit:
            self.graphs.pop(next(iter(self.graphs)))
        self.graphs[key] = g
        return g


def forward_packed_graphed(model, cache, ids, types, pos, cu, longest, **kw):
    """model.forward_packed(ids, types, pos, cu, longest, **kw) replayed from a HIP graph: the token stream is padded
    to a step of 256 tokens (2048 beyond 4096) and the longest sequence to a step of 64, the sequences themselves run
    at their real lengths.  Returns None when no graph is available (the caller then calls forward_packed)."""
    t = int(ids.shape[0])
    step = 256 if t <= 4096 else 2048
    t_pad = (t + step) // step * step                     # at least one token beyond the stream
    max_len = (int(longest) + 63) // 64 * 64
    n_seq = int(cu.shape[0]) - 1

    def fwd(i, ty, p, c):
        return model.forward_packed(i, ty, p, c, max_len, **kw)

    def padded(x):
        out = torch.zeros((t_pad,), dtype=torch.int32, device=x.device)
        out[:t] = x
        return out
    inputs = [padded(ids), padded(types), padded(pos), cu]
    g = cache.get((type(model).__name__, n_seq, t_pad, max_len, tuple(sorted(kw.items()))), fwd, inputs)
    if g is None:
        return None
    return g(*inputs).clone()


class BiEncoder(nn.Module):
    """BERT -> mean pooling over real tokens -> (optional) L2 normalisation."""

    def __init__(self, c: BertConfig):
        super().__init__()
        self.bert = BertEncoder(c)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, normalize=True):
        h = self.bert(input_ids, token_type_ids, attention_mask).float()
        if attention_mask is None:
            pooled = h.mean(dim=1)
        else:
            m = attention_mask.to(h.dtype)[:, :, None]
            pooled = (h * m).sum(dim=1) / m.sum(dim=1).clamp(min=1e-9)
        return F.normalize(pooled, p=2, dim=1) if normalize else pooled

    def forward_packed(self, ids, types, pos, cu_seqlens, max_seq_len, normalize=True):
        """Sentence embeddings of an unpadded batch (float32 weights): no padding position is
        computed; everything between the GEMMs runs in librag_hip kernels."""
        import rag_hip
        h = self.bert.forward_packed(ids, types, pos, cu_seqlens, max_seq_len)
        return rag_hip.enc_mean_pool_dev(h.float(), cu_seqlens, normalize)


class CrossEncoderModel(nn.Module):
    """BERT -> tanh pooler on [CLS] -> linear head; one logit per pair when num_labels == 1."""

    def __init__(self, c: BertConfig):
        super().__init__()
        self.bert = BertEncoder(c)
        self.pooler = nn.Linear(c.hidden_size, c.hidden_size)
        self.classifier = nn.Linear(c.hidden_size, c.num_labels)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None):
        h = self.bert(input_ids, token_type_ids, attention_mask)
        logits = self.classifier(torch.tanh(self.pooler(h[:, 0])))
        return logits.float().squeeze(-1) if logits.shape[-1] == 1 else logits.float()

    def forward_packed(self, ids, types, pos, cu_seqlens, max_seq_len, tiles=None):
        """Logits of an unpadded batch (every sequence non-empty, first token [CLS])."""
        b = self.bert
        if b.word.weight.dtype in (torch.bfloat16, torch.float16) or (b.split_gemm and b.fused_split_tails):
            cls = b.forward_packed(ids, types, pos, cu_seqlens, max_seq_len, first_token_only=True, tiles=tiles)
        else:
            cls = b.forward_packed(ids, types, pos, cu_seqlens, max_seq_len)[cu_seqlens[:-1].long()]
        logits = self.classifier(torch.tanh(self.pooler(cls)))
        return logits.float().squeeze(-1) if logits.shape[-1] == 1 else logits.float()


# ---- checkpoints ---------------------------------------------------------------------------
def _read_weights(directory):
    st = os.path.join(directory, 'model.safetensors')
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    pt = os.path.join(directory, 'pytorch_model.bin')
    if os.path.exists(pt):
        return torch.load(pt, map_location='cpu', weights_only=True)
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {directory}")


def _convert_hf_state(sd, n_layers):
    """Hugging Face BertModel / BertForSequenceClassification names -> this module's."""
    sd = {re.sub(r'^(bert\.|0\.auto_model\.|auto_model\.)', '', k): v for k, v in sd.items()}
    out = {}
    e = 'embeddings.'
    out['bert.word.weight'] = sd[e + 'word_embeddings.weight']
    out['bert.position.weight'] = sd[e + 'position_embeddings.weight']
    out['bert.token_type.weight'] = sd[e + 'token_type_embeddings.weight']
    out['bert.emb_norm.weight'] = sd[e + 'LayerNorm.weight']
    out['bert.emb_norm.bias'] = sd[e + 'LayerNorm.bias']
    for i in range(n_layers):
        s, d = f'encoder.layer.{i}.', f'bert.layers.{i}.'
        for part in ('weight', 'bias'):
            out[d + 'qkv.' + part] = torch.cat([sd[s + f'attention.self.{n}.{part}'] for n in ('query', 'key', 'value')])
            out[d + 'attn_out.' + part] = sd[s + 'attention.output.dense.' + part]
            out[d + 'attn_norm.' + part] = sd[s + 'attention.output.LayerNorm.' + part]
            out[d + 'ffn_in.' + part] = sd[s + 'intermediate.dense.' + part]
            out[d + 'ffn_out.' + part] = sd[s + 'output.dense.' + part]
            out[d + 'ffn_norm.' + part] = sd[s + 'output.LayerNorm.' + part]
    for part in ('weight', 'bias'):
        if 'pooler.dense.' + part in sd:
            out['pooler.' + part] = sd['pooler.dense.' + part]
        if 'classifier.' + part in sd:
            out['classifier.' + part] = sd['classifier.' + part]
    return out


# ---- tokenizers ----------------------------------------------------------------------------
class HashTokenizer:
    """Deterministic stand-in when no vocabulary exists: lower-cased ``\\w+`` words hashed
    into the id range above the special tokens."""
    cls_id, sep_id, pad_id = 101, 102, 0
    _word = re.compile(r'\w+')

    def __init__(self, vocab_size):
        if vocab_size <= 200:
            raise ValueError("hash tokenizer needs a vocabulary above the special-token ids")
        self.vocab_size = vocab_size
        self.base = min(1000, vocab_size // 2)

    def encode(self, text):
        span = self.vocab_size - self.base
        return [self.base + zlib.crc32(w.encode('utf-8')) % span for w in self._word.findall(text.lower())]


class FileTokenizer:
    """WordPiece tokenizer read from ``tokenizer.json`` or ``vocab.txt`` (``tokenizers`` lib)."""

    def __init__(self, directory):
        from tokenizers import Tokenizer, BertWordPieceTokenizer
        tj = os.path.join(directory, 'tokenizer.json')
        if os.path.exists(tj):
            self.tok = Tokenizer.from_file(tj)
            self.tok.no_padding()
            self.tok.no_truncation()
        else:
            vocab = os.path.join(directory, 'vocab.txt')
            if not os.path.exists(vocab):
                raise FileNotFoundError(f"no tokenizer.json / vocab.txt under {directory}")
            lower = True
            cfg = os.path.join(directory, 'tokenizer_config.json')
            if os.path.exists(cfg):
                with open(cfg, 'r', encoding='utf-8') as f:
                    lower = json.load(f).get('do_lower_case', True)
            self.tok = BertWordPieceTokenizer(vocab, lowercase=lower)
        self.cls_id = self.tok.token_to_id('[CLS]')
        self.sep_id = self.tok.token_to_id('[SEP]')
        self.pad_id = self.tok.token_to_id('[PAD]') or 0

    def encode(self, text):
        return self.tok.encode(text, add_special_tokens=False).ids


def _resolve_dir(spec):
    if os.path.isdir(spec):
        return spec
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(spec, local_files_only=True)
    except Exception as e:
        raise FileNotFoundError(
            f"model {spec!r} is not a local directory and is not in the local Hugging Face cache "
            f"({type(e).__name__}); pass a directory or a 'random:<arch>' spec") from e


# Weight initialisations of the `random:<arch>` models.  'hf-default' is what transformers gives an untrained model
# (every matrix N(0, 0.02), no biases, unit LayerNorm gains): attention is nearly uniform and the 50 logits of a query
# differ by ~1e-3.  'trained-scale' has the magnitudes of a trained checkpoint (matrix gains around one, biases,
# LayerNorm gains spread around one), so that every term of the forward moves the output and logits spread over units.
INITS = ('hf-default', 'trained-scale')


def _init_random(model, kind, seed, init):
    gen = torch.Generator().manual_seed(seed + (0 if kind == 'bi' else 1))
    with torch.no_grad():
        for name, p in model.named_parameters():
            if init == 'hf-default':
                if name.endswith('norm.weight'):
                    p.fill_(1.0)
                elif name.endswith('bias'):
                    p.zero_()
                else:
                    p.copy_(torch.randn(p.shape, generator=gen) * 0.02)
            else:
                if name.endswith('norm.weight'):
                    p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=gen))
                elif name.endswith('bias'):
                    p.copy_(0.2 * torch.randn(p.shape, generator=gen))
                elif p.dim() == 2 and name.split('.')[-2] in ('word', 'position', 'token_type'):
                    p.copy_(0.5 * torch.randn(p.shape, generator=gen))
                else:
                    p.copy_(torch.randn(p.shape, generator=gen) / math.sqrt(p.shape[-1]))


def load_model(spec, kind, device, dtype=torch.float32, seed=0, native=False, split_gemm=False, split_dtype=None,
               init='hf-default'):
    """kind: 'bi' or 'cross'.  Returns (module in eval mode on `device`, tokenizer, config).
    native=True runs attention and residual+LayerNorm through the librag_hip kernels (GPU,
    bfloat16 or float32, left-aligned attention masks; shapes they do not cover raise);
    native='auto' does so whenever the model sits on a GPU and has a head size the kernels
    cover.  split_gemm=True (float32 weights, packed forward) runs the GEMMs and attention as
    split-precision products on the 16-bit matrix cores with fp32 accumulation; split_dtype =
    torch.bfloat16 (default: results to ~1e-5 relative) or torch.float16 (to fp32 round-off;
    activations and weights must stay below 65504).  init: how a `random:` model's weights are drawn (INITS)."""
    cls = BiEncoder if kind == 'bi' else CrossEncoderModel
    if spec.startswith('random:'):
        arch = spec.split(':', 1)[1]
        if arch not in ARCHS:
            raise ValueError(f"unknown random architecture {arch!r}; known: {sorted(ARCHS)}")
        cfg = ARCHS[arch]
        model = cls(cfg)
        if init not in INITS:
            raise ValueError(f"unknown initialisation {init!r}; known: {INITS}")
        _init_random(model, kind, seed, init)
        tok = HashTokenizer(cfg.vocab_size)
    else:
        directory = _resolve_dir(spec)
        cfg = BertConfig.from_json(os.path.join(directory, 'config.json'))
        model = cls(cfg)
        state = _convert_hf_state(_read_weights(directory), cfg.num_hidden_layers)
        missing, unexpected = model.load_state_dict(state, strict=False)
        bad = [m for m in missing if not (kind == 'bi' and m.startswith(('pooler', 'classifier')))]
        if bad:
            raise RuntimeError(f"checkpoint under {directory} lacks weights: {bad[:5]}")
        tok = FileTokenizer(directory)
    model.eval().to(device=device, dtype=dtype)
    if native == 'auto':            # on the GPU whenever the kernels cover the architecture
        native = (torch.device(device).type == 'cuda' and dtype in (torch.bfloat16, torch.float16, torch.float32)
                  and cfg.hidden_size // cfg.num_attention_heads in (32, 64))
    model.bert.native = bool(native)
    # float16 weights (the screening forward of the batched rerank): residual + LayerNorm behind the output projection
    # inside the feed-forward kernel (one pass and one rounding fewer: 155.1 against 156.8 ms of rerank per 1024 queries)
    model.bert.fused_norm_mlp = dtype == torch.float16
    model.bert.split_gemm = bool(split_gemm)
    if split_dtype is not None:
        model.bert.split_dtype = split_dtype
    return model, tok, cfg


def pad_batch(seqs, pad_id, device):
    """list of id lists -> (ids, mask) int64 tensors padded to the longest."""
    n = max((len(s) for s in seqs), default=0)
    n = max(n, 1)
    ids = torch.full((len(seqs), n), pad_id, dtype=torch.int64)
    mask = torch.zeros((len(seqs), n), dtype=torch.int64)
    for i, s in enumerate(seqs):
        ids[i, :len(s)] = torch.tensor(s, dtype=torch.int64)
        mask[i, :len(s)] = 1
    return ids.to(device), mask.to(device)
