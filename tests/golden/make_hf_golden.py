#!/usr/bin/env python3
"""Generate the Hugging Face checkpoints + outputs under tests/golden/hf_tiny_* (build container only).

The reference's two forwards are sentence-transformers pipelines over Hugging Face BERT models
(src/indexing/embedder.py:40-42,65-71: BertModel -> mean pooling over the attention mask -> L2
normalisation; src/reranker/rerank.py:77,83: BertForSequenceClassification with one label on
tokenizer(text_a, text_b)).  sentence-transformers is not installed here, `transformers` is: this
script builds tiny random-initialised models of both kinds with the library's own classes, saves them
with save_pretrained (safetensors + config.json + vocab.txt + tokenizer_config.json), and records
the library's tokenisation and outputs for a handful of texts and pairs.  The tests then load the
directories through this repository's own loader, tokenizer and forwards (CPU and GPU) and must
reproduce those numbers.

Usage:  python tests/golden/make_hf_golden.py
"""
import json
import os

import torch
from transformers import BertConfig, BertForSequenceClassification, BertModel, BertTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))

WORDS = ("aspirin ibuprofen metformin warfarin dose daily tablet mg pain fever nausea liver kidney heart blood pressure "
         "treat treats treatment effect effects side adult adults children pregnancy warning interaction type diabetes "
         "what is the of and for with in a to can cause causes take taken first line chronic disease monitor").split()
PIECES = ["##s", "##ing", "##ed", "##ment", "##tion", "##al", "##ly", "##in", "##ol", "##e", "##a", "##o", "##i", "##t", "##n",
          "##r", "##m", "##d", "##p", "##l", "##c", "##x", "##y", "##u", "##b", "##g", "##h", "##f", "##k", "##v", "##w", "##z",
          "##1", "##2", "##5", "##0"]
LETTERS = list("abcdefghijklmnopqrstuvwxyz0125")
PUNCT = [".", ",", "?", "-", "(", ")", ":"]

TEXTS = [
    "What treats high blood pressure in adults?",
    "aspirin side effects",
    "Metformin is the first line treatment for type 2 diabetes.",
    "warfarin interaction with ibuprofen can cause bleeding",
    "",
    "x",
    "Daily dose: 5 mg tablet (children), monitor liver and kidney.",
    "zzzq unknownword aspirinol",
]
PAIRS = [
    ("what treats high blood pressure", "ibuprofen is taken for pain and fever"),
    ("what treats high blood pressure", "treatment of blood pressure in adults with a daily tablet"),
    ("aspirin side effects", "aspirin can cause nausea"),
    ("aspirin side effects", ""),
    ("metformin dose", "Metformin is the first line treatment for type 2 diabetes. " * 3),
    ("q", "warfarin interaction warning: monitor blood"),
]


def write_vocab(path):
    vocab = list(dict.fromkeys(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + PUNCT + LETTERS + sorted(set(WORDS)) + PIECES))
    with open(path, 'w', encoding='utf-8') as f:
        f.write('\n'.join(vocab) + '\n')
    return len(vocab)


def config(vocab_size, **kw):
    return BertConfig(vocab_size=vocab_size, hidden_size=32, num_hidden_layers=2, num_attention_heads=1,
                      intermediate_size=64, max_position_embeddings=96, type_vocab_size=2, hidden_act='gelu',
                      layer_norm_eps=1e-12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **kw)


def randomise(model, seed):
    """Weights far from the library's 0.02-normal initialisation, so that every term of the forward (biases,
    LayerNorm gains, token types, positions) moves the output."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'LayerNorm.weight' in name:
                p.copy_(1.0 + 0.3 * torch.randn(p.shape, generator=g))
            elif name.endswith('bias'):
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
            elif 'embeddings' in name:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.25 * torch.randn(p.shape, generator=g))


def main():
    torch.manual_seed(0)
    out = {'note': 'outputs of Hugging Face transformers %s for the checkpoints saved beside this file'
                   % __import__('transformers').__version__}
    # ---- bi-encoder: BertModel, mean pooling over the mask, L2 normalisation
    d_bi = os.path.join(HERE, 'hf_tiny_bi')
    os.makedirs(d_bi, exist_ok=True)
    n_vocab = write_vocab(os.path.join(d_bi, 'vocab.txt'))
    tok = BertTokenizer(os.path.join(d_bi, 'vocab.txt'), do_lower_case=True)
    tok.save_pretrained(d_bi)
    bi = BertModel(config(n_vocab), add_pooling_layer=False).eval()
    randomise(bi, 1)
    bi.save_pretrained(d_bi, safe_serialization=True)
    enc = tok(TEXTS, padding=True, truncation=True, max_length=96, return_tensors='pt')
    with torch.no_grad():
        h = bi(**enc).last_hidden_state
        m = enc['attention_mask'].unsqueeze(-1).to(h.dtype)
        pooled = (h * m).sum(1) / m.sum(1).clamp(min=1e-9)
        emb = torch.nn.functional.normalize(pooled, p=2, dim=1)
    out['bi'] = {
        'texts': TEXTS,
        'token_ids': [tok.encode(t, add_special_tokens=False) for t in TEXTS],
        'input_ids': [[int(v) for v, k in zip(row, mk) if k] for row, mk in zip(enc['input_ids'].tolist(),
                                                                               enc['attention_mask'].tolist())],
        'pooled': pooled.tolist(), 'embeddings': emb.tolist(),
        'special_ids': {'cls': tok.cls_token_id, 'sep': tok.sep_token_id, 'pad': tok.pad_token_id},
    }
    # ---- cross-encoder: BertForSequenceClassification, one label
    d_ce = os.path.join(HERE, 'hf_tiny_cross')
    os.makedirs(d_ce, exist_ok=True)
    write_vocab(os.path.join(d_ce, 'vocab.txt'))
    tok2 = BertTokenizer(os.path.join(d_ce, 'vocab.txt'), do_lower_case=True)
    tok2.save_pretrained(d_ce)
    ce = BertForSequenceClassification(config(n_vocab, num_labels=1)).eval()
    randomise(ce, 2)
    ce.save_pretrained(d_ce, safe_serialization=True)
    enc2 = tok2([a for a, _ in PAIRS], [b for _, b in PAIRS], padding=True, truncation='longest_first', max_length=96,
                return_tensors='pt')
    with torch.no_grad():
        logits = ce(**enc2).logits.squeeze(-1)
    out['cross'] = {
        'pairs': [list(p) for p in PAIRS],
        'input_ids': [[int(v) for v, k in zip(row, mk) if k] for row, mk in zip(enc2['input_ids'].tolist(),
                                                                               enc2['attention_mask'].tolist())],
        'token_type_ids': [[int(v) for v, k in zip(row, mk) if k] for row, mk in zip(enc2['token_type_ids'].tolist(),
                                                                                    enc2['attention_mask'].tolist())],
        'logits': logits.tolist(),
    }
    with open(os.path.join(HERE, 'hf_golden.json'), 'w') as f:
        json.dump(out, f)
    for d in (d_bi, d_ce):
        for name in sorted(os.listdir(d)):
            print(d, name, os.path.getsize(os.path.join(d, name)))


if __name__ == '__main__':
    main()
