"""The two transformer forwards against Hugging Face `transformers`.

tests/golden/hf_tiny_bi and hf_tiny_cross are checkpoints written by `transformers`' own
save_pretrained (BertModel; BertForSequenceClassification with one label), hf_golden.json holds that
library's tokenisation and outputs for them (tests/golden/make_hf_golden.py, run in the build
container).  sentence-transformers, which the reference calls (src/indexing/embedder.py:40-42,65-71;
src/reranker/rerank.py:77,83), runs exactly these classes underneath: Transformer -> mean pooling over
the attention mask -> L2 normalisation, and tokenizer(a, b) -> logits.  Here the directories go through
this repository's loader (safetensors reader, state-dict conversion), WordPiece tokenizer and forwards.
"""
import json
import os
import shutil

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')
BI_DIR, CE_DIR = os.path.join(GOLD, 'hf_tiny_bi'), os.path.join(GOLD, 'hf_tiny_cross')


@pytest.fixture(scope='module')
def gold():
    with open(os.path.join(GOLD, 'hf_golden.json')) as f:
        return json.load(f)


def _vocab_only_dir(src, tmp_path):
    """The checkpoint with vocab.txt but no tokenizer.json: the loader's second tokenizer route."""
    dst = tmp_path / 'vocab_only'
    dst.mkdir()
    for name in ('config.json', 'model.safetensors', 'vocab.txt', 'tokenizer_config.json'):
        shutil.copy(os.path.join(src, name), dst / name)
    return str(dst)


def test_wordpiece_tokenizer_matches_transformers(gold, tmp_path):
    from indexing import bert
    for d in (BI_DIR, _vocab_only_dir(BI_DIR, tmp_path)):
        tok = bert.FileTokenizer(d)
        sp = gold['bi']['special_ids']
        assert (tok.cls_id, tok.sep_id, tok.pad_id) == (sp['cls'], sp['sep'], sp['pad'])
        for text, want in zip(gold['bi']['texts'], gold['bi']['token_ids']):
            assert tok.encode(text) == want, text


def test_bi_encoder_checkpoint_matches_transformers_on_cpu(gold):
    from indexing import bert
    model, tok, cfg = bert.load_model(BI_DIR, 'bi', 'cpu')
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.intermediate_size) == (32, 2, 1, 64)
    seqs = gold['bi']['input_ids']
    ids, mask = bert.pad_batch(seqs, tok.pad_id, 'cpu')
    with torch.no_grad():
        pooled = model(ids, None, mask, normalize=False)
        emb = model(ids, None, mask, normalize=True)
    assert float((pooled - torch.tensor(gold['bi']['pooled'])).abs().max()) < 2e-6
    assert float((emb - torch.tensor(gold['bi']['embeddings'])).abs().max()) < 1e-6


def test_medical_embedder_from_checkpoint_directory(gold, tmp_path):
    """MedicalEmbedder(<directory>): text in, the sentence-transformers pipeline's embeddings out."""
    from indexing.embedder import MedicalEmbedder
    for d in (BI_DIR, _vocab_only_dir(BI_DIR, tmp_path)):
        e = MedicalEmbedder(d, device='cpu')
        assert e.embedding_dim == 32 and e.max_length == 96
        for text, want in zip(gold['bi']['texts'], gold['bi']['input_ids']):
            assert e.tokenize(text) == want
        got = e.encode(gold['bi']['texts'])
        assert got.dtype == np.float32 and got.shape == (len(gold['bi']['texts']), 32)
        assert float(np.abs(got - np.asarray(gold['bi']['embeddings'], np.float32)).max()) < 1e-6


def test_cross_encoder_checkpoint_matches_transformers_on_cpu(gold):
    from indexing import bert
    from reranker.rerank import CrossEncoderReranker
    model, tok, cfg = bert.load_model(CE_DIR, 'cross', 'cpu')
    assert cfg.num_labels == 1
    ids, mask = bert.pad_batch(gold['cross']['input_ids'], tok.pad_id, 'cpu')
    types, _ = bert.pad_batch(gold['cross']['token_type_ids'], 0, 'cpu')
    with torch.no_grad():
        logits = model(ids, types, mask)
    want = torch.tensor(gold['cross']['logits'])
    assert float((logits - want).abs().max()) < 2e-6
    # text pairs through the reranker class: tokenizer(a, b) and the forward
    rr = CrossEncoderReranker(CE_DIR, device='cpu')
    for (a, b), w_ids, w_types in zip(gold['cross']['pairs'], gold['cross']['input_ids'], gold['cross']['token_type_ids']):
        assert rr.encode_pair(a, b) == (w_ids, w_types)
    got = rr.score_pairs([tuple(p) for p in gold['cross']['pairs']])
    assert max(abs(x - y) for x, y in zip(got, gold['cross']['logits'])) < 2e-6


def test_loader_rejects_a_checkpoint_that_lacks_weights(tmp_path):
    from indexing import bert
    from safetensors.torch import load_file, save_file
    d = tmp_path / 'broken'
    d.mkdir()
    for name in ('config.json', 'vocab.txt'):
        shutil.copy(os.path.join(CE_DIR, name), d / name)
    sd = load_file(os.path.join(CE_DIR, 'model.safetensors'))
    sd.pop('classifier.weight')
    save_file(sd, str(d / 'model.safetensors'))
    with pytest.raises(RuntimeError, match='lacks weights'):
        bert.load_model(str(d), 'cross', 'cpu')
    with pytest.raises(FileNotFoundError):
        bert.load_model(str(tmp_path / 'nowhere'), 'cross', 'cpu')


# ---- the same numbers from the GPU paths ------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('split_gemm,half,tol', [(False, None, 2e-6), (True, torch.float16, 2e-6), (True, torch.bfloat16, 3e-5)])
def test_bi_encoder_checkpoint_matches_transformers_on_gpu(gold, split_gemm, half, tol):
    """Unpadded native forward (librag_hip kernels between the GEMMs; GEMMs in fp32 or as split products)."""
    from indexing import bert
    dev = torch.device('cuda:0')
    model, tok, _ = bert.load_model(BI_DIR, 'bi', dev, native=True, split_gemm=split_gemm, split_dtype=half)
    ids, mask = bert.pad_batch(gold['bi']['input_ids'], tok.pad_id, dev)
    with torch.no_grad():
        emb = model.forward_packed(*model.bert.pack(ids, mask), normalize=True)
        pooled = model.forward_packed(*model.bert.pack(ids, mask), normalize=False)
    assert float((pooled.cpu() - torch.tensor(gold['bi']['pooled'])).abs().max()) < 4 * tol
    assert float((emb.cpu() - torch.tensor(gold['bi']['embeddings'])).abs().max()) < tol


@pytest.mark.gpu
def test_medical_embedder_and_reranker_from_directories_on_gpu(gold):
    from indexing.embedder import MedicalEmbedder
    from reranker.rerank import CrossEncoderReranker
    e = MedicalEmbedder(BI_DIR, device='cuda')
    assert e.model.bert.native and e.model.bert.split_gemm
    got = e.encode(gold['bi']['texts'])
    assert float(np.abs(got - np.asarray(gold['bi']['embeddings'], np.float32)).max()) < 3e-5      # bf16 halves
    rr = CrossEncoderReranker(CE_DIR, device='cuda')
    assert rr.model.bert.native and rr.model.bert.split_gemm and rr.model.bert.split_dtype == torch.float16
    logits = rr.score_pairs([tuple(p) for p in gold['cross']['pairs']])
    assert max(abs(x - y) for x, y in zip(logits, gold['cross']['logits'])) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['fp32-split-fp16', 'fp32-plain', 'bf16'])
def test_cross_encoder_checkpoint_matches_transformers_on_gpu(gold, kind):
    from indexing import bert
    dev = torch.device('cuda:0')
    if kind == 'bf16':
        model, tok, _ = bert.load_model(CE_DIR, 'cross', dev, dtype=torch.bfloat16, native=True)
        tol = 0.15                      # bfloat16 weights and activations: logits of magnitude ~1-3
    elif kind == 'fp32-plain':
        model, tok, _ = bert.load_model(CE_DIR, 'cross', dev, native=True)
        tol = 5e-6
    else:
        model, tok, _ = bert.load_model(CE_DIR, 'cross', dev, native=True, split_gemm=True, split_dtype=torch.float16)
        tol = 5e-6
    ids, mask = bert.pad_batch(gold['cross']['input_ids'], tok.pad_id, dev)
    types, _ = bert.pad_batch(gold['cross']['token_type_ids'], 0, dev)
    with torch.no_grad():
        packed = model.bert.pack(ids, mask, types)
        logits = model.forward_packed(*packed)
        padded = model(ids, types, mask)
    want = torch.tensor(gold['cross']['logits'])
    assert float((logits.cpu() - want).abs().max()) < tol
    assert float((padded.cpu() - want).abs().max()) < tol
