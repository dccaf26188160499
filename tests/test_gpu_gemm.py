"""GPU: the fused encoder GEMM (bias / GELU / residual+LayerNorm tails) against fp64 PyTorch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(m, k, n, seed):
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn((m, k), generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn((n, k), generator=g, device=dev) * 0.05).to(torch.bfloat16)
    b = (torch.randn((n,), generator=g, device=dev) * 0.1).to(torch.bfloat16)
    return x, w, b, g, dev


@pytest.mark.parametrize('m,k,n', [(1000, 384, 1152), (257, 384, 384), (128, 1536, 384), (5, 64, 768), (4097, 384, 1536)])
def test_bias_and_gelu_tails(m, k, n):
    import rag_hip
    x, w, b, g, dev = _data(m, k, n, m + k + n)
    ref = x.double() @ w.double().t() + b.double()
    y = rag_hip.enc_linear_dev(x, w, b)
    # bf16 output of an fp32 accumulation: half an ulp of bf16 relative, plus accumulation noise
    assert float(((y.double() - ref).abs() / (ref.abs() + 1.0)).max()) < 6e-3
    y = rag_hip.enc_linear_dev(x, w, b, rag_hip.EPILOGUE_BIAS_GELU)
    gref = torch.nn.functional.gelu(ref)
    assert float(((y.double() - gref).abs() / (gref.abs() + 1.0)).max()) < 6e-3


@pytest.mark.parametrize('m,k', [(1000, 384), (300, 1536), (1, 64)])
def test_residual_layernorm_tail(m, k):
    import rag_hip
    x, w, b, g, dev = _data(m, k, 384, m + k)
    res = torch.randn((m, 384), generator=g, device=dev).to(torch.bfloat16)
    gamma = (1 + 0.1 * torch.randn((384,), generator=g, device=dev)).to(torch.bfloat16)
    beta = (0.1 * torch.randn((384,), generator=g, device=dev)).to(torch.bfloat16)
    y = rag_hip.enc_linear_dev(x, w, b, rag_hip.EPILOGUE_BIAS_RESIDUAL_LAYERNORM, res, gamma, beta, 1e-12)
    v = x.double() @ w.double().t() + b.double() + res.double()
    ref = torch.nn.functional.layer_norm(v, (384,), gamma.double(), beta.double(), 1e-12)
    assert float((y.double() - ref).abs().max()) < 3e-2
    # the unfused route (library GEMM, then the residual+LayerNorm kernel) rounds the GEMM output to
    # bf16 first; the fused tail must be at least as close to the exact result
    lin = (x @ w.t() + b)
    unfused = rag_hip.enc_add_layernorm_dev(lin, res, gamma, beta, 1e-12)
    assert float((y.double() - ref).abs().mean()) <= float((unfused.double() - ref).abs().mean()) * 1.05


def test_linear_argument_errors():
    import rag_hip
    x, w, b, g, dev = _data(8, 48, 384, 1)
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.enc_linear_dev(x, w, b)                       # k not a multiple of 64
    x, w, b, g, dev = _data(8, 64, 200, 1)
    with pytest.raises(rag_hip.RagHipError):
        rag_hip.enc_linear_dev(x, w, b)                       # n not a multiple of 384
