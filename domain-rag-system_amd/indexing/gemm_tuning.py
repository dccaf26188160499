"""hipBLASLt solution choice for the encoder GEMMs.

The transformer GEMMs of the path stay with the vendor library (PyTorch -> hipBLASLt).  Its
default heuristic is not the fastest solution for the skinny shapes of a 384-wide encoder
(K = 384; measured 17 % slower over the four GEMMs of a MiniLM layer at 1.2 M tokens), so the
path can run with PyTorch's TunableOp reading a table of per-shape solutions that was measured
once on an MI355X (``tunableop_gfx950.csv`` next to this file).  ``BatchSearcher`` keeps the
token count of every cross-encoder mini-batch on a small grid, so the table stays small.

Without the table (or when the library versions recorded in it differ from the installed
ones) PyTorch silently uses the default heuristic: results are the same, only slower.
"""
import os

import torch

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tunableop_gfx950.csv')


def enable(tune=False, table=TABLE):
    """Use the recorded solutions (tune=False) or measure and record new ones (tune=True,
    slow: seconds per new GEMM shape; PyTorch writes the table when the process ends).
    Returns True when TunableOp is active."""
    if not torch.cuda.is_available():
        return False
    if not tune and not os.path.exists(table):
        return False
    t = torch.cuda.tunable
    t.enable(True)
    t.set_filename(table)
    t.tuning_enable(bool(tune))
    if tune:
        t.set_max_tuning_duration(1500)
        t.set_max_tuning_iterations(30)
    else:
        t.read_file(table)
    return True


@torch.no_grad()
def tune_cross_encoder(model, budget=1 << 20, bucket=1 << 17, seq=64):
    """Run the packed cross-encoder forward once for every token count on the mini-batch grid."""
    dev = next(model.parameters()).device
    for m in range(bucket, budget + 1, bucket):
        n_seq = m // seq
        ids = torch.randint(1000, model.bert.config.vocab_size, (m,), device=dev, dtype=torch.int32)
        types = torch.zeros((m,), dtype=torch.int32, device=dev)
        pos = (torch.arange(m, device=dev, dtype=torch.int32) % seq).contiguous()
        cu = (torch.arange(n_seq + 1, device=dev, dtype=torch.int32) * seq).contiguous()
        model.forward_packed(ids, types, pos, cu, seq)
        torch.cuda.synchronize()
