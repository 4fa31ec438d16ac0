"""CPU: rank metrics (recstudio_amd.eval) against values recorded from the reference's
BaseRetriever._test_step (tests/golden/topk.npz)."""
import numpy as np
import torch

import oracle
from recstudio_amd import eval as rs_eval

T = torch.from_numpy


def test_rank_metrics_match_reference(golden):
    g = golden('topk')
    items = T(g['items'])
    B = items.shape[0]
    for tgt, rating, pre in ((T(g['tgt1']), torch.ones(B, 1), 'm1_'), (T(g['tgt2']), T(g['rat2']), 'm2_')):
        hits = oracle.test_step_hits(tgt, items)
        for cutoff in (5, 10):
            for name, fn in rs_eval.get_rank_metrics(['ndcg', 'recall', 'precision', 'map', 'mrr', 'hit']):
                np.testing.assert_allclose(fn(hits, rating, cutoff).numpy(), g[f'{pre}{name}@{cutoff}'], rtol=1e-6,
                                           err_msg=f'{pre}{name}@{cutoff}')
