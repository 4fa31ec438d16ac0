"""SASRec fixture (BASELINE.json configs[2]): the REAL reference's SASRecQueryEncoder
(recstudio/model/seq/sasrec.py:8-67) + InnerProductScorer + SampledSoftmaxLoss / BinaryCrossEntropyLoss on a fixed batch with
fixed weights: the tower's output, the losses and the gradients of the TIED item table (history gather + positives +
negatives), of the position table and of one Transformer weight.  Writes tests/golden/sasrec.npz.  Runs only where
/root/reference exists."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    from make_golden import OUT, import_reference, np_
    _, scorer, loss_func, _, _ = import_reference()
    from recstudio.model.seq.sasrec import SASRecQueryEncoder
    torch.set_num_threads(1)
    torch.manual_seed(7)
    N, d, L, B, n, heads, hidden, layers = 211, 32, 9, 12, 6, 2, 48, 2
    item = torch.nn.Embedding(N, d, padding_idx=0)
    with torch.no_grad():
        item.weight.mul_(0.3)
        item.weight[0] = 0
    enc = SASRecQueryEncoder('item_id', d, L, heads, hidden, 0.0, 'gelu', 1e-12, layers, item)
    enc.train()                                               # training pooling ('last'); dropout 0: deterministic
    g = torch.Generator().manual_seed(8)
    seqlen = torch.randint(1, L + 1, (B,), generator=g)
    seqlen[0], seqlen[1] = L, 1
    hist = torch.randint(1, N, (B, L), generator=g)
    hist[torch.arange(L).view(1, -1) >= seqlen.view(-1, 1)] = 0
    pos = torch.randint(1, N, (B,), generator=g)
    neg = torch.randint(1, N, (B, n), generator=g)
    neg[2, :3] = hist[2, 0]                                    # an item that is history AND negative: the tie matters
    log_pos = torch.randn(B, generator=g) * 0.3 - 4
    log_neg = torch.randn(B, n, generator=g) * 0.3 - 4
    batch = {'in_item_id': hist, 'seqlen': seqlen}
    out = {'N': N, 'd': d, 'L': L, 'heads': heads, 'hidden': hidden, 'layers': layers,
           'hist': np_(hist), 'seqlen': np_(seqlen), 'pos': np_(pos), 'neg': np_(neg), 'log_pos': np_(log_pos), 'log_neg': np_(log_neg)}
    for k, v in enc.state_dict().items():
        out['w::' + k] = np_(v)
    ip = scorer.InnerProductScorer()
    for tag, loss_fn in (('ssm', loss_func.SampledSoftmaxLoss()), ('bce', loss_func.BinaryCrossEntropyLoss())):
        enc.zero_grad()
        query = enc(batch)
        pos_score = ip(query, item(pos))
        neg_score = ip(query, item(neg))
        loss = loss_fn(None, pos_score, log_pos, neg_score, log_neg)
        loss.backward()
        out[tag + '_query'], out[tag + '_pos_score'], out[tag + '_neg_score'] = np_(query), np_(pos_score), np_(neg_score)
        out[tag + '_loss'] = np_(loss)
        out[tag + '_item_grad'] = np_(item.weight.grad)
        out[tag + '_posemb_grad'] = np_(enc.position_emb.weight.grad)
        out[tag + '_lin1_grad'] = np_(enc.transformer_layer.layers[0].linear1.weight.grad)
    enc.eval()
    with torch.no_grad():
        out['eval_query'] = np_(enc(batch))
    np.savez_compressed(os.path.join(OUT, 'sasrec.npz'), **out)
    print('sasrec.npz', os.path.getsize(os.path.join(OUT, 'sasrec.npz')), {k: float(out[k + '_loss']) for k in ('ssm', 'bce')})


if __name__ == '__main__':
    main()
