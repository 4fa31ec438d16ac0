"""Accuracy-anchor fixture: the REAL reference's own training run, recorded epoch by epoch.

``quickstart.run('BPR', 'ml-100k')`` with the reference's stock configuration (README.md:125-208: d = 64, B = 512, n = 1,
Adam 1e-3, xavier_normal, seed 2022, early stopping on ndcg@5 with patience 10) on the CPU of the build container, to
the end.  The published run (README.md:198-208, the authors' GPU, 2023) stops after 35 epochs on test ndcg@10 = 0.2442 /
recall@20 = 0.3530; this file records what the same code does HERE -- the per-epoch training loss and validation
metrics, the stopping epoch and the test metrics -- which the product's fit must track.  Writes
tests/golden/fit_bpr_ml100k.npz.  Runs only where /root/reference exists."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    from make_golden import OUT, import_reference
    import_reference()
    from recstudio.model.basemodel import recommender
    from recstudio.quickstart import run
    rows = []
    orig = recommender.Recommender.training_epoch_end

    def spy(self, output_list):
        res = orig(self, output_list)
        rows.append(dict(self.logged_metrics))
        return res
    recommender.Recommender.training_epoch_end = spy
    (model, datasets), (val_result, test_result) = run('BPR', 'ml-100k', verbose=False,
                                                      train={'gpu': None, 'accelerator': 'cpu', 'num_threads': 8})
    out = {'epochs': np.array(len(rows)), 'train_loss': np.array([r['train_loss_0'] for r in rows], dtype=np.float64)}
    for k in ('ndcg@5', 'recall@5'):
        out['val_' + k] = np.array([r[k] for r in rows], dtype=np.float64)
    for k, v in test_result.items():
        out['test_' + k] = np.array(float(v))
    out['n_train'] = np.array(len(datasets[0]))
    np.savez_compressed(os.path.join(OUT, 'fit_bpr_ml100k.npz'), **out)
    for k, v in out.items():
        print(k, v)


if __name__ == '__main__':
    main()
