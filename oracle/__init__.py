"""CPU oracle for the RecStudio retriever hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``recstudio_amd/`` may import this package.  The only legal
importers are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- and there only as the checker / the timed baseline,
never as the thing shipped.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the real reference
(``/root/reference``, Python/PyTorch) in the build container, records its
outputs into ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks
every function here against those fixtures.  The device random stream
(Philox4x32-10 as driven by torch-ROCm's ``distribution_nullary_kernel``) is a
third-party algorithm (PyTorch 2.10 + rocRAND 7.x, not under /root/reference);
it is restated in ``oracle/philox.py`` from the published Random123 algorithm,
pinned by the Random123 known-answer vectors on CPU and by ``torch.randint`` /
``torch.rand`` on ``cuda`` in the ``-m gpu`` tests.
"""
from .philox import (philox4x32_10, rng_grid_threads, rng_counter_offset,
                     device_randint, device_rand)
from .path import (UniformSampler, PopularSamplerModel, popular_tables, sasrec_query,
                   searchsorted_left, masked_uniform_from_u, inner_product_score, cosine_score, euclidean_score,
                   norm_score, gmf_score,
                   bpr_loss, sampled_softmax_loss, softmax_loss, bce_loss, weighted_bpr_loss, weighted_bce_loss,
                   hinge_loss, info_nce_loss, nce_loss, ccl_loss,
                   retriever_forward, dense_grads, topk_with_history,
                   rank_metrics, seq_gather, test_step_hits)
