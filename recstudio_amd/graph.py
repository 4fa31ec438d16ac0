"""HIP-graph capture of the launch-bound small-batch training step.

``GraphedBPRStep`` captures a BPR training step once (``torch.cuda.CUDAGraph`` = hipGraph on ROCm) and replays it
per batch.  Measured on this stack (tools/exp_graph.py, B = 4096, n = 64): the two kernels of the step already keep
the GPU busy (eager 65 us per step, 38 us for the forward alone) and a replay is no cheaper than two ctypes
launches (72 us; a one-kernel graph replays in 10 us vs 4 us for an eager launch) -- so nothing in the package uses
it by default; it exists for callers that embed the step in a larger captured region.  Random numbers stay
torch's: the Philox offset lives in a device word that the graph itself advances (``rsa_fused_args.offset_dev`` +
``rsa_rng_advance``), and ``sync_generator()`` mirrors the consumption into ``torch.cuda.default_generators`` so
that code running afterwards continues the same stream a non-captured run would have left.
"""
import torch

from . import _native as nat
from . import ops, rng
from .fused import _sampler_kind


class GraphedBPRStep:
    """``step(user_ids, pos_ids) -> loss`` for nn.Embedding user / item tables, stock BPRLoss, num_neg % 64 == 0.

    mode='grads': leaves ``neg_ids``, ``dneg``, ``dpos``, ``query_grad`` (per-query user-row gradient) and the
    row-sparse item-gradient ``rows`` in ``self.out`` (static buffers, overwritten by the next replay);
    mode='sgd': applies ``-lr *`` gradient to the touched rows of both tables in place (see fused.bpr_sgd_step)."""

    def __init__(self, item_weight, user_weight, num_neg, batch_size, sampler, mode='grads', lr=0.0):
        if num_neg % 64:
            raise ValueError('GraphedBPRStep: num_neg must be a multiple of 64 (the single-launch BPR path)')
        kind = _sampler_kind(sampler)
        if kind in (None, nat.SAMPLER_GIVEN):
            raise TypeError('GraphedBPRStep needs a UniformSampler or PopularSamplerModel')
        self.iw, self.uw = item_weight.data, user_weight.data
        dev = self.iw.device
        self.n, self.B, self.mode, self.lr = int(num_neg), int(batch_size), mode, float(lr)
        self.kw = {'sampler': kind}
        if kind == nat.SAMPLER_POPULAR:
            self.kw.update(sampler.lookup_kwargs())
        unroll = 4 if kind == nat.SAMPLER_POPULAR else rng.randint_unroll(1, self.iw.shape[0])
        cu, mt = rng.device_props(dev)
        self.increment = rng.counter_offset(self.B * self.n, rng.grid_threads(self.B * self.n, cu, mt), unroll)
        if not torch.cuda.is_initialized():
            torch.cuda.init()
        self.gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        self.seed = int(self.gen.initial_seed())
        self.offset0 = int(self.gen.get_offset())
        self.offset_dev = torch.tensor([self.offset0], dtype=torch.int64, device=dev)
        self.replays = 0
        self.uid = torch.ones(self.B, dtype=torch.int64, device=dev)
        self.pos = torch.ones(self.B, dtype=torch.int64, device=dev)
        self.step_t = torch.zeros(1, dtype=torch.float32, device=dev)     # 0 during the warm-up: tables untouched
        self.out = None
        # warm-up on a side stream (one-time allocations of the library happen here, not under capture), then capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                self._body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.offset_dev.fill_(self.offset0)           # the warm-up draws are not part of the stream
        self.step_t.fill_(-self.lr)                   # read by the captured kernels at replay time
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self._body()

    def _body(self):
        o = ops.fused_forward(self.iw, self.uw, self.n, query_index=self.uid, pos_ids=self.pos, out=self.out,
                              want_logp=False, fused_bpr=True, want_query_grad=True,
                              rng_state=(self.seed, self.offset_dev), **self.kw)
        self.out = o
        if self.mode == 'sgd':
            ops.fused_backward(self.iw, self.uw, o['neg_ids'], o['dneg'], query_index=self.uid, pos_ids=self.pos,
                               dpos=o['dpos'], upstream=self.step_t, dense_item_grad=True, item_grad_out=self.iw,
                               want_query_grad=False)
            ops.scatter_add_rows(o['query_grad'] * self.step_t, self.uid, self.uw.shape[0], out=self.uw)
        else:
            _, rows, _ = ops.fused_backward(self.iw, self.uw, o['neg_ids'], o['dneg'], query_index=self.uid,
                                            pos_ids=self.pos, dpos=o['dpos'], dense_item_grad=False,
                                            row_item_grad=True, want_query_grad=False)
            o['rows'] = rows
        ops.rng_advance(self.offset_dev, self.increment)

    def step(self, user_ids, pos_ids):
        self.uid.copy_(user_ids, non_blocking=True)
        self.pos.copy_(pos_ids, non_blocking=True)
        self.graph.replay()
        self.replays += 1
        return self.out['loss']

    __call__ = step

    def sync_generator(self):
        """Make the torch generator reflect the numbers the replays consumed."""
        self.gen.set_offset(self.offset0 + self.replays * self.increment)
