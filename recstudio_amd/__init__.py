"""recstudio_amd -- MI355X (gfx950) native embedding-lookup + negative-sampling + scoring hot path
behind RecStudio's Sampler / scorer / loss / BaseRetriever plugin surface.

Host code is Python on PyTorch-ROCm (device memory, streams, generator state); every
computation on the path runs in hand-written HIP kernels reached through the C ABI declared in
``include/recstudio_amd.h`` (``librecstudio_amd.so``, loaded with ctypes).  There is no CPU
fallback: calling an op without the built library or with CPU tensors raises.
"""
from . import _native, ops, rng                                  # noqa: F401
from .sampler import (Sampler, UniformSampler, MaskedUniformSampler, PopularSamplerModel,    # noqa: F401
                      RetrieverSampler)
from .scorer import InnerProductScorer, CosineScorer, EuclideanScorer, NormScorer, GMFScorer   # noqa: F401
from .loss_func import (FullScoreLoss, PairwiseLoss, PointwiseLoss, BPRLoss,   # noqa: F401
                        SampledSoftmaxLoss, SoftmaxLoss, BinaryCrossEntropyLoss, WeightedBPRLoss,
                        WeightedBinaryCrossEntropyLoss, HingeLoss, NCELoss, CCLLoss, InfoNCELoss)
from .fused import retriever_scores                               # noqa: F401
from .dataset import TripletDataset, SeqDataset, DataSampler, SortedDataSampler   # noqa: F401
from .retriever import (BaseRetriever, TwoTowerRecommender, ItemTowerRecommender, BPR, SASRec,   # noqa: F401
                        default_config, seed_everything)
from . import eval                                                # noqa: F401
from . import fused                                               # noqa: F401

__version__ = '0.1.0'
