"""clt_training_amd: MI355X-native SAE / cross-layer-transcoder train-step hot path.

Mirrors the operator boundary of EleutherAI/clt-training (package ``sparsify``):
``fused_encoder``, ``decoder_impl``, ``SparseCoder``, ``MidDecoder``,
``CrossLayerRunner``, ``SparseCoderConfig``/``TrainConfig`` and ``Trainer``
(sparsify/__init__.py:3-18), backed by hand-written gfx950 HIP kernels behind the
C ABI of ``include/clt_mi355.h``.  There is no CPU fallback."""
__version__ = "0.1.0"

from .config import SaeConfig, SparseCoderConfig, TrainConfig, TranscoderConfig  # noqa: E402,F401
from .runner import CrossLayerRunner  # noqa: E402,F401
from .sparse_coder import ForwardOutput, MidDecoder, Sae, SparseCoder  # noqa: E402,F401
from .trainer import SaeTrainer, Trainer  # noqa: E402,F401

__all__ = ["Sae", "SaeConfig", "SaeTrainer", "SparseCoder", "SparseCoderConfig", "CrossLayerRunner", "Trainer",
           "TrainConfig", "TranscoderConfig", "MidDecoder", "ForwardOutput"]
