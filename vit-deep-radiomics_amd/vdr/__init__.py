"""vdr — host side of the MI355X-native ViT dense-descriptor / CLS-feature path.

Importing this package does not touch the GPU; constructing an Engine / model does and raises if
libvdr.so or a HIP device is missing (there is no CPU fallback).
"""
from ._lib import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RESID, EPI_SWIGLU, OUT_CLS, OUT_DENSE, OUT_ENCODER,  # noqa: F401
                   OUT_PATCH_EMBED, OUT_TOKENS, VdrError, load, source_id)
from .engine import Engine, VdrConfig  # noqa: F401
from .model import (ARCHS, TransformerNoduleBimodalClassifier, TransformerNoduleClassifier, VitDescriptorModel, extract_dense, get_dense_descriptor,  # noqa: F401
                    load_model)
from . import pipeline, prep  # noqa: F401,E402
