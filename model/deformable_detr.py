"""Alias of egtr_amd.deformable_detr under the reference's module path (model/deformable_detr.py)."""
from egtr_amd.deformable_detr import *  # noqa: F401,F403
from egtr_amd.deformable_detr import (DeformableDetrConfig, DeformableDetrDecoder, DeformableDetrEncoder,  # noqa: F401
                                      DeformableDetrForObjectDetection, DeformableDetrLoss,
                                      DeformableDetrHungarianMatcher, DeformableDetrMLPPredictionHead,
                                      DeformableDetrModel, DeformableDetrMultiheadAttention,
                                      DeformableDetrMultiscaleDeformableAttention, DeformableDetrPreTrainedModel,
                                      MultiScaleDeformableAttentionFunction, inverse_sigmoid)
from egtr_amd.feature_extraction import (DeformableDetrFeatureExtractor,  # noqa: F401
                                         DeformableDetrFeatureExtractorWithAugmentor,
                                         DeformableDetrFeatureExtractorWithAugmentorNoCrop)
