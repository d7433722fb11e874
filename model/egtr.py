"""Alias of egtr_amd.egtr under the reference's module path (model/egtr.py)."""
from egtr_amd.egtr import (DetrForSceneGraphGeneration, DetrSceneGraphGenerationOutput,  # noqa: F401
                           SceneGraphGenerationLoss)
