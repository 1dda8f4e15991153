"""Drop-in replacement for the reference's ``unet1d`` package (denoiser only).

``from unet1d import UNet1DConditionModel`` / ``from unet1d.unet_1d_condition import
UNet1DConditionModel`` / ``from unet1d.embeddings import TextTimeEmbedding`` keep
working for the reference's ``model.py`` (imports at model.py:6-7); the forward runs
on the MI355X HIP engine (libns2vc_hip.so).
"""
from .unet_1d_condition import UNet1DConditionModel, UNet1DConditionOutput  # noqa: F401
