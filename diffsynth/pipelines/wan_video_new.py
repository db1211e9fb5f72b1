"""`diffsynth.pipelines.wan_video_new` names used by the reference [R infinicube/videogen/inference.py:26]."""
from infinicube_amd.videogen.pipeline import ModelConfig, WanVideoPipeline

__all__ = ["ModelConfig", "WanVideoPipeline"]
