"""genpercept_amd — MI355X (gfx950) native engine for GenPercept's inference path (one-step `genpercept`, multi-step `marigold` /
`rgb_blending`).

Python host mirroring `genpercept.GenPerceptPipeline` over a C-ABI HIP library (include/genpercept_hip.h); `DDIMSchedulerCustomized`
is the host-side (scalar) mirror of src/customized_modules/ddim.py for callers that have no diffusers.
"""
from .pipeline import GenPerceptOutput, GenPerceptPipeline  # noqa: F401
from .scheduler import DDIMSchedulerCustomized  # noqa: F401

__all__ = ["GenPerceptPipeline", "GenPerceptOutput", "DDIMSchedulerCustomized"]
