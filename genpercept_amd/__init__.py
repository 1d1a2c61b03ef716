"""genpercept_amd — MI355X (gfx950) native engine for GenPercept's one-step inference path.

Python host mirroring `genpercept.GenPerceptPipeline` over a C-ABI HIP library (include/genpercept_hip.h).
"""
from .pipeline import GenPerceptOutput, GenPerceptPipeline  # noqa: F401

__all__ = ["GenPerceptPipeline", "GenPerceptOutput"]
