"""Import-path shim: `from genpercept import GenPerceptPipeline` (run.py:33, infer.py:30 of the reference; its
genpercept/__init__.py:18) resolves to the MI355X engine's pipeline.  Nothing else of the reference's `genpercept` package is provided
here: its model classes (genpercept.models.*) wrap diffusers modules, which this engine replaces -- pass checkpoint directories or
state dicts to GenPerceptPipeline instead."""
from genpercept_amd.pipeline import GenPerceptOutput, GenPerceptPipeline  # noqa: F401

__all__ = ["GenPerceptPipeline", "GenPerceptOutput"]
