"""`from genpercept.genpercept_pipeline import GenPerceptPipeline, GenPerceptOutput` (the reference's module path) -> the HIP engine's."""
from genpercept_amd.pipeline import GenPerceptOutput, GenPerceptPipeline  # noqa: F401
