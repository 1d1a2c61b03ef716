"""Import-path shim: `from genpercept import GenPerceptPipeline` (run.py:33, infer.py:30 of the reference; its
genpercept/__init__.py:18) resolves to the MI355X engine's pipeline, and so does `genpercept.genpercept_pipeline`.

run.py:49,51 / infer.py:46,48 also import `genpercept.models.dpt_head` and `genpercept.models.custom_unet` at module top (the classes
run.py instantiates itself and then hands to the pipeline, which only needs their `state_dict()`).  This package does not carry those
files -- they are the reference's -- so its `__path__` is EXTENDED with every other `genpercept/` package directory found on `sys.path`
(a reference checkout): this directory comes first, hence `genpercept.genpercept_pipeline` and `GenPerceptPipeline` stay the engine's,
while `genpercept.models.*`, `genpercept.util.*`, `genpercept.losses.*` resolve to the reference's files when a checkout is on the path
(and raise the usual ModuleNotFoundError when none is).  The scan is redone on every sub-module import, so the checkout may be appended to
`sys.path` after this package was imported.  `python -m genpercept_amd.dropin <reference>/run.py ...` sets the path order up (the
script's own directory would otherwise come FIRST and shadow this shim)."""
import os
import sys

from genpercept_amd.pipeline import GenPerceptOutput, GenPerceptPipeline  # noqa: F401

__all__ = ["GenPerceptPipeline", "GenPerceptOutput"]

_HERE = os.path.dirname(os.path.abspath(__file__))


class _ShimPath:
    """`__path__` of the shim: this directory, then every other `<sys.path entry>/genpercept/` that is a regular package (has an
    `__init__.py`), in `sys.path` order.  The import system only iterates it (like a namespace package's `_NamespacePath`)."""

    def _dirs(self):
        out = [_HERE]
        for entry in list(sys.path):
            if not isinstance(entry, str):
                continue
            cand = os.path.abspath(os.path.join(entry or os.getcwd(), "genpercept"))
            if cand not in out and os.path.isfile(os.path.join(cand, "__init__.py")):
                out.append(cand)
        return out

    def __iter__(self):
        return iter(self._dirs())

    def __len__(self):
        return len(self._dirs())

    def __getitem__(self, i):
        return self._dirs()[i]

    def __contains__(self, item):
        return item in self._dirs()

    def __repr__(self):
        return f"_ShimPath({self._dirs()!r})"


__path__ = _ShimPath()
