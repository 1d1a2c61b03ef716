"""Run one of the reference's drivers (run.py, infer.py) unchanged on the MI355X engine:

    python -m genpercept_amd.dropin /path/to/GenPercept/run.py --checkpoint ... --input_rgb_dir ... --mode depth

`python run.py` puts the script's own directory at sys.path[0], AHEAD of PYTHONPATH, so the reference's `genpercept/` package would shadow
this repository's shim whatever PYTHONPATH says.  This launcher fixes the order -- [this repository, the script's directory, the rest] -- and
then executes the script as `__main__` with its own argv.  Under that order `from genpercept import GenPerceptPipeline` (run.py:33,
infer.py:30) is the engine's pipeline while `genpercept.models.dpt_head` / `genpercept.models.custom_unet` (run.py:49,51, infer.py:46,48)
and `src.*` still come from the checkout (genpercept/__init__.py extends its `__path__` with the checkout's package directory)."""
import os
import runpy
import sys


def path_order(script: str, path=None):
    """sys.path for running `script`: this repository's root first, then the script's directory, then what was there (without duplicates)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sdir = os.path.dirname(os.path.abspath(script))
    rest = [p for p in (sys.path if path is None else path) if os.path.abspath(p or os.getcwd()) not in (root, sdir)]
    return [root, sdir] + rest


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        raise SystemExit("usage: python -m genpercept_amd.dropin <reference script (run.py / infer.py)> [its arguments ...]")
    script = argv[0]
    if not os.path.isfile(script):
        raise SystemExit(f"genpercept_amd.dropin: no such script: {script}")
    sys.path[:] = path_order(script)
    for name in [m for m in sys.modules if m == "genpercept" or m.startswith("genpercept.")]:
        del sys.modules[name]  # (a `genpercept` imported under another path order must not leak into the script)
    sys.argv = [script] + list(argv[1:])
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
