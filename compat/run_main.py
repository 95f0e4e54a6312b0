"""Run the reference's UNMODIFIED main.py with the hot-path modules served by ragroute_amd.

    RAGROUTE_REFERENCE_DIR=/path/to/ragroute python compat/run_main.py --dataset medrag --routing all --disable-llm

`python /path/to/ragroute/main.py` itself puts the reference checkout FIRST on sys.path (the script's directory), in front of
anything PYTHONPATH names, so `import ragroute` would find the reference's own package.  This launcher's directory is first
instead: `ragroute` resolves to compat/ragroute (router / data_source / rerank from ragroute_amd, every other submodule from the
reference through the package's extended __path__), then main.py is executed as `__main__` byte for byte."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
ref = os.environ.get("RAGROUTE_REFERENCE_DIR")
if not ref or not os.path.isfile(os.path.join(ref, "main.py")):
    raise SystemExit("set RAGROUTE_REFERENCE_DIR to a checkout of sacs-epfl/ragroute (the directory holding main.py)")
for p in (REPO, HERE):
    if p in sys.path:
        sys.path.remove(p)
    sys.path.insert(0, p)          # HERE ends up first, REPO second
sys.argv[0] = os.path.join(ref, "main.py")
runpy.run_path(sys.argv[0], run_name="__main__")
