"""Drop-in shim: put this directory's parent on PYTHONPATH *before* the reference checkout and set
RAGROUTE_REFERENCE_DIR to that checkout; `python $RAGROUTE_REFERENCE_DIR/main.py ...` then runs unchanged, with

    ragroute.router / ragroute.data_source / ragroute.rerank      -> ragroute_amd (MI355X kernels)
    every other ragroute.* module (config, http_server, ragroute, queue_manager, llm_message, models, ...)
                                                                   -> the reference's own files

    PYTHONPATH=/path/to/this/repo/compat:/path/to/this/repo RAGROUTE_REFERENCE_DIR=/path/to/ragroute \\
        python /path/to/ragroute/main.py --dataset medrag --routing all --disable-llm
"""
import os

_ref = os.environ.get("RAGROUTE_REFERENCE_DIR")
if _ref:
    _pkg = os.path.join(_ref, "ragroute")
    if os.path.isdir(_pkg) and _pkg not in __path__:
        __path__.append(_pkg)  # modules not provided here resolve to the reference's package directory
