"""Drop-in shim: with this directory's parent FIRST on sys.path and RAGROUTE_REFERENCE_DIR set to the reference checkout, the
reference's main.py runs unchanged (compat/run_main.py arranges both: `python main.py` itself would put the checkout first), with

    ragroute.router / ragroute.data_source / ragroute.rerank      -> ragroute_amd (MI355X kernels)
    every other ragroute.* module (config, http_server, ragroute, queue_manager, llm_message, models, ...)
                                                                   -> the reference's own files

    RAGROUTE_REFERENCE_DIR=/path/to/ragroute python /path/to/this/repo/compat/run_main.py --dataset medrag --routing all --disable-llm
"""
import os

_ref = os.environ.get("RAGROUTE_REFERENCE_DIR")
if _ref:
    _pkg = os.path.join(_ref, "ragroute")
    if os.path.isdir(_pkg) and _pkg not in __path__:
        __path__.append(_pkg)  # modules not provided here resolve to the reference's package directory
