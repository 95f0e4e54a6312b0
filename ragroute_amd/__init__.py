"""ragroute_amd — MI355X-native retrieval hot path for RAGRoute (route -> per-source top-k -> merge)."""
from ._lib import RagrouteHipError  # noqa: F401

__version__ = "0.1.0"
