"""MI355X-native HyperVLA action-prediction path.  ``from hypervla.model import HyperVLA``
is the drop-in for the reference's import of the same name."""
from .config import FULL, MID, SMALL_E, TINY, Geometry, default_config, generated_leaves  # noqa: F401

__all__ = ["FULL", "MID", "SMALL_E", "TINY", "Geometry", "default_config", "generated_leaves"]
