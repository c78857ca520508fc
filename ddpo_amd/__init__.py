"""ddpo_amd — MI355X-native DDPO hot path (HIP kernels behind the reference's Python surfaces)."""
from . import lib  # noqa: F401
