"""``from model.lib import libsurf, librf`` -- the import the reference plugins use
(model/model_surf.py:2, model/model_rf.py:2)."""
from . import librf, libsurf  # noqa: F401
