"""``infinicube.utils``: this repo's MI355X-native ``buffer_utils`` / ``semantic_utils`` first, the reference
checkout's other utility modules (wds_utils, fileio_utils, depth_utils, ...) through the extended search path."""
import importlib.util
import os
import pkgutil
import sys

__path__ = pkgutil.extend_path(__path__, __name__)


def overlay_reference(shim_module_name: str, shim_globals: dict, provided) -> None:
    """If a reference checkout also provides ``infinicube/utils/<name>.py``, execute it under a private name and
    copy its public names into the shim module first, so callers that import names this repo does NOT re-implement
    (e.g. WAYMO_VISUALIZATION_TYPES_BLUE_SKY, read_semantic_buffer_from_file) keep working; the names in
    ``provided`` are then (re)bound to the MI355X-native versions by the shim itself."""
    leaf = shim_module_name.rsplit(".", 1)[1]
    here = os.path.dirname(os.path.abspath(__file__))
    for d in __path__:
        f = os.path.join(d, leaf + ".py")
        if os.path.abspath(d) == here or not os.path.isfile(f):
            continue
        private = f"{__name__}._reference_{leaf}"
        spec = importlib.util.spec_from_file_location(private, f)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[private] = mod
        saved = {k: shim_globals[k] for k in provided if k in shim_globals}
        try:
            spec.loader.exec_module(mod)
        except BaseException:       # e.g. a package the reference module needs is missing: fail like the reference would
            sys.modules.pop(private, None)
            raise
        for k, v in vars(mod).items():
            if not k.startswith("_") and k not in provided:
                shim_globals.setdefault(k, v)
        shim_globals.update(saved)
        return
