"""Make the reference's import paths resolve to the MI355X implementation (option A), so
drivers written against crockwell/pixelsynth (`demo.py`, `create_vid.py`, `evaluation/*.py`) pick up the HIP
path without source edits:

    import pixelsynth_amd.compat; pixelsynth_amd.compat.install_reference_aliases()
"""
import importlib
import importlib.util
import sys
import types

ALIASES = {
    "models.projection.z_buffer_manipulator": "pixelsynth_amd.projection.z_buffer_manipulator",
    "models.layers.z_buffer_layers": "pixelsynth_amd.layers.z_buffer_layers",
    "models.lmconv.locally_masked_convolution": "pixelsynth_amd.lmconv.locally_masked_convolution",
    "models.lmconv.layers": "pixelsynth_amd.lmconv.layers",
    "models.lmconv.model": "pixelsynth_amd.lmconv.model",
    "models.lmconv.masking": "pixelsynth_amd.lmconv.masking",
    "models.lmconv.sample": "pixelsynth_amd.lmconv.sample",
    "models.lmconv.utils": "pixelsynth_amd.lmconv.utils",
    "models.vqvae2.vqvae": "pixelsynth_amd.vqvae2.vqvae",
}


def _parent_package(pkg):
    """The reference's own package when its tree is importable (demo.py run from a checkout: everything that is NOT
    aliased -- models.networks, models.base_model, ... -- must keep resolving through the real package's __path__);
    an empty namespace module only when there is nothing to import."""
    if pkg in sys.modules:
        return sys.modules[pkg]
    try:
        return importlib.import_module(pkg)
    except Exception:   # ImportError, or a package __init__ that needs what this machine lacks
        sys.modules.pop(pkg, None)
    m = types.ModuleType(pkg)
    m.__path__ = []
    spec = None
    try:
        spec = importlib.util.find_spec(pkg)
    except Exception:
        pass
    if spec is not None and spec.submodule_search_locations:
        m.__path__ = list(spec.submodule_search_locations)   # the real directory: non-aliased submodules still resolve
    sys.modules[pkg] = m
    return m


def install_reference_aliases(force=False):
    """Register the mirrors under the reference's module names."""
    installed = []
    for ref, ours in ALIASES.items():
        if ref in sys.modules and not force:
            continue
        parts = ref.split(".")
        for i in range(1, len(parts)):
            pkg = ".".join(parts[:i])
            parent = _parent_package(pkg)
            if i > 1:
                setattr(sys.modules[".".join(parts[:i - 1])], parts[i - 1], parent)
        mod = importlib.import_module(ours)
        sys.modules[ref] = mod
        setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)
        installed.append(ref)
    return installed
