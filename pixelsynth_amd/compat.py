"""Make the reference's import paths resolve to the MI355X implementation (option A), so
drivers written against crockwell/pixelsynth (`demo.py`, `create_vid.py`, `evaluation/*.py`) pick up the HIP
path without source edits:

    import pixelsynth_amd.compat; pixelsynth_amd.compat.install_reference_aliases()
"""
import importlib
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


def install_reference_aliases(force=False):
    """Register the mirrors under the reference's module names.  Parent packages that are not importable
    (the reference tree absent) are created as empty namespace modules."""
    installed = []
    for ref, ours in ALIASES.items():
        if ref in sys.modules and not force:
            continue
        parts = ref.split(".")
        for i in range(1, len(parts)):
            pkg = ".".join(parts[:i])
            if pkg not in sys.modules:
                m = types.ModuleType(pkg)
                m.__path__ = []
                sys.modules[pkg] = m
        mod = importlib.import_module(ours)
        sys.modules[ref] = mod
        setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)
        installed.append(ref)
    return installed
