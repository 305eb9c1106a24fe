"""Import shim: `import forge_amd` loads the package stored in ./stable-diffusion-webui-forge_amd/."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stable-diffusion-webui-forge_amd")
_spec = importlib.util.spec_from_file_location("forge_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["forge_amd"] = _pkg
_spec.loader.exec_module(_pkg)
