"""dir_amd -- importable name of the package that lives in ./details-in-recommendation_amd/.

The directory name mandated for this repo contains hyphens and cannot be an import name; this module
sets __path__ so that `import dir_amd`, `import dir_amd.ops`, `from dir_amd.deepfm import DeepFM` resolve
to the files in that directory (a module with __path__ is a package to the import system).
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "details-in-recommendation_amd")]
__version__ = "0.1.0"

from dir_amd._lib import load as load_library, library_path  # noqa: E402,F401
