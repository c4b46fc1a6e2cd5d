"""skani_amd -- MI355X-native ANI engine: host-side mirror of skani's library interface over the C ABI of
libskani_hip.so (include/skani_hip.h).  There is no CPU path: importing works anywhere, but creating a
Context requires the built HIP library and a gfx950 device, and fails loudly otherwise."""
import os

from . import _binding
from .api import (SEED_AVX2, SEED_SCALAR, Context, GenomeSet, MapParams, SketchParams, SketchSet, SkaniHipError,
                  fastx_to_multiple_sketch_rewrite, fastx_to_sketches, use_learned_ani)
from .formats import load_database, load_sketch_files, save_database
from .search import SketchDB, build_db, open_db, search

__all__ = ["load_database", "load_sketch_files", "save_database", "Context", "GenomeSet", "SketchSet", "SketchParams", "MapParams", "SkaniHipError", "fastx_to_sketches",
           "fastx_to_multiple_sketch_rewrite", "SketchDB", "open_db", "build_db", "search",
           "use_learned_ani", "SEED_SCALAR", "SEED_AVX2", "library_path"]


def library_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "libskani_hip.so")
