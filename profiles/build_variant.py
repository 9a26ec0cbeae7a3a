"""Builds an experiment variant of the library: lib/var_<name>.so with extra -D flags (loaded through YCGE_LIB).
    python profiles/build_variant.py <name> [-DFLAG=1 ...]"""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from yetanotherconsolegameengine_amd import build as b
name, flags = sys.argv[1], sys.argv[2:]
print("built", b.build_variant(name, flags, force=True))
