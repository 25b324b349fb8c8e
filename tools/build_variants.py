#!/usr/bin/env python3
"""Build kernel-variant libraries openwakeword_amd/libowwhip_<tag>.so for A/B runs (tools/ab.sh).
usage: tools/build_variants.py tag=DEF1,DEF2 ..."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from openwakeword_amd import _build
for spec in sys.argv[1:]:
    tag, _, defs = spec.partition("=")
    out = os.path.join(_build.HERE, f"libowwhip_{tag}.so")
    _build.build(force=True, out=out, defines=tuple(d for d in defs.split(",") if d))
    print("built", out)
