#!/usr/bin/env python3
"""Regenerate the AIR code of every machine:
   dvt_circuits_amd/csrc/gen/air_<m>.inc (product) and oracle/gen/air_<m>.c (oracle)."""
import importlib
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from tools.airgen import emit  # noqa: E402

MACHINES = ["toy", "rv32"]


def main():
    for name in MACHINES:
        try:
            mod = importlib.import_module(f"tools.airgen.{name}")
        except ModuleNotFoundError:
            continue
        m = mod.build()
        for path, text in (
            (os.path.join(ROOT, "dvt_circuits_amd", "csrc", "gen", f"air_{name}.inc"), emit.emit_cpp(m)),
            (os.path.join(ROOT, "oracle", "gen", f"air_{name}.c"), emit.emit_c(m)),
            (os.path.join(ROOT, "dvt_circuits_amd", "csrc", "gen", f"{name}_cols.h"), emit.emit_cols_header(m)),
            (os.path.join(ROOT, "dvt_circuits_amd", "csrc", "gen", f"{name}_rels.h"), emit.emit_rels_header(m)),
        ):
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                f.write(text)
            print("wrote", os.path.normpath(path))
        for ch in m.chips:
            print(f"  {name}.{ch.name}: main {ch.main_width} prep {ch.prep_width} constraints {len(ch.constraints)} interactions {len(ch.interactions)}")


if __name__ == "__main__":
    main()
