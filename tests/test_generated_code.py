"""The generated AIR code (product C++ templates, oracle C, column ids) and the Poseidon2 constant table are in sync
with their generators: a change to tools/airgen/*.py or tools/gen_poseidon2_rc.py without regenerating fails here, and
so does a hand edit of a generated file."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_air_code_is_up_to_date():
    from tools.airgen import emit

    for name in ("toy", "rv32"):
        m = importlib.import_module(f"tools.airgen.{name}").build()
        for path, text in ((("dvt_circuits_amd", "csrc", "gen", f"air_{name}.inc"), emit.emit_cpp(m)),
                           (("oracle", "gen", f"air_{name}.c"), emit.emit_c(m)),
                           (("dvt_circuits_amd", "csrc", "gen", f"{name}_cols.h"), emit.emit_cols_header(m)),
                           (("dvt_circuits_amd", "csrc", "gen", f"{name}_rels.h"), emit.emit_rels_header(m))):
            with open(os.path.join(ROOT, *path)) as f:
                assert f.read() == text, f"{'/'.join(path)} is stale: run python tools/gen_air.py"


def test_cpu_chip_shape():
    """the numbers DESIGN.md quotes: 96 main columns = 12 sponge blocks exactly, 28 interactions = 14 batches = 13 batch
    columns + phi = 56 permutation columns = 7 sponge blocks exactly"""
    from tools.airgen import rv32

    cpu = next(c for c in rv32.build().chips if c.name == "cpu")
    assert cpu.main_width == 96 and cpu.main_width % 8 == 0 and len(cpu.interactions) == 28
    assert 4 * ((len(cpu.interactions) + 1) // 2) == 56


def test_poseidon2_constants_are_up_to_date(tmp_path):
    path = os.path.join(ROOT, "dvt_circuits_amd", "csrc", "poseidon2_rc.inc")
    before = open(path).read()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_poseidon2_rc.py")], stdout=subprocess.DEVNULL)
    assert open(path).read() == before, "poseidon2_rc.inc was stale (it has been regenerated: rebuild and commit)"
