"""CPU tests of the host side above the prover call (reference src/main.rs:430-478): typed JSON ->
CBOR -> SP1Stdin buffer, and the CLI's verbs / exit-code contract.  The input files under
tests/golden/ are data files of the reference (examples/, test_vectors/: see README_host_inputs.json).
serde_cbor itself is absent (parity unpinned); the byte layout is cross-checked against the
independent Python encoder below, which restates RFC 8949 + the reference's serde attributes."""
import json
import os
import struct
import subprocess
import sys

import pytest

from dvt_circuits_amd import capi
from tests import guests

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "dvt_circuits_amd", "dvt_prover_host")

# (name, kind, arg): the reference's structs in declaration order (crates/dkg/src/types.rs:26-203)
SETTINGS = [("n", "u8"), ("k", "u8"), ("gen_id", "hex")]
COMMITMENT = [("hash", "hex", "auth"), ("pubkey", "hex"), ("signature", "hex", "auth")]
GENERATION = [("base_pubkeys", "vhex"), ("base_hash", "hex"), ("partial_pubkey", "hex"), ("message_cleartext", "str"), ("message_signature", "hex")]
SCHEMAS = {
    "finalization": [("settings", SETTINGS), ("generations", [GENERATION]), ("aggregate_pubkey", "hex")],
    "bad-share": [("base_hashes", "vhex"),
                  ("initial_commitment", [("hash", "hex"), ("settings", SETTINGS), ("base_pubkeys", "vhex")]),
                  ("seeds_exchange_commitment", [("initial_commitment_hash", "hex"),
                                                 ("ssecret", [("dst_base_hash", "hex"), ("shared_secret", "hex")]),
                                                 ("commitment", COMMITMENT)])],
    "bad-partial-key": [("settings", SETTINGS), ("generations", [[("base_pubkeys", "vhex"), ("base_hash", "hex")]]),
                        ("bad_partial", [("settings", SETTINGS), ("data", GENERATION), ("commitment", COMMITMENT)])],
    "bad-encrypted-share": [("sender_pubkey", "hex"), ("sender_encr_pubkey", "hex"), ("receiver_encr_seckey", "hex"), ("encrypted_data", "str"),
                            ("settings", SETTINGS), ("base_hashes", "vhex"), ("sender_base_pubkeys", "vhex"), ("receiver_base_pubkeys", "vhex")],
}


def head(major, n):
    m = major << 5
    if n < 24:
        return bytes([m | n])
    for code, size in ((24, 1), (25, 2), (26, 4), (27, 8)):
        if n < 1 << (8 * size):
            return bytes([m | code]) + n.to_bytes(size, "big")


def text(s):
    b = s.encode()
    return head(3, len(b)) + b


def enc(schema, obj, auth):
    fields = [f for f in schema if len(f) == 2 or auth]
    out = head(5, len(fields))
    for f in fields:
        name, kind = f[0], f[1]
        v = obj[name]
        out += text(name)
        if kind == "u8":
            out += head(0, v)
        elif kind in ("hex",):
            out += text(v.lower())
        elif kind == "str":
            out += text(v)
        elif kind == "vhex":
            out += head(4, len(v)) + b"".join(text(x.lower()) for x in v)
        elif isinstance(kind, list) and kind and isinstance(kind[0], list):
            out += head(4, len(v)) + b"".join(enc(kind[0], x, auth) for x in v)
        else:
            out += enc(kind, v, auth)
    return out


def c_encode(typ, js: bytes, auth=False):
    try:
        return 0, capi.stdin_from_json(typ, js, auth)
    except capi.DvtError as e:
        return e.code, str(e)


CASES = [("finalization", "finalization_example.json", False), ("finalization", "finalization_no_auth_report1.json", False),
         ("bad-share", "share_no_auth_bad_secret_key.json", False), ("bad-partial-key", "bad_partial_key_no_auth.json", False),
         ("bad-encrypted-share", "bad_encrypted_share_auth.json", True)]


@pytest.mark.parametrize("typ,name,auth", CASES)
def test_stdin_buffer_matches_independent_encoder(typ, name, auth):
    js = open(os.path.join(GOLD, name), "rb").read()
    rc, buf = c_encode(typ, js, auth)
    assert rc == 0, buf
    want = enc(SCHEMAS[typ], json.loads(js), auth)
    assert struct.unpack("<Q", buf[:8])[0] == len(want) == len(buf) - 8      # bincode framing of Vec<u8>
    assert buf[8:] == want


def test_stdin_input_errors():
    js = json.load(open(os.path.join(GOLD, "finalization_example.json")))
    bad = dict(js)
    del bad["aggregate_pubkey"]
    rc, msg = c_encode("finalization", json.dumps(bad).encode())
    assert rc == capi.DVT_ERR_INPUT and "aggregate_pubkey" in msg
    bad = json.loads(json.dumps(js))
    bad["generations"][1]["base_hash"] = bad["generations"][1]["base_hash"][:-2]
    rc, msg = c_encode("finalization", json.dumps(bad).encode())
    assert rc == capi.DVT_ERR_INPUT and "generations[1].base_hash" in msg
    bad = json.loads(json.dumps(js))
    bad["settings"]["n"] = 256          # n is a u8 in the reference (types.rs:30-32)
    assert c_encode("finalization", json.dumps(bad).encode())[0] == capi.DVT_ERR_INPUT
    assert c_encode("no-such-type", b"{}")[0] == capi.DVT_ERR_INPUT
    assert c_encode("finalization", b"{ not json")[0] == capi.DVT_ERR_INPUT
    # the 48-byte BLS commitment key of the stale examples/ files is rejected like the reference's typed parse would
    rc, msg = c_encode("bad-share", open(os.path.join(GOLD, "share_no_auth_bad_secret_key.json"), "rb").read().replace(b'"pubkey": "', b'"pubkey": "00'))
    assert rc == capi.DVT_ERR_INPUT and "pubkey" in msg


def test_cli_execute_contract(tmp_path):
    """`execute` reads the typed JSON, feeds the guest one stdin buffer, and maps guest failure to exit code 1"""
    elf_ok = tmp_path / "ok.elf"
    elf_ok.write_bytes(guests.hint_sum())
    elf_bad = tmp_path / "bad.elf"
    elf_bad.write_bytes(guests.exit_with(1))
    inp = os.path.join(GOLD, "finalization_example.json")
    r = subprocess.run([CLI, "execute", "--type", "finalization", "-i", inp, "--elf", str(elf_ok), "--show-report"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rc, buf = c_encode("finalization", open(inp, "rb").read())
    assert f"input len: {len(buf) - 8}" in r.stdout and "total instructions" in r.stdout
    # the guest really consumed that buffer: its committed word is the sum of the buffer's words
    rc2, rep, pv, err = capi.execute(guests.hint_sum(), [buf])
    padded = buf + b"\0" * (-len(buf) % 4)
    assert rc2 == 0 and struct.unpack("<I", pv)[0] == sum(struct.unpack("<%dI" % (len(padded) // 4), padded)) & 0xFFFFFFFF
    r = subprocess.run([CLI, "execute", "--type", "finalization", "-i", inp, "--elf", str(elf_bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "Verification failed" in r.stderr
    r = subprocess.run([CLI, "execute", "--type", "bad-share", "-i", inp, "--elf", str(elf_ok)], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to read input" in r.stderr     # wrong type for this file
    env = dict(os.environ, DVT_ELF_DIR=str(tmp_path))
    (tmp_path / "finalization.elf").write_bytes(guests.hint_sum())
    assert subprocess.run([CLI, "execute", "--type", "finalization", "-i", inp], env=env, capture_output=True).returncode == 0


SCHEMA_FILES = {"finalization": "finalization_spec.json", "bad-share": "share_exchange_spec.json",
           "bad-partial-key": "bad_partial_key_spec.json", "bad-encrypted-share": "bad_encrypted_partial_key_spec.json"}


def schema_errors(schema: bytes, doc: bytes):
    import ctypes as C

    lib = capi.load()
    lib.dvt_json_schema_validate.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_char_p)]
    err = C.c_char_p()
    rc = lib.dvt_json_schema_validate(schema, len(schema), doc, len(doc), C.byref(err))
    msg = err.value.decode() if err.value else ""
    if err.value:
        lib.dvt_free(C.cast(err, C.c_void_p))
    return rc, msg.split("\n") if msg else []


@pytest.mark.parametrize("typ,name,auth", CASES)
def test_reference_inputs_satisfy_the_reference_schemas(typ, name, auth):
    """--json-schema-file (src/main.rs:509-541) on the reference's own data: its inputs validate against its schemas
    (tests/golden/spec_json = spec/json of the reference)"""
    schema = open(os.path.join(GOLD, "spec_json", SCHEMA_FILES[typ]), "rb").read()
    rc, errs = schema_errors(schema, open(os.path.join(GOLD, name), "rb").read())
    assert rc == 0, errs


def test_schema_violations_are_reported():
    schema = open(os.path.join(GOLD, "spec_json", "finalization_spec.json"), "rb").read()
    js = json.load(open(os.path.join(GOLD, "finalization_example.json")))

    def errs(mutate):
        d = json.loads(json.dumps(js))
        mutate(d)
        rc, e = schema_errors(schema, json.dumps(d).encode())
        assert rc == capi.DVT_ERR_INPUT
        return e

    e = errs(lambda d: d.pop("settings"))
    assert len(e) == 1 and '"settings" is a required property' in e[0]
    e = errs(lambda d: d["settings"].__setitem__("n", "3"))
    assert e == ['$.settings.n: is not of type "integer"']
    e = errs(lambda d: d["settings"].__setitem__("k", -1))
    assert e == ["$.settings.k: is less than the minimum"]
    e = errs(lambda d: d["generations"][1].__setitem__("base_hash", d["generations"][1]["base_hash"][:-1] + "g"))
    assert len(e) == 1 and e[0].startswith("$.generations[1].base_hash: does not match")
    e = errs(lambda d: d.__setitem__("aggregate_pubkey", d["aggregate_pubkey"] + "00"))   # too long AND off-pattern: both reported
    assert len(e) == 2 and all(x.startswith("$.aggregate_pubkey") for x in e)
    e = errs(lambda d: d.__setitem__("generations", {}))
    assert e == ['$.generations: is not of type "array"']
    assert schema_errors(b"{ nope", b"{}")[0] == capi.DVT_ERR_INPUT and schema_errors(schema, b"[1,")[0] == capi.DVT_ERR_INPUT


def test_vector_harness_and_cli_schema_flag(tmp_path):
    """tools/run_vectors.py = the reference's script/run.sh: scenario -> scratch file, cmd_extra_args (with their
    --key=value spelling and schema paths relative to the reference root), exit code against the expectation"""
    meta = json.load(open(os.path.join(GOLD, "README_host_inputs.json")))
    vec_dir, elf_dir, root = tmp_path / "vectors", tmp_path / "elf", tmp_path / "root"
    for d in (vec_dir, elf_dir, root / "spec" / "json"):
        d.mkdir(parents=True)
    for f in os.listdir(os.path.join(GOLD, "spec_json")):
        (root / "spec" / "json" / f).write_bytes(open(os.path.join(GOLD, "spec_json", f), "rb").read())
    for name, m in meta.items():
        if m["params"]:
            (vec_dir / name).write_text(json.dumps({"scenario": json.load(open(os.path.join(GOLD, name))), "params": m["params"]}))
    for typ in SCHEMA_FILES:                                   # stand-in guests: accept every input
        (elf_dir / (typ + ".elf")).write_bytes(guests.hint_sum())
    env = dict(os.environ, DVT_ELF_DIR=str(elf_dir))
    run = lambda *extra: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_vectors.py"), "--cwd", str(root), str(vec_dir), *extra],
                                        env=env, capture_output=True, text=True)
    r = run()
    # the stand-in guests exit 0 on everything: vectors that expect 0 pass, vectors that expect the guest to reject fail
    want_pass = sorted(n for n, m in meta.items() if m["params"] and m["params"]["expected_exit_code"] == 0)
    want_fail = sorted(n for n, m in meta.items() if m["params"] and m["params"]["expected_exit_code"] == 1)
    assert sorted(l.split("/")[-1] for l in r.stdout.splitlines() if l.startswith("[PASS]")) == want_pass, r.stdout
    assert sorted(l.split("/")[-1].split(" ")[0] for l in r.stdout.splitlines() if l.startswith("[FAIL]")) == want_fail
    assert r.returncode == 1 and f"passed {len(want_pass)}  failed {len(want_fail)}" in r.stdout
    r = run("--filter", "share_no_auth")
    assert r.returncode == 0 and "passed 1  failed 0  skipped 3" in r.stdout
    # with guests that reject everything the expectations flip; a schema violation alone is exit code 1 as well
    for typ in SCHEMA_FILES:
        (elf_dir / (typ + ".elf")).write_bytes(guests.exit_with(1))
    assert f"passed {len(want_fail)}  failed {len(want_pass)}" in run().stdout
    bad = json.load(open(os.path.join(GOLD, "finalization_example.json")))
    bad["settings"]["n"] = "three"
    (tmp_path / "bad.json").write_text(json.dumps(bad))
    (elf_dir / "finalization.elf").write_bytes(guests.hint_sum())
    r = subprocess.run([CLI, "execute", "--type=finalization", "--json-schema-file=spec/json/finalization_spec.json", "-i", str(tmp_path / "bad.json")],
                       cwd=root, env=env, capture_output=True, text=True)
    assert r.returncode == 1 and "Validation error in" in r.stderr and "JSON validation failed" in r.stderr


@pytest.mark.gpu
def test_cli_prove_then_verify(tmp_path):
    """the reference's `prove` verb end to end (src/main.rs:448-478): JSON -> stdin -> core proof at the default
    output path, exit code 1 when the guest rejects the input, and stock verification of the saved file"""
    import shutil

    (tmp_path / "finalization.elf").write_bytes(guests.hint_sum())
    inp = tmp_path / "in.json"
    shutil.copy(os.path.join(GOLD, "finalization_example.json"), inp)
    env = dict(os.environ, DVT_ELF_DIR=str(tmp_path))
    r = subprocess.run([CLI, "prove", "--type", "finalization", "-i", str(inp)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    proof_path = str(inp) + "_proof.bin"
    assert f"Proof saved to: {proof_path}" in r.stdout and os.path.getsize(proof_path) > 1000
    r = subprocess.run([CLI, "verify", "--type", "finalization", "-i", proof_path], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and "Proof verified" in r.stdout, r.stderr
    # the file is the C-ABI's proof: the library verifier accepts it under the same key and returns the committed sum
    p = capi.Prover("{}")
    pk, vk = p.setup(guests.hint_sum())
    ok, ec, pv, why = capi.verify(vk, open(proof_path, "rb").read())
    buf = capi.stdin_from_json("finalization", inp.read_bytes())
    padded = buf + b"\0" * (-len(buf) % 4)
    assert ok and struct.unpack("<I", pv)[0] == sum(struct.unpack("<%dI" % (len(padded) // 4), padded)) & 0xFFFFFFFF, why
    p.pk_free(pk)
    p.close()
    # a guest that exits non-zero: no proof, exit code 1 (what script/run.sh:82-89 observes for the negative vectors)
    (tmp_path / "finalization.elf").write_bytes(guests.exit_with(1))
    out = tmp_path / "neg.bin"
    r = subprocess.run([CLI, "prove", "--type", "finalization", "-i", str(inp), "-o", str(out)], env=env, capture_output=True, text=True)
    assert r.returncode == 1 and "Proof generation failed" in r.stderr and not out.exists()


def test_commitment_hash_formula_and_synthetic_inputs():
    """(1) KAT from the reference's own data: every base_hash of its finalization example is
    SHA-256(gen_id || n || k || len || base_pubkeys) (crates/dkg/src/verification.rs:151-175);
    (2) tools/gen_dkg_input.py builds n-participant inputs with that formula, which pass the reference's schema and the
    host encoder, and whose encoded size follows SURVEY.md section 8(a1)'s model (about 170 KB at n = 255, k = 2)"""
    import hashlib

    from tools import gen_dkg_input

    ex = json.load(open(os.path.join(GOLD, "finalization_example.json")))
    st = ex["settings"]
    for g in ex["generations"]:
        h = hashlib.sha256(bytes.fromhex(st["gen_id"]) + bytes([st["n"], st["k"], len(g["base_pubkeys"])]) + b"".join(bytes.fromhex(p) for p in g["base_pubkeys"]))
        assert h.hexdigest() == g["base_hash"].lower()
    schema = open(os.path.join(GOLD, "spec_json", "finalization_spec.json"), "rb").read()
    small = gen_dkg_input.finalization(3, 2)
    assert gen_dkg_input.finalization(3, 2) == small and gen_dkg_input.finalization(3, 2, seed=7) != small
    assert [g["base_hash"] for g in small["generations"]] == sorted(g["base_hash"] for g in small["generations"])
    assert schema_errors(schema, json.dumps(small).encode())[0] == 0
    rc, buf = c_encode("finalization", json.dumps(small).encode())
    assert rc == 0 and abs(len(buf) - len(capi.stdin_from_json("finalization", open(os.path.join(GOLD, "finalization_example.json"), "rb").read()))) < 16
    big = gen_dkg_input.finalization(255, 2)
    assert schema_errors(schema, json.dumps(big).encode())[0] == 0
    rc, buf = c_encode("finalization", json.dumps(big).encode())
    assert rc == 0 and 160_000 < len(buf) < 185_000
    assert buf[8:] == enc(SCHEMAS["finalization"], big, False)
