"""TEST TOOLING: the reference's finalization check restated in Python over tools/bls12_381.py — what the guest
`finalization_prove` runs inside the zkVM (reference crates/finalization_prove/src/main.rs:8-32 ->
crates/dkg/src/verification.rs:211-331, crates/dkg/src/dkg_math.rs:160-248).  Used to certify that
tools/gen_dkg_input.py emits inputs a DKG-verifying guest accepts, and pinned by the reference's own real vectors
(tests/golden/finalization_example.json = examples/finalization_test.json, finalization_no_auth_report1.json =
test_vectors/no_auth/finalization/report-1.json, which the reference's harness expects to exit 0).

verify_finalization(doc) returns None when the input is accepted, else the reason (the reference's error text class)."""
import hashlib

from tools import bls12_381 as B


def commitment_hash(gen_id: bytes, n: int, k: int, base_pubkeys) -> bytes:
    """reference crates/dkg/src/verification.rs:151-175: SHA-256(gen_id || n || k || len(base_pubkeys) as u8 || pubkeys...)"""
    return hashlib.sha256(gen_id + bytes([n, k, len(base_pubkeys) & 0xFF]) + b"".join(base_pubkeys)).digest()


def evaluate_polynomial(cfs, x: int):
    """Horner over G1 points (dkg_math.rs:160-174)"""
    if not cfs:
        return None
    y = cfs[-1]
    for c in reversed(cfs[:-1]):
        y = B.E1.add(B.E1.mul(y, x), c)
    return y


def lagrange_interpolation(ys, xs):
    """value at 0 of the interpolating polynomial (dkg_math.rs:178-227); raises ValueError as the reference returns Err"""
    k = len(xs)
    if k == 0 or k != len(ys):
        raise ValueError("invalid inputs")
    if k == 1:
        return ys[0]
    a = 1
    for x in xs:
        a = a * x % B.R
    if a == 0:
        raise ValueError("zero secret share id")
    r = None
    for i in range(k):
        b = xs[i]
        for j in range(k):
            if j != i:
                v = (xs[j] - xs[i]) % B.R
                if v == 0:
                    raise ValueError("duplicate secret share id")
                b = b * v % B.R
        li0 = a * pow(b, B.R - 2, B.R) % B.R
        r = B.E1.add(r, B.E1.mul(ys[i], li0))
    return r


def agg_coefficients(vectors, ids):
    """dkg_math.rs:230-248: sum the verification vectors coefficient-wise, evaluate the sum polynomial at every id"""
    final = []
    for i in range(len(vectors[0])):
        s = None
        for v in vectors:
            s = B.E1.add(s, v[i])
        final.append(s)
    return [evaluate_polynomial(final, x) for x in ids]


def verify_finalization(doc, check_signatures=True):
    B.ensure_ready()
    st = doc["settings"]
    n, k, gen_id = st["n"], st["k"], bytes.fromhex(st["gen_id"])
    gens = doc["generations"]
    if len(gens) != n:
        return "Invalid number of generations"
    if any(g["message_cleartext"] != gens[0]["message_cleartext"] for g in gens):
        return "Invalid message cleartext"
    hm = B.hash_to_g2(gens[0]["message_cleartext"].encode())          # one hash-to-G2 for all (verification.rs:233-235)
    for g in gens:
        try:
            sig = B.g2_decompress(bytes.fromhex(g["message_signature"]))
            key = B.g1_decompress(bytes.fromhex(g["partial_pubkey"]))
        except ValueError as e:
            return f"Invalid point: {e}"
        if check_signatures and not B.pairing_product_is_one([(key, hm), (B.E1.neg(B.G1), sig)]):
            return "Invalid signature " + g["message_signature"]
        pks = [bytes.fromhex(p) for p in g["base_pubkeys"]]
        if commitment_hash(gen_id, n, k, pks).hex() != g["base_hash"]:
            return "Invalid initial commitment hash " + g["base_hash"]
    ordered = sorted(gens, key=lambda g: bytes.fromhex(g["base_hash"]))
    try:
        vectors = [[B.g1_decompress(bytes.fromhex(p)) for p in g["base_pubkeys"]] for g in ordered]
        partial = [B.g1_decompress(bytes.fromhex(g["partial_pubkey"])) for g in ordered]
        agg = B.g1_decompress(bytes.fromhex(doc["aggregate_pubkey"]))
    except ValueError as e:
        return f"Invalid point: {e}"
    ids = list(range(1, n + 1))
    try:
        computed = lagrange_interpolation(agg_coefficients(vectors, ids), ids)
        if B.g1_compress(computed) != B.g1_compress(agg):
            return "Computed key does not match aggregate public key (verification vectors)"
        computed = lagrange_interpolation(partial, ids)
        if B.g1_compress(computed) != B.g1_compress(agg):
            return "Computed key does not match aggregate public key (partial keys)"
    except ValueError as e:
        return str(e)
    return None
