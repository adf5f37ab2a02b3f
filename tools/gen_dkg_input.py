#!/usr/bin/env python3
"""Synthetic n-participant `finalization` input in the reference's JSON format (crates/dkg/src/types.rs:176-203,
examples/finalization_test.json), for workloads larger than the reference's own example (SURVEY.md section 8d,
configs 3-4).

finalization(n, k, real_keys=True) runs a real (k, n) DKG over BLS12-381 (tools/bls12_381.py, pinned by the reference's
KATs): n polynomials of k random scalars, base_pubkeys_i[j] = a_ij G1, commitment hashes, generations ordered by hash
-> ids 1..n, partial keys S_i G1 with S_i = sum_j f_j(id_i), signatures S_i H(msg), aggregate sum_j a_j0 G1
(SURVEY.md section 8d).  Such an input is accepted by the reference's finalization check (restated in
tools/dkg_verify.py; tests/test_bls_tooling.py).  real_keys=False keeps the structure, sizes and commitment hashes
(crates/dkg/src/verification.rs:151-175) but fills the 48- / 96-byte points with pseudo-random bytes: fast (no curve
arithmetic), enough to size a workload for the synthetic guests, rejected by a DKG-verifying guest.

Randomness: SHA-256 in counter mode over seed = 0xD17C0DE5 + n (sha256(seed_le64 || ctr_le64)), as SURVEY.md prescribes.

    python tools/gen_dkg_input.py --n 255 --k 2 > finalization_n255.json"""
import argparse
import hashlib
import json
import struct
import sys

MESSAGE = "Sign with new partial key"


class Stream:
    def __init__(self, seed):
        self.seed, self.ctr, self.buf = seed, 0, b""

    def take(self, n):
        while len(self.buf) < n:
            self.buf += hashlib.sha256(struct.pack("<QQ", self.seed, self.ctr)).digest()
            self.ctr += 1
        out, self.buf = self.buf[:n], self.buf[n:]
        return out


def finalization_real(n, k, seed=None):
    """a valid instance: see the module docstring"""
    from tools import bls12_381 as B

    assert 1 <= k <= n <= 255
    B.ensure_ready()
    rnd = Stream(0xD17C0DE5 + n if seed is None else seed)
    gen_id = rnd.take(16)
    polys = [[int.from_bytes(rnd.take(48), "big") % B.R for _ in range(k)] for _ in range(n)]
    gens = []
    for coeffs in polys:
        pubkeys = [B.g1_compress(B.E1.mul(B.G1, a)) for a in coeffs]
        h = hashlib.sha256(gen_id + bytes([n, k, len(pubkeys)]) + b"".join(pubkeys)).digest()
        gens.append(dict(coeffs=coeffs, pubkeys=pubkeys, hash=h))
    gens.sort(key=lambda g: g["hash"])
    hm = B.hash_to_g2(MESSAGE.encode())
    out = []
    for i, g in enumerate(gens):
        x = i + 1
        share = sum(sum(a * pow(x, j, B.R) for j, a in enumerate(q["coeffs"])) for q in gens) % B.R      # S_i = sum_j f_j(id_i)
        out.append({
            "base_pubkeys": [p.hex() for p in g["pubkeys"]],
            "base_hash": g["hash"].hex(),
            "partial_pubkey": B.g1_compress(B.E1.mul(B.G1, share)).hex(),
            "message_cleartext": MESSAGE,
            "message_signature": B.g2_compress(B.E2.mul(hm, share)).hex(),
        })
    agg = B.E1.mul(B.G1, sum(g["coeffs"][0] for g in gens) % B.R)
    return {"settings": {"n": n, "k": k, "gen_id": gen_id.hex()}, "generations": out, "aggregate_pubkey": B.g1_compress(agg).hex()}


def finalization(n, k, seed=None, real_keys=False):
    if real_keys:
        return finalization_real(n, k, seed)
    assert 1 <= k <= n <= 255
    rnd = Stream(0xD17C0DE5 + n if seed is None else seed)
    gen_id = rnd.take(16)
    gens = []
    for _ in range(n):
        pubkeys = [rnd.take(48) for _ in range(k)]
        h = hashlib.sha256(gen_id + bytes([n, k, len(pubkeys)]) + b"".join(pubkeys)).digest()
        gens.append({
            "base_pubkeys": [p.hex() for p in pubkeys],
            "base_hash": h.hex(),
            "partial_pubkey": rnd.take(48).hex(),
            "message_cleartext": MESSAGE,
            "message_signature": rnd.take(96).hex(),
        })
    gens.sort(key=lambda g: g["base_hash"])
    return {"settings": {"n": n, "k": k, "gen_id": gen_id.hex()}, "generations": gens, "aggregate_pubkey": rnd.take(48).hex()}


def bad_encrypted_share(n, k, seed=None):
    """Synthetic `bad-encrypted-share` input (crates/dkg/src/types.rs:182-203): n base hashes, k + k base pubkeys, a
    178-byte share message (crates/bad_encrypted_share_prove/src/main.rs:150-176: gen_id[16] || type = 3 || secret[32]
    || commit_hash[32] || commit_pubkey[33] || commit_sig[64]) as the hex string `encrypted_data`.  Same caveat as
    finalization(): field sizes and structure are the reference's, the key material is pseudo-random bytes."""
    assert 1 <= k <= n <= 255
    rnd = Stream(0xD17C0DE5 + 1000 + n if seed is None else seed)
    gen_id = rnd.take(16)
    message = gen_id + bytes([3]) + rnd.take(32) + rnd.take(32) + rnd.take(33) + rnd.take(64)
    return {
        "sender_pubkey": rnd.take(33).hex(),
        "sender_encr_pubkey": rnd.take(48).hex(),
        "receiver_encr_seckey": rnd.take(32).hex(),
        "encrypted_data": message.hex(),
        "settings": {"n": n, "k": k, "gen_id": gen_id.hex()},
        "base_hashes": sorted(rnd.take(32).hex() for _ in range(n)),
        "sender_base_pubkeys": [rnd.take(48).hex() for _ in range(k)],
        "receiver_base_pubkeys": [rnd.take(48).hex() for _ in range(k)],
    }


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, required=True)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=None)
    ap.add_argument("--type", choices=("finalization", "bad-encrypted-share"), default="finalization")
    ap.add_argument("--real-keys", action="store_true", help="finalization: a valid DKG instance over BLS12-381 (slow: pure-Python curve arithmetic)")
    a = ap.parse_args()
    doc = finalization(a.n, a.k, a.seed, a.real_keys) if a.type == "finalization" else bad_encrypted_share(a.n, a.k, a.seed)
    json.dump(doc, sys.stdout, indent=1)
    sys.stdout.write("\n")
