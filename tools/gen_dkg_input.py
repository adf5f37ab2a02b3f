#!/usr/bin/env python3
"""Synthetic n-participant `finalization` input in the reference's JSON format (crates/dkg/src/types.rs:176-203,
examples/finalization_test.json), for workloads larger than the reference's own example (SURVEY.md section 8d,
configs 3-4).

What is real: the structure, every field size, the commitment hashes base_hash = SHA-256(gen_id || n || k || len ||
base_pubkeys...) (crates/dkg/src/verification.rs:151-175; tests/test_host_stdin.py pins this formula against the
reference's example file) and the ordering of the generations by that hash.  What is NOT: the 48- / 96-byte "points"
are pseudo-random bytes, not BLS12-381 elements (this image has no pairing library and the current-source guests that
would check them cannot be built here), so the file exercises the host path and sizes the workload; a DKG-verifying
guest would reject it.

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


def finalization(n, k, seed=None):
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
    a = ap.parse_args()
    json.dump((finalization if a.type == "finalization" else bad_encrypted_share)(a.n, a.k, a.seed), sys.stdout, indent=1)
    sys.stdout.write("\n")
