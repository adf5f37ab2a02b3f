#!/usr/bin/env python3
"""Writes the guest ELFs the host CLI looks up in $DVT_ELF_DIR (the reference embeds its guests at build time, reference
build.rs:56-73 / src/main.rs:115-118): `finalization.elf` = the re-stated finalization guest of tests/guests_finalization.py.

    python tools/build_guests.py OUT_DIR [NMAX KMAX]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    from tests import guests_finalization

    out = sys.argv[1]
    nmax, kmax = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 8)
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "finalization.elf"), "wb") as f:
        f.write(guests_finalization.finalization(nmax=nmax, kmax=kmax))
    print("wrote", os.path.join(out, "finalization.elf"))


if __name__ == "__main__":
    main()
