#!/usr/bin/env python3
"""Exit-code harness over the reference's test-vector format (what script/run.sh does with jq + bash, reference
script/run.sh:44-104): every *.json under the given directories holds

    {"scenario": <the typed input>, "params": {"cmd_extra_args": "execute --type=... --json-schema-file=...",
                                               "expected_exit_code": 0|1, "disabled": false}}

The scenario is written to a scratch file, `dvt_prover_host <cmd_extra_args> --input-file <scratch>` runs in `--cwd`
(the vectors name their schema files relative to the reference's root), and the process exit code is compared with the
expectation.  Guests come from $DVT_ELF_DIR/<type>.elf (the reference embeds them at build time).

    python tools/run_vectors.py [--filter REGEX] [--cwd DIR] [--host PATH] DIR...

Exit code 0 when every enabled vector met its expectation."""
import argparse
import json
import os
import re
import shlex
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--filter", default="")
    ap.add_argument("--cwd", default=".")
    ap.add_argument("--host", default=os.path.join(ROOT, "dvt_circuits_amd", "dvt_prover_host"))
    args = ap.parse_args()
    files = sorted(os.path.join(d, f) for top in args.dirs for d, _, fs in os.walk(top) for f in fs if f.endswith(".json"))
    passed = failed = skipped = disabled = 0
    failures = []
    for path in files:
        if args.filter and not re.search(args.filter, path):
            skipped += 1
            continue
        with open(path) as f:
            vec = json.load(f)
        params = vec.get("params") or {}
        if params.get("disabled") in (True, "true"):
            disabled += 1
            continue
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as t:
            json.dump(vec["scenario"], t)
        try:
            r = subprocess.run([args.host] + shlex.split(params["cmd_extra_args"]) + ["--input-file", t.name], cwd=args.cwd,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        finally:
            os.unlink(t.name)
        want = int(params["expected_exit_code"])
        if r.returncode == want:
            print(f"[PASS] {path}")
            passed += 1
        else:
            print(f"[FAIL] {path} (expected exit code: {want}, got {r.returncode})")
            failed += 1
            failures.append(path)
    print(f"passed {passed}  failed {failed}  skipped {skipped}  disabled {disabled}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
