#!/usr/bin/env python3
"""tools/ab.py -- A/B timing of library builds on one GPU: interleaved bench.py runs (methodology: same device, same
process sequence, medians over rounds), blur serialised so that every stage time is a stand-alone kernel time.

usage: python tools/ab.py [--rounds N] name=path/to/lib.so [name=path ...]     ("cur" = the in-tree build is always included)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(lib, serial):
    env = dict(os.environ)
    if lib:
        env["VSLAM_AMD_LIB"] = lib
    if serial:
        env["VSLAM_AMD_SERIAL_BLUR"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline",
                          "--no-optin"], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise SystemExit("bench failed for %s:\n%s\n%s" % (lib, out.stdout[-2000:], out.stderr[-2000:]))
    return json.loads(line[-1])


def main():
    args = sys.argv[1:]
    rounds = 3
    if args and args[0] == "--rounds":
        rounds = int(args[1]); args = args[2:]
    variants = [("cur", None)] + [tuple(a.split("=", 1)) for a in args]
    variants = [(n, os.path.abspath(p) if p else None) for n, p in variants]
    acc = {n: [] for n, _ in variants}
    for r in range(rounds):
        for n, p in variants:
            acc[n].append((run(p, True), run(p, False)))
    med = lambda xs: sorted(xs)[len(xs) // 2]
    stages = list(acc["cur"][0][0]["stage_ms"])
    print("%-8s %9s %9s | " % ("variant", "overlap", "serial") + " ".join("%9s" % s[:9] for s in stages))
    for n, _ in variants:
        ser = [a[0] for a in acc[n]]; ovl = [a[1] for a in acc[n]]
        print("%-8s %9.3f %9.3f | " % (n, med([o["ms_per_step"] for o in ovl]), med([s["ms_per_step"] for s in ser])) +
              " ".join("%9.4f" % med([s["stage_ms"][st] for s in ser]) for st in stages))


if __name__ == "__main__":
    main()
