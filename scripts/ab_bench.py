#!/usr/bin/env python3
"""A/B the bench line under environment switches in ONE GPU call (every variant: `bench.py` without the slow legs, parity leg on).

    python scripts/ab_bench.py [--config C4] [--repeats 20] base: launches:SICP_LM=launches nowindow:SICP_TAIL_WINDOW=0

Each argument is NAME:VAR=VALUE[,VAR=VALUE...] (NAME: alone = the defaults).  Prints one row per variant -- iterations/s, ms per
step, instrumented match / tail / selection times, solver evaluations per iteration, parity verdict -- and writes the JSON lines
to gpurun_out/ab/<NAME>.json."""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C4")
    ap.add_argument("--repeats", type=int, default=20)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    out = ROOT / "gpurun_out" / "ab"
    out.mkdir(parents=True, exist_ok=True)
    rows = []
    for spec in a.variants:
        name, _, envs = spec.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        f = out / f"{name}.json"
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", a.config, "--repeats", str(a.repeats), "--steps",
                            str(a.steps), "--no-cpu-baseline", "--no-end-to-end", "--no-bruteforce-leg", "--out", str(f)],
                           env=env, capture_output=True, text=True)
        if r.returncode != 0 or not f.exists():
            rows.append((name, envs, None, r.stderr.strip().splitlines()[-1:] or ["failed"]))
            continue
        rows.append((name, envs, json.loads(f.read_text()), None))
    print(f"{'variant':14s} {'it/s':>9s} {'us/step':>8s} {'p10':>7s} {'p90':>7s} {'match':>7s} {'tail':>7s} {'select':>7s} {'evals':>6s} parity  switches")
    for name, envs, d, err in rows:
        if d is None:
            print(f"{name:14s} FAILED: {err[0]}")
            continue
        k, rs = d["kernels_instrumented"], d["repeat_stats"]
        print(f"{name:14s} {d['value']:9.0f} {d['ms_per_step'] * 1e3:8.2f} {rs['ms_per_step_p10'] * 1e3:7.2f} {rs['ms_per_step_p90'] * 1e3:7.2f} "
              f"{k['match']['avg_ms'] * 1e3:7.2f} {k['solve']['avg_ms'] * 1e3:7.2f} {k['reject_select']['avg_ms'] * 1e3:7.2f} "
              f"{d['solver']['normal_eq_evaluations_per_iteration']:6.2f} {str(d.get('parity', {}).get('ok')):6s}  {envs}")


if __name__ == "__main__":
    main()
