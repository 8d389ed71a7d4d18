"""A / B timing of two builds of libconex.so on the SAME box (box-to-box spread of the headline kernels
is +-3 %, a change worth keeping is often smaller): the C4 step and its two kernels, alternating
A B A B ..., one subprocess per run.

    python tools/ab_bench.py scratch/libconex_a.so scratch/libconex_b.so [rounds] [--workload c5]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
import conex_amd.kkt as kk
kk.LIB_PATH = os.environ["CXK_AB_LIB"]
sys.argv = ["bench.py", "--no-cpu", "--no-newton-step", "--cold-copies", "0"] + sys.argv[1:]
import runpy
runpy.run_path(os.path.join(%r, "bench.py"), run_name="__main__")
""" % (ROOT, ROOT)


def run(lib, extra):
    env = dict(os.environ, CXK_AB_LIB=os.path.abspath(lib))
    out = subprocess.run([sys.executable, "-c", CHILD] + extra, env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise SystemExit(out.stderr[-2000:])
    d = json.loads(line[-1])
    return d["ms_per_step"] * 1e3, d.get("roofline", {}).get("kernel_ms", 0) * 1e3, d.get("roofline_tree", {}).get("kernel_ms", 0) * 1e3


def main():
    a, b = sys.argv[1], sys.argv[2]
    rest = sys.argv[3:]
    rounds = int(rest[0]) if rest and rest[0].isdigit() else 3
    extra = [x for x in rest if not x.isdigit()]
    res = {a: [], b: []}
    for _ in range(rounds):
        for lib in (a, b):
            res[lib].append(run(lib, extra))
    for lib in (a, b):
        r = res[lib]
        print("%-28s step %s us | assembly %s us | tree %s us" % (
            os.path.basename(lib), " ".join("%.2f" % x[0] for x in r), " ".join("%.2f" % x[1] for x in r),
            " ".join("%.2f" % x[2] for x in r)))


if __name__ == "__main__":
    main()
