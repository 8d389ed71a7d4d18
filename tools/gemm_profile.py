"""Isolated rates of the fp64 MFMA GEMM (cxk_gemm_f64) at the supernode shapes SURVEY 8(d) names.

SYRK-shaped trailing updates  C (n_s x n_s, lower) -= L L^T  with L n_s x s  (M = N = n_s, K = s,
TB, alpha = -1, beta = 1, lower_only) for n_s in {64, 128, 200}, s in {64, 128}; the separator
update  U (s x s) = off^T off  over K = n_s (TA); and cubes for reference.  Every line carries the
arithmetic intensity of the call, the roof that binds it (fp64 MFMA peak as MEASURED by
tools/mfma_f64_peak.hip, HBM 8 TB/s) and the achieved fraction of that roof.

  python tools/gemm_profile.py > gpurun_out/r02/gemm_rates.jsonl
  rocprofv3 --kernel-trace --stats ... -- python3 tools/gemm_profile.py --quick
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from conex_amd import kkt

MFMA_PEAK_TF = 77.5   # profiles/r02/mfma_f64_peak.jsonl (v_mfma_f64_16x16x4 back to back, every CU)
HBM_PEAK_GBS = 8000.0


def run(name, M, N, K, batch, ta, tb, alpha, beta, lower, reps):
    rng = np.random.default_rng(M + N + K)
    A = rng.uniform(-1, 1, (batch, M, K))
    B = rng.uniform(-1, 1, (batch, K, N))
    C0 = rng.uniform(-1, 1, (batch, M, N)) if beta != 0 else None
    _, ms = kkt.gemm_f64(A, B, C0, ta=ta, tb=tb, alpha=alpha, beta=beta, lower_only=lower, reps=reps)
    useful = (M * (N + 1) / 2 if lower else M * N) * K * 2.0 * batch          # flops of the entries kept
    out_elems = (M * (N + 1) / 2 if lower else M * N) * batch
    bytes_ = 8.0 * ((M * K + K * N) * batch + out_elems * (2 if beta != 0 else 1))
    tf = useful / (ms * 1e-3) / 1e12
    gbs = bytes_ / (ms * 1e-3) / 1e9
    ai = useful / bytes_
    bound = "mfma" if ai >= MFMA_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9) else "hbm"
    frac = tf / MFMA_PEAK_TF if bound == "mfma" else gbs / HBM_PEAK_GBS
    print(json.dumps({"shape": name, "M": M, "N": N, "K": K, "batch": batch, "ms": ms, "useful_tflops": tf,
                      "GBps": gbs, "flop_per_byte": ai, "bound": bound, "frac_of_bound": frac,
                      "frac_of_mfma_peak": tf / MFMA_PEAK_TF}), flush=True)


def main():
    quick = "--quick" in sys.argv
    reps = 3 if quick else 10
    for ns in (64, 128, 200):
        for s in (64, 128):
            batch = 2048 if ns == 64 else (1024 if ns == 128 else 512)
            run(f"syrk ns={ns} s={s}: C -= L L^T (lower)", ns, ns, s, batch, False, True, -1.0, 1.0, True, reps)
    for s in (64, 128):
        for ns in (128, 200):
            run(f"separator update s={s} ns={ns}: U = off^T off", s, s, ns, 1024, True, False, 1.0, 0.0, False, reps)
    for n, batch in ((64, 4096), (128, 1024), (200, 512), (1024, 4)):
        run(f"cube n={n}", n, n, n, batch, False, False, 1.0, 0.0, False, reps)


if __name__ == "__main__":
    main()
