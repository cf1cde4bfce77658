"""The batched (bf16-screened) path against the exact path on STRUCTURED dictionaries: does any signal come back with a
support that differs from csmp_omp_batch's WITHOUT having been flagged `uncertain` (and re-solved)?  The screening
certificate's default error bound is a statistical model of independent bf16 roundings; few-valued, partial-DCT,
sign and common-component dictionaries are where roundings could add coherently.
Prints one JSON line per (dictionary, signal family, certificate mode).  usage: python tools/probe_structured.py [quick]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from csmp_pkg import load

cs = load()


KINDS = ("few_valued", "partial_dct", "common_component", "signs", "one_magnitude")


def signals(A, k, nsig, family, rng):
    M, N = A.shape
    B = np.empty((M, nsig), order="F")
    for s in range(nsig):
        sup = rng.choice(N, size=k, replace=False)
        if family == "pm1":
            x = rng.choice(np.array([-1.0, 1.0]), size=k)
        elif family == "gauss":
            x = rng.standard_normal(k)
        elif family == "neartie":  # coefficients within 0.2 % of each other: the exact correlations nearly tie
            x = rng.choice(np.array([-1.0, 1.0]), size=k) * (1.0 + 2e-3 * rng.random(k))
        else:  # "decay": a dominant atom and a tail -- |c| / |r| is large at the first steps
            x = rng.choice(np.array([-1.0, 1.0]), size=k) * 0.5 ** np.arange(k)
        b = A[:, sup] @ x
        e = rng.standard_normal(M)
        B[:, s] = b + e * (5e-3 / np.linalg.norm(e))
    return B


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    M, N = (512, 4096) if quick else (1024, 8192)
    nsig = 256
    rng = np.random.default_rng(20261003)
    eps = float(np.finfo(np.float32).eps)
    for name in KINDS:
        A = cs.structured_dictionary(name, M, N, rng=rng)
        d = cs.Dictionary(A)
        for family, k in (("pm1", 24), ("gauss", 24), ("decay", 8), ("pm1", 2), ("neartie", 2), ("neartie", 6)):
            B = signals(A.astype(np.float64), k, nsig, family, rng)
            i2, v2, n2 = d.ctx.omp_batch(B, k, eps)
            for mode in ("default", "rigorous", "int8") + (("round2",) if os.environ.get("CSMP_PROBE_ROUND2") else ()):
                d.ctx.set_option("batch_cert", 1 if mode == "rigorous" else 0)
                d.ctx.set_option("batch_screen", 1 if mode == "int8" else 0)  # int8 operands of the screening GEMM (statistical bound)
                if mode == "round2":  # (experiments build only: the statistical bound without the coherent term)
                    os.environ["CSMP_CERT_NOREL"] = "1"
                else:
                    os.environ.pop("CSMP_CERT_NOREL", None)
                idx, val, nnz = d.ctx.omp_batch_mfma(B, k, eps)
                st = d.ctx.batch_stats()
                bad = [s for s in range(nsig) if nnz[s] != n2[s] or not np.array_equal(idx[:, s], i2[:, s])]
                coef = float(np.abs(val - v2).max())
                print(json.dumps({"dict": name, "M": M, "N": N, "signals": family, "k": k, "cert": mode, "nsig": nsig,
                                  "supports_differing_from_exact_path": len(bad), "uncertain": st["uncertain"], "illcond": st["illcond"],
                                  "resolved_exactly": st["resolved_exactly"], "max_coef_diff": coef}), flush=True)
        d.close()


if __name__ == "__main__":
    main()
