"""Run every per-kernel parity check in one process and print a table (used on the GPU box)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def main():
    import kernel_checks as kc
    only = sys.argv[1:]
    nfail = 0
    for chk in kc.ALL_CHECKS:
        if only and not any(o in chk.__name__ for o in only):
            continue
        t0 = time.time()
        try:
            rows = chk()
            torch.cuda.synchronize()
        except Exception:
            print("EXC  %s\n%s" % (chk.__name__, traceback.format_exc()))
            nfail += 1
            try:
                torch.cuda.synchronize()
            except Exception as e:
                print("device error after %s: %s -- aborting" % (chk.__name__, e))
                break
            continue
        for name, err, tol in rows:
            ok = err <= tol
            nfail += (not ok)
            print("%s %-52s err=%.3e tol=%.1e" % ("ok  " if ok else "FAIL", name, err, tol))
        print("---- %s: %.1fs" % (chk.__name__, time.time() - t0))
    print("TOTAL FAILURES: %d" % nfail)
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
