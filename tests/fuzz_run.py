"""Long-running randomised differential run (GPU box; test infrastructure, not collected by pytest):
    python tests/fuzz_run.py [seconds] [first_seed] [size]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import plastid_amd as pa
from oracle import oracle
import fuzz_cases, test_gpu_parity as T

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
size = sys.argv[3] if len(sys.argv) > 3 else "small"
t0 = time.time(); ok = bad = 0
while time.time() - t0 < budget:
    case = fuzz_cases.random_case(seed, pa, size)
    try:
        fuzz_cases.run_case(pa, oracle, case, T.spec_for, T.engine_for)
        ok += 1
    except AssertionError as e:
        bad += 1
        print("MISMATCH", str(e)[:300], flush=True)
    except Exception:
        bad += 1
        print("ERROR seed", seed, case["mapping"], case["size_filter"], case["knobs"], case["layout"], flush=True)
        traceback.print_exc(limit=3)
    seed += 1
print("cases ok=%d bad=%d last_seed=%d" % (ok, bad, seed - 1))
