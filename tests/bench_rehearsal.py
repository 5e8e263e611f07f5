"""REHEARSAL wrapper around bench.py -- test infrastructure, never used by the product or by a measured run.

Runs bench.main() with the oracle-backed stand-in engine of tests/oracle_engine.py in place of the HIP engine, so that the
multi-rank control flow of the one-job mode (range-addressable generation, partition, parity gates, all-reduces, the
JSON line) can be exercised in a container without GPUs.  bench.py itself has no switch for this: the stand-in is
patched in from here.  The line such a run prints says `rehearsal` and carries no value.

    python tests/bench_rehearsal.py --gpus 2 --scale 0.002 ...      (spawns its ranks like bench.py does)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402
from tests.oracle_engine import Engine as OracleEngine  # noqa: E402

bench.ENGINE_OVERRIDE = OracleEngine

if __name__ == "__main__":
    bench.main()
