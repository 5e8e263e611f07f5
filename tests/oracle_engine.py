"""TEST INFRASTRUCTURE -- a stand-in for ``plastid_amd.engine.Engine`` that counts with the oracle, so that the
multi-rank CONTROL FLOW of ``bench.py --gpus N`` (self-spawn, per-rank generation, partition, parity gates,
all-reduces, the JSON line) can be rehearsed in a container without GPUs (``tests/bench_rehearsal.py``,
``PC_BENCH_BACKEND=gloo``).  It is never imported by the product or by a measuring run; a line produced with it says
"rehearsal" and carries no value."""
import numpy as np

from oracle import oracle
from plastid_amd.packing import concat_file_major

_KINDS = {0: "fiveprime", 1: "threeprime", 2: "center", 3: "variable", 4: "stratified"}


class Plan(object):
    def __init__(self, eng, tid, start, end, strand, out_off, out_step, row_stride, out_elems, rows):
        self.eng = eng
        self.seg = dict(tid=np.asarray(tid, np.int32), start=np.asarray(start, np.int64), end=np.asarray(end, np.int64),
                        strand=np.asarray(strand, np.uint8), out_off=np.asarray(out_off, np.int64),
                        out_step=np.asarray(out_step, np.int8), row_stride=np.asarray(row_stride, np.int64))
        self.out_elems, self.rows = int(out_elems), int(rows)
        self.positions = int((self.seg["end"] - self.seg["start"]).sum())
        self.tiles = 0
        self.device_ptr = 0
        self._out = None

    def launch(self, dtype):
        sg = self.seg
        arrays, _ = oracle.count_segments(self.eng.aln, self.eng.spec(), sg["tid"], sg["start"], sg["end"], sg["strand"])
        out = np.zeros(self.out_elems, np.float64 if self.eng.kind == "center" else np.int64)
        for s, a in enumerate(arrays):
            a2 = a.reshape(self.rows, -1)
            n = a2.shape[1]
            for r in range(self.rows):
                if sg["out_step"][s] == 0:
                    out[sg["out_off"][s] + r * sg["row_stride"][s]] += a2[r].sum()
                else:
                    out[sg["out_off"][s] + r * sg["row_stride"][s] + int(sg["out_step"][s]) * np.arange(n)] = a2[r]
        self._out = out.astype(dtype)

    def read(self, out=None):
        if out is None:
            return self._out.copy()
        out[:] = self._out
        return out

    def total(self):
        return self._out.sum()

    def close(self):
        pass


class Engine(object):
    def __init__(self, device=0):
        self.aln, self.kind, self.args, self.size_filter, self.rows = None, "fiveprime", {}, None, 1

    def set_alignments(self, files, ntid=None):
        self.aln = concat_file_major(files)

    def set_mapping(self, kind, param=0, fw=None, rc=None, min_len=25, max_len=35, **kw):
        self.kind = _KINDS[kind]
        self.args = dict(param=param, fw=fw, rc=rc, min_len=min_len, max_len=max_len)
        self.rows = max_len - min_len + 1 if self.kind == "stratified" else 1

    def spec(self):
        sp = oracle.mapping_spec(self.kind, self.args.get("param", 0), None, self.args.get("min_len", 25), self.args.get("max_len", 35),
                                 size_filter=self.size_filter) if self.kind in ("fiveprime", "threeprime", "center") else None
        if sp is None:   # offset tables as the factory handed them over
            sp = {"kind": oracle.KIND_NAMES[self.kind], "param": 0, "fw": np.asarray(self.args["fw"], np.int32),
                  "rc": np.asarray(self.args["rc"], np.int32), "min_len": int(self.args["min_len"]), "max_len": int(self.args["max_len"]),
                  "size_filter": self.size_filter}
        return sp

    def set_size_filter(self, min_len=None, max_len=-1):
        self.size_filter = None if min_len is None else (min_len, max_len)

    def plan(self, *a):
        return Plan(self, *a)

    def sync(self):
        pass

    def set_profiling(self, level):
        pass

    def last_timing(self):
        return {"total": 0.0, "worklist": 0.0, "hist": 0.0, "long": 0.0, "gather": 0.0, "zero": 0.0}

    def last_algorithmic_bytes(self):
        return 0

    def close(self):
        pass
