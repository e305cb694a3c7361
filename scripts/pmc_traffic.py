#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only -- see scripts/gpu_r2g.sh) over
scripts/dev/lz4s_exp.py to bytes per launch of the dominant kernel and per input byte, and write profiles/traffic.json entries.
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced reads
(MI355X_MICROARCH.md, HBM section): the read side is doubled here and the raw figure kept beside it."""
import csv, json, sys
from pathlib import Path
out_dir, stream, chunks, kernel = Path(sys.argv[1]), sys.argv[2], int(sys.argv[3]), sys.argv[4]
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = out_dir / f"{stream}_{name}_counter_collection.csv"
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(kernel) and r["Counter_Name"] == name]
    res[name] = (sum(vals) / len(vals) * 1024.0, len(vals))        # bytes per launch (mean over the launches seen)
inp = chunks * 8 * 1024 * 1024
entry = {"fetch_bytes_per_input_byte": round(2.0 * res["FETCH_SIZE"][0] / inp, 4), "fetch_bytes_per_input_byte_raw_counter": round(res["FETCH_SIZE"][0] / inp, 4),
         "write_bytes_per_input_byte": round(res["WRITE_SIZE"][0] / inp, 4), "launches_measured": res["FETCH_SIZE"][1], "chunks_per_launch": chunks,
         "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), {kernel}, {chunks} x 8 MiB chunks per launch, stream {stream}; "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-byte requests as 64)"}
tf = Path(__file__).resolve().parents[1] / "profiles" / "traffic.json"
allv = json.loads(tf.read_text()) if tf.exists() else {}
if "source" in allv:      # round-1 format (one global entry): superseded
    allv = {}
allv[f"{kernel}:{stream}"] = entry
tf.write_text(json.dumps(allv, indent=1) + "\n")
print(json.dumps({f"{kernel}:{stream}": entry}))
