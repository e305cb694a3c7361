#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, kernel-trace only -- scripts/pmc.sh) over
scripts/dev/lz4s_exp.py to bytes per launch of the dominant kernel and per input byte, and write profiles/traffic.json entries.
FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 bytes per
lane); other access widths and WRITE_SIZE are uncalibrated (MI355X_MICROARCH.md, HBM section).  This kernel's one wide streaming read is the input
itself -- N bytes, every byte once, global_load_dwordx4 -- so N/2 of it is missing from the counter: fetch = raw + N/2.  The rest of the raw figure
(4-byte prefetch touches, reloads of spilled registers) is taken at face value.  The raw counter and the figure with EVERY fetch doubled are kept
beside it."""
import csv, json, sys
from pathlib import Path
out_dir, stream, chunks, kernel = Path(sys.argv[1]), sys.argv[2], int(sys.argv[3]), sys.argv[4]
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = out_dir / f"{stream}_{name}_counter_collection.csv"
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(kernel) and r["Counter_Name"] == name]
    res[name] = (sum(vals) / len(vals) * 1024.0, len(vals))        # bytes per launch (mean over the launches seen)
inp = chunks * 8 * 1024 * 1024
raw_f, wr = res["FETCH_SIZE"][0] / inp, res["WRITE_SIZE"][0] / inp
entry = {"fetch_bytes_per_input_byte": round(raw_f + 0.5, 4), "fetch_bytes_per_input_byte_raw_counter": round(raw_f, 4),
         "fetch_bytes_per_input_byte_all_doubled": round(2.0 * raw_f, 4), "write_bytes_per_input_byte": round(wr, 4),
         "launches_measured": res["FETCH_SIZE"][1], "chunks_per_launch": chunks,
         "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), {kernel}, {chunks} x 8 MiB chunks per launch, stream {stream}; "
                   "fetch = raw counter + half of the input (gfx950 tallies the 128-byte requests of the coalesced 16-byte-per-lane stream read at 64 bytes, "
                   "MI355X_MICROARCH.md; other widths -- prefetch touches, spill reloads -- at face value)"}
tf = Path(__file__).resolve().parents[1] / "profiles" / "traffic.json"
allv = json.loads(tf.read_text()) if tf.exists() else {}
if "source" in allv:      # round-1 format (one global entry): superseded
    allv = {}
allv[f"{kernel}:{stream}"] = entry
tf.write_text(json.dumps(allv, indent=1) + "\n")
print(json.dumps({f"{kernel}:{stream}": entry}))
