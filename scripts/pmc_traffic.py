#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_sum passes (separate runs, kernel-trace only -- scripts/pmc.sh) over
scripts/dev/lz4s_exp.py to HBM bytes per launch of the dominant kernel and per input byte, and write profiles/traffic.json entries.

FETCH_SIZE / WRITE_SIZE are reported in KiB; FETCH_SIZE = TCC_EA0_RDREQ x 64 B.  On gfx950 a wide coalesced streaming read (16 bytes per lane) asks the
memory side for whole 128-byte lines in ONE request, which the counter tallies at 64 bytes (MI355X_MICROARCH.md, HBM section: "double it"); other access
widths are uncalibrated there.  This kernel has exactly two kinds of reads of the stream, both of the same N input bytes:
  * the block load, global_load_dwordx4, fully coalesced -- a 128-byte request per line it misses;
  * the prefetch touches of the NEXT block, one global_load_dword per 128-byte line, issued a block ahead so that the lines wait in the XCD's L2.  A
    touch fetches the 64-byte half it hits; the block load later fetches the other half with a second 64-byte request.
So a touched line costs two requests (2 x 64 B, both tallied in full), an untouched one a single request (128 B, tallied at 64), every line moves 128
bytes either way, and with L = N / 128 lines and R requests the touched fraction is t = R / L - 1.  The calibration the guide asks for is the pair of
counters itself: FETCH_SIZE / N must equal (1 + t) / 2, and the build without touches must show R = L and FETCH_SIZE = N / 2 (profiles/r4_pmc_traffic.txt
holds both checks).  fetch = FETCH_SIZE + (1 - t) x N / 2: the halved tally corrected for the lines that came in through the wide load only.  The
guide's blanket rule (FETCH_SIZE + N / 2) is kept beside it; it double-counts the touched lines."""
import csv, json, sys
from pathlib import Path
out_dir, stream, chunks, kernel = Path(sys.argv[1]), sys.argv[2], int(sys.argv[3]), sys.argv[4]
res = {}
for name in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum"):
    f = out_dir / f"{stream}_{name}_counter_collection.csv"
    if not f.exists():
        continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(kernel) and r["Counter_Name"] == name]
    res[name] = (sum(vals) / len(vals), len(vals))        # mean per launch over the launches seen
inp = chunks * 8 * 1024 * 1024
raw_f, wr = res["FETCH_SIZE"][0] * 1024.0 / inp, res["WRITE_SIZE"][0] * 1024.0 / inp
entry = {"fetch_bytes_per_input_byte_raw_counter": round(raw_f, 4), "write_bytes_per_input_byte": round(wr, 4),
         "fetch_bytes_per_input_byte_blanket_rule": round(raw_f + 0.5, 4), "launches_measured": res["FETCH_SIZE"][1], "chunks_per_launch": chunks}
if "TCC_EA0_RDREQ_sum" in res:
    per_line = res["TCC_EA0_RDREQ_sum"][0] / (inp / 128.0)
    t = min(max(per_line - 1.0, 0.0), 1.0)
    entry["read_requests_per_128B_line"] = round(per_line, 4)
    entry["lines_first_fetched_by_a_prefetch_touch"] = round(t, 4)
    entry["raw_counter_predicted_from_requests"] = round((1.0 + t) / 2.0 + max(per_line - 2.0, 0.0) / 2.0, 4)
    entry["fetch_bytes_per_input_byte"] = round(raw_f + 0.5 * (1.0 - t), 4)
    how = ("fetch = FETCH_SIZE + half of the lines that came in through the wide block load only (gfx950 tallies a 128-byte request of a coalesced "
           "16-byte-per-lane read at 64 bytes, MI355X_MICROARCH.md; a line a prefetch touch fetched first costs two 64-byte requests tallied in full: "
           f"TCC_EA0_RDREQ = {per_line:.3f} requests per 128-byte line, scripts/pmc_traffic.py)")
else:
    entry["fetch_bytes_per_input_byte"] = round(raw_f + 0.5, 4)
    how = "fetch = FETCH_SIZE + half of the input (blanket rule of MI355X_MICROARCH.md: no request count was taken)"
entry["source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_sum (separate passes), {kernel}, {chunks} x 8 MiB chunks per launch, stream {stream}; " + how
tf = Path(__file__).resolve().parents[1] / "profiles" / "traffic.json"
allv = json.loads(tf.read_text()) if tf.exists() else {}
if "source" in allv:      # round-1 format (one global entry): superseded
    allv = {}
allv[f"{kernel}:{stream}"] = entry
tf.write_text(json.dumps(allv, indent=1) + "\n")
print(json.dumps({f"{kernel}:{stream}": entry}))
