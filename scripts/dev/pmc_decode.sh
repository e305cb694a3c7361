#!/bin/bash
# HBM-side counters of the decode kernels on scripts/decode_bench.py's own launches (separate rocprofv3 --pmc passes): scripts/dev/pmc_decode.sh OUTDIR KIND FRAMES
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$1; KIND=${2:-ours}; N=${3:-1024}; mkdir -p $OUT
PMC_T=100 $R/scripts/pmc.sh $OUT dec_$KIND "FETCH_SIZE" "WRITE_SIZE" -- python $R/scripts/decode_bench.py --kind $KIND --frames $N --steps 2 --warmup 1 --no-cpu-baseline
python3 - $OUT $KIND $N <<'PY'
import csv, json, sys
from pathlib import Path
out, kind, n = Path(sys.argv[1]), sys.argv[2], int(sys.argv[3])
tot = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = out / f"dec_{kind}_{name}_counter_collection.csv"
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and r["Kernel_Name"].split("(")[0].split(".")[0] in ("sky_lz4f_scan", "sky_lz4_decode", "sky_lz4_parse", "sky_lz4_link", "sky_lz4_decode_seq", "sky_lz4_resolve", "sky_lz4_chain"):
            k = r["Kernel_Name"].split("(")[0]
            per.setdefault(k, []).append(float(r["Counter_Value"]))
    tot[name] = {k: sum(v) / len(v) * 1024.0 for k, v in per.items()}      # KiB -> bytes, mean per launch
raw = n * 8 * 1024 * 1024
entry = {"fetch_bytes_per_output_byte_raw_counter": round(sum(tot["FETCH_SIZE"].values()) / raw, 4), "write_bytes_per_output_byte": round(sum(tot["WRITE_SIZE"].values()) / raw, 4),
         "per_kernel_fetch": {k: round(v / raw, 4) for k, v in tot["FETCH_SIZE"].items()}, "per_kernel_write": {k: round(v / raw, 4) for k, v in tot["WRITE_SIZE"].items()}}
entry["bytes_per_output_byte"] = round(entry["fetch_bytes_per_output_byte_raw_counter"] + entry["write_bytes_per_output_byte"], 4)
entry["source"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), sky_lz4f_scan + sky_lz4_decode (+ parse / link or resolve / chain), {n} frames of 8 MiB, kind {kind}; raw counters, per decoded "
                   "byte (the decoder's reads are 16-byte per-lane window loads and match-source gathers, not the wide coalesced stream read whose requests gfx950 tallies at half: no correction applied)")
tf = Path(__file__).resolve().parent if False else Path(sys.argv[1])
p = Path(__import__("os").environ.get("GRAFT_REPO_ROOT", "/root/repo")) / "profiles" / "traffic_decode.json"
allv = json.loads(p.read_text()) if p.exists() else {}
allv[f"{kind}:{n}"] = entry
p.write_text(json.dumps(allv, indent=1) + "\n")
(out / "traffic_decode.json").write_text(json.dumps(allv, indent=1) + "\n")
print(json.dumps({f"{kind}:{n}": entry}))
PY
