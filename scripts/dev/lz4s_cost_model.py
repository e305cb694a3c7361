#!/usr/bin/env python3
"""LDS cost model of sky_lz4s_compress's probe and parse (development tool, CPU only).

The sequential model of the parse (tests/model/lz4s_model.c) traces every visit -- slice, visit number, which regions hold a valid table candidate,
how many 16-byte extension steps the match takes -- and this script replays the trace the way the kernel executes it: 16 waves of 64 slices, a wave's
iteration t runs the lanes that have a t-th visit, every LDS access priced as measured on an MI355X (scripts/dev/lds_rate.hip,
profiles/r2_lds_rate.txt; CU cycles per wave-level instruction with 16 waves at random addresses):

    aligned ds_read_b32 / b64 / b128 (table rows)   11      (rows are 16-byte aligned: a quarter / half / all of the banks)
    misaligned ds_read_b128                          1 per ACTIVE lane  (+ 1 issue cycle)

It prints the modelled LDS cycles per 64 KiB block for probe-all and for the parse, per region and in total, next to the measured phase table when
one is given (the window "probe-all + parse + wait for the slowest parse" is what they should be compared with: the phase is throughput-bound, the
waves share one LDS).  Usage: python scripts/dev/lz4s_cost_model.py [--mib 16] [--phases profiles/r2_lz4s_phases.txt]"""
import argparse
import ctypes as C
import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from skyplane_amd import synth  # noqa: E402
from tests.model import lz4smodel  # noqa: E402

ROW = 11.0          # one table-row read (any width), 16 waves, random rows
PER_LANE = 1.0      # misaligned ds_read_b128: cycles per active lane
ISSUE = 1.0         # ... plus its issue slot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=16)
    ap.add_argument("--phases", default=str(ROOT / "profiles" / "r2_lz4s_phases.txt"))
    a = ap.parse_args()
    lib = lz4smodel._lib()
    cap = 1 << 16
    trace = (C.c_uint32 * cap)()
    C.c_void_p.in_dll(lib, "lz4s_dbg_trace").value = C.addressof(trace)
    C.c_uint32.in_dll(lib, "lz4s_dbg_trace_cap").value = cap
    ntrace = C.c_uint32.in_dll(lib, "lz4s_dbg_ntrace")
    data = synth.silesia_like(a.mib << 20, config_id=2)
    nblk = 0
    probe = np.zeros(4)
    parse = np.zeros(4)
    parts = np.zeros(4)                     # W, row, candidates, extension
    visits = valid = 0
    for b in range(0, data.size, 65536):
        blk = np.ascontiguousarray(data[b:b + 65536])
        ntrace.value = 0
        lib.lz4s_model_block(blk.ctypes.data, blk.size, None, None)
        t = np.frombuffer(trace, np.uint32, ntrace.value)
        sl, vis, vm, ext = t & 1023, (t >> 10) & 15, (t >> 14) & 15, t >> 18
        nblk += 1
        visits += t.size
        valid += int(sum(((vm >> k) & 1).sum() for k in range(4)))
        for w in range(16):
            q = w >> 2
            probe[q] += 64 * ROW                                               # every lane probes its 64 positions: 64 row reads per wave
            m = (sl >> 6) == w
            for it in range(int(vis[m].max()) + 1 if m.any() else 0):
                mm = m & (vis == it)
                act = int(mm.sum())
                if not act:
                    continue
                c_w = PER_LANE * act + ISSUE
                c_row = ROW
                c_cand = sum(PER_LANE * int(((vm[mm] >> k) & 1).sum()) + ISSUE for k in range(q + 1))
                e = ext[mm]
                c_ext = sum(2 * (PER_LANE * int((e > s).sum()) + ISSUE) for s in range(int(e.max()))) if act else 0
                parse[q] += c_w + c_row + c_cand + c_ext
                parts += (c_w, c_row, c_cand, c_ext)
    probe /= nblk; parse /= nblk; parts /= nblk
    print(f"{nblk} blocks of the Silesia-like stream: {visits / nblk / 1024:.2f} visits per slice, {valid / max(visits, 1):.2f} valid table candidates per visit")
    print("modelled LDS cycles per block          region 0   region 1   region 2   region 3      total")
    print("  probe-all (64 row reads per lane)  " + " ".join(f"{x:10.0f}" for x in probe) + f" {probe.sum():10.0f}")
    print("  parse                              " + " ".join(f"{x:10.0f}" for x in parse) + f" {parse.sum():10.0f}")
    print(f"    of which: 16 bytes around the position {parts[0]:.0f}, table rows {parts[1]:.0f}, candidates {parts[2]:.0f}, extension {parts[3]:.0f}")
    print(f"  probe-all + parse                                                               {probe.sum() + parse.sum():10.0f}")
    p = Path(a.phases)
    if p.exists():
        vals = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r"^\s+(.+?)\s{2,}(\d+)\s+[\d.]+%", p.read_text(), re.M)}
        keys = [k for k in vals if k.startswith(("probe-all", "parse", "scan 1"))]
        if keys:
            print(f"measured ({p.name}): " + ", ".join(f"{k} {vals[k]:.0f}" for k in keys) + f"  ->  window {sum(vals[k] for k in keys):.0f} wave-cycles")


if __name__ == "__main__":
    main()
