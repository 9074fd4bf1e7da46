#!/usr/bin/env python3
"""Turn rocprofv3 FETCH_SIZE / WRITE_SIZE summaries (tools/rocprof_summary.py --pmc output) plus the
kernel-trace stats into profiles/pmc_traffic.json: measured HBM bytes per work unit for each kernel.

FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied at 64 B);
WRITE_SIZE is used as reported (it matched the known byte count of the NTT kernels to 6 %).
usage: tools/pmc_traffic.py <pmc_fetch.txt> <pmc_write.txt> <units.json> [<pmc_inst.txt>] > profiles/pmc_traffic.json
units.json maps kernel short name -> average work units per dispatch (from a bench line's profiler output).

With the SQ_INSTS_VALU pass (pmc_inst.txt) and the GRBM_GUI_ACTIVE column of the fetch pass, each kernel also gets its
VALU issue occupancy: wave-instructions x 4 cycles (a wave64 instruction occupies its SIMD's 16 lanes for 4 cycles; the
FP64 fma/mul/add/rndne the kernels are made of issue at that full rate) / (1024 SIMDs x shader cycles of the dispatch).
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (GUI_ACTIVE / 8 / wall time = the 1.9-2.0 GHz the part sustains under
this load, MI355X_MICROARCH.md "DVFS give-back").
"""
import json
import re
import sys


def parse(path, counter):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(?:void )?hipbfv::([a-z_0-9]+)", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^\s+(\S+)\s+([0-9.]+)", line)
        if m and cur and m.group(1) == counter:
            # template variants of one kernel (e.g. mul_mid<..,true/false>) are one profiler record in bench.py: sum them
            key = cur.replace("_kernel", "")
            out[key] = out.get(key, 0.0) + float(m.group(2))
    return out


def main():
    fetch = parse(sys.argv[1], "FETCH_SIZE")
    write = parse(sys.argv[2], "WRITE_SIZE")
    units = json.load(open(sys.argv[3]))
    valu = parse(sys.argv[4], "SQ_INSTS_VALU") if len(sys.argv) > 4 else {}
    gui = parse(sys.argv[1], "GRBM_GUI_ACTIVE")
    res = {}
    for k in fetch:
        if k not in units or k not in write:
            continue
        rd = fetch[k] * 1024 * 2
        wr = write[k] * 1024
        res[k] = {
            "fetch_bytes_per_dispatch": rd,
            "write_bytes_per_dispatch": wr,
            "units_per_dispatch": units[k],
            "hbm_bytes_per_unit": (rd + wr) / units[k],
        }
        if k in valu and gui.get(k):
            cycles = gui[k] / 8.0
            res[k]["valu_wave_insts_per_dispatch"] = valu[k]
            res[k]["shader_cycles_per_dispatch"] = cycles
            res[k]["valu_issue_frac"] = round(valu[k] * 4.0 / (1024.0 * cycles), 4)
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 on gfx950; SQ_INSTS_VALU and GRBM_GUI_ACTIVE for the VALU issue occupancy", "kernels": res}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
