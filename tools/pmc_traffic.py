#!/usr/bin/env python3
"""Turn rocprofv3 PMC summaries (tools/rocprof_summary.py --pmc output) into an entry of profiles/pmc_traffic.json:
measured HBM bytes per work unit and VALU issue occupancy of each kernel of ONE workload.

usage: tools/pmc_traffic.py <workload-key> <pmc_fetch.txt> <pmc_write.txt> <units.json> [<pmc_inst.txt>] [--into profiles/pmc_traffic.json]

  workload-key   bench.py's pmc_workload_key(): e.g. mulrelin_n8192, mulrelin_n16384, ntt_n8192, mulrelin_n8192_bits54-54-54-56
  units.json     kernel short name -> average work units per dispatch (from the bench line's profiler output)

FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied at 64 B); WRITE_SIZE is used
as reported (it matched the known byte count of the NTT kernels to 6 %).

VALU issue occupancy = issue cycles / (1024 SIMDs x shader cycles of the dispatch).  Shader cycles = GRBM_GUI_ACTIVE / 8
(reported summed over the 8 XCDs; GUI_ACTIVE / 8 / wall time = the 1.9-2.0 GHz the part sustains under this load).  Issue
cycles per wave64 instruction by class (MI355X_MICROARCH.md "Wave scheduling": a SIMD-32 issues a 32-bit VALU instruction
over 2 cycles; FP64 runs at half that rate): 4 for FP64 and 64-bit integer multiplies, 2 for everything else.  FP64
instructions = SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, 64-bit integer = SQ_INSTS_VALU_INT64, when the instruction-class pass
collected them; v_rndne_f64 / v_cvt are in none of those classes, so
`valu_issue_frac` (the unclassified remainder priced at 2 cycles) is a lower bound and `valu_issue_frac_upper` prices the
remainder at 4.  Without class counters (older passes) every instruction is priced at 4 and the entry says so.
--source-hash H records the hash of the kernel sources the passes ran on (bench.kernel_source_hash(); bench.py prints the
PMC-derived fields as null when the sources have changed since).
--into merges the entry into an existing multi-workload file (and refreshes the top-level "kernels" alias of the headline
workload mulrelin_n8192) and writes it back; otherwise the single entry is printed.
"""
import json
import re
import sys

SOURCE = ("rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / WRITE_SIZE / SQ_INSTS_VALU* (separate passes, tools/gpu_pmc_report.sh); "
          "FETCH_SIZE x2 on gfx950")
F64 = ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")
# 64-bit integer multiplies (v_mad_u64_u32, the Harvey / Barrett arithmetic of the integer policy) issue at the FP64 rate too
# (measured 50 lane-ops/clk/CU against 59-61 for v_fma_f64: profiles/r01_microbench_instruction_rates.txt)
SLOW = ("SQ_INSTS_VALU_INT64",)
OTHER = ("SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_CVT")


# rocprof kernel name -> the profiler record bench.py reports it under (the fused multiply+relinearize kernels run in the
# key-switch head / tail slots of the pipeline)
ALIAS = {"mulrelin_head": "ks_head", "mulrelin_tail": "ks_tail", "ks_mid_int": "ks_mid", "dot_plain_tab": "plain", "dot_plain2": "plain", "dot_plain": "plain"}


def parse(path, counter):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(?:void )?hipbfv::([a-z_0-9]+)", line)
        if m:
            cur = m.group(1)
            continue
        if not line.startswith(" "):
            cur = None
            continue
        m = re.match(r"^\s+(\S+)\s+([0-9.]+)", line)
        if m and cur and m.group(1) == counter:
            # template variants of one kernel (e.g. mul_mid<..,true/false>) are one profiler record in bench.py: sum them
            key = ALIAS.get(cur.replace("_kernel", ""), cur.replace("_kernel", ""))
            out[key] = out.get(key, 0.0) + float(m.group(2))
    return out


def entry(fetch_p, write_p, units, inst_p):
    fetch = parse(fetch_p, "FETCH_SIZE")
    write = parse(write_p, "WRITE_SIZE")
    gui = parse(fetch_p, "GRBM_GUI_ACTIVE")
    valu = parse(inst_p, "SQ_INSTS_VALU") if inst_p else {}
    classes = {c: parse(inst_p, c) for c in F64 + SLOW + OTHER} if inst_p else {}
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
            have_classes = any(k in classes[c] for c in F64)
            if have_classes:
                f64 = sum(classes[c].get(k, 0.0) for c in F64 + SLOW)
                other = sum(classes[c].get(k, 0.0) for c in OTHER)
                rest = max(0.0, valu[k] - f64 - other)  # v_rndne_f64, moves, bit ops, compares ...
                res[k]["valu_f64_wave_insts_per_dispatch"] = f64
                res[k]["valu_int_cvt_wave_insts_per_dispatch"] = other
                res[k]["valu_unclassified_wave_insts_per_dispatch"] = rest
                res[k]["valu_issue_frac"] = round((4.0 * f64 + 2.0 * (other + rest)) / (1024.0 * cycles), 4)
                res[k]["valu_issue_frac_upper"] = round((4.0 * (f64 + rest) + 2.0 * other) / (1024.0 * cycles), 4)
                res[k]["valu_pricing"] = "4 cycles per FP64 / INT64 wave64 instruction, 2 per other; unclassified (v_rndne_f64, moves, logic) at 2 (frac) or 4 (upper)"
            else:
                res[k]["valu_issue_frac"] = round(valu[k] * 4.0 / (1024.0 * cycles), 4)
                res[k]["valu_pricing"] = "4 cycles per wave64 instruction (no class counters in this pass: upper bound)"
    return {"kernels": res}


def main():
    argv = sys.argv[1:]
    into = None
    if "--into" in argv:
        i = argv.index("--into")
        into = argv[i + 1]
        del argv[i : i + 2]
    source_hash = None
    if "--source-hash" in argv:  # bench.kernel_source_hash() of the tree the passes ran on (tools/gpu_pmc_report.sh records it on the GPU box)
        i = argv.index("--source-hash")
        source_hash = argv[i + 1]
        del argv[i : i + 2]
    key, fetch_p, write_p, units_p = argv[:4]
    inst_p = argv[4] if len(argv) > 4 else None
    e = entry(fetch_p, write_p, json.load(open(units_p)), inst_p)
    if not into:
        json.dump(e, sys.stdout, indent=1)
        return
    try:
        doc = json.load(open(into))
    except Exception:
        doc = {}
    doc.setdefault("workloads", {})
    doc["source"] = SOURCE
    doc["workloads"][key] = e
    # bench.py reports these figures only while the kernels are the ones measured here
    doc.setdefault("source_hash", {})[key] = source_hash
    if "mulrelin_n8192" in doc["workloads"]:
        doc["kernels"] = doc["workloads"]["mulrelin_n8192"]["kernels"]  # alias: the headline workload
    json.dump(doc, open(into, "w"), indent=1)


if __name__ == "__main__":
    main()
