"""The committed benchmark evidence keeps the driver's contract: one JSON object per line with the metric BASELINE.json
names, a `roofline` object for the dominant kernel, a `cpu_baseline` object for the oracle, and -- for the headline
workload -- PMC-derived traffic and VALU issue occupancy that agree with profiles/pmc_traffic.json (tools/pmc_traffic.py)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lines():
    # the lines of the newest profile set that carry the full contract (older sets predate some fields and stay as history)
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json"))) if "valu" in json.load(open(f))]
    assert files, "no committed bench lines"
    return [(os.path.basename(f), json.load(open(f))) for f in files]


def _newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    assert files, pattern
    return files[-1]


def test_bench_lines_follow_the_contract():
    for name, d in _lines():
        for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                         ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
            assert isinstance(d[key], typ), (name, key)
        assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "u64", name
        # one fixed database sharded by row (pir) or one fixed batch (--total-batch) is strong scaling; --batch per GPU is weak
        strong = "total_batch" in d["config"] or (d["metric"].startswith("pir_") and not name.startswith(("r01_", "r02_")))  # rounds 1-2 ran pir as per-GPU replicas
        assert d["scaling"] == ("strong" if strong else "weak"), name
        assert "workload" in d["config"] and "model" not in d["config"], name
        r = d["roofline"]
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s"), name
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3, name
        c = d["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"], name
        assert "bit-exact" in d["parity"] or "==" in d["parity"] or "decrypts" in d["parity"] or "decoded" in d["parity"], name


def test_headline_line_carries_the_measured_traffic_and_valu_occupancy():
    d = json.load(open(_newest("r*_bench_mulrelin_n8192.json")))
    doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    pmc = doc["workloads"]["mulrelin_n8192"]["kernels"]
    assert doc["kernels"] == pmc  # the top-level alias is the headline workload
    k = d["roofline"]["kernel"]
    per_launch = pmc[k]["hbm_bytes_per_unit"] * pmc[k]["units_per_dispatch"]
    assert abs(d["roofline"]["traffic"] - per_launch) / per_launch < 0.01
    # measured traffic within a few per cent of the kernel's own algorithmic reads + writes: no wasted re-reads
    own = d["roofline"]["kernel_hbm"]["bytes_per_launch"] if "kernel_hbm" in d["roofline"] else d["roofline"]["algorithmic_bytes_per_launch"]
    assert 0.95 < d["roofline"]["traffic"] / own < 1.10
    if "kernel_hbm" in d["roofline"]:
        # roofline.achieved is priced on SURVEY 8(d)'s compulsory bytes per op (48*K*N), not on the kernel's own traffic
        c = d["config"]
        assert d["roofline"]["algorithmic_bytes_per_unit"] == 48 * (c["coeff_modulus_primes"] - 1) * c["poly_modulus_degree"]
        assert abs(d["roofline"]["algorithmic_bytes_per_launch"] - d["roofline"]["algorithmic_bytes_per_unit"] * d["roofline"]["units_per_launch"]) <= 1
        assert d["roofline"]["frac"] < d["roofline"]["kernel_hbm"]["frac"]
    assert d["valu"]["kernel"] == k and abs(d["valu"]["frac"] - pmc[k]["valu_issue_frac"]) < 1e-6
    assert 0.0 < d["valu"]["frac"] < 1.0


def test_every_profiled_workload_has_bench_line_and_kernel_stats():
    doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key, w in doc["workloads"].items():
        assert w["kernels"], key
        for k, rec in w["kernels"].items():
            assert rec["hbm_bytes_per_unit"] > 0 and rec["units_per_dispatch"] > 0, (key, k)
            if "valu_issue_frac" in rec:
                assert 0.0 < rec["valu_issue_frac"] <= rec.get("valu_issue_frac_upper", 1.0) <= 1.0, (key, k)


def test_pmc_traffic_tool_reproduces_the_committed_file(tmp_path):
    p = _newest("r*mulrelin_n8192_pmc_fetch.txt")[: -len("pmc_fetch.txt")]
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    units = tmp_path / "units.json"
    units.write_text(json.dumps({k: v["units_per_dispatch"] for k, v in committed["kernels"].items()}))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), "mulrelin_n8192", p + "pmc_fetch.txt", p + "pmc_write.txt",
                          str(units), p + "pmc_inst.txt"], capture_output=True, text=True, check=True).stdout
    assert json.loads(out) == committed["workloads"]["mulrelin_n8192"]


def test_a_rank_that_dies_takes_the_self_spawned_launch_down_at_once():
    """`python bench.py --gpus N` spawns its ranks; a rank without a device (here: every rank, there is no GPU) must end the
    whole launch within seconds, not after the rendezvous timeout of the survivors (600 s, measured the hard way)."""
    import subprocess
    import sys
    import time

    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                         capture_output=True, text=True, timeout=240, env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    assert out.returncode != 0
    assert "no GPU for LOCAL_RANK" in out.stderr
    assert time.time() - t0 < 200


def test_dry_run_validates_an_eight_gpu_launch_without_touching_a_device():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--workload", "pir", "--n", "16384", "--batch", "1024"],
                       env=env, capture_output=True, text=True, timeout=300)
    d = json.loads(p.stdout.splitlines()[-1])
    assert p.returncode == 0 and d["ok"] and len(d["ranks"]) == 8
    assert d["ranks"][7]["database_rows"] == [896, 1024] and d["ranks"][0]["database_bytes"] == 128 * 1024 * 8 * 16384 * 8  # 128 GiB of the 1 TiB database


def test_power_leg_parses_rocm_smi_and_survives_its_absence(tmp_path, monkeypatch):
    """bench.power_leg samples `rocm-smi` while the steps keep running: the text it parses is the tool's (ROCm 7.2) and nothing it
    prints -- or its absence -- may fail the benchmark."""
    import bench

    fake = tmp_path / "rocm-smi"
    fake.write_text("#!/bin/sh\ncat <<'EOF'\n"
                    "GPU[0]\t\t: fclk clock level: 0: (1250Mhz)\nGPU[0]\t\t: mclk clock level: 0: (2000Mhz)\nGPU[0]\t\t: sclk clock level: 1: (2156Mhz)\n"
                    "=================================== Power Consumption ====================================\n"
                    "GPU[0]\t\t: Current Socket Graphics Package Power (W): 1374.0\n"
                    "======================================= Power Cap ========================================\n"
                    "GPU[0]\t\t: Max Graphics Package Power (W): 1400.0\nEOF\n")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    steps = []
    rec = bench.power_leg(lambda: steps.append(1), lambda: None)
    assert rec["package_w"] == 1374.0 and rec["sclk_mhz"] == 2156 and rec["cap_w"] == 1400.0 and rec["samples"] == 3 and steps
    fake.write_text("#!/bin/sh\necho nothing useful\n")
    assert bench.power_leg(lambda: None, lambda: None) is None


def test_default_line_is_compact_and_ends_in_the_summary_of_every_metric_quantity():
    """The driver stores the TAIL of the printed line (2000 characters): the last key is `summary`, it carries the three quantities
    BASELINE.json's metric names (mul+relin at n = 8192 and 16384, NTTs/s), the per-key worst cases and the program workloads, and the
    whole line stays below 10 KB.  Replayed on a committed full line (round 4's, 16 KB) through the functions bench.py's main() uses."""
    sys.path.insert(0, ROOT)
    import bench

    full = json.load(open(_newest("r*_bench_default.json")))
    full.pop("summary", None)
    line = dict(full)
    line["secondary"] = {k: (bench.compact_secondary(v) if "algorithmic_bytes_per_unit" in (v.get("roofline") or {}) else v) for k, v in full["secondary"].items()}
    line["summary"] = bench.summary_of(line)
    bench.trim_headline(line)
    text = json.dumps(line)
    assert len(text) < 10240, len(text)  # (11 secondary workloads since r06: the three per-key jobs; what matters is the 2000-character tail below)
    assert list(line)[-1] == "summary"
    tail = text[-2000:]
    assert tail.index('"summary"') >= 0
    summ = json.loads(tail[tail.index('"summary"') + len('"summary": '):-1])
    for key in ("mulrelin_n8192", "mulrelin_n16384", "ntt_n8192", "3x54", "keys64", "keys4096", "n16384_keys1024", "chi_sq_1024", "chi_sq_128", "dot_prod",
                "pir_2p17"):
        e = summ[key]
        assert set(e) == {"value", "ms_per_step", "frac", "whole_op_frac", "traffic_ratio", "parity_ok"}, key
        assert e["value"] > 0 and e["parity_ok"] is True, key
    # a failed or skipped secondary job is visible in the summary and never passes for a measurement
    assert bench.summary_entry({"error": "OutOfMemoryError: ..."}) == {"error": "OutOfMemoryError: ...", "parity_ok": False}
    assert bench.summary_entry({"skipped": "needs 140 GiB"})["parity_ok"] is False


# summary key of the default line -> workload key of its evidence (kernel trace + three PMC passes); keys64 shares the kernels and the
# command shape of keys4096 (one PMC set covers the per-key path), the 128-set chi_sq run is the 1024-set workload at another batch
SUMMARY_EVIDENCE = {
    "mulrelin_n8192": "mulrelin_n8192", "mulrelin_n16384": "mulrelin_n16384", "ntt_n8192": "ntt_n8192", "3x54": "mulrelin_n8192_bits54-54-54-56",
    "keys64": "mulrelin_n8192_keys4096", "keys4096": "mulrelin_n8192_keys4096", "n16384_keys1024": "mulrelin_n16384_keys1024",
    "chi_sq_1024": "chi_sq_n16384", "chi_sq_128": "chi_sq_n16384", "dot_prod": "dot_prod_n16384", "pir_2p17": "pir_n16384",
}


def test_every_summary_workload_has_its_kernel_trace_and_pmc_passes_for_the_tree_sources():
    """every quantity the newest default line's `summary` claims has, under profiles/, the rocprofv3 kernel trace and
    the three PMC passes of its workload, merged into pmc_traffic.json under the kernel-source hash of THIS tree (bench.py refuses
    stale PMC figures at run time; this refuses a stale evidence set at commit time)."""
    import pytest

    if os.environ.get("HIPBFV_LIB"):
        pytest.skip("a variant library is loaded (its build flags are part of the source hash): the evidence belongs to the default build")
    sys.path.insert(0, ROOT)
    import bench

    newest = _newest("r*_final_bench_default.json")
    tag = os.path.basename(newest)[: -len("_bench_default.json")]
    line = json.load(open(newest))
    doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    tree = bench.kernel_source_hash()
    assert set(line["summary"]) == set(SUMMARY_EVIDENCE), sorted(set(line["summary"]) ^ set(SUMMARY_EVIDENCE))
    for skey, wkey in SUMMARY_EVIDENCE.items():
        assert line["summary"][skey]["parity_ok"] is True, skey
        for part in ("kernel_stats", "pmc_fetch", "pmc_write", "pmc_inst"):
            f = os.path.join(ROOT, "profiles", f"{tag}_{wkey}_{part}.txt")
            assert os.path.exists(f) and os.path.getsize(f) > 200, f
        assert wkey in doc["workloads"] and doc["workloads"][wkey]["kernels"], wkey
        assert doc["source_hash"].get(wkey) == tree, (wkey, doc["source_hash"].get(wkey), tree)
    if "kernel_source_hash" in line:
        assert line["kernel_source_hash"] == tree
