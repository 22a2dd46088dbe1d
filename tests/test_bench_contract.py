"""bench.py: the pieces that can be checked without a GPU (the PMC-summary staleness guard) and, with -m gpu, the JSON line
the driver parses (one small run as a subprocess)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_pmc_summary_is_used_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    b = _bench()
    digest = b.kernel_source_digest()
    assert len(digest) == 16 and digest == b.kernel_source_digest()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    (tmp_path / "echoglad_amd").mkdir()
    os.symlink(os.path.join(ROOT, "echoglad_amd", "csrc"), tmp_path / "echoglad_amd" / "csrc")
    assert b.pmc_traffic("k_gcn_layer")[0] is None                     # no summary at all
    (prof / "r09_pmc.json").write_text(json.dumps({"kernel_source_digest": "0" * 16, "k_gcn_layer": {"hbm_bytes_per_launch": 1.0}}))
    traffic, why = b.pmc_traffic("k_gcn_layer")
    assert traffic is None and "stale" in why
    (prof / "r09_pmc.json").write_text(json.dumps({"kernel_source_digest": digest, "k_gcn_layer": {"hbm_bytes_per_launch": 7.5e8}}))
    assert b.pmc_traffic("k_gcn_layer") == (750000000, "profiles/r09_pmc.json")
    # other *_pmc.json files next to the real one (the training step's summary sorts AFTER r09_pmc.json; round 2's HEAD
    # picked it and reported traffic = null) are never candidates for the inference kernel's traffic
    (prof / "r09_train_pmc.json").write_text(json.dumps({"k_bn_act_fwd": {"hbm_bytes_per_launch": 3.5e9}}))
    (prof / "zz_pmc.json").write_text("{}")
    assert b.pmc_traffic("k_gcn_layer") == (750000000, "profiles/r09_pmc.json")
    # ... and the training summary has its own digest (over the training kernels' sources) and its own reader
    assert b.train_pmc_traffic()[0] is None and "stale" in b.train_pmc_traffic()[1]
    (prof / "r09_train_pmc.json").write_text(json.dumps({"kernel_source_digest": b.kernel_source_digest(b.TRAIN_KERNEL_SOURCES),
                                                         "hbm_bytes_per_step": 5.0e10}))
    assert b.train_pmc_traffic() == (50000000000, "profiles/r09_train_pmc.json")
    assert b.pmc_traffic("k_gcn_layer") == (750000000, "profiles/r09_pmc.json")


def test_committed_pmc_summary_matches_the_committed_kernel_sources():
    """profiles/r*_pmc.json of this round must have been measured on the dominant kernel's current sources."""
    b = _bench()
    traffic, src = b.pmc_traffic("k_gcn_layer")
    assert traffic is not None, src
    assert 5.9e8 <= traffic <= 1.2e9            # >= the algorithmic 590 MB per launch, and not absurdly above it


@pytest.mark.gpu
def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                        "--no-other-configs", "--repeats", "2", "--spinup", "0.05"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "repeats", "distributed"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"]
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0 < rf["frac"] < 1
    assert d["repeats"]["n"] == 2 and d["repeats"]["min"] <= d["repeats"]["max"]
    assert d["eager_ms_per_step"] > 0 and "replay" in d["config"]["launch"]
    assert len(lines[0]) <= 1600 and os.path.exists(os.path.join(ROOT, d["details"]))
    full = json.load(open(os.path.join(ROOT, d["details"])))
    assert full["value"] == d["value"] and "avg_launch_timing" in full["roofline"]


def test_headline_of_a_full_result_fits_the_drivers_tail_window():
    """The driver keeps the last 2000 characters of the run's output: the stdout line must fit there as a WHOLE JSON object
    with the numbers a reader needs (round 5's line was 7 KB and arrived truncated)."""
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))
    full["eager_ms_per_step"] = 0.8312
    full["cpu_baseline"]["sample_batch"] = 2
    h = b.headline_of(full, "gpurun_out/bench_details.json")
    line = json.dumps(h)
    assert len(line) <= 1536, len(line)          # 1.5 KB
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "eager_ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "cfg4_train"):
        assert key in h, key
    assert h["roofline"]["frac"] == full["roofline"]["frac"] and h["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert h["cfg4_train"]["traffic_bytes_per_step"] == full["other_configs"]["cfg4_train"]["traffic_bytes_per_step"]
    assert h["cpu_baseline"]["sample_batch"] == 2 and h["other_ms_per_step"]["cfg2_b1"] > 0


def test_only_the_result_line_reaches_stdout():
    """bench.py prints ONE JSON line.  Native libraries write to stdout as well (RCCL's version banner, block-buffered: behind a
    pipe it lands after the result): guard_stdout() points descriptor 1 at stderr, emit() writes the line to the real stdout."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent(f'''
        import ctypes, importlib.util, sys
        spec = importlib.util.spec_from_file_location("bench_module", {os.path.join(ROOT, "bench.py")!r})
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        b.guard_stdout()
        ctypes.CDLL(None).printf(b"native banner line\\n")
        print("a python print")
        b.emit({{"ok": 1}})
    ''')
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    assert r.stdout == '{"ok": 1}\n'
    assert "native banner line" in r.stderr and "a python print" in r.stderr
