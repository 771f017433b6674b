"""The bench line's schema: the committed line of the round (profiles/bench_rNN_metric.json, written by
`python bench.py` on the GPU box through tools/final_evidence.sh) must say what work it timed and carry counter-based
bandwidth next to the byte model.  CPU test: it reads the committed file, it does not run the bench."""
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest_line():
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "bench_r[0-9][0-9]_metric.json"))
                   if int(re.search(r"bench_r(\d\d)_metric", f).group(1)) >= 6)
    if not files:
        pytest.skip("no round-6+ bench line committed yet")
    text = open(files[-1]).read().strip().splitlines()
    return json.loads([ln for ln in text if ln.startswith("{")][-1]), files[-1]


def test_contract_keys_of_the_metric_line():
    d, path = newest_line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, (key, path)
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "metric" in d["config"]["workload"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_the_line_says_what_work_it_timed():
    d, _ = newest_line()
    assert d["config"]["depth_output"] is False and d["config"]["tile_cull"] is True
    for key, depth, cull in (("value_with_depth", True, True), ("value_reference_lists", False, False),
                             ("value_reference_lists_with_depth", True, False)):
        v = d[key]
        assert v["depth_output"] is depth and v["tile_cull"] is cull and 0 < v["value"] <= d["value"] * 1.02, key
    # the worst case (every piece of render work the reference does) still clears BASELINE.json's 300 / s target
    assert d["value_reference_lists_with_depth"]["value"] > 300
    assert d["forward_only_renders_per_s"] > d["value"]
    assert re.fullmatch(r"[0-9a-f]{64}", d["build_id"])


def test_three_regions_and_counter_based_stage_bandwidth():
    d, _ = newest_line()
    reg = d["regions"]
    assert len(reg["ms_per_step"]) == 3
    assert reg["min_ms_per_step"] <= reg["median_ms_per_step"] <= reg["max_ms_per_step"]
    assert abs(reg["median_ms_per_step"] - d["ms_per_step"]) < 1e-6
    st = d["stages"]
    for name in ("preprocess", "sort", "composite_fwd", "composite_bwd", "contrib_reduce", "geometry_bwd", "tile_cull"):
        s = st[name]
        assert "GBps" not in s and s["model_GBps"] >= 0            # the byte model is labelled as a model
        assert s["traffic_bytes"] > 0 and s["traffic_GBps"] > 0, name       # counter-based, from profiles/traffic_rNN.json
        assert s["traffic_GBps"] < 8000.0
    ph = d["pipeline_hbm"]
    assert ph["traffic_bytes_per_step"] > 0 and 0 < ph["traffic_GBps"] < 8000.0 and "traffic_r" in ph["traffic_source"]
