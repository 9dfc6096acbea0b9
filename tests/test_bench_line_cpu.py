"""bench.py's output contract, checked without a GPU: the LAST stdout line is one JSON object of at most 4096 bytes that
carries the contract keys, `roofline` and `cpu_baseline`; everything else goes to the detail file.  (Round 5's line grew
to 19.8 KB and the driver, which keeps the tail of stdout, could no longer parse it.)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _fat_result(n_gpus=1):
    long = "x" * 700
    res = {
        "metric": bench.METRIC, "value": 14.383251354965758, "unit": "layers/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5,
        "ms_per_step": 973.3543309848756, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] at reduced depth: " + bench.llama_workload_text(
                       2, {"trade_off_factor": 40.0}, "bf16") + "; comm " + long[:300],
                   "model_dtype": "bfloat16", "layers_per_step": 14, "blocks": 2, "parallelism": "single"},
        "spread": 0.03015176698551412,
        "phases_ms": {"A_accumulate": 22.0, "B_eigh": 394.6, "C_factors": 5.7, "D_metrics": 534.3, "other_host_and_gaps": 7.2},
        "roofline": {"bound": "hbm", "achieved": 6612.2483644264075, "peak": 8000.0, "unit": "GB/s", "frac": 0.8265310455533009,
                     "traffic": 35651584.0, "kernel": "sytrd_symv2_kernel " + long[:200], "n": 4096, "k": 2048, "launches": 256,
                     "avg_launch_us": 18.1, "algorithmic_bytes_per_launch": 1.2e8, "solver_frac": 0.6, "eigh_ms": 45.0,
                     "of": "the (4096, 2048) direct eigendecompositions inside one headline step", "traffic_stale": False,
                     "traffic_source": long, "solver_note": long, "note": long},
        "cpu_baseline": {"value": 0.045, "unit": "layers/s", "cores": 16, "kind": "port", "sample": long[:200],
                         "workload": long, "s_per_layer": {"q_o": 5.5, "k_v": 0.8, "gate_up": 65.0, "down": 10.7},
                         "physical_cores": 128, "usable_cpus": 16},
        "c2_layers_per_s": 20.166334000489435, "c2_ms_per_step": 49.58759484870825, "c3_s": 17.2, "c3_layers_per_s": 2.85,
        "c1_cpu_s": 3.1, "fwd_gflops": {"r256": 780000, "r512": 1010000, "r1024": 1250000},
        "fwd_vs_lib_pair": {"r256": 1.16, "r512": 1.04, "r1024": 0.92},
        "detail": {"kernels": {f"k{i}": {"ms": 0.123456789, "note": long} for i in range(40)},
                   "c4_shapes": {"f32": {f"s{i}": long for i in range(20)}}},
    }
    if n_gpus > 1:
        res.update({"comm_ms": 12.3, "b_eigh_ms_max": 101.0, "d_metrics_ms_max": 88.0, "rccl_ranks": n_gpus,
                    "cov_collective": "reduce"})
    return res


def _run(capsys, tmp_path, n_gpus):
    detail = tmp_path / "bench_detail.json"
    rc = bench.main(["--gpus", str(n_gpus), "--steps", "20", "--warmup", "5", "--detail", str(detail)],
                    measure_fn=lambda args: _fat_result(args.gpus))
    assert rc == 0
    out = capsys.readouterr().out.rstrip("\n").split("\n")
    return out, detail


def test_last_line_is_compact_json_with_the_contract_keys(capsys, tmp_path):
    for n_gpus in (1, 8):
        out, detail = _run(capsys, tmp_path, n_gpus)
        last = out[-1]
        assert len(last.encode()) <= 4096, len(last.encode())
        line = json.loads(last)
        for key in bench.CONTRACT_KEYS + ("roofline", "cpu_baseline"):
            assert key in line, key
        assert line["config"]["workload"].startswith("BASELINE configs[3]")
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "of"):
            assert key in line["roofline"], key
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in line["cpu_baseline"], key
        assert "detail" in line and "kernels" not in line
        for key in ("c2_layers_per_s", "c2_ms_per_step", "c3_s", "c1_cpu_s", "phases_ms"):
            assert key in line, key
        if n_gpus > 1:
            for key in ("comm_ms", "b_eigh_ms_max", "d_metrics_ms_max", "rccl_ranks"):
                assert key in line, key
        # the detail file holds everything, the contract keys included
        full = json.loads(detail.read_text())
        assert "kernels" in full["detail"] and full["value"] == _fat_result()["value"]


def test_compact_line_never_exceeds_the_limit_even_with_absurd_blocks():
    res = _fat_result()
    res["config"]["workload"] = "w" * 5000
    res["roofline"]["kernel"] = "k" * 3000
    for i in range(50):
        res[f"extra_{i}"] = "e" * 200
    text = bench.compact_line(res, os.path.join(ROOT, "bench_detail.json"))
    assert len(text.encode()) <= bench.LINE_LIMIT
    line = json.loads(text)
    for key in bench.CONTRACT_KEYS:
        assert key in line
    assert isinstance(line["roofline"], dict) and isinstance(line["cpu_baseline"], dict)


def test_ranks_other_than_zero_print_nothing(capsys, tmp_path):
    rc = bench.main(["--gpus", "1", "--detail", str(tmp_path / "d.json")], measure_fn=lambda args: None)
    assert rc == 0 and capsys.readouterr().out == ""
